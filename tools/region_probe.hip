// tools/region_probe.hip -- developer tool: is every region of the 288 GB equally fast?  Allocates N chunks of G GiB, and times a streaming
// read, a streaming write and an in-place read-modify-write of each chunk, then a copy between every pair of the first four.
//   hipcc -O3 --offload-arch=gfx950 -o region_probe region_probe.hip ;  ./region_probe [N=9] [G=28]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void k_read(size_t n, const double2* __restrict__ in, double* __restrict__ out) {
  double acc = 0;
  const size_t per = 1024;
  for (size_t p = blockIdx.x; p * per < n; p += gridDim.x)
#pragma unroll
    for (int u = 0; u < 4; u++) { const size_t i = p * per + u * 256 + threadIdx.x; if (i < n) { const double2 v = in[i]; acc += v.x + v.y; } }
  if (acc == 1.2345e300) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_write(size_t n, double2* __restrict__ out) {
  const size_t per = 1024;
  for (size_t p = blockIdx.x; p * per < n; p += gridDim.x)
#pragma unroll
    for (int u = 0; u < 4; u++) { const size_t i = p * per + u * 256 + threadIdx.x; if (i < n) out[i] = make_double2(1.0, 2.0); }
}
__global__ __launch_bounds__(256) void k_copy(size_t n, const double2* __restrict__ in, double2* __restrict__ out) {
  const size_t per = 1024;
  for (size_t p = blockIdx.x; p * per < n; p += gridDim.x)
#pragma unroll
    for (int u = 0; u < 4; u++) { const size_t i = p * per + u * 256 + threadIdx.x; if (i < n) { double2 v = in[i]; v.x += 1.0; out[i] = v; } }
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 9;
  const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 28) << 30, n = bytes / 16;
  std::vector<char*> buf(N);
  double* out; CK(hipMalloc(&out, 4096));
  for (int i = 0; i < N; i++) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 0, bytes)); }
  CK(hipDeviceSynchronize());
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto t = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); launch(); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 2; };
  const dim3 g(256 * 8), blk(256);
  for (int i = 0; i < N; i++) {
    const float r = t([&] { hipLaunchKernelGGL(k_read, g, blk, 0, 0, n, (const double2*)buf[i], out); });
    const float w = t([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, n, (double2*)buf[i]); });
    const float u = t([&] { hipLaunchKernelGGL(k_copy, g, blk, 0, 0, n, (const double2*)buf[i], (double2*)buf[i]); });
    printf("chunk %d at %p: read %6.0f GB/s  write %6.0f GB/s  update in place %6.0f GB/s (r+w)\n", i, (void*)buf[i], bytes / r / 1e6, bytes / w / 1e6, 2.0 * bytes / u / 1e6);
  }
  const int M = N < 5 ? N : 5;
  for (int i = 0; i < M; i++) {
    printf("copy from chunk %d to:", i);
    for (int j = 0; j < M; j++) {
      if (i == j) { printf("     -"); continue; }
      const float c = t([&] { hipLaunchKernelGGL(k_copy, g, blk, 0, 0, n, (const double2*)buf[i], (double2*)buf[j]); });
      printf(" %5.0f", 2.0 * bytes / c / 1e6);
    }
    printf("  GB/s (r+w)\n");
  }
  return 0;
}
