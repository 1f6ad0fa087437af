// tools/region_probe.hip -- developer tool: is every region of the 288 GB equally fast?  Allocates N chunks of G GiB, and times a streaming
// read, a streaming write and an in-place read-modify-write of each chunk, then a copy between every pair of the first four.
//   hipcc -O3 --offload-arch=gfx950 -o region_probe region_probe.hip ;  ./region_probe [N=9] [G=28] [S: map of S-GiB pieces instead of the pair copies; -P: re-roll test with pads of P MiB; 0: allocation flags; 1 G1 G2 ..: chunks stitched from granules of G MiB with the virtual-memory API]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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
  if (argc > 3 && atoi(argv[3]) == 0) {   // allocation flags: does a physically contiguous (or another kind of) allocation write at a rate of its own?
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto t = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); launch(); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 2; };
    const unsigned flags[3] = {hipDeviceMallocDefault, hipDeviceMallocContiguous, hipDeviceMallocUncached};
    const char* names[3] = {"default", "contiguous", "uncached"};
    double* out2; CK(hipMalloc(&out2, 4096));
    for (int round = 0; round < N; round++)
      for (int f = 0; f < 3; f++) {
        char* p = nullptr;
        if (hipExtMallocWithFlags((void**)&p, bytes, flags[f]) != hipSuccess) { (void)hipGetLastError(); printf("%-10s allocation failed\n", names[f]); continue; }
        const float w = t([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, n, (double2*)p); });
        const float r = t([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, n, (const double2*)p, out2); });
        printf("%-10s %p write %5.0f read %5.0f GB/s\n", names[f], (void*)p, bytes / w / 1e6, bytes / r / 1e6);
        CK(hipFree(p));
      }
    return 0;
  }
  if (argc > 4 && atoi(argv[3]) == 1) {   // virtual-memory API: a chunk stitched from granules of argv[4] MiB, mapped in order / shuffled
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto t = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); launch(); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 2; };
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("allocation granularity %zu bytes\n", gran);
    double* out2; CK(hipMalloc(&out2, 4096));
    for (int ai = 4; ai < argc; ai++) {
      const size_t g = (size_t)atoi(argv[ai]) << 20;
      const size_t ng = bytes / g;
      for (int shuffled = 0; shuffled < 2; shuffled++) {
        std::vector<hipMemGenericAllocationHandle_t> h(ng);
        bool ok = true;
        for (size_t i = 0; i < ng && ok; i++) ok = hipMemCreate(&h[i], g, &prop, 0) == hipSuccess;
        if (!ok) { printf("hipMemCreate failed\n"); return 1; }
        void* va = nullptr; CK(hipMemAddressReserve(&va, ng * g, 0, nullptr, 0));
        std::vector<size_t> order(ng);
        for (size_t i = 0; i < ng; i++) order[i] = i;
        if (shuffled) { unsigned long long x = 88172645463325252ull; for (size_t i = ng - 1; i > 0; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::swap(order[i], order[x % (i + 1)]); } }
        for (size_t i = 0; i < ng; i++) CK(hipMemMap((char*)va + i * g, g, 0, h[order[i]], 0));
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, ng * g, &acc, 1));
        const size_t nn = ng * g / 16;
        const float w = t([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, nn, (double2*)va); });
        const float r = t([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, nn, (const double2*)va, out2); });
        const float u = t([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, nn, (const double2*)va, (double2*)va); });
        printf("granules of %5zu MiB, %s: write %5.0f read %5.0f update %5.0f GB/s\n", g >> 20, shuffled ? "shuffled" : "in order", ng * g / w / 1e6, ng * g / r / 1e6, 2.0 * ng * g / u / 1e6);
        CK(hipMemUnmap(va, ng * g)); CK(hipMemAddressFree(va, ng * g));
        for (size_t i = 0; i < ng; i++) CK(hipMemRelease(h[i]));
      }
    }
    return 0;
  }
  if (argc > 3 && atoi(argv[3]) < 0) {   // re-roll: does the same memory behind a small pad write at another rate?
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto t = [&](auto launch) { launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); launch(); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 2; };
    const size_t pad = (size_t)(-atoi(argv[3])) << 20;
    std::vector<char*> held;
    for (int k = 0; k < N; k++) {   // k held chunks in front (never freed), then a fresh one probed 3 times with a pad allocated between the tries
      printf("with %d chunk(s) held:", k);
      std::vector<char*> pads;
      for (int r = 0; r < 4; r++) {
        char* p; CK(hipMalloc(&p, bytes));
        const float w = t([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, n, (double2*)p); });
        printf("  %p %4.0f", (void*)p, bytes / w / 1e6);
        if (r == 3) { held.push_back(p); break; }
        CK(hipFree(p));
        char* q; CK(hipMalloc(&q, pad)); pads.push_back(q);
      }
      for (char* q : pads) CK(hipFree(q));
      printf("\n");
    }
    return 0;
  }
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
  if (argc > 3) {   // map: the write rate of every S-GiB piece of every chunk, then of 1-GiB allocations of their own
    const size_t sub = (size_t)atoi(argv[3]) << 30;
    for (int i = 0; i < N; i++) {
      printf("chunk %d write GB/s per %d GiB:", i, atoi(argv[3]));
      for (size_t o = 0; o + sub <= bytes; o += sub) {
        const float w = t([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, sub / 16, (double2*)(buf[i] + o)); });
        printf(" %4.0f", sub / w / 1e6);
      }
      printf("\n");
    }
    for (int i = 0; i < N; i++) {   // how long must a write be for a slow chunk to show?
      printf("chunk %d write GB/s over its first 1, 2, 4, 8, 16, all GiB:", i);
      for (size_t len = (size_t)1 << 30; ; len *= 2) {
        const size_t l = len < bytes ? len : bytes;
        const float w = t([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, l / 16, (double2*)buf[i]); });
        printf(" %4.0f", l / w / 1e6);
        if (l == bytes) break;
      }
      printf("   last 8 GiB: ");
      { const size_t l = (size_t)8 << 30; const float w = t([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, l / 16, (double2*)(buf[i] + bytes - l)); }); printf("%4.0f", l / w / 1e6); }
      printf("\n");
    }
    for (int i = 0; i < N; i++) CK(hipFree(buf[i]));
    std::vector<char*> small;
    for (int i = 0; i < (argc > 4 ? 250 : 0); i++) { char* p; if (hipMalloc(&p, (size_t)1 << 30) != hipSuccess) break; small.push_back(p); }
    printf("%zu allocations of 1 GiB, write GB/s of each:\n", small.size());
    for (size_t i = 0; i < small.size(); i++) {
      const float w = t([&] { hipLaunchKernelGGL(k_write, g, blk, 0, 0, ((size_t)1 << 30) / 16, (double2*)small[i]); });
      printf(" %4.0f%s", ((size_t)1 << 30) / w / 1e6, i % 25 == 24 ? "\n" : "");
    }
    printf("\n");
    return 0;
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
