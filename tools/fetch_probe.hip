// tools/fetch_probe.hip -- developer tool (not product): what does a partial-line read cost on gfx950?
// Reads PIECE contiguous bytes out of every STRIDE bytes of a buffer far larger than the Infinity Cache, 16 B per lane,
// and reports time per piece next to the plain streaming read of the same buffer.  If a 32-B piece per 128-B line costs
// as much as streaming the whole line, the halo-ring entries of the DSS-on-read kernels (32 B each) are priced by the
// line; if it costs a quarter, by the sector.  Run under `rocprofv3 --pmc FETCH_SIZE` to calibrate the counter on the
// same shapes.   hipcc -O3 --offload-arch=gfx950 -o fetch_probe fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// lanes_per_piece = PIECE / 16 consecutive lanes read one piece; pieces are STRIDE bytes apart, starting at byte OFF of the stride
template <int PIECE, int STRIDE, int OFF>
__global__ __launch_bounds__(256) void k_pieces(size_t npieces, const char* __restrict__ buf, double* __restrict__ out) {
  constexpr int LPP = PIECE / 16;
  double acc = 0.0;
  const size_t nl = npieces * LPP;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (size_t)gridDim.x * blockDim.x) {
    const size_t piece = i / LPP, l = i % LPP;
    const double2 v = *reinterpret_cast<const double2*>(buf + piece * STRIDE + OFF + l * 16);
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[threadIdx.x] = acc;
}
// the same with 4 independent loads in flight per lane and iteration
template <int PIECE, int STRIDE, int OFF>
__global__ __launch_bounds__(256) void k_pieces4(size_t npieces, const char* __restrict__ buf, double* __restrict__ out) {
  constexpr int LPP = PIECE / 16;
  double acc = 0.0;
  const size_t nl = npieces * LPP, g = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * g < nl; i += 4 * g) {
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const size_t ii = i + u * g; v[u] = *reinterpret_cast<const double2*>(buf + (ii / LPP) * STRIDE + OFF + (ii % LPP) * 16); }
#pragma unroll
    for (int u = 0; u < 4; u++) acc += v[u].x + v[u].y;
  }
  for (; i < nl; i += g) { const double2 v = *reinterpret_cast<const double2*>(buf + (i / LPP) * STRIDE + OFF + (i % LPP) * 16); acc += v.x + v.y; }
  if (acc == 1.2345e300) out[threadIdx.x] = acc;
}

// streaming mix of NR reads to one write, 16 B per lane: what HBM sustains for a kernel that reads NR fields and writes one
template <int NR>
__global__ __launch_bounds__(256) void k_mix(size_t n /* double2 per array */, const double2* __restrict__ in, double2* __restrict__ out) {
  const size_t g = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += g) {
    double2 a = in[i];
#pragma unroll
    for (int r = 1; r < NR; r++) { const double2 b = in[i + r * n]; a.x += b.x; a.y += b.y; }
    out[i] = a;
  }
}
// the same with every block working through contiguous 16 KB pieces (block-contiguous instead of grid-strided addresses)
template <int NR>
__global__ __launch_bounds__(256) void k_mix_blk(size_t n, const double2* __restrict__ in, double2* __restrict__ out) {
  const size_t per = 1024;   // double2 per piece: 16 KB
  for (size_t p = blockIdx.x; p * per < n; p += gridDim.x) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const size_t i = p * per + u * 256 + threadIdx.x;
      if (i < n) {
        double2 a = in[i];
#pragma unroll
        for (int r = 1; r < NR; r++) { const double2 b = in[i + r * n]; a.x += b.x; a.y += b.y; }
        out[i] = a;
      }
    }
  }
}

// The access shape of k_dss_patch on the reference's [e][q][k][p] layout: a block = 16 elements x PIECE contiguous bytes (its levels of one
// tracer), one tracer after the other (9216 B further on), the 18 chunks of an element handled by other blocks at other times; read one
// such stream, write another.  PIECE = 512 is the kernel's shape (4 levels); larger pieces = more levels per block.
template <int PIECE>
__global__ __launch_bounds__(256) void k_pieces_rw(int nelem, int qsize, const char* __restrict__ in, char* __restrict__ out) {
  constexpr int LPP = PIECE / 16, EPB = 256 / LPP;               // lanes per piece, elements per block
  constexpr int NCH = 9216 / PIECE;                              // pieces per (element, tracer)
  const int nblk_e = nelem / EPB;
  for (int b = blockIdx.x; b < nblk_e * NCH; b += gridDim.x) {
    const int ch = b / nblk_e, eb = b - ch * nblk_e;             // chunk-major walk, as the kernels do
    if ((int)threadIdx.x >= EPB * LPP) continue;                  // (pieces that do not fill the block)
    const int e = eb * EPB + threadIdx.x / LPP, l = threadIdx.x % LPP;
    size_t off = (size_t)e * qsize * 9216 + (size_t)ch * PIECE + l * 16;
    for (int q = 0; q < qsize; q++, off += 9216) {
      const double2 v = *reinterpret_cast<const double2*>(in + off);
      *reinterpret_cast<double2*>(out + off) = make_double2(v.x + 1.0, v.y);
    }
  }
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 8) << 30;
  char* buf; double* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMemset(buf, 0, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, size_t npieces, int piece, int stride, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-34s %8.3f ms  useful %7.1f GB/s  as-if-whole-stride %7.1f GB/s  (%zu pieces of %d B every %d B)\n", name, ms,
           (double)npieces * piece / ms / 1e6, (double)npieces * stride / ms / 1e6, npieces, piece, stride);
  };
  const dim3 grid(256 * 16), blk(256);
#define RUN(K, P, S, O) timeit(#K "<" #P "," #S "," #O ">", bytes / S, P, S, [&] { hipLaunchKernelGGL((K<P, S, O>), grid, blk, 0, 0, bytes / S, (const char*)buf, out); })
  RUN(k_pieces, 128, 128, 0);
  RUN(k_pieces4, 128, 128, 0);
  RUN(k_pieces4, 64, 128, 0);
  RUN(k_pieces4, 64, 128, 32);   // straddles the two 64-B halves
  RUN(k_pieces4, 32, 128, 0);
  RUN(k_pieces4, 32, 128, 48);   // straddles the two 64-B halves
  RUN(k_pieces4, 32, 64, 0);
  RUN(k_pieces4, 16, 128, 0);
  RUN(k_pieces4, 16, 64, 0);
  RUN(k_pieces4, 16, 32, 0);
  RUN(k_pieces4, 128, 256, 0);
  RUN(k_pieces4, 128, 256, 64);  // 128 B straddling two lines
  RUN(k_pieces4, 128, 512, 0);
  RUN(k_pieces4, 32, 512, 0);
  {
    // 1 GiB per array; inputs and output inside `buf` (needs >= 5 GiB)
    const size_t n = ((size_t)1 << 30) / 16;
    const double2* in = (const double2*)buf; double2* o = (double2*)(buf + ((size_t)4 << 30));
    auto mix = [&](const char* name, int nr, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      printf("%-34s %8.3f ms  %7.1f GB/s total (%d reads : 1 write)\n", name, ms, (double)(nr + 1) * n * 16 / ms / 1e6, nr);
    };
    if (bytes >= ((size_t)5 << 30)) {
      mix("k_mix<1> grid-stride", 1, [&] { hipLaunchKernelGGL(k_mix<1>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix<2> grid-stride", 2, [&] { hipLaunchKernelGGL(k_mix<2>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix<3> grid-stride", 3, [&] { hipLaunchKernelGGL(k_mix<3>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix_blk<1> 16KB pieces", 1, [&] { hipLaunchKernelGGL(k_mix_blk<1>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix_blk<2> 16KB pieces", 2, [&] { hipLaunchKernelGGL(k_mix_blk<2>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix_blk<3> 16KB pieces", 3, [&] { hipLaunchKernelGGL(k_mix_blk<3>, grid, blk, 0, 0, n, in, o); });
      mix("k_mix<1> 512 blocks", 1, [&] { hipLaunchKernelGGL(k_mix<1>, dim3(512), blk, 0, 0, n, in, o); });
      mix("k_mix<2> 512 blocks", 2, [&] { hipLaunchKernelGGL(k_mix<2>, dim3(512), blk, 0, 0, n, in, o); });
      mix("k_mix<2> 2048 blocks", 2, [&] { hipLaunchKernelGGL(k_mix<2>, dim3(2048), blk, 0, 0, n, in, o); });
    }
  }
  if (bytes >= ((size_t)8 << 30)) {
    // 3.5 GB per stream: 10 800 elements x 35 tracers x 9216 B
    const int nelem = 10800, qsize = 35;
    const size_t fb = (size_t)nelem * qsize * 9216;
    auto rw = [&](const char* name, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
      printf("%-34s %8.3f ms  %7.1f GB/s (read + write of %.1f GB each)\n", name, ms, 2.0 * fb / ms / 1e6, fb / 1e9);
    };
    rw("k_pieces_rw<512>  (4 levels)", [&] { hipLaunchKernelGGL(k_pieces_rw<512>, dim3(256 * 8), blk, 0, 0, nelem, qsize, (const char*)buf, buf + ((size_t)4 << 30)); });
    rw("k_pieces_rw<1024> (8 levels)", [&] { hipLaunchKernelGGL(k_pieces_rw<1024>, dim3(256 * 8), blk, 0, 0, nelem, qsize, (const char*)buf, buf + ((size_t)4 << 30)); });
    rw("k_pieces_rw<3072> (24 levels)", [&] { hipLaunchKernelGGL(k_pieces_rw<3072>, dim3(256 * 8), blk, 0, 0, nelem, qsize, (const char*)buf, buf + ((size_t)4 << 30)); });
  }
  return 0;
}