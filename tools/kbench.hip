// tools/kbench.hip -- kernel micro-benchmark harness (developer tool, not product): times stream/slab copies and the
// non-gathering slab kernels on ne120-sized synthetic arrays with hipEvents (the DSS-on-read kernels need a real mesh:
// use bench.py + tools/ab_bench.sh for those).   hipcc -O3 --offload-arch=gfx950 -o kbench kbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../transport_se_amd/csrc/tse_kernels.h"
using namespace tse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int SLAB_THREADS = 320;   // one element per block: 72 levels x 4 rows = 288 active lanes
__global__ __launch_bounds__(SLAB_THREADS) void k_copy_slab(int qsize, const double* __restrict__ in, double* __restrict__ out) {
  const int e = blockIdx.x, tid = threadIdx.x, k = tid >> 2, j = tid & 3;
  if (k >= NLEV) return;
  size_t so = ((size_t)e * qsize * NLEV + k) * 16 + j * 4;
  for (int q = 0; q < qsize; q++) { double v[4]; load4(in + so, v); store4(out + so, v); so += (size_t)NLEV * 16; }
}
__global__ void k_copy_stream(size_t n, const double2* __restrict__ in, double2* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main(int argc, char** argv) {
  int ne = argc > 1 ? atoi(argv[1]) : 120, qsize = argc > 2 ? atoi(argv[2]) : 35;
  int nelem = 6 * ne * ne;
  size_t lev = (size_t)nelem * NLEV * 16, trc = lev * qsize;
  double *Q, *T, *vn0, *dp, *divdp, *divdp_proj, *qmin, *qmax, *dp0, *Dinv, *m1, *m2, *m3, *m4;
  CK(hipMalloc(&Q, trc * 8)); CK(hipMalloc(&T, trc * 8)); CK(hipMalloc(&vn0, 2 * lev * 8)); CK(hipMalloc(&dp, lev * 8));
  CK(hipMalloc(&divdp, lev * 8)); CK(hipMalloc(&divdp_proj, lev * 8)); CK(hipMalloc(&qmin, (size_t)nelem * qsize * NLEV * 8));
  CK(hipMalloc(&qmax, (size_t)nelem * qsize * NLEV * 8)); CK(hipMalloc(&dp0, NLEV * 8)); CK(hipMalloc(&Dinv, (size_t)nelem * 64 * 8));
  CK(hipMalloc(&m1, (size_t)nelem * 16 * 8)); CK(hipMalloc(&m2, (size_t)nelem * 16 * 8)); CK(hipMalloc(&m3, (size_t)nelem * 16 * 8)); CK(hipMalloc(&m4, (size_t)nelem * 16 * 8));
  // synthetic but well-conditioned data
  {
    std::vector<double> h(lev);
    for (size_t i = 0; i < lev; i++) h[i] = 1000.0 + (i % 97);
    CK(hipMemcpy(dp, h.data(), lev * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < lev; i++) h[i] = 1e-6 * ((i * 7) % 13 - 6);
    CK(hipMemcpy(divdp, h.data(), lev * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(divdp_proj, h.data(), lev * 8, hipMemcpyHostToDevice));
    std::vector<double> v(2 * lev);
    for (size_t i = 0; i < 2 * lev; i++) v[i] = 1e4 * (((i * 31) % 101) / 50.0 - 1.0);
    CK(hipMemcpy(vn0, v.data(), 2 * lev * 8, hipMemcpyHostToDevice));
    std::vector<double> g((size_t)nelem * 64);
    for (size_t i = 0; i < g.size(); i++) g[i] = (i % 4 == 0 || i % 4 == 3) ? 7.0 + (i % 5) : 0.3;
    CK(hipMemcpy(Dinv, g.data(), g.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> mm((size_t)nelem * 16, 0.02);
    CK(hipMemcpy(m1, mm.data(), mm.size() * 8, hipMemcpyHostToDevice));
    for (auto& x : mm) x = 50.0; CK(hipMemcpy(m2, mm.data(), mm.size() * 8, hipMemcpyHostToDevice));
    for (auto& x : mm) x = 1e-3; CK(hipMemcpy(m3, mm.data(), mm.size() * 8, hipMemcpyHostToDevice));
    for (auto& x : mm) x = 250.0; CK(hipMemcpy(m4, mm.data(), mm.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> d0(NLEV, 1000.0); CK(hipMemcpy(dp0, d0.data(), NLEV * 8, hipMemcpyHostToDevice));
    std::vector<double> qq((size_t)9216 * 64);
    for (size_t i = 0; i < qq.size(); i++) qq[i] = 1000.0 * (0.2 + 0.6 * ((i * 13) % 29) / 29.0);
    for (size_t off = 0; off < trc; off += qq.size()) CK(hipMemcpy(Q + off, qq.data(), std::min(qq.size(), trc - off) * 8, hipMemcpyHostToDevice));
    size_t mmn = (size_t)nelem * qsize * NLEV;
    std::vector<double> lo(mmn, 0.1), hi(mmn, 0.9);
    CK(hipMemcpy(qmin, lo.data(), mmn * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(qmax, hi.data(), mmn * 8, hipMemcpyHostToDevice));
  }
  Dvv_t D; double dv[16] = {-3, -0.809, 0.309, -0.5, 4.045, 0, -1.118, 1.545, -1.545, 1.118, 0, -4.045, 0.5, -0.309, 0.809, 3};
  for (int i = 0; i < 16; i++) D.d[i] = dv[i];
  double* dvvd; CK(hipMalloc(&dvvd, 128)); CK(hipMemcpy(dvvd, dv, 128, hipMemcpyHostToDevice));
  GeoPtrs G{Dinv, m1, m2, m3, m4, dvvd};
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int r = 0; r < 3; r++) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  double fb = trc * 8.0;
  timeit("copy_stream (R+W)", 2 * fb, [&] { hipLaunchKernelGGL(k_copy_stream, dim3(256 * 8), dim3(256), 0, 0, trc / 2, (const double2*)Q, (double2*)T); });
  timeit("copy_slab (R+W)", 2 * fb, [&] { hipLaunchKernelGGL(k_copy_slab, dim3(nelem), dim3(SLAB_THREADS), 0, 0, qsize, Q, T); });
  timeit("k_advance<0>", 2 * fb, [&] { hipLaunchKernelGGL(k_advance<0>, dim3(flat_blocks(nelem)), dim3(FLAT_THREADS), 0, 0, nelem, D, G, qsize, 37.5, 1e13, Q, (const double*)nullptr, T, vn0, dp, divdp, divdp_proj, qmin, qmax, dp0, GatherArgs{nullptr, nullptr, (size_t)(nelem + 1) * 16 * NLEV, nullptr}); });
  timeit("k_advance<1>", 2 * fb, [&] { hipLaunchKernelGGL(k_advance<1>, dim3(flat_blocks(nelem)), dim3(FLAT_THREADS), 0, 0, nelem, D, G, qsize, 37.5, 1e13, Q, (const double*)nullptr, T, vn0, dp, divdp, divdp_proj, qmin, qmax, dp0, GatherArgs{nullptr, nullptr, (size_t)(nelem + 1) * 16 * NLEV, nullptr}); });
  timeit("k_qminmax", fb, [&] { hipLaunchKernelGGL(k_qminmax, dim3(flat_blocks(nelem)), dim3(FLAT_THREADS), 0, 0, nelem, qsize, 0.0, Q, dp, divdp_proj, qmin, qmax); });
  return 0;
}
