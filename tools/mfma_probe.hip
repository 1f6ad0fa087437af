// Operand layout and summation order of v_mfma_f64_4x4x4f64 on gfx950, found by experiment (the MFMA question):
//   hipcc -O2 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.hip && ./tools/mfma_probe
// Result (MI355X, ROCm 7.2): with lane = 16*k + 4*g + i,   D[16*i + 4*g + j] = sum_k A[16*k + 4*g + i] * B[16*k + 4*g + j],
// accumulated as a sequential FMA chain over k = 0,1,2,3 (all 24 orders of {1e16, 1, -1e16, 1} agree with fma(a3,b3,fma(a2,b2,fma(a1,b1,a0*b0)))).
// I.e. the contraction index sits in lane bits 4-5, the 4 independent products in bits 2-3, the free index in bits 0-1 -- and moves to
// bits 4-5 in the result.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
  D[l] = d;
}
int main() {
  double *A, *B, *D;
  hipMalloc(&A, 64 * 8); hipMalloc(&B, 64 * 8); hipMalloc(&D, 64 * 8);
  double a[64], b[64], d[64];
  // which (A lane, B lane) pairs contribute to which output lane
  int da[64][4], db[64][4], cnt[64] = {0};
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      for (int i = 0; i < 64; i++) { a[i] = 0; b[i] = 0; }
      a[la] = 1; b[lb] = 1;
      hipMemcpy(A, a, 512, hipMemcpyHostToDevice); hipMemcpy(B, b, 512, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D);
      hipMemcpy(d, D, 512, hipMemcpyDeviceToHost);
      for (int i = 0; i < 64; i++) if (d[i] != 0 && cnt[i] < 4) { da[i][cnt[i]] = la; db[i][cnt[i]] = lb; cnt[i]++; }
    }
  for (int i = 0; i < 64; i++) {
    printf("D lane %2d = sum of", i);
    for (int t = 0; t < cnt[i]; t++) printf("  A[%2d]*B[%2d]", da[i][t], db[i][t]);
    printf("\n");
  }
  // order of the sum over k inside the instruction: D lane 0 = A[0]B[0] + A[16]B[16] + A[32]B[32] + A[48]B[48] in some order
  const double v[4] = {1e16, 1.0, -1e16, 1.0};
  int perm[24][4], np = 0;
  for (int p0 = 0; p0 < 4; p0++) for (int p1 = 0; p1 < 4; p1++) for (int p2 = 0; p2 < 4; p2++) for (int p3 = 0; p3 < 4; p3++)
    if (p0 != p1 && p0 != p2 && p0 != p3 && p1 != p2 && p1 != p3 && p2 != p3) { perm[np][0] = p0; perm[np][1] = p1; perm[np][2] = p2; perm[np][3] = p3; np++; }
  for (int t = 0; t < np; t++) {
    for (int i = 0; i < 64; i++) { a[i] = 0; b[i] = 0; }
    for (int kk = 0; kk < 4; kk++) { a[da[0][kk]] = v[perm[t][kk]]; b[db[0][kk]] = 1.0; }
    hipMemcpy(A, a, 512, hipMemcpyHostToDevice); hipMemcpy(B, b, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D);
    hipMemcpy(d, D, 512, hipMemcpyDeviceToHost);
    // sequential fma chain in term order 0,1,2,3 for comparison
    double seq = 0.0; for (int kk = 0; kk < 4; kk++) seq = __builtin_fma(v[perm[t][kk]], 1.0, seq);
    printf("terms (%g, %g, %g, %g): D = %.17g, sequential fma chain = %.17g\n", v[perm[t][0]], v[perm[t][1]], v[perm[t][2]], v[perm[t][3]], d[0], seq);
  }
  return 0;
}
