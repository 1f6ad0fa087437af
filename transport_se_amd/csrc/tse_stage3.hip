// tse_stage3.hip -- the stage-3 kernel k_advance<2,3> (tse_kernels.h) in a translation unit of its own, because it is built with
// another instruction scheduler than the rest of the library: -mllvm -amdgpu-sched-strategy=max-ilp.
//
// k_advance<2,3> is the one kernel that holds two waves per SIMD (224 registers) AND keeps the vector units busy for more than half of
// its run time (19 000 VALU instructions per wave): with so little to interleave, the latency of its dependent fp64 chains shows.
// LLVM's default AMDGPU strategy schedules for occupancy first; "max-ilp" orders for instruction-level parallelism inside the
// register budget the launch bounds allow: 22.55 -> 22.08 ms per launch on one box, interleaved runs, same bits
// (profiles/r03_ab_sched_strategy.txt).  The strategy is a per-module compiler option, and for the three-wave kernels it is a loss
// (k_advance<0,0> 142 -> 196 registers, 11.5 -> 12.2 ms; k_advance<1,1> 168 -> 204, 14.5 -> 15.4), hence this file.
#include "tse_kernels.h"

namespace tse {

// launch k_advance<2,3,true,psz> over `blocks` (patch, chunk) blocks; the arguments are the kernel's
void launch_advance23(int psz, unsigned blocks, hipStream_t stream, int nelemd, const Dvv_t& D, const GeoPtrs& G, int qsize, double dt, double nu_q,
                      const double* B, const double* lapT, double* C, const double* vn0, const double* dp, const double* divdp,
                      const double* divdp_proj, double* qmin, double* qmax, const double* dp0, const GatherArgs& ga) {
  if (psz == 32)
    hipLaunchKernelGGL((k_advance<2, 3, true, 32>), dim3(blocks), dim3(Patch<32>::THREADS), 0, stream, nelemd, D, G, qsize, dt, nu_q, B, lapT, C, vn0, dp, divdp,
                       divdp_proj, qmin, qmax, dp0, ga);
  else if (psz == 24)
    hipLaunchKernelGGL((k_advance<2, 3, true, 24>), dim3(blocks), dim3(Patch<24>::THREADS), 0, stream, nelemd, D, G, qsize, dt, nu_q, B, lapT, C, vn0, dp, divdp,
                       divdp_proj, qmin, qmax, dp0, ga);
  else
    hipLaunchKernelGGL((k_advance<2, 3, true, 16>), dim3(blocks), dim3(Patch<16>::THREADS), 0, stream, nelemd, D, G, qsize, dt, nu_q, B, lapT, C, vn0, dp, divdp,
                       divdp_proj, qmin, qmax, dp0, ga);
}

}  // namespace tse
