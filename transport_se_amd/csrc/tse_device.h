// tse_device.h -- device-side building blocks of the gfx950 tracer-advection kernels.
//
// Thread layout used by every slab kernel ("row-per-lane"): one lane owns one row j (4 GLL points, i = 0..3,
// 32 contiguous bytes) of a 4x4 slab; the 4 lanes of a DPP quad own the 4 rows of one slab, so a 64-wide
// wavefront covers 16 slabs (16 consecutive levels) and reads/writes 2 KiB contiguous per instruction pair
// (two global_load_dwordx4 per lane).  Contractions along i are in-register; contractions along j and the
// 16-point reductions of the limiter are quad_perm DPP moves (no LDS, no ds_bpermute): two-stage butterflies over the quad.
#pragma once
#include <hip/hip_runtime.h>

#define NP 4
#define NLEV 72
#define NLEVP 73

namespace tse {

constexpr double RREARTH = 1.0 / 6.376e6;  // physical_constants.F90:22,34

struct Dvv_t { double d[16]; };  // d[l*4+i] = Dvv(i,l), passed by value (lives in SGPRs)

// one DPP quad_perm move of a double (two v_mov_b32_dpp); CTRL = quad_perm selector byte
template <int CTRL>
__device__ __forceinline__ double dppq(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// broadcast quad lane M to the 4 lanes of the quad
template <int M>
__device__ __forceinline__ double quad_bcast(double x) { return dppq<M * 0x55>(x); }
// A sum over the 4 rows of a slab, out_j = sum_m M(j,m) f_m, as a two-stage butterfly: the lane gets the value of its mirror row 3-j, forms
// the contribution of that pair of rows {j, 3-j} to its neighbour row j^1, and the two neighbours swap these partial sums -- 2 cross-lane
// moves and 4 multiply-adds per value where own row + three rotations take 3 and 4.  c[] = M(j,j), M(j,3-j), M(j^1,j), M(j^1,3-j).
__device__ __forceinline__ double quad_matvec(const double c[4], double f) {
#pragma clang fp contract(off)
  const double g = dppq<0x1B>(f);                    // row 3-j: quad_perm [3,2,1,0]
  const double send = fma(c[3], g, c[2] * f);        // rows {j, 3-j} seen from row j^1
  return fma(c[1], g, fma(c[0], f, dppq<0xB1>(send)));   // rows {j^1, 3-(j^1)}, then the lane's own pair
}
// butterfly reductions over the 4 lanes of a quad; every lane ends with the bit-identical result
__device__ __forceinline__ double quad_sum(double x) { x += dppq<0xB1>(x); x += dppq<0x4E>(x); return x; }
__device__ __forceinline__ double quad_min(double x) { x = fmin(x, dppq<0xB1>(x)); x = fmin(x, dppq<0x4E>(x)); return x; }
__device__ __forceinline__ double quad_max(double x) { x = fmax(x, dppq<0xB1>(x)); x = fmax(x, dppq<0x4E>(x)); return x; }

__device__ __forceinline__ void load4(const double* __restrict__ p, double v[4]) {
  const double2* q = reinterpret_cast<const double2*>(p);
  double2 a = q[0], b = q[1];
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ __forceinline__ void store4(double* __restrict__ p, const double v[4]) {
  double2* q = reinterpret_cast<double2*>(p);
  q[0] = make_double2(v[0], v[1]);
  q[1] = make_double2(v[2], v[3]);
}

// per-lane geometry of row j of element e
struct RowGeo {
  double Di11[4], Di21[4], Di12[4], Di22[4];  // Dinv(a,b,i,j)
  double metdet[4], rmetdet[4], spheremp[4];
  // coefficients of the lane in the order quad_matvec takes them: rows j, 3-j as seen from row j, then as seen from row j^1
  double dcol[4];  // M(j,m) = Dvv(m, j)   (sum over the row index in d/dy)
  double drow[4];  // M(j,m) = Dvv(j, m)   (weak divergence, derivative_mod.F90:2066-2070)
};

// dvv: a device copy of Dvv.  The lane's column/row of Dvv are loaded from it rather than selected out of the by-value
// kernel argument: a select chain over D.d[] by the lane's j is turned into a dynamically indexed read of D, which puts D
// into scratch memory and its 16 values into 32 VGPRs for the whole kernel (instead of SGPRs).
__device__ __forceinline__ void load_row_geo(RowGeo& g, const double* __restrict__ dvv, const double* __restrict__ Dinv,
                                             const double* __restrict__ metdet, const double* __restrict__ rmetdet,
                                             const double* __restrict__ spheremp, int e, int j) {
  const double* di = Dinv + ((size_t)e * 16 + j * 4) * 4;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double2 a = reinterpret_cast<const double2*>(di)[2 * i], b = reinterpret_cast<const double2*>(di)[2 * i + 1];
    g.Di11[i] = a.x; g.Di21[i] = a.y; g.Di12[i] = b.x; g.Di22[i] = b.y;
  }
  load4(metdet + (size_t)e * 16 + j * 4, g.metdet);
  load4(rmetdet + (size_t)e * 16 + j * 4, g.rmetdet);
  load4(spheremp + (size_t)e * 16 + j * 4, g.spheremp);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = r < 2 ? j : (j ^ 1), m = (r & 1) ? 3 - j : j;   // M(row, m)
    g.dcol[r] = dvv[row * 4 + m]; g.drow[r] = dvv[m * 4 + row];
  }
}

// dx[l] = sum_i Dvv(i,l) a[i]  (in-register) ;  dy[i] = sum_m Dvv(m,j) b(i,m)  (quad broadcast of rows)
__device__ __forceinline__ void deriv_xy(const Dvv_t& D, const RowGeo& g, const double a[4], const double b[4],
                                         double dx[4], double dy[4]) {
#pragma unroll
  for (int l = 0; l < 4; l++) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) s = s + D.d[l * 4 + i] * a[i];
    dx[l] = s;
  }
  // (the four rows meet pairwise, quad_matvec -- the reference adds them in row order 1..np, derivative_mod.F90:2395-2406; the same four terms)
#pragma unroll
  for (int i = 0; i < 4; i++) dy[i] = quad_matvec(g.dcol, b[i]);
}

// divergence_sphere (derivative_mod.F90:2364-2414) of v = (v1,v2) at the lane's 4 points
__device__ __forceinline__ void divergence_sphere_row(const Dvv_t& D, const RowGeo& g, const double v1[4],
                                                      const double v2[4], double div[4]) {
  double gv1[4], gv2[4], dx[4], dy[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    gv1[i] = g.metdet[i] * (g.Di11[i] * v1[i] + g.Di12[i] * v2[i]);
    gv2[i] = g.metdet[i] * (g.Di21[i] * v1[i] + g.Di22[i] * v2[i]);
  }
  deriv_xy(D, g, gv1, gv2, dx, dy);
#pragma unroll
  for (int i = 0; i < 4; i++) div[i] = (dx[i] + dy[i]) * (g.rmetdet[i] * RREARTH);
}

// laplace_sphere_wk = divergence_sphere_wk(gradient_sphere(s))  (derivative_mod.F90:1660-1700,2027-2097,2418-2460) in a
// register-lean form: with v = rrearth*(ds/dx, ds/dy),
//   vtemp = Dinv * (Dinv^T v)  =  [A B; B C] v,  A = Di11^2+Di12^2, B = Di11*Di21+Di12*Di22, C = Di21^2+Di22^2
// so only the 3 entries of the symmetric tensor spheremp*rrearth^2*[A B; B C] are kept per point (12 doubles per
// lane instead of 20); same operator as derivative_mod.F90:2418-2460, products re-associated.
struct LapGeo { double A[4], B[4], C[4], dcol[4], drow[4]; };
__device__ __forceinline__ void make_lap_geo(LapGeo& L, const RowGeo& g) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const double s = g.spheremp[i] * (RREARTH * RREARTH);
    L.A[i] = s * (g.Di11[i] * g.Di11[i] + g.Di12[i] * g.Di12[i]);
    L.B[i] = s * (g.Di11[i] * g.Di21[i] + g.Di12[i] * g.Di22[i]);
    L.C[i] = s * (g.Di21[i] * g.Di21[i] + g.Di22[i] * g.Di22[i]);
    L.dcol[i] = g.dcol[i]; L.drow[i] = g.drow[i];
  }
}
// The first Laplacian of stage 3 is formed in TWO kernels -- k_lap1 for the values other patches and ranks read, k_advance<2,3> for
// its own slots -- and both must give a point the SAME bits (otherwise the result would depend on where patch and rank boundaries
// lie).  What the compiler fuses on its own is decided per inlined copy, so these routines are compiled without implicit contraction
// and spell their fused operations out.
// dp of a stage: dp - c * divdp_proj (c = rhs_multiplier * dt; prim_advection_mod.F90:750-761)
__device__ __forceinline__ double dp_stage(double dp, double c, double dvp) {
#pragma clang fp contract(off)
  return fma(-c, dvp, dp);
}
// Q = Qdp * (1/dp) at the lane's 4 points
__device__ __forceinline__ void lap_q_of(const double qdp[4], const double rdp[4], double q[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = qdp[i] * rdp[i];
}
__device__ __forceinline__ void laplace_lean_row(const Dvv_t& D, const LapGeo& L, const double s[4], double lap[4]) {
#pragma clang fp contract(off)
#ifdef TSE_NO_CONTRACTION   // A/B build: see k_advance
  for (int i = 0; i < 4; i++) lap[i] = (L.A[i] + L.B[i] + L.C[i]) * s[i];
  return;
#endif
  // written point-by-point so that few cross-lane values are live at a time (register pressure)
  double w1[4], w2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const double dx = fma(D.d[i * 4 + 3], s[3], fma(D.d[i * 4 + 2], s[2], fma(D.d[i * 4 + 1], s[1], D.d[i * 4] * s[0])));
    const double dy = quad_matvec(L.dcol, s[i]);
    w1[i] = fma(L.B[i], dy, L.A[i] * dx);
    w2[i] = fma(L.C[i], dy, L.B[i] * dx);
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    // the x-sum and the y-sum as two independent chains (9 operations per point; the reference pairs each x-term with one y-term and
    // subtracts the pair, derivative_mod.F90:2066-2070: the same eight products)
    const double a = fma(w1[3], D.d[3 * 4 + m], fma(w1[2], D.d[2 * 4 + m], fma(w1[1], D.d[1 * 4 + m], w1[0] * D.d[0 * 4 + m])));
    lap[m] = -a - quad_matvec(L.drow, w2[m]);
  }
}

// limiter_optim_iter_full (prim_advection_mod.F90:976-1094) on one slab spread over a quad.
// x[i] = ptens/dpmass at the lane's 4 points, c[i] = sphweights*dpmass, sumc = sum(c) over the slab.
// minp/maxp are relaxed in place (intent(inout) in the reference).  The 16-point sums are tree sums
// (4 in-lane + quad butterfly) instead of the reference's serial k1 loop.
#ifdef TSE_LIMITER_STATS
__device__ unsigned long long g_lim_hist[20];  // [it] slabs converged at iteration it (16 = never), [17] wave-iterations, [18] waves
#endif
// returns true when the bounds were relaxed (the only case in which minp/maxp change)
__device__ __forceinline__ bool limiter8_quad(double x[4], const double c[4], double sumc, double& minp, double& maxp) {
  const double tol_limiter = (double)5e-14f;
#ifdef TSE_LIMITER_STATS
  int my_it = 16, wave_it = 0;
#endif
  if (!(sumc > 0.0)) return false;  // whole quad takes the same branch
  double mass = quad_sum(((c[0] * x[0] + c[1] * x[1]) + c[2] * x[2]) + c[3] * x[3]);
  const bool lo = mass < minp * sumc, hi = mass > maxp * sumc;
  if (__any(lo | hi)) {   // (rare: the division is skipped by the whole wave when no slab of it relaxes a bound)
    const double r = mass / sumc;
    if (lo) minp = r;
    if (hi) maxp = r;
  }
  const double tol_mass = tol_limiter * fabs(mass);
  for (int iter = 1; iter <= NP * NP - 1; iter++) {
    // clip to [minp,maxp]; the removed mass is sum (x - clipped)*c: (x-maxp)*c above, -(minp-x)*c below, +0 inside --
    // the same terms in the same order as the reference's two branches (:1037-1046), with v_min/v_max instead of
    // compare+select pairs
    double addmass = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double xc = fmin(fmax(x[i], minp), maxp);
      addmass = addmass + (x[i] - xc) * c[i];
      x[i] = xc;
    }
    addmass = quad_sum(addmass);
    const bool done = fabs(addmass) <= tol_mass;
#ifdef TSE_LIMITER_STATS
    wave_it = iter;
    if (done && my_it == 16) my_it = iter;
#endif
    if (__all(done)) break;  // wave-uniform exit; slabs already converged are left untouched below
    // redistribute over the points that are not pinned at the bound the mass moves towards (:1052-1078).  After the clip
    // x is inside [minp,maxp], so "x < maxp" (mass moving up) / "x > minp" (down) is just x != bound.  The selection is
    // carried as an exact 0.0/1.0 factor (one v_cndmask on the high word) folded into FMAs: fma(1,c,w) = w+c and
    // fma(1,inc,x) = x+inc round exactly like the reference's additions, fma(0,.,x) = x -- instead of 64-bit selects
    // (two v_cndmask each) around every add.
    const double bound = addmass > 0.0 ? maxp : minp;
    double m[4], w = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      m[i] = __hiloint2double(x[i] != bound ? 0x3ff00000 : 0, 0);
      w = fma(m[i], c[i], w);
    }
    w = quad_sum(w);
    // w == 0: every point is pinned (e.g. minp == maxp); the reference then applies addmass/weightssum to nobody, and an
    // infinite increment must not reach the 0-factor FMAs
    const double inc = (done || w <= 0.0) ? 0.0 : addmass / w;
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = fma(m[i], inc, x[i]);
  }
#ifdef TSE_LIMITER_STATS
  if ((threadIdx.x & 3) == 0) atomicAdd(&g_lim_hist[my_it], 1ull);
  if ((threadIdx.x & 63) == 0) { atomicAdd(&g_lim_hist[17], (unsigned long long)wave_it); atomicAdd(&g_lim_hist[18], 1ull); }
#endif
  return lo | hi;
}

}  // namespace tse
