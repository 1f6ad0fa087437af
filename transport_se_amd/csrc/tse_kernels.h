// tse_kernels.h -- the gfx950 kernels of the tracer hot path (included once by tse_api.hip).
//
// HBM layout (all fp64, point index p = j*4+i fastest):
//   tracer fields  Qdp[tl][e][q][k][p]
//   pre-DSS scratch T, B: tracer-major planes; inside a plane the 72 levels are cut into NCHUNK chunks of CL = 4 levels and
//                  a chunk holds, level fastest, T[q][kc][slot][pos(p)][kk] (points perimeter first): the element slots (elements regrouped into
//                  patches of <= 16 neighbouring elements, tse_api.hip), then one all-zero slot (target of empty DSS
//                  contributions), then the received halo columns [col][kk].  Plane stride `tps` doubles, `cse` entries
//                  (points / halo columns) per chunk.  A plane is < 4 GB, so the DSS-on-read kernels address it with one
//                  uniform base + 32-bit byte offsets.
//   level fields   dp, divdp, divdp_proj, omega_p, dp3d [e][k][p]; vn0[e][k][c][p]; eta_dot_dpdn[e][73][p]
//   bounds         qmin/qmax[e][kc][q (rounded up to a multiple of 4)][kk] (mm_idx): the 4 levels of a chunk fastest, then the tracer, so that the 32 bytes a
//                  (patch x chunk) block needs per element and tracer share their line with the next three tracers
//   metric         Dinv[e][p][4], metdet/rmetdet/spheremp/rspheremp[e][p]
// Every slab kernel uses the row-per-lane layout of tse_device.h: thread = (slab (e,k), row j), looping over the tracers so
// that everything that depends on (e,k) only -- Vstar, dp, dp_star, the metric rows -- is computed once and stays in
// registers for all qsize tracers.
#pragma once
#include <type_traits>
#include "tse_device.h"

namespace tse {

// The plain slab kernels (k_divdp, k_qminmax, k_advance<*,0>, k_lap1<0>) run over the flattened slab index s = e*NLEV + k: a
// 72-level element is 4.5 waves, so element-sized blocks idle 10% of their lanes and -- worse -- come in units of 5 waves,
// which leaves SIMD wave slots empty whenever the register budget allows 2 or 3 waves per SIMD (8 or 12 per CU).
constexpr int FLAT_THREADS = 256;
// Scratch layout: CL levels per chunk.  The DSS-on-read kernels work on blocks of (patch of 16 element slots) x (one chunk):
// 16 slots x 4 levels x 4 rows = 256 lanes, and a slot's 16 points x 4 levels are 512 contiguous bytes.
constexpr int CL = 4;
constexpr int NCHUNK = NLEV / CL;
static_assert(NLEV % CL == 0 && CL % 2 == 0, "chunks hold whole level pairs");
constexpr int PS = 16;        // element slots per patch of the STORAGE tiling (4 x 4 elements): slot = patch * PS + position
// Block shapes of the DSS-on-read kernels.  A block owns a patch of PSZ element slots -- 4 x 4, 6 x 4 or 8 x 4 elements: 256, 384 or
// 512 lanes -- whatever the storage tiling is: its tables (PatchSet, tse_api.hip) name the storage slot of every element and of
// every halo-ring entry.  A wider patch has a shorter ring per element (0.375, 0.29, 0.25 of a field in whole lines), a narrower
// one leaves room for more blocks per CU; each kernel takes the shape that suits its register and LDS budget.
template <int PSZ> struct Patch {
  static_assert(PSZ == 16 || PSZ == 24 || PSZ == 32, "patch shapes: 4x4, 6x4, 8x4 elements");
  static constexpr int PS = PSZ, THREADS = PSZ * 16;
  // halo-ring entries (distinct (element, point) pairs outside the patch: 68, 84, 100 for the full shapes; tse_api.hip gives a
  // patch fewer rows if its ring would not fit); lanes 2r, 2r+1 load entry r
  static constexpr int NRMAX = PSZ == 16 ? 96 : PSZ == 24 ? 112 : 128;
  // Entries of one LDS buffer (an entry = the CL levels of a point = 32 bytes = 8 of the 64 banks): the own points, the ring, one all-zero
  // entry.  The own points are SKEWED (lds_own_entry): a slot takes LDS_SLOT = 20 entries instead of 16, and point (j, i) of a slot sits at
  // position 4j + ((i + j) & 3).  A wave reads, in one ds_read_b64, the same edge of four slots (its rows' neighbour values): in the
  // natural order an east or west edge is points 3,7,11,15 / 0,4,8,12 -- two bank groups -- and the four slots, 512 bytes apart, fall on
  // the same two: 8 cycles for the 2 the 512 bytes need.  Skewed, the four points of any edge are in four different groups and slots
  // s, s+1 in complementary halves of the banks: 2 cycles.  (SQ_LDS_BANK_CONFLICT: more than half of the LDS cycles of the four
  // gathering kernels before; profiles/r03_ab_lds_skew.txt.)
  static constexpr int LDS_SLOT = 20;
  static constexpr int LDS_RING = PSZ * LDS_SLOT, LDS_ZERO = LDS_RING + NRMAX, LDS_ENT = LDS_ZERO + 1;
  static_assert(2 * NRMAX <= THREADS, "one 16-byte ring load per lane");
};
__host__ __device__ inline int lds_own_entry(int sl, int p) { return sl * 20 + (p & ~3) + (((p & 3) + (p >> 2)) & 3); }   // (20 = Patch<>::LDS_SLOT)
inline int patch_nrmax(int psz) { return psz == 16 ? Patch<16>::NRMAX : psz == 24 ? Patch<24>::NRMAX : Patch<32>::NRMAX; }
struct Scr { size_t tps; unsigned cse; };   // plane stride (doubles), entries per chunk
// Inside a slot the 16 points are stored in a PER-SLOT order (nibble p of the slot's 64-bit word pperm[slot] = position of
// point p).  The memory system moves whole 128-byte lines (tools/fetch_probe.hip: 32 bytes out of every line cost what the
// line costs), a position holds the CL = 4 levels of a point = 32 bytes, so a slot is four lines of four points -- and what a
// neighbouring patch's halo ring reads from a slot is one EDGE of the element (4 points).  tse_init gives every edge that
// some patch or neighbour rank reads a line of its own (slot_perm, tse_api.hip), so a ring edge is one line instead of the
// two that three of the four edges straddled with one fixed perimeter-first order.
__host__ __device__ __forceinline__ int ppos(unsigned long long perm, int p) { return (int)((perm >> (4 * p)) & 15ull); }
// qmin/qmax(k,q,e) of prim_advection_mod (:459) in the device layout [e][k / CL][q][k % CL]
// (the tracer count of the bounds layout is rounded up to a multiple of 4: the 4 levels of 4 consecutive tracers are one aligned
// 128-byte line, so that the kernels that emit bounds can write whole lines)
__host__ __device__ __forceinline__ int mm_qpad(int qsize) { return (qsize + 3) & ~3; }
__device__ __forceinline__ size_t mm_idx(int e, int q, int k, int qsize) { return (((size_t)e * NCHUNK + k / CL) * mm_qpad(qsize) + q) * CL + (k % CL); }
// Blocks are dealt round-robin to the 8 XCDs, so logical block = (blockIdx % 8) * (gridDim/8) + blockIdx / 8 gives every
// XCD a contiguous range of slabs (and the ranges coincide with the element ranges the DSS kernels walk per XCD).
struct SlabId { int e, k; bool live; };
__device__ __forceinline__ SlabId flat_slab(int nwork, const int* __restrict__ order = nullptr) {
  const int per = gridDim.x >> 3, lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  SlabId s;
  const int n = nwork * NLEV, gs = lb * (FLAT_THREADS / 4) + (threadIdx.x >> 2);
  s.live = gs < n;
  const int g = s.live ? gs : n - 2 + (gs & 1);   // idle tail lanes recompute one of the last two slabs (their level parity
                                                  // is kept: lanes l and l+4 stay a level pair) and store nothing
  const int slot = g / NLEV;
  s.k = g - slot * NLEV;
  s.e = order ? order[slot] : slot;
  return s;
}
inline int flat_blocks(int nwork) { return 8 * ((nwork * NLEV * 4 + 8 * FLAT_THREADS - 1) / (8 * FLAT_THREADS)); }

struct GeoPtrs {
  const double* Dinv; const double* metdet; const double* rmetdet; const double* spheremp; const double* rspheremp;
  const double* dvv;   // device copy of deriv%Dvv (see load_row_geo)
};

// ---------------------------------------------------------------------------------------------------
// divdp = divdp_proj = divergence_sphere(vn0)   (prim_advection_mod.F90:614-623)
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(FLAT_THREADS) void k_divdp(int nelemd, Dvv_t D, GeoPtrs G, const double* __restrict__ vn0,
                                                        double* __restrict__ divdp, double* __restrict__ divdp_proj) {
  const SlabId sid = flat_slab(nelemd);
  const int e = sid.e, k = sid.live ? sid.k : NLEV, j = threadIdx.x & 3, kc = sid.k;
  RowGeo g;
  load_row_geo(g, G.dvv, G.Dinv, G.metdet, G.rmetdet, G.spheremp, e, j);
  double v1[4], v2[4], div[4];
  load4(vn0 + (((size_t)e * NLEV + kc) * 2 + 0) * 16 + j * 4, v1);
  load4(vn0 + (((size_t)e * NLEV + kc) * 2 + 1) * 16 + j * 4, v2);
  divergence_sphere_row(D, g, v1, v2, div);
  if (k < NLEV) {
    store4(divdp + ((size_t)e * NLEV + k) * 16 + j * 4, div);
    store4(divdp_proj + ((size_t)e * NLEV + k) * 16 + j * 4, div);
  }
}

// Single calls of the element-local operators on one 4x4 slab per element (the public derivative_mod functions the path is
// built from: divergence_sphere, derivative_mod.F90:2364-2414; laplace_sphere_wk, :2418-2460), through the same device
// routines the fused kernels use.  thread = (element, row j); OP 0: out = divergence_sphere(in[e][2][16]),
// OP 1: out = laplace_sphere_wk(in[e][16]).
template <int OP>
__global__ __launch_bounds__(256) void k_elem_op(int nelemd, Dvv_t D, GeoPtrs G, const double* __restrict__ in, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, j = t & 3;
  const int e = min(t >> 2, nelemd - 1);   // whole quads stay active (the DPP moves need all four rows)
  RowGeo g;
  load_row_geo(g, G.dvv, G.Dinv, G.metdet, G.rmetdet, G.spheremp, e, j);
  double r[4];
  if (OP == 0) {
    double v1[4], v2[4];
    load4(in + ((size_t)e * 2 + 0) * 16 + j * 4, v1); load4(in + ((size_t)e * 2 + 1) * 16 + j * 4, v2);
    divergence_sphere_row(D, g, v1, v2, r);
  } else {
    double sv[4];
    LapGeo L;
    make_lap_geo(L, g);
    load4(in + (size_t)e * 16 + j * 4, sv);
    laplace_lean_row(D, L, sv, r);
  }
  if ((t >> 2) < nelemd) store4(out + (size_t)e * 16 + j * 4, r);
}

// Per-element tracer mass sum_k sum_p spheremp(p) * Qdp(p,k,q) -- the element's share of global_integral behind the "Q, Q diss"
// diagnostics (global_norms_mod.F90:39-86, prim_state_mod.F90:352-385) -- in a FIXED order (points 0..15 inside a level, then
// levels 0..71), so that an element's partial does not depend on which rank or block computed it; the cross-element sum is
// done by the caller with an exact (order-independent) summation.  block = (element, tracer), thread = level.
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(128) void k_elem_mass(int qsize, const double* __restrict__ Q, const double* __restrict__ spheremp, double* __restrict__ out) {
  __shared__ double lev[NLEV];
  const int e = blockIdx.x / qsize, q = blockIdx.x - e * qsize, k = threadIdx.x;
  if (k < NLEV) {
    const double* x = Q + (((size_t)e * qsize + q) * NLEV + k) * 16;
    const double* w = spheremp + (size_t)e * 16;
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < 16; p++) s = s + w[p] * x[p];
    lev[k] = s;
  }
  __syncthreads();
  if (k == 0) {
    double s = 0.0;
    for (int l = 0; l < NLEV; l++) s = s + lev[l];
    out[blockIdx.x] = s;
  }
}

// The element's share of the two integrals behind the "Q<q>,Q diss, dQ^2/dt:" line of prim_printstate (prim_state_mod.F90:352-385):
// prim_diag_scalars (:604-655) forms, per point, Qmass = sum_k Qdp and Qvar = sum_k Qdp*Q with Q = Qdp/dp,
// dp = (hyai(k+1)-hyai(k))*ps0 + (hybi(k+1)-hybi(k))*ps_v (prim_driver_mod.F90:810-815), and global_integral
// (global_norms_mod.F90:74-80) adds da*h over the element's points, i fastest.  Same operations in the same order (no contraction), so
// that an element's partial is a fixed number whatever computes it.  block = (element, tracer); lanes 0..15 = points.
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(64) void k_elem_qdiag(int qsize, const double* __restrict__ Q, const double* __restrict__ spheremp, const double* __restrict__ ps_v,
                                                   const double* __restrict__ hyai, const double* __restrict__ hybi, double ps0,
                                                   double* __restrict__ mass_out, double* __restrict__ var_out,
                                                   double* __restrict__ min_out, double* __restrict__ max_out /* element min/max of Q (the qv= line, :184-192) */) {
#pragma clang fp contract(off)
  __shared__ double hm[16], hv[16], hn[16], hx[16];
  const int e = blockIdx.x / qsize, q = blockIdx.x - e * qsize, p = threadIdx.x;
  if (p < 16) {
    const double* x = Q + ((size_t)e * qsize + q) * NLEV * 16 + p;
    const double ps = ps_v[(size_t)e * 16 + p];
    double m = 0.0, v = 0.0, mn = 1e300, mx = -1e300;
    for (int k = 0; k < NLEV; k++) {
      const double dpk = (hyai[k + 1] - hyai[k]) * ps0 + (hybi[k + 1] - hybi[k]) * ps;
      const double qd = x[(size_t)k * 16], qq = qd / dpk;
      m = m + qd;
      v = v + qd * qq;
      mn = fmin(mn, qq); mx = fmax(mx, qq);
    }
    hm[p] = m; hv[p] = v; hn[p] = mn; hx[p] = mx;
  }
  __syncthreads();
  if (p == 0) {
    const double* w = spheremp + (size_t)e * 16;
    double jm = 0.0, jv = 0.0, mn = hn[0], mx = hx[0];
    for (int i = 0; i < 16; i++) { jm = jm + w[i] * hm[i]; jv = jv + w[i] * hv[i]; mn = fmin(mn, hn[i]); mx = fmax(mx, hx[i]); }
    mass_out[blockIdx.x] = jm; var_out[blockIdx.x] = jv; min_out[blockIdx.x] = mn; max_out[blockIdx.x] = mx;
  }
}

// ---------------------------------------------------------------------------------------------------
// element min/max of Q = Qdp/dp, dp = derived%dp - rhs_multiplier*dt*divdp_proj  (prim_advection_mod.F90:750-775)
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(FLAT_THREADS) void k_qminmax(int nelemd, int qsize, double rdt /* rhs_multiplier*dt */,
                                                          const double* __restrict__ Qn0, const double* __restrict__ dp,
                                                          const double* __restrict__ divdp_proj,
                                                          double* __restrict__ qmin, double* __restrict__ qmax) {
  const SlabId sid = flat_slab(nelemd);
  const int e = sid.e, k = sid.live ? sid.k : NLEV, j = threadIdx.x & 3, kc = sid.k;
  double dpk[4], dv[4];
  const size_t lo = ((size_t)e * NLEV + kc) * 16 + j * 4;
  load4(dp + lo, dpk); load4(divdp_proj + lo, dv);
  // Q = Qdp * (1/dp): every kernel that forms element bounds (here, k_lap1, k_advance<1>, the emissions of k_dss_patch<1> and
  // k_remap) multiplies by the reciprocal of dp, computed once per (element, level) -- so the cached bounds a step inherits from
  // its predecessor's last kernel and the bounds recomputed here are the same bits (the reference divides, :764-775: <= 1 ulp
  // in a limiter bound)
#pragma unroll
  for (int i = 0; i < 4; i++) dpk[i] = 1.0 / (dpk[i] - rdt * dv[i]);
  for (int q = 0; q < qsize; q++) {
    double x[4];
    load4(Qn0 + (((size_t)e * qsize + q) * NLEV + kc) * 16 + j * 4, x);
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = x[i] * dpk[i];
    double mn = quad_min(fmin(fmin(x[0], x[1]), fmin(x[2], x[3])));
    double mx = quad_max(fmax(fmax(x[0], x[1]), fmax(x[2], x[3])));
    if (j == 0 && k < NLEV) {
      qmin[mm_idx(e, q, k, qsize)] = mn;
      qmax[mm_idx(e, q, k, qsize)] = mx;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// neighbor_minmax / min-max half of biharmonic_wk_scalar_minmax (viscosity_mod.F90:748-816,389-432):
// min/max over the element and its <= 8 neighbours.  nbr[e][8]: >= 0 local element, -1 none,
// <= -2 remote: column -(v+2) of the received halo (layer index = (q*NLEV+k), min set then max set).
// Block = (patch of the scratch layout, tile of 64 level-pair entries); a wave serves two element slots.  Each wave publishes
// its elements' tiles in LDS; of the 8 neighbours of an element two thirds (84 of 128 in a full 4x4 patch) are slots of the
// same patch and come from there, the rest -- other patches, the received halo -- from global memory.  The neighbour
// structure is wave-uniform (scalar loads and branches).  8 GB per launch at ne120/q35: 1.9 ms; one element per block with all
// 8 neighbours from global memory (L2) took 2.45 ms, 16 waves with one slot each 2.15, 4 waves with four slots 2.0.
constexpr int MM_TILE = 64;
inline int nbr_patch_blocks(int npatch, int qsize) { return 8 * ((npatch + 7) / 8) * ((mm_qpad(qsize) * NLEV / 2 + MM_TILE - 1) / MM_TILE); }
template <int WB /* waves per block; a wave serves PS / WB element slots */>
__global__ __launch_bounds__(WB * 64) void k_nbr_minmax_patch(int npatch, int qsize, const int* __restrict__ nbr, const int* __restrict__ pslots,
                                                             const int* __restrict__ slot_of, const double* __restrict__ in_min,
                                                             const double* __restrict__ in_max, double* __restrict__ out_min,
                                                             double* __restrict__ out_max, const double* __restrict__ recvbuf, int nlyr_halo) {
  __shared__ double2 smn[PS][MM_TILE], smx[PS][MM_TILE];
  constexpr int R = PS / WB;
  const int m = mm_qpad(qsize) * NLEV;   // entries per element of the bounds layout
  // the 8 XCDs each take a contiguous range of patches and walk it tile by tile: the patches whose elements a block reads
  // from global memory ran the same tile on the same XCD a few blocks earlier
  const int npx = (npatch + 7) >> 3, x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int tile = i / npx, pi = x * npx + (i - tile * npx);
  if (pi >= npatch) return;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int l = 2 * (tile * MM_TILE + lane);
  const bool inr = l < m;   // m is even: l + 1 < m as well
  int e[R];
  double2 mn[R], mx[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    e[r] = __builtin_amdgcn_readfirstlane(pslots[pi * PS + w + r * WB]);   // -1: hole
    mn[r] = make_double2(0., 0.); mx[r] = mn[r];
    if (e[r] >= 0 && inr) { mn[r] = *reinterpret_cast<const double2*>(in_min + (size_t)e[r] * m + l); mx[r] = *reinterpret_cast<const double2*>(in_max + (size_t)e[r] * m + l); }
  }
  // the neighbours that are not slots of this patch -- other patches, the received halo -- are loaded and reduced BEFORE the barrier,
  // with the own values in flight: one memory latency per block instead of two (the block lives for little more than that); min and
  // max are exact, so the order in which the 9 values meet does not matter
  double2 a[R], b[R];
  int lsl[R][8];   // >= 0: neighbour d of slot r is slot lsl of this patch (from LDS, behind the barrier); wave-uniform
#pragma unroll
  for (int r = 0; r < R; r++) {
    a[r] = mn[r]; b[r] = mx[r];
#pragma unroll
    for (int d = 0; d < 8; d++) {
      lsl[r][d] = -1;
      if (e[r] < 0) continue;
      const int n = nbr[e[r] * 8 + d];   // wave-uniform
      double2 x = mn[r], y = mx[r];      // no neighbour in this direction
      if (n >= 0) {
        const int sl = slot_of[n];
        if (sl / PS == pi) lsl[r][d] = sl - pi * PS;
        else if (inr) { x = *reinterpret_cast<const double2*>(in_min + (size_t)n * m + l); y = *reinterpret_cast<const double2*>(in_max + (size_t)n * m + l); }
      } else if (n <= -2 && inr) {
        const double* h = recvbuf + (size_t)(-(n + 2)) * nlyr_halo + l;
        x = *reinterpret_cast<const double2*>(h); y = *reinterpret_cast<const double2*>(h + m);
      }
      a[r].x = fmin(a[r].x, x.x); a[r].y = fmin(a[r].y, x.y);
      b[r].x = fmax(b[r].x, y.x); b[r].y = fmax(b[r].y, y.y);
    }
  }
#pragma unroll
  for (int r = 0; r < R; r++) { smn[w + r * WB][lane] = mn[r]; smx[w + r * WB][lane] = mx[r]; }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (e[r] < 0) continue;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      if (lsl[r][d] < 0) continue;
      const double2 x = smn[lsl[r][d]][lane], y = smx[lsl[r][d]][lane];
      a[r].x = fmin(a[r].x, x.x); a[r].y = fmin(a[r].y, x.y);
      b[r].x = fmax(b[r].x, y.x); b[r].y = fmax(b[r].y, y.y);
    }
    if (inr) { *reinterpret_cast<double2*>(out_min + (size_t)e[r] * m + l) = a[r]; *reinterpret_cast<double2*>(out_max + (size_t)e[r] * m + l) = b[r]; }
  }
}

// ---------------------------------------------------------------------------------------------------
// DSS on read.  In the whole-step path the DSS'd field between two RK stages is never written to memory: the consuming
// slab kernel assembles rspheremp*DSS(T) for its own row from the producer's pre-DSS scratch -- the lane's 4 own points plus
// up to 8 neighbour edge/corner values -- with the same table, the same summation order (S, E, N, W edges, then the
// corner) and the same inverse-mass multiply as k_dss_patch, so the value is bit-identical to what the DSS pass would have
// stored.  That removes one read+write pass over the tracers per fused hand-over.
//
// Where the neighbour values come from.  Each of them is some other element's own value, so a block that owns a PATCH of
// neighbouring elements already loads most of what its slabs need: block = (patch of <= 16 element slots, one chunk of 4
// levels).  Per tracer every lane loads its own points (two 16-byte loads shared with the lane that holds the other level
// of the pair), the block loads the patch's HALO RING -- the edge/corner columns of elements outside the patch, 92 of them
// for a full 4x4 patch against 320 neighbour values inside -- with one more 16-byte load per lane, everything is published
// to LDS, and after one workgroup barrier the 20 neighbour values of a slab are LDS reads.  Compared with gathering them
// from global memory this fetches 0.36 instead of 1.25 times the field on top of the own values: the gathers of
// concurrently running blocks did not meet in the XCD's L2 (blocks drift apart by several tracers), so they were HBM traffic.
// LDS is double-buffered over the tracers; one barrier per tracer separates the publish from the reads, and the buffer
// written for tracer q+2 was last read before the barrier of tracer q+1.
//
// Who reads what (as before): the quad shares the neighbour values -- every lane is responsible for 5 of the slab's 20 (the
// middle rows take, besides their own two, the third contribution of point 0 and the contributions of points 1 and 2 of
// the edge row next to them) and hands them over with one quad_perm DPP move each.
//
// plist/npwork: the patches this launch walks (nullptr: patches 0..npwork).  A multi-rank step launches every slab kernel
// twice: first over the patches (plain kernels: elements, order/nwork) that touch another rank, so that their halo can travel
// while the second launch computes the interior (tse_api.hip).
constexpr int NER = 48;            // elements around a patch whose bounds the stage-3 kernel reads (full 4x4, 6x4, 8x4 patches: 20, 24, 28; a two-deep band
                                   // along a rank boundary: every received (element, direction) pair is an entry of its own, about 40)
struct GatherArgs {
  Scr S;
  const int* slot_of;              // element -> slot of the scratch layout
  const int* order; int nwork;     // plain slab kernels: element list of this launch
  const double* rspheremp;
  const int* pslots;               // [npatch][PS] element of a slot, -1 = hole
  const unsigned* pring;           // [npatch][NRMAX] ring entry -> entry index within a chunk (slot*16+p, zero slot, halo column)
  const unsigned short* plds;      // [npatch][PS][16][3] DSS contribution -> LDS entry (own-patch point, ring entry, zero entry)
  const int* plist; int npwork;    // patch list of this launch
  // The stage's extra DSS variable (divdp_proj / eta_dot_dpdn / omega_p, prim_advection_mod.F90:911-919,943-957) rides along as
  // one more plane (index qsize) of the scratch fields, exactly as edgeAdv_p1 carries it behind the tracers:
  const double* var_in;  int var_in_lev;    // producer: plane qsize of the output := spheremp * var_in[e][var_in_lev][p] (levels 0..71)
  double* var_out;       int var_out_lev;   // consumer: rspheremp*DSS(plane qsize of the gathered input) -> var_out[e][var_out_lev][p]
  // stage 3 forms the neighbour min/max of the element bounds itself (min/max half of biharmonic_wk_scalar_minmax):
  double* divdp_out;               // stage 1 (k_advance<0,0>): divdp = divergence_sphere(vn0) is formed here and stored (no k_divdp pass)
  const int* pering;               // [npatch][NER] elements around the patch (>= nelemd: received entry nelemd + i, stored behind the local elements)
  const unsigned char* pnb;        // [npatch][PS][8] neighbour d of a slot -> entry of the bounds image (slot, PS + ring entry, 255 = none)
  const unsigned long long* pperm; // [slot] point order inside the slot (ppos)
  const unsigned char* pexp;       // [slot] exported lines of the slot (row_store_setup); null: every point is stored
};
// bounds image of the stage-3 kernel: [buffer][element entry: the patch's slots, then the element ring][min|max][level of the chunk]
template <int PSZ> struct BoundsLds { static constexpr int ENT = PSZ + NER; double v[2][ENT][2][CL]; };   // 6-8 KB
template <int PSZ> struct PatchLds { double v[2][Patch<PSZ>::LDS_ENT][CL]; };   // 2 x 11.3 | 15.9 | 20.5 KB

__device__ __forceinline__ double swz_xor4(double x) {   // value of lane ^ 4 (the other level of the pair)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F); hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F);   // and 0x1f, or 0, xor 4
  return __hiloint2double(hi, lo);
}

// block -> (patch, chunk), lane -> (slot of the patch, level of the chunk, row).  The 8 XCDs each take a contiguous range of
// patches and walk it chunk by chunk, so that the patches whose own values are a block's ring are in flight on the same XCD.
struct PatchId { int patch, tslot /* patch * PSZ + position: index into the patch tables */, slot /* storage slot of the element */, e, k, j; bool live, any; };
inline int patch_blocks(int npwork) { return 8 * ((npwork + 7) / 8) * NCHUNK; }
template <int PSZ>
__device__ __forceinline__ PatchId patch_slab(const GatherArgs& A) {
  const int npx = (A.npwork + 7) >> 3, x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int kc = i / npx, pi = x * npx + (i - kc * npx);
  PatchId P;
  P.any = pi < A.npwork;
  P.patch = P.any ? (A.plist ? A.plist[pi] : pi) : 0;
  const int sl = threadIdx.x >> 4;
  P.k = kc * CL + ((threadIdx.x >> 2) & (CL - 1));
  P.j = threadIdx.x & 3;
  P.tslot = P.patch * PSZ + sl;
  const int e = A.pslots[P.tslot];
  P.live = e >= 0;
  P.e = P.live ? e : A.pslots[P.patch * PSZ];   // a hole recomputes the patch's first element and stores nothing
  P.slot = A.slot_of[P.e];
  return P;
}

struct GatherRaw { double2 w[2], r; };   // the lane's two own loads and its ring load
struct RowGather {
  unsigned own, own1;  // plane-relative byte offsets of T[.][kc][slot][pos(j*4 + (odd ? 1 : 0) + {0,2})][kk & ~1]
  unsigned ring;       // plane-relative byte offset of the lane's half of a ring entry
  unsigned lw, lw1, lwr;  // LDS byte offsets (buffer 0) where the lane publishes its two own loads / its ring load
  unsigned lr[5];      // LDS byte offsets (buffer 0) of the lane's five neighbour values
  double rs[4];
};
template <int PSZ>
__device__ __forceinline__ void gather_setup(RowGather& R, PatchLds<PSZ>& L, const GatherArgs& A, const PatchId& P) {
  constexpr int NRMAX = Patch<PSZ>::NRMAX, LDS_ZERO = Patch<PSZ>::LDS_ZERO;
  if (threadIdx.x < 2 * CL) L.v[threadIdx.x / CL][LDS_ZERO][threadIdx.x % CL] = 0.0;   // target of absent contributions (read after the first barrier)
  const int j = P.j, kk = P.k & (CL - 1), kc = P.k / CL, sl = P.tslot - P.patch * PSZ;
  const bool edge = (j == 0) | (j == 3), odd = kk & 1;
  const int jt = j == 1 ? 0 : (j == 2 ? 3 : j);   // the edge row a middle row helps
  // (row, point, contribution index) of the row's five fetches
  const int rw[5] = {j, j, edge ? j : jt, edge ? j : jt, edge ? j : jt};
  const int pt[5] = {0, edge ? 0 : 3, edge ? 3 : 0, edge ? 3 : 1, edge ? 3 : 2};
  const int cn[5] = {0, edge ? 1 : 0, edge ? 0 : 2, edge ? 1 : 0, edge ? 2 : 0};
  const unsigned chunk0 = (unsigned)kc * A.S.cse;                       // first entry of the chunk
  // the two lanes of a level pair share the row's 4 points {0,2} (even level) / {1,3} (odd): the pair's first load covers points 0 and 1,
  // its second points 2 and 3 -- where a row is a line of the slot, 64 contiguous bytes per instruction (with {0,1} / {2,3}, the form
  // up to round 4, each instruction touched two 32-byte pieces 32 bytes apart: 0.4 ms per step, profiles/r04_ab_pair_adjacent.txt)
  const int p0 = j * 4 + (odd ? 1 : 0), p1 = p0 + 2;
  const unsigned long long perm = A.pperm[P.slot];
  R.own = ((chunk0 + (unsigned)P.slot * 16 + ppos(perm, p0)) * CL + (kk & ~1)) * 8u;
  R.own1 = ((chunk0 + (unsigned)P.slot * 16 + ppos(perm, p1)) * CL + (kk & ~1)) * 8u;
  R.lw = (unsigned)(lds_own_entry(sl, p0) * CL + (kk & ~1)) * 8u;
  R.lw1 = (unsigned)(lds_own_entry(sl, p1) * CL + (kk & ~1)) * 8u;
  unsigned short le[5];
#pragma unroll
  for (int m = 0; m < 5; m++) le[m] = A.plds[((size_t)P.tslot * 16 + rw[m] * 4 + pt[m]) * 3 + cn[m]];
  // ring: lanes 2r, 2r+1 load the two level pairs of ring entry r (unused entries of the table hold the zero slot); the lanes
  // beyond the table repeat its last entry (same address, same LDS word, same value)
  const int r = min((int)(threadIdx.x >> 1), NRMAX - 1), half = threadIdx.x & 1;
  const unsigned ent = A.pring[(size_t)P.patch * NRMAX + r];
  R.ring = ((chunk0 + ent) * CL + half * 2) * 8u;
  R.lwr = (unsigned)((Patch<PSZ>::LDS_RING + r) * CL + half * 2) * 8u;
#pragma unroll
  for (int m = 0; m < 5; m++) R.lr[m] = ((unsigned)le[m] * CL + kk) * 8u;
  load4(A.rspheremp + (size_t)P.e * 16 + j * 4, R.rs);
}
// loads only, all 3 in flight together; every use comes later.
// The empty asm keeps the offsets opaque inside the tracer loop: otherwise their zero-extension is hoisted out of the loop
// (two registers per offset) and the loads fall back from "SGPR base + 32-bit VGPR offset" to 64-bit VALU address math.
__device__ __forceinline__ void gather_issue(RowGather& R, const GatherArgs& A, const double* __restrict__ src, int q, GatherRaw& raw) {
  asm volatile("" : "+v"(R.own), "+v"(R.own1), "+v"(R.ring));
  const char* pq = reinterpret_cast<const char*>(src + (size_t)q * A.S.tps);   // wave-uniform
  raw.w[0] = *reinterpret_cast<const double2*>(pq + R.own);
  raw.w[1] = *reinterpret_cast<const double2*>(pq + R.own1);
  raw.r = *reinterpret_cast<const double2*>(pq + R.ring);
}
// publish the lane's loads of tracer q in LDS buffer `b` and keep its own 4 values (level pair exchange: keep my level's half
// of what I loaded, send the other half to lane ^ 4).  Holes publish their copy into their own (unreferenced) entries.
template <int PSZ>
__device__ __forceinline__ void gather_publish(const RowGather& R, PatchLds<PSZ>& L, int b, int k, const GatherRaw& raw, double v[4]) {
  const bool odd = k & 1;
  char* base = reinterpret_cast<char*>(&L.v[b][0][0]);
  *reinterpret_cast<double2*>(base + R.lw) = raw.w[0];
  *reinterpret_cast<double2*>(base + R.lw1) = raw.w[1];
  *reinterpret_cast<double2*>(base + R.lwr) = raw.r;
  const double r0 = swz_xor4(odd ? raw.w[0].x : raw.w[0].y), r1 = swz_xor4(odd ? raw.w[1].x : raw.w[1].y);
  v[0] = odd ? r0 : raw.w[0].x; v[1] = odd ? raw.w[0].y : r0; v[2] = odd ? r1 : raw.w[1].x; v[3] = odd ? raw.w[1].y : r1;
  // the values must have left `raw` before the next tracer's loads are issued into it
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
}
// all LDS writes of the workgroup have landed; the loads of the next tracer stay in flight (no vmcnt wait)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory"); }

// The tracer loop of the double-buffered kernels: tracer 0, then PAIRS (results alternate between the register sets A and B; a step
// issues the stores of the tracer before it).  The compiler emits one copy of the step body per call site and decides FMA contraction
// per copy, so a tracer's last bits depend on WHICH copy computed it.  With exactly three copies in fixed roles -- tracer 0, the first
// and the second step of a pair -- tracer i is computed by the same copy whatever qsize is: tracers 1-4 of a 35-tracer run are bit
// for bit the 4-tracer run (tests/test_gpu_baseline_configs.py).  An even tracer count therefore ends with a SURPLUS second step that
// repeats the last tracer (it issues that tracer's stores; its own result is dropped) instead of a fourth copy of the body -- the
// form before round 4, which gave the last tracer of an even count other last bits than the same tracer inside a longer run.
#define TSE_TRACER_PAIRS(step, put, qsize, A, B)                          \
  do {                                                                    \
    step(0, nullptr, 0, A);                                               \
    for (int q_ = 1; q_ < (qsize); q_ += 2) {                             \
      step(q_, &A, q_ - 1, B);                                            \
      step(q_ + 1 < (qsize) ? q_ + 1 : q_, &B, q_, A);                    \
    }                                                                     \
    if ((qsize) & 1) put(A, (qsize) - 1);                                 \
  } while (0)

// Element bounds out of a patch kernel: staged in LDS over 4 consecutive tracers and written as whole 128-byte lines (the layout keeps
// the 4 levels of 4 consecutive tracers of an (element, chunk) in one aligned line: mm_idx).  One 8-byte store per (element, level,
// tracer) -- what these kernels did before -- leaves every line to be filled by four tracers at four different times, and the memory
// system pays for a partial line as for a whole one and more: 1.6 ms of k_lap1's 12.8 and 1.65 of k_dss_patch's 16.8 for 6 % of their
// bytes (profiles/r03_ab_bounds_lines.txt).
template <int PSZ> struct BoundsStage { double v[2][2][PSZ][4][CL]; };   // [group parity][min|max][slot][tracer & 3][level]: 8 KB (4x4 patch)
template <int PSZ>
__device__ __forceinline__ void stage_init(BoundsStage<PSZ>& S) {   // (the pad tracers of the last group are written too: defined values)
  double2* p = reinterpret_cast<double2*>(&S.v[0][0][0][0][0]);
  p[threadIdx.x] = make_double2(0., 0.); p[threadIdx.x + PSZ * 16] = make_double2(0., 0.);   // PSZ*16 lanes x 2 x 16 B = 2*2*PSZ*16*8
}
template <int PSZ>
__device__ __forceinline__ void stage_bounds(BoundsStage<PSZ>& S, int q, double mn, double mx) {   // lane = (slot, level, row): rows 0 and 1 write
  const int t = threadIdx.x, sl = t >> 4, kk = (t >> 2) & (CL - 1), j = t & 3;
  if (j == 0) S.v[(q >> 2) & 1][0][sl][q & 3][kk] = mn;
  if (j == 1) S.v[(q >> 2) & 1][1][sl][q & 3][kk] = mx;
}
// the two lines per slot of tracers 4g .. 4g+3: one 16-byte store per lane; call behind a workgroup barrier that follows their stage_bounds
template <int PSZ>
__device__ __forceinline__ void flush_bounds(const BoundsStage<PSZ>& S, int g, const int* __restrict__ pslots, int patch, int kchunk, int qsize,
                                             double* __restrict__ mn_out, double* __restrict__ mx_out) {
  const int t = threadIdx.x, a = t / (PSZ * 8), sl = (t >> 3) % PSZ, piece = t & 7;
  const int el = pslots[patch * PSZ + sl];
  if (el < 0) return;   // hole
  const double2 v = *reinterpret_cast<const double2*>(&S.v[g & 1][a][sl][piece >> 1][(piece & 1) * 2]);
  double* dst = (a ? mx_out : mn_out) + (((size_t)el * NCHUNK + kchunk) * mm_qpad(qsize) + g * 4) * CL + piece * 2;
  *reinterpret_cast<double2*>(dst) = v;
}
// the reference's order per point: edge contributions (S, E, N, W) first, then the corner; an absent one adds +0.0
template <int PSZ>
__device__ __forceinline__ void gather_sum(const RowGather& R, const PatchLds<PSZ>& L, int b, int j, const double v[4], double out[4]) {
  const bool edge = (j == 0) | (j == 3);
  const char* base = reinterpret_cast<const char*>(&L.v[b][0][0]);
  const double f0 = *reinterpret_cast<const double*>(base + R.lr[0]), f1 = *reinterpret_cast<const double*>(base + R.lr[1]),
               f2 = *reinterpret_cast<const double*>(base + R.lr[2]), f3 = *reinterpret_cast<const double*>(base + R.lr[3]),
               f4 = *reinterpret_cast<const double*>(base + R.lr[4]);
  // quad hand-over: lanes 0 and 3 receive what lanes 1 and 2 fetched for them: quad_perm [1,1,2,2]
  const double d2 = dppq<0xA5>(f2), d3 = dppq<0xA5>(f3), d4 = dppq<0xA5>(f4);
  // "this contribution belongs to my row" as an exact 1.0/0.0 factor folded into FMAs (fma(1,x,t) = t + x rounds like the
  // addition, fma(0,x,t) = t; every x is a finite field value) instead of a 64-bit select around every add
  const double em = edge ? 1.0 : 0.0, nm = edge ? 0.0 : 1.0;
  double t0 = v[0] + f0; t0 = fma(em, f1, t0); t0 = fma(em, d2, t0);
  const double t1 = fma(em, d3, v[1]);
  const double t2 = fma(em, d4, v[2]);
  double t3 = fma(nm, f1, v[3]); t3 = fma(em, f2, t3); t3 = fma(em, f3, t3); t3 = fma(em, f4, t3);
  out[0] = R.rs[0] * t0; out[1] = R.rs[1] * t1; out[2] = R.rs[2] * t2; out[3] = R.rs[3] * t3;
}
// DSS of the extra plane on read, before the tracer loop of a DSS-on-read kernel (LDS buffer 1: the loop starts on buffer 0, and
// its barrier of tracer 0 separates these reads from the publish of tracer 1 into buffer 1).  All lanes of the block call it.
template <int PSZ>
__device__ __forceinline__ void gather_var_plane(RowGather& R, PatchLds<PSZ>& L, const GatherArgs& A, const double* __restrict__ src, int plane,
                                                 int j, int k, double x[4]) {
  GatherRaw raw;
  double own[4];
  gather_issue(R, A, src, plane, raw);
  gather_publish(R, L, 1, k, raw, own);
  lds_barrier();
  gather_sum(R, L, 1, j, own, x);
}

// The slab kernels' output into the scratch layout: the lane's 4 values (points i of row j at level k) leave as two 16-byte
// stores shared with the lane that holds the other level of the pair (even level: points 0,2 for both levels; odd: 1,3),
// instead of four 8-byte stores.  All lanes must call it (the swizzle needs both lanes of a pair); `live` gates the stores.
struct RowStore { unsigned o0, o1; bool s0, s1; };   // plane-relative offsets (doubles) of the lane's two stores, and whether each is made
// pexp (k_lap1<1> only): lines of the slot that hold points some other patch or rank reads (tse_api.hip: slot_perm puts every exported
// edge into the slot's first lines); the store of a point beyond them is dropped -- nobody reads the first Laplacian of such a point
// from memory (k_advance<2,3> forms the Laplacian of its own slots itself), so k_lap1 writes a quarter of the field instead of all of it
__device__ __forceinline__ RowStore row_store_setup(Scr S, const unsigned long long* __restrict__ pperm, int slot, int j, int k,
                                                    const unsigned char* __restrict__ pexp = nullptr) {
  const int p0 = j * 4 + ((k & 1) ? 1 : 0), p1 = p0 + 2;   // (points {0,2} / {1,3} of the row: as the loads, gather_setup)
  const unsigned base = (unsigned)(k / CL) * S.cse + (unsigned)slot * 16;
  const unsigned long long perm = pperm[slot];
  const int lim = pexp ? 4 * (int)pexp[slot] : 16;   // positions below `lim` are stored (whole lines)
  return RowStore{(base + ppos(perm, p0)) * CL + ((k & (CL - 1)) & ~1), (base + ppos(perm, p1)) * CL + ((k & (CL - 1)) & ~1),
                  ppos(perm, p0) < lim, ppos(perm, p1) < lim};
}
__device__ __forceinline__ void store_row_pair(double* __restrict__ plane /* &T[q][0] */, const RowStore& R, int k, bool live, const double v[4]) {
  const bool odd = k & 1;
  const double r0 = swz_xor4(odd ? v[0] : v[1]), r1 = swz_xor4(odd ? v[2] : v[3]);
  if (live) {
    if (R.s0) *reinterpret_cast<double2*>(plane + R.o0) = odd ? make_double2(r0, v[1]) : make_double2(v[0], r0);
    if (R.s1) *reinterpret_cast<double2*>(plane + R.o1) = odd ? make_double2(r1, v[3]) : make_double2(v[2], r1);
  }
}
// received halo -> the halo columns of every tracer plane of a scratch field (only before a DSS-on-read consumer)
// thread = (one chunk of one plane, halo column), columns fastest: a chunk's halo columns are contiguous in the scratch layout
// (32 bytes each), so the stores are whole lines; the loads are 32-byte pieces of the [col][layer] buffer
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_unpack_halo(int ncol, int ng /* layers / CL: (tracer or extra-variable plane, chunk) pairs */, const double* __restrict__ recvbuf,
                              int nlyr_halo, double* __restrict__ dst, Scr S, unsigned halo0 /* entry index of halo column 0 within a chunk */) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ncol * (unsigned)ng) return;
  const unsigned g = t / (unsigned)ncol, col = t - g * (unsigned)ncol, q = g / NCHUNK, kc = g - q * NCHUNK;
  const double* in = recvbuf + (size_t)col * nlyr_halo + (size_t)g * CL;   // layer q*NLEV + kc*CL
  double* out = dst + (size_t)q * S.tps + ((size_t)kc * S.cse + halo0 + col) * CL;
  static_assert(CL == 4, "two 16-byte moves per chunk entry");
  const double2 a = *reinterpret_cast<const double2*>(in), b = *reinterpret_cast<const double2*>(in + 2);
  *reinterpret_cast<double2*>(out) = a; *reinterpret_cast<double2*>(out + 2) = b;
}
// received element bounds (compact min/max exchange: [col][min set | max set]) -> behind the local elements of qmin / qmax, where the
// stage-3 kernel's element ring finds them (entry nelemd + col)
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_unpack_minmax(int ncol, int m /* even */, const double* __restrict__ recvbuf, double* __restrict__ qmin_tail, double* __restrict__ qmax_tail) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, h = (unsigned)m / 2;
  if (t >= (unsigned)ncol * h) return;
  const unsigned col = t / h, l = 2 * (t - col * h);
  const double* in = recvbuf + (size_t)col * 2 * m + l;
  *reinterpret_cast<double2*>(qmin_tail + (size_t)col * m + l) = *reinterpret_cast<const double2*>(in);
  *reinterpret_cast<double2*>(qmax_tail + (size_t)col * m + l) = *reinterpret_cast<const double2*>(in + m);
}
// the all-zero slot of every chunk of every plane (after a caller used the scratch field as a plain buffer)
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_zero_slot(int qsize, double* __restrict__ dst, Scr S, unsigned zero0 /* entry index of the zero slot */) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= qsize * NCHUNK * 16 * CL) return;
  const int w = t % (16 * CL), c = (t / (16 * CL)) % NCHUNK, q = t / (16 * CL * NCHUNK);
  dst[(size_t)q * S.tps + ((size_t)c * S.cse + zero0) * CL + w] = 0.0;
}

// ---------------------------------------------------------------------------------------------------
// the fused euler_step advance (prim_advection_mod.F90:834-902) for one RK stage:
//   Vstar = vn0/dp, dp_star = dp - dt*divdp, Qtens = Qdp - dt*div(Vstar*Qdp) [+ biharmonic], limiter8, *spheremp.
// RHS = rhs_multiplier.  RHS==1 folds in the local min/max update (:781-793).  RHS==2 folds in the second
// Laplacian of the biharmonic and its scaling (viscosity_mod.F90:419-423 + prim_advection_mod.F90:813-826);
// `lap` then holds rspheremp*DSS(laplace_sphere_wk(Q)).
// GIN: DSS on read (whole-step path; block = patch x chunk, see above).  1: the tracer input is rspheremp*DSS of the previous
// stage's pre-DSS scratch (passed in Qn0, scratch layout); 3 (RHS == 2): in addition the DSS'd first Laplacian is assembled here --
// its own slots' values are FORMED here from the stage-2 tracers just assembled (OWNLAP below), only the halo ring comes from `lap`
// (the exported lines k_lap1<1> stored) -- and so are the element bounds.  Stage 3 never needs the DSS'd stage-2 tracers in memory.
// Register tiers (512 VGPRs per SIMD lane): 128 -> 4 waves, 168 -> 3, 256 -> 2.  Forcing the stage-2 DSS-on-read kernel
// (170) into the 3-wave tier with amdgpu_waves_per_eu costs 2 spills and gains nothing measurable.
template <int RHS, int GIN = 0, bool DB = (GIN != 0), int PSZ = 16 /* block shape (GIN != 0): Patch<PSZ> */>
#ifndef TSE_ADV2_WPE
#define TSE_ADV2_WPE 1   // A/B: minimum waves per SIMD asked of the compiler for k_advance<2,3,.,16> (3 = the 168-register tier)
#endif
__global__ __launch_bounds__(GIN ? Patch<PSZ>::THREADS : FLAT_THREADS, (GIN == 3 && PSZ == 16) ? TSE_ADV2_WPE : 1) void k_advance(int nelemd, Dvv_t D, GeoPtrs G, int qsize, double dt, double nu_q,
                                                          const double* __restrict__ Qn0, const double* __restrict__ lap,
                                                          double* __restrict__ Tout, const double* __restrict__ vn0,
                                                          const double* __restrict__ dp, const double* __restrict__ divdp,
                                                          const double* __restrict__ divdp_proj, double* __restrict__ qmin,
                                                          double* __restrict__ qmax, const double* __restrict__ dp0, GatherArgs GA) {
  static_assert(GIN == 0 || GIN == 1 || (GIN == 3 && RHS == 2), "plain inputs, gathered tracers, or gathered tracers and Laplacian");
  static_assert(GIN != 0 || PSZ == 16, "the plain kernels have no block shape");
  __shared__ PatchLds<PSZ> lds_[GIN == 3 ? 2 : 1];   // (unused and removed by the compiler when GIN == 0)
  __shared__ BoundsLds<PSZ> bnd_;                    // (GIN == 3 only)
  // Stage 3 forms the first Laplacian and the element bounds of its patch's OWN slots itself (OWNLAP; below): k_lap1 only has to
  // leave what other patches and ranks read
  constexpr bool OWNLAP = GIN == 3;
  constexpr int BND_ENT = BoundsLds<PSZ>::ENT, LDS_ZERO = Patch<PSZ>::LDS_ZERO;
  static_assert(BND_ENT * 4 <= Patch<PSZ>::THREADS, "one 16-byte load per lane fills the bounds image");
  constexpr bool NBR = GIN == 3;                // the limiter bounds are the min/max over the element and its neighbours of qmin/qmax, formed here
  int e, k, kc, slot;
  const int j = threadIdx.x & 3;
  PatchId pid{};
  if (GIN) {
    pid = patch_slab<PSZ>(GA);
    if (!pid.any) return;   // whole block (uniform): before any barrier
    e = pid.e; kc = pid.k; k = pid.live ? pid.k : NLEV; slot = pid.slot;
  } else {
    const SlabId sid = flat_slab(GA.nwork, GA.order);
    e = sid.e; kc = sid.k; k = sid.live ? sid.k : NLEV; slot = GA.slot_of[e];
  }
  const RowStore RS = row_store_setup(GA.S, GA.pperm, slot, j, kc);
  RowGather RG;
  if (GIN) gather_setup<PSZ>(RG, lds_[0], GA, pid);
  double vdss[4] = {0, 0, 0, 0};   // the previous stage's extra variable, DSS'd on read (stage 2: divdp_proj, which this stage's dp needs)
  if (GIN && GA.var_out) {
    gather_var_plane(RG, lds_[0], GA, Qn0, qsize, j, kc, vdss);
    if (k < NLEV) store4(GA.var_out + ((size_t)e * GA.var_out_lev + k) * 16 + j * 4, vdss);
  }
  // per-(e,k,row) constants, computed once and reused for every tracer:
  //   a1,a2 : metdet*Dinv*Vstar  (contravariant flux per unit Qdp: gv = a*Qdp, derivative_mod.F90:2386-2391)
  //   rm    : dt*rmetdet*rrearth ; dps = dp_star ; rdps = 1/dp_star ; c = spheremp*dp_star ; rdpk = 1/dp (RHS 1)
  double a1[4], a2[4], rm[4], dps[4], rdps[4], c[4], spm[4], rdpk[4], dcol[4];
  LapGeo L;  // only used by the Laplacian of stage 3 (RHS == 2)
  {
    RowGeo g;
    load_row_geo(g, G.dvv, G.Dinv, G.metdet, G.rmetdet, G.spheremp, e, j);
    const size_t lo = ((size_t)e * NLEV + kc) * 16 + j * 4;
    double dpk[4], vs1[4], vs2[4], t0[4] = {0, 0, 0, 0}, t1[4];
    const bool mkdiv = RHS == 0 && GIN == 0 && GA.divdp_out;   // divdp = divdp_proj = divergence_sphere(vn0) (prim_advection_mod.F90:614-623)
    load4(dp + lo, dpk);
    load4(vn0 + (((size_t)e * NLEV + kc) * 2 + 0) * 16 + j * 4, vs1);
    load4(vn0 + (((size_t)e * NLEV + kc) * 2 + 1) * 16 + j * 4, vs2);
    // Whole-step path: stage 1 forms divdp for its slab and stores it (derived%divdp is an output of the step); stage 3 (GIN == 3) forms
    // it again from the vn0 it holds for Vstar anyway instead of reading the field back -- the same routine on the same inputs, the same
    // bits, 0.8 GB less to read at ne120.  (Stage 2 keeps the load: the divergence in its prologue costs k_advance<1,1> its third wave,
    // 164 -> 174 registers.)
    if (mkdiv || GIN == 3) {
      divergence_sphere_row(D, g, vs1, vs2, t1);
      if (mkdiv && k < NLEV) store4(GA.divdp_out + lo, t1);
    } else load4(divdp + lo, t1);
    if (GIN && GA.var_out && RHS == 1) {   // divdp_proj = what was just assembled (the array is being written by this launch)
#pragma unroll
      for (int i = 0; i < 4; i++) t0[i] = vdss[i];
    } else if (RHS != 0) load4(divdp_proj + lo, t0);
    if (GA.var_in) {   // this stage's extra variable, weighted, as plane qsize of the output
      double vin[4], w[4];
      if (mkdiv) {
#pragma unroll
        for (int i = 0; i < 4; i++) vin[i] = t1[i];   // (stage 1's extra variable is divdp_proj = the divergence just formed)
      } else load4(GA.var_in + ((size_t)e * GA.var_in_lev + kc) * 16 + j * 4, vin);
#pragma unroll
      for (int i = 0; i < 4; i++) w[i] = g.spheremp[i] * vin[i];
      store_row_pair(Tout + (size_t)qsize * GA.S.tps, RS, kc, k < NLEV, w);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      dpk[i] = dp_stage(dpk[i], RHS * dt, t0[i]);   // (shared with k_lap1: the two must agree on dp of stage 3 to the bit)
      dps[i] = dpk[i] - dt * t1[i];
      rdps[i] = 1.0 / dps[i];
      rdpk[i] = 1.0 / dpk[i];
      double u1 = vs1[i] * rdpk[i], u2 = vs2[i] * rdpk[i];   // Vstar = vn0/dp
      a1[i] = g.metdet[i] * (g.Di11[i] * u1 + g.Di12[i] * u2);
      a2[i] = g.metdet[i] * (g.Di21[i] * u1 + g.Di22[i] * u2);
      rm[i] = dt * (g.rmetdet[i] * RREARTH);
      spm[i] = g.spheremp[i];
      c[i] = spm[i] * dps[i];
      dcol[i] = g.dcol[i];
    }
    if (RHS == 2) make_lap_geo(L, g);
  }
  // NBR: per tracer the block loads the bounds of its patch's elements and of the element ring (lane = 16 bytes: entry, min|max,
  // level pair) into LDS with the tracer's other loads; after the barrier a lane reads its element's and its 8 neighbours' values
  unsigned bsrc = 0, bdst = 0, bnb[3] = {0, 0, 0};
  const double* bbase = qmin;
  if (NBR) {
    const int t = threadIdx.x, u = min(t >> 2, BND_ENT - 1), w = (t >> 1) & 1, h = t & 1;   // lanes beyond the image repeat its last entry
    // (the bounds of the patch's own slots are formed in this kernel: the lanes that would load them repeat the first ring entry --
    // same address, same LDS word, same value)
    const int ul = u < PSZ ? PSZ : u;
    int el = GA.pering[pid.patch * NER + (ul - PSZ)];
    if (el < 0) el = pid.e;   // hole
    bsrc = (unsigned)((((size_t)el * NCHUNK + kc / CL) * mm_qpad(qsize)) * CL + h * 2);   // + q*CL: entry index in qmin / qmax (< 2^32: the arrays are < 32 GB)
    bbase = w ? qmax : qmin;
    bdst = (unsigned)(((ul * 2 + w) * CL + h * 2) * 8);
    // the 9 entries of a slab (its element, then the 8 neighbours) are shared out over the quad: row j takes entries j, j+4 (and 8)
    const int sl = pid.live ? pid.tslot - pid.patch * PSZ : 0, kk = kc & (CL - 1);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int d = j + 4 * i;   // 0: the element itself; d >= 1: neighbour d-1
      int n = sl;
      if (d >= 1 && d < 9) { const int v = GA.pnb[((size_t)pid.patch * PSZ + sl) * 8 + (d - 1)]; if (v != 255) n = v; }
      bnb[i] = (unsigned)((n * 2 * CL + kk) * 8);
    }
  }
  const double sumc = quad_sum(((c[0] + c[1]) + c[2]) + c[3]);
  double visc[4];
#pragma unroll
  for (int i = 0; i < 4; i++) visc[i] = RHS == 2 ? (-3.0 * dt * nu_q * dp0[kc]) / spm[i] : 0.0;  // -rhs_viss*dt*nu_q*dp0/spheremp

  // Memory schedule of one tracer step (gfx950: loads and stores share one counter and return out of order between the
  // two kinds, so a wait for loads also drains every store issued before it; and a store's data registers may not be
  // overwritten until it has completed):
  //   wait for this tracer's inputs -> issue the PREVIOUS tracer's stores -> issue the NEXT tracer's loads -> compute.
  // With DB the results go to a second register set (A/B alternate): the stores issued at the top of a step then have the
  // whole step to drain and nothing waits for them (worth its 12 registers where few waves fit: the DSS-on-read kernels).
  struct Out { double x[4], mn, mx; bool ch; };   // ch: the bounds differ from what qmin/qmax already hold
  auto put = [&](const Out& o, int q) {
    // pre-DSS output in the scratch layout (chunks of 4 levels, slots, points perimeter first): what the next stage's blocks
    // load as their own points and as their halo ring
    store_row_pair(Tout + (size_t)q * GA.S.tps, RS, kc, k < NLEV, o.x);
    // Whole-step path (GIN != 0): nothing reads the bounds after stage 2 or stage 3 -- stage 3 recomputes the element min/max
    // (prim_advection_mod.F90:796-809: qmin = minval(...), not min(qmin, ...)), the next step starts from fresh ones (:765-779) -- so only
    // the plain kernels (stage 1 of the whole-step path, whose relaxed bounds stage 2 reuses, :781-793; every stage of the per-stage
    // API, where qmin/qmax are visible state) write them back
    if (GIN == 0 && k < NLEV && j == 0 && o.ch) { const size_t m = mm_idx(e, q, k, qsize); qmin[m] = o.mn; qmax[m] = o.mx; }
  };
  GatherRaw graw;
  double2 lring = make_double2(0., 0.);                  // OWNLAP: the lane's ring load of the first Laplacian
  double2 braw = make_double2(0., 0.);                   // NBR: the lane's piece of the bounds image                                 // raw own / ring loads of the gathered input(s) (DSS on read)
  double qnx[4] = {0, 0, 0, 0}, lsx[4] = {0, 0, 0, 0}, minx = 0.0, maxx = 0.0;   // plainly loaded inputs of the next tracer
  auto fetch = [&](int q) {   // loads only
    const size_t so = (((size_t)e * qsize + q) * NLEV + kc) * 16 + j * 4, mi = mm_idx(e, q, kc, qsize);
    if (GIN) gather_issue(RG, GA, Qn0, q, graw);              // GIN == 3: graw <- Qn0 (tracers), graw2 <- lap
    if (OWNLAP) {   // of the first Laplacian only the patch's halo ring comes from memory (what k_lap1 left of it: the exported lines)
      asm volatile("" : "+v"(RG.ring));
      lring = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(lap + (size_t)q * GA.S.tps) + RG.ring);
    }
    if (GIN == 0) load4(Qn0 + so, qnx);
    if (RHS == 2 && GIN != 3) load4(lap + so, lsx);
    if (NBR) braw = *reinterpret_cast<const double2*>(bbase + ((size_t)bsrc + (size_t)q * CL));
    else { minx = qmin[mi]; maxx = qmax[mi]; }
  };
  if (GIN == 3 && threadIdx.x < 2 * CL) lds_[GIN == 3 ? 1 : 0].v[threadIdx.x / CL][LDS_ZERO][threadIdx.x % CL] = 0.0;
  fetch(0);
  // Memory schedule with DSS on read: wait for this tracer's own/ring loads -> publish them in LDS -> issue the previous
  // tracer's stores and the next tracer's loads -> workgroup barrier -> neighbour values from LDS -> compute.
  auto step = [&](int q, const Out* prev, int qprev, Out& cur) {
    double qn[4], ls[4] = {0, 0, 0, 0}, own[4], own2[4], minp = minx, maxp = maxx;
    if (GIN) gather_publish(RG, lds_[0], q & 1, kc, graw, own);
    if (OWNLAP) {
      *reinterpret_cast<double2*>(reinterpret_cast<char*>(&lds_[GIN == 3 ? 1 : 0].v[q & 1][0][0]) + RG.lwr) = lring;
      asm volatile("" : : : "memory");   // (written before the next tracer's load reuses lring)
    }
    if (NBR) {
      *reinterpret_cast<double2*>(reinterpret_cast<char*>(&bnd_.v[q & 1][0][0][0]) + bdst) = braw;
      asm volatile("" : : : "memory");   // (written before the next tracer's load reuses braw)
    }
    if (GIN == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) qn[i] = qnx[i];
      asm volatile("" : "+v"(qn[0]), "+v"(qn[1]), "+v"(qn[2]), "+v"(qn[3]) : : "memory");   // the wait belongs here, not below
    }
    if (RHS == 2 && GIN != 3) {
#pragma unroll
      for (int i = 0; i < 4; i++) ls[i] = lsx[i];
      asm volatile("" : "+v"(ls[0]), "+v"(ls[1]), "+v"(ls[2]), "+v"(ls[3]) : : "memory");
    }
    asm volatile("" : "+v"(minp), "+v"(maxp) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (prev) put(*prev, qprev);
    fetch(q + 1 < qsize ? q + 1 : q);   // branch-free: the last step re-reads its own tracer
    __builtin_amdgcn_sched_barrier(0);
    if (GIN) {
      lds_barrier();
      gather_sum(RG, lds_[0], q & 1, j, own, qn);
      if (OWNLAP) {
        // What k_lap1 does for a slab (prim_advection_mod.F90:750-761,796-809 + viscosity_mod.F90:378-389), for the patch's own slots:
        // Q = Qdp/dp, element min/max, first weak Laplacian -- published in the second image / the bounds image, whose ring parts came
        // from memory.  Same routines on the same assembled tracers as in k_lap1 (lap_q_of, laplace_lean_row: no contraction left to
        // the compiler), so a point's Laplacian is the same bits whether its own block formed it or a neighbour read it from T.
        double x1[4];
        lap_q_of(qn, rdpk, x1);
        const double emn = quad_min(fmin(fmin(x1[0], x1[1]), fmin(x1[2], x1[3]))), emx = quad_max(fmax(fmax(x1[0], x1[1]), fmax(x1[2], x1[3])));
        laplace_lean_row(D, L, x1, own2);
        const int sl = threadIdx.x >> 4, kk = kc & (CL - 1);
        char* im = reinterpret_cast<char*>(&lds_[GIN == 3 ? 1 : 0].v[q & 1][0][0]);
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<double*>(im + (lds_own_entry(sl, j * 4 + i) * CL + kk) * 8) = own2[i];
        if (j == 0) { bnd_.v[q & 1][sl][0][kk] = emn; bnd_.v[q & 1][sl][1][kk] = emx; }
        lds_barrier();
        gather_sum(RG, lds_[GIN == 3 ? 1 : 0], q & 1, j, own2, ls);
      }
    }
    if (NBR) {   // viscosity_mod.F90:429-432 on the element bounds k_lap1 left in qmin/qmax
      const char* b = reinterpret_cast<const char*>(&bnd_.v[q & 1][0][0][0]);
      double mn = *reinterpret_cast<const double*>(b + bnb[0]), mx = *reinterpret_cast<const double*>(b + bnb[0] + CL * 8);
#pragma unroll
      for (int i = 1; i < 3; i++) {
        mn = fmin(mn, *reinterpret_cast<const double*>(b + bnb[i]));
        mx = fmax(mx, *reinterpret_cast<const double*>(b + bnb[i] + CL * 8));
      }
      minp = quad_min(mn); maxp = quad_max(mx);
    }
    double bih[4] = {0, 0, 0, 0};
    if (RHS == 2) {   // ls = rspheremp*DSS(first Laplacian): second Laplacian and the biharmonic scaling
      laplace_lean_row(D, L, ls, bih);
    }
    double gv1[4], gv2[4], x[4], dx[4], dy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { gv1[i] = a1[i] * qn[i]; gv2[i] = a2[i] * qn[i]; }
    // d/dx in-register, d/dy across the quad (deriv_xy with the lane's dcol)
#ifdef TSE_NO_CONTRACTION   // A/B build (tools/ab): the contractions replaced by a copy -- an upper bound on what a faster contraction could buy
#pragma unroll
    for (int i = 0; i < 4; i++) { dx[i] = gv1[i]; dy[i] = gv2[i]; }
#else
#pragma unroll
    for (int l = 0; l < 4; l++) {
      double sm = 0.0;
#pragma unroll
      for (int i = 0; i < 4; i++) sm = sm + D.d[l * 4 + i] * gv1[i];
      dx[l] = sm;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) dy[i] = quad_matvec(dcol, gv2[i]);
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = qn[i] - rm[i] * (dx[i] + dy[i]);   // Qtens = Qdp - dt*div
    bool changed = false;
    if (RHS == 1) {
      double q0 = qn[0] * rdpk[0], q1 = qn[1] * rdpk[1], q2 = qn[2] * rdpk[2], q3 = qn[3] * rdpk[3];
      const double lmn = quad_min(fmin(fmin(q0, q1), fmin(q2, q3))), lmx = quad_max(fmax(fmax(q0, q1), fmax(q2, q3)));
      changed = (lmn < minp) | (lmx > maxp);
      minp = fmin(minp, lmn);
      maxp = fmax(maxp, lmx);
    }
    if (RHS == 2) {
#pragma unroll
      for (int i = 0; i < 4; i++) x[i] = fma(visc[i], bih[i], x[i]);   // -rhs_viss*dt*nu_q*dp0*Qtens_biharmonic/spheremp (prim_advection_mod.F90:813-826)
    }
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = x[i] * rdps[i];
    changed |= limiter8_quad(x, c, sumc, minp, maxp);
#pragma unroll
    for (int i = 0; i < 4; i++) cur.x[i] = c[i] * x[i];   // spheremp * (x*dp_star)
    cur.mn = minp; cur.mx = maxp; cur.ch = changed;   // unchanged bounds are not written back (3.5 GB per launch)
  };
  Out A, B;
  if (DB) {
    TSE_TRACER_PAIRS(step, put, qsize, A, B);
  } else {   // plenty of waves (4 per SIMD): store at the end of the step, 12 registers less (one copy of the body)
    for (int q = 0; q < qsize; q++) { step(q, nullptr, 0, A); put(A, q); }
  }
}

// ---------------------------------------------------------------------------------------------------
// stage-3 prologue (prim_advection_mod.F90:750-761,796-809 + viscosity_mod.F90:378-389):
// Q = Qdp/dp, element min/max, first weak Laplacian (pre-DSS) -> Bout
// GIN == 1 (whole-step path; block = patch x chunk): Qn0 is the stage-2 pre-DSS scratch; the DSS'd Qdp is assembled on read
// and not stored (k_advance<2,3> of stage 3 assembles it again itself).
template <int GIN = 0, int PSZ = 16>
__global__ __launch_bounds__(GIN ? Patch<PSZ>::THREADS : FLAT_THREADS) void k_lap1(int nelemd, Dvv_t D, GeoPtrs G, int qsize, double rdt,
                                                       const double* __restrict__ Qn0, double* __restrict__ Bout,
                                                       const double* __restrict__ dp, const double* __restrict__ divdp_proj,
                                                       double* __restrict__ qmin, double* __restrict__ qmax, GatherArgs GA) {
  __shared__ PatchLds<PSZ> lds_;
  __shared__ BoundsStage<PSZ> stg_;   // (GIN only)
  int e, k, kc, slot;
  const int j = threadIdx.x & 3;
  PatchId pid{};
  if (GIN) {
    pid = patch_slab<PSZ>(GA);
    if (!pid.any) return;   // whole block (uniform): before any barrier
    e = pid.e; kc = pid.k; k = pid.live ? pid.k : NLEV; slot = pid.slot;
  } else {
    const SlabId sid = flat_slab(GA.nwork, GA.order);
    e = sid.e; kc = sid.k; k = sid.live ? sid.k : NLEV; slot = GA.slot_of[e];
  }
  const RowStore RS = row_store_setup(GA.S, GA.pperm, slot, j, kc, GIN ? GA.pexp : nullptr);
  LapGeo L;
  {
    RowGeo g;
    load_row_geo(g, G.dvv, G.Dinv, G.metdet, G.rmetdet, G.spheremp, e, j);
    make_lap_geo(L, g);
  }
  const size_t lo = ((size_t)e * NLEV + kc) * 16 + j * 4;
  double dpk[4], dv[4];
  load4(dp + lo, dpk); load4(divdp_proj + lo, dv);
#pragma unroll
  for (int i = 0; i < 4; i++) dpk[i] = 1.0 / dp_stage(dpk[i], rdt, dv[i]);   // (the same routine as in k_advance<2,3>: tse_device.h)
  RowGather RG;
  GatherRaw graw;
  if (GIN) {
    gather_setup<PSZ>(RG, lds_, GA, pid);
    if (GA.var_out) {   // the previous stage's extra variable (stage 3: eta_dot_dpdn), DSS'd on read
      double vdss[4];
      gather_var_plane(RG, lds_, GA, Qn0, qsize, j, kc, vdss);
      if (k < NLEV) store4(GA.var_out + ((size_t)e * GA.var_out_lev + k) * 16 + j * 4, vdss);
    }
    gather_issue(RG, GA, Qn0, 0, graw);
  }
  if (!GIN) {
    for (int q = 0; q < qsize; q++) {
      const size_t so = (((size_t)e * qsize + q) * NLEV + kc) * 16 + j * 4;
      double x[4], l1[4];
      load4(Qn0 + so, x);
      lap_q_of(x, dpk, x);
      double mn = quad_min(fmin(fmin(x[0], x[1]), fmin(x[2], x[3])));
      double mx = quad_max(fmax(fmax(x[0], x[1]), fmax(x[2], x[3])));
      laplace_lean_row(D, L, x, l1);
      store_row_pair(Bout + (size_t)q * GA.S.tps, RS, kc, k < NLEV, l1);   // scratch layout, as T
      if (k < NLEV && j == 0) { qmin[mm_idx(e, q, k, qsize)] = mn; qmax[mm_idx(e, q, k, qsize)] = mx; }
    }
    return;
  }
  // DSS on read.  Memory schedule of one tracer step (gfx950: one counter for loads and stores, which return out of order
  // between the two kinds, so a wait for loads also drains every store issued before it; and a store's data registers may
  // not be overwritten until it has completed):
  //   wait for this tracer's own/ring loads -> publish in LDS -> issue the PREVIOUS tracer's stores -> issue the NEXT tracer's
  //   loads -> workgroup barrier -> neighbour values from LDS, sum -> compute.
  // The results go to a second register set (A/B alternate), so that the stores issued at the top of a step drain during
  // the whole step and nothing waits for them.
  struct Out { double l[4], mn, mx; };
  auto put = [&](const Out& o, int q) {   // stores of tracer q
    store_row_pair(Bout + (size_t)q * GA.S.tps, RS, kc, k < NLEV, o.l);
  };
  stage_init(stg_);   // (ordered before its first use by the first tracer's barrier)
  auto step = [&](int q, const Out* prev, int qprev, Out& cur) {
    double x[4], own[4];
    gather_publish(RG, lds_, q & 1, kc, graw, own);
    __builtin_amdgcn_sched_barrier(0);
    if (prev) put(*prev, qprev);
    gather_issue(RG, GA, Qn0, q + 1 < qsize ? q + 1 : q, graw);   // branch-free: the last step re-reads its own tracer
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    if ((q & 3) == 0 && q) flush_bounds(stg_, (q >> 2) - 1, GA.pslots, pid.patch, kc / CL, qsize, qmin, qmax);   // the element bounds of the 4 tracers before
    gather_sum(RG, lds_, q & 1, j, own, x);
    lap_q_of(x, dpk, x);
    cur.mn = quad_min(fmin(fmin(x[0], x[1]), fmin(x[2], x[3])));
    cur.mx = quad_max(fmax(fmax(x[0], x[1]), fmax(x[2], x[3])));
    stage_bounds(stg_, q, cur.mn, cur.mx);
    laplace_lean_row(D, L, x, cur.l);
  };
  Out A, B;
  TSE_TRACER_PAIRS(step, put, qsize, A, B);
  lds_barrier();
  flush_bounds(stg_, (qsize - 1) >> 2, GA.pslots, pid.patch, kc / CL, qsize, qmin, qmax);
}

// ---------------------------------------------------------------------------------------------------
// DSS as a gather (edgeVpack + bndry_exchangeV + edgeVunpack, edge_mod.F90:366-511,648-742), fused with the
// inverse mass matrix (prim_advection_mod.F90:929-960) and, for MODE 1, with qdp_time_avg (:645-662).
// Each element adds its neighbours' values in the reference's fixed order (all S, E, N, W, then SW, SE, NE, NW),
// so results do not depend on how elements are distributed over GPUs.
//   tab[e][16][3] = {source element (>=0 local, -1 none, <=-2 remote column -(v+2)), source point}  (level fields; the tracer
//   kernels use the patch tables derived from it: tse_api.hip)
// Tracer-field DSS (k_dss_patch): source in the scratch layout written by k_advance/k_lap1, destination in the standard
// layout dst[e][q][k][p].
// MODE 0: dst = rspheremp * DSS(src)                                   (prim_advection_mod.F90:929-960)
// MODE 1: ... fused with qdp_time_avg: dst = (Qn0 + 2*that)/3         (:645-662)
// Lane mapping of the level-field DSS (k_dss_lvl): work = (XCD range of elements, flattened (element slot, unit)) with UNITS
// lanes per element.  144 lanes per element do not fill whole waves, so the lanes of a block run across element boundaries:
// no idle lanes except in a range's last block, and blocks of 4 waves instead of 5.
constexpr int DSS_FLAT_THREADS = 256;
template <int UNITS>
inline int dss_blocks_per_xcd(int nelemd) { return (((nelemd + 7) >> 3) * UNITS + DSS_FLAT_THREADS - 1) / DSS_FLAT_THREADS; }
struct DssLane { int slot, r, qc; bool live; };
template <int UNITS>
__device__ __forceinline__ DssLane dss_lane(int nelemd) {
  const int S8 = (nelemd + 7) >> 3;                                             // elements per XCD range (as in `order`)
  const int B8 = (S8 * UNITS + DSS_FLAT_THREADS - 1) / DSS_FLAT_THREADS;        // blocks per XCD range and tracer chunk
  const int xcd = blockIdx.x & 7, it = blockIdx.x >> 3;
  const int bi = it % B8, g = bi * DSS_FLAT_THREADS + threadIdx.x, idx = g / UNITS;
  DssLane l;
  l.qc = it / B8; l.r = g - idx * UNITS; l.slot = xcd * S8 + idx; l.live = idx < S8 && l.slot < nelemd;
  return l;
}
// Tracer DSS pass (per-stage API, and the last pass of the whole-step path): block = (patch, chunk) exactly like the
// DSS-on-read slab kernels -- own points and the patch's halo ring are loaded once per tracer, published in LDS, and the
// neighbour contributions are LDS reads; remote contributions sit in the halo columns of the scratch planes (k_unpack_halo).
// mn_out/mx_out (MODE 1 only, may be null): element min/max of Q = Qdp/dp of the field just written, i.e. what the
// next tracer step's first stage would compute with k_qminmax (prim_advection_mod.F90:764-775) -- saves that pass.
template <int MODE, int PSZ = 16>
__global__ __launch_bounds__(Patch<PSZ>::THREADS) void k_dss_patch(int qsize, const double* __restrict__ src, double* __restrict__ dst,
                                                            const double* __restrict__ Qn0, const double* __restrict__ dpnext,
                                                            double* __restrict__ mn_out, double* __restrict__ mx_out, GatherArgs GA) {
  __shared__ PatchLds<PSZ> lds_;
  __shared__ BoundsStage<PSZ> stg_;   // (MODE 1 with mn_out only)
  const PatchId pid = patch_slab<PSZ>(GA);
  if (!pid.any) return;   // whole block (uniform): before any barrier
  const int e = pid.e, kc = pid.k, k = pid.live ? pid.k : NLEV, j = pid.j;
  RowGather RG;
  GatherRaw graw;
  gather_setup<PSZ>(RG, lds_, GA, pid);
  double dn[4] = {1, 1, 1, 1}, q0x[4] = {0, 0, 0, 0};
  if (MODE == 1 && mn_out) {   // 1/dp of the next step's stage 1 (the bounds are Qdp * (1/dp) everywhere: k_qminmax)
    load4(dpnext + ((size_t)e * NLEV + kc) * 16 + j * 4, dn);
#pragma unroll
    for (int i = 0; i < 4; i++) dn[i] = 1.0 / dn[i];
  }
  auto fetch = [&](int q) {
    gather_issue(RG, GA, src, q, graw);
    if (MODE == 1) load4(Qn0 + (((size_t)e * qsize + q) * NLEV + kc) * 16 + j * 4, q0x);
  };
  struct Out { double x[4], mn, mx; };
  auto put = [&](const Out& o, int q) {
    if (k < NLEV) {
      store4(dst + (((size_t)e * qsize + q) * NLEV + k) * 16 + j * 4, o.x);
    }
  };
  const bool emit = MODE == 1 && mn_out;   // (kernel argument: uniform)
  if (emit) stage_init(stg_);
  // wait for this tracer's loads -> publish in LDS -> issue the PREVIOUS tracer's stores and the NEXT tracer's loads ->
  // workgroup barrier -> neighbour values from LDS -> sum (results alternate between two register sets, see k_lap1)
  auto step = [&](int q, const Out* prev, int qprev, Out& cur) {
    double own[4], qa[4], x[4];
    gather_publish(RG, lds_, q & 1, kc, graw, own);
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 4; i++) qa[i] = q0x[i];
      asm volatile("" : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]) : : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if (prev) put(*prev, qprev);
    fetch(q + 1 < qsize ? q + 1 : q);   // branch-free: the last step re-reads its own tracer
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    if (emit && (q & 3) == 0 && q) flush_bounds(stg_, (q >> 2) - 1, GA.pslots, pid.patch, kc / CL, qsize, mn_out, mx_out);   // the bounds of the 4 tracers before
    gather_sum(RG, lds_, q & 1, j, own, x);
#pragma unroll
    for (int i = 0; i < 4; i++) cur.x[i] = MODE == 1 ? (qa[i] + 2 * x[i]) / 3 : x[i];   // (Qdp(n0) + (rkstage-1)*Qdp(np1))/rkstage
    if (emit) {
      double y[4];
#pragma unroll
      for (int i = 0; i < 4; i++) y[i] = cur.x[i] * dn[i];
      cur.mn = quad_min(fmin(fmin(y[0], y[1]), fmin(y[2], y[3])));
      cur.mx = quad_max(fmax(fmax(y[0], y[1]), fmax(y[2], y[3])));
      stage_bounds(stg_, q, cur.mn, cur.mx);
    }
  };
  if (GA.var_out) {   // the last stage's extra variable (omega_p), DSS'd on read
    double vdss[4];
    gather_var_plane(RG, lds_, GA, src, qsize, j, kc, vdss);
    if (k < NLEV) store4(GA.var_out + ((size_t)e * GA.var_out_lev + k) * 16 + j * 4, vdss);
  }
  fetch(0);
  Out A, B;
  TSE_TRACER_PAIRS(step, put, qsize, A, B);
  if (emit) {
    lds_barrier();
    flush_bounds(stg_, (qsize - 1) >> 2, GA.pslots, pid.patch, kc / CL, qsize, mn_out, mx_out);
  }
}

// DSS of one level field (divdp_proj, eta_dot_dpdn(1:nlev), omega_p): dst = rspheremp * DSS(spheremp * src)
// (prim_advection_mod.F90:911-919,943-957), out of place (the neighbours read src).  src/dst carry src_lev/dst_lev levels per
// element (eta_dot_dpdn: nlev+1; the extra level is copied through), so no staging copies are needed and the caller just
// swaps the two buffers.  Lanes flattened over (element slot, level pair, row); all gathers issued before any use.
constexpr int LVL_UNITS = (NLEV / 2) * 4;   // lanes per element: 36 level pairs (k, k+36) x 4 rows
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(DSS_FLAT_THREADS) void k_dss_lvl(int nelemd, const int2* __restrict__ tab, const double* __restrict__ rspheremp,
                                                              const double* __restrict__ spheremp, const double* __restrict__ src, int src_lev,
                                                              double* __restrict__ dst, int dst_lev, const double* __restrict__ recvbuf,
                                                              int nlyr_halo, int lyr0, const int* __restrict__ order) {
  // spheremp*var is a stored (rounded) product in the reference (prim_advection_mod.F90:913-918) and in the whole-step path's extra
  // plane, so the products below must not be contracted into the sums that follow: both routes then leave the same bits
#pragma clang fp contract(off)
  const DssLane ln = dss_lane<LVL_UNITS>(nelemd);
  if (!ln.live) return;
  const int e = order[ln.slot], k = ln.r >> 2, j = ln.r & 3;   // the lane does levels k and k + NLEV/2 (the table lookup is shared)
  constexpr int NS = 8, H = NLEV / 2;
  const int si[NS] = {0, 0, 0, 1, 2, 3, 3, 3}, sc[NS] = {0, 1, 2, 0, 0, 0, 1, 2};
  int2 tt[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) tt[s] = tab[((size_t)e * 16 + j * 4 + si[s]) * 3 + sc[s]];
  double v0[4], v1[4], sm[4], rs[4];
  const double* own = src + ((size_t)e * src_lev + k) * 16 + j * 4;
  load4(own, v0); load4(own + (size_t)H * 16, v1);
  load4(spheremp + (size_t)e * 16 + j * 4, sm);
  load4(rspheremp + (size_t)e * 16 + j * 4, rs);
  // every slot is one unconditional pair of loads (a load inside a divergent branch is waited for in the branch): an empty
  // slot re-reads the lane's own point with weight 0, a remote one reads the halo (already weighted by the sender) with 1
  const double* ap[NS]; const double* wp[NS]; size_t st[NS]; double wc[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int2 t = tt[s];
    ap[s] = own; wp[s] = spheremp + (size_t)e * 16 + j * 4; st[s] = (size_t)H * 16; wc[s] = 0.0;
    if (t.x >= 0) { ap[s] = src + ((size_t)t.x * src_lev + k) * 16 + t.y; wp[s] = spheremp + (size_t)t.x * 16 + t.y; wc[s] = -1.0; }
    else if (t.x <= -2) { ap[s] = recvbuf + (size_t)(-(t.x + 2)) * nlyr_halo + lyr0 + k; st[s] = H; wc[s] = 1.0; }
  }
  double a0[NS], a1[NS], w[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) { a0[s] = ap[s][0]; a1[s] = ap[s][st[s]]; w[s] = wp[s][0]; }
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const double ws = wc[s] < 0.0 ? w[s] : wc[s];   // local: the neighbour's spheremp; remote: 1; empty: 0
    a0[s] = wc[s] == 0.0 ? 0.0 : ws * a0[s];        // (an empty slot adds +0.0 exactly, whatever it read)
    a1[s] = wc[s] == 0.0 ? 0.0 : ws * a1[s];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) { v0[i] = sm[i] * v0[i]; v1[i] = sm[i] * v1[i]; }
  v0[0] = v0[0] + a0[0]; v0[0] = v0[0] + a0[1]; v0[0] = v0[0] + a0[2];
  v0[1] = v0[1] + a0[3];
  v0[2] = v0[2] + a0[4];
  v0[3] = v0[3] + a0[5]; v0[3] = v0[3] + a0[6]; v0[3] = v0[3] + a0[7];
  v1[0] = v1[0] + a1[0]; v1[0] = v1[0] + a1[1]; v1[0] = v1[0] + a1[2];
  v1[1] = v1[1] + a1[3];
  v1[2] = v1[2] + a1[4];
  v1[3] = v1[3] + a1[5]; v1[3] = v1[3] + a1[6]; v1[3] = v1[3] + a1[7];
#pragma unroll
  for (int i = 0; i < 4; i++) { v0[i] = rs[i] * v0[i]; v1[i] = rs[i] * v1[i]; }
  double* out = dst + ((size_t)e * dst_lev + k) * 16 + j * 4;
  store4(out, v0); store4(out + (size_t)H * 16, v1);
  if (k == H - 1 && src_lev > NLEV && dst_lev > NLEV) {   // the interface below the last level is not DSS'd: copy it
    double x[4];
    load4(src + ((size_t)e * src_lev + NLEV) * 16 + j * 4, x);
    store4(dst + ((size_t)e * dst_lev + NLEV) * 16 + j * 4, x);
  }
}

// pack the rank-boundary columns of a [e][nlyr][16] field into sendbuf[col][nlyr_halo] (layer fastest, the
// reference's buf(nlyr,nbuf) layout, edge_mod.F90:150,177-196); send_src[col] = {element, point}
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_pack(int ncol, int nlyr, const int2* __restrict__ send_src, const double* __restrict__ src,
                       const double* __restrict__ scale_in, double* __restrict__ sendbuf, int nlyr_halo, int lyr0,
                       int src_lyr /* layers per element of src (>= nlyr; eta_dot_dpdn carries nlev+1) */) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ncol * (unsigned)nlyr) return;
  const unsigned col = t / (unsigned)nlyr, l = t - col * (unsigned)nlyr;
  const int2 s = send_src[col];
  double a = src[((size_t)s.x * src_lyr + l) * 16 + s.y];
  if (scale_in) a = scale_in[(size_t)s.x * 16 + s.y] * a;
  sendbuf[(size_t)col * nlyr_halo + lyr0 + l] = a;
}
// the same from a scratch field (send_src_s[col] = {slot, position}): thread = (column, chunk of a plane) moves the chunk's
// 4 levels -- 32 contiguous bytes on both sides -- layers fastest, so the stores are whole lines
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_pack_scratch(int ncol, int ng /* layers / CL */, const int2* __restrict__ send_src_s, const double* __restrict__ src,
                               double* __restrict__ sendbuf, int nlyr_halo, Scr S) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (unsigned)ncol * (unsigned)ng) return;
  const unsigned col = t / (unsigned)ng, g = t - col * (unsigned)ng, q = g / NCHUNK, kc = g - q * NCHUNK;
  const int2 sp = send_src_s[col];
  const double* in = src + (size_t)q * S.tps + ((size_t)kc * S.cse + (size_t)sp.x * 16 + sp.y) * CL;
  double* out = sendbuf + (size_t)col * nlyr_halo + (size_t)g * CL;
  const double2 a = *reinterpret_cast<const double2*>(in), b = *reinterpret_cast<const double2*>(in + 2);
  *reinterpret_cast<double2*>(out) = a; *reinterpret_cast<double2*>(out + 2) = b;
}
// element-constant min/max fields packed the way neighbor_minmax does (viscosity_mod.F90:764-774); two entries per thread
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_pack_minmax(int ncol, int m /* even */, const int2* __restrict__ send_src, const double* __restrict__ qmin,
                              const double* __restrict__ qmax, double* __restrict__ sendbuf, int nlyr_halo, int lyr0) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x, h = (unsigned)m / 2;
  if (t >= (unsigned)ncol * h) return;
  const unsigned col = t / h, l = 2 * (t - col * h);
  const int e = send_src[col].x;
  double* out = sendbuf + (size_t)col * nlyr_halo + lyr0 + l;
  *reinterpret_cast<double2*>(out) = *reinterpret_cast<const double2*>(qmin + (size_t)e * m + l);
  *reinterpret_cast<double2*>(out + m) = *reinterpret_cast<const double2*>(qmax + (size_t)e * m + l);
}

// qdp_time_avg alone (prim_advection_mod.F90:645-662), for the stage-by-stage API
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_time_avg(size_t n, int rkstage, const double* __restrict__ Qn0, double* __restrict__ Qnp1) {
  size_t t = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (t >= n) return;
  double2 a = *reinterpret_cast<const double2*>(Qn0 + t), b = *reinterpret_cast<double2*>(Qnp1 + t);
  b.x = (a.x + (rkstage - 1) * b.x) / rkstage;
  b.y = (a.y + (rkstage - 1) * b.y) / rkstage;
  *reinterpret_cast<double2*>(Qnp1 + t) = b;
}

// ---------------------------------------------------------------------------------------------------
// vertical_remap + remap_Q_ppm (prim_advection_mod.F90:1242-1330, 98-356).  Block = element, 4 waves; TWO blocks share a CU.
// Phase 1 (grid part, once per column): dp3d, ps_v, target dp, interface sums, bracket search kid/z2, the PPM grid
// coefficients per level -> LDS.  Phase 2: thread = (tracer q, column p) streams down the column keeping a 5-cell window of
// cell means in registers (kid(k) >= k-1 by construction of the search, :160-166, so the in-place update never overtakes
// the reads).
// Everything the column loop reads is cut to < 80 KB of LDS, so that two elements fit a CU and one element's grid phase (serial
// scans on 16 lanes, ~14 000 divisions) runs beside the other's column loop:
//   * the ten grid coefficients of compute_ppm_grids (:221-260) enter the column arithmetic only through five combinations:
//       da(j)  = c1 (c2 (a(j+1)-a(j)) + c3 (a(j)-a(j-1)))                       = e1 (a(j+1)-a(j)) + e2 (a(j)-a(j-1))
//       ai(j)  = a(j) + c4 d + c5 (c6 (c7-c8) d - c9 dma(j+1) + c10 dma(j)),  d = a(j+1)-a(j)
//              = a(j) + f3 d - f8 dma(j+1) + f9 dma(j)
//     with e1 = c1 c2, e2 = c1 c3, f3 = c4 + c5 c6 (c7-c8), f8 = c5 c9, f9 = c5 c10 formed once per (level, column): 47 KB instead
//     of 95 KB, five LDS reads and three flops less per level and tracer (products re-associated: relative 1e-16 per level);
//   * dpo(kid(k)) is stored per NEW level with "kid(k) == k+1" in its sign bit (no kid array, no select of two dpo values);
//   * 1/dp of the next step (bounds emission) goes through a global level field (rdp_g; an L1/L2-resident 9 KB per element);
//   * the scratch of phase 1 (old interface pressures, hybrid-coefficient differences) lies where the coefficients go afterwards.
// Measured (profiles/r03_ab_remap_two_blocks.txt): 16.5 -> 16.0 ms per launch at ne120/q35 -- the launch is bound by the vector
// issue of the column loop (two waves per SIMD in both forms), not by the phases following one another.
constexpr int REMAP_THREADS = 256;   // 4 waves: one per SIMD and element
#ifndef TSE_REMAP_PF
#define TSE_REMAP_PF 8
#endif
constexpr int REMAP_PF = TSE_REMAP_PF;  // column loads kept in flight per thread = levels per unrolled block = levels per segment task
static_assert(NLEV % REMAP_PF == 0, "whole blocks");
static_assert(REMAP_PF % CL == 0, "a block of REMAP_PF levels holds whole chunks of the bounds layout");
constexpr int REMAP_SEG_MAX = 3;   // at most this many tracers of an element go through segment tasks (LDS for their mass prefixes)
// tracers left over after whole rounds of `slots` tracer slots; more than REMAP_SEG_MAX of them take one more (partly idle) round
__host__ __device__ inline int remap_left(int qsize, int slots, int nt) {
  const int left = qsize % slots;
  return nt == 1 && left <= REMAP_SEG_MAX ? left : 0;
}
struct RemapLds {
  double cd[NLEV + 2][2][16];      // [j][e1|e2][p], cell j = 0..NLEV+1 (stage 1 of compute_ppm)                      18.9 KB
  double ca[NLEV + 1][3][16];      // [j][f3|f8|f9][p], interface j = 0..NLEV (stage 2)                               28.0 KB
  double rdpo[NLEV + 4][16];       // 1/dpo, index j+1, j = -1..NLEV+2 (the column loop multiplies instead of dividing)
  double dsel[NLEV + 4][16];       // phase 1: dpo, index j+1.  Column loop (lockstep form): row k-1 = dpo(kid(k)), negated where kid(k) == k+1
  double z2[NLEV][16];
  double mpre[REMAP_SEG_MAX][NLEV / REMAP_PF - 1][16];   // segment tasks: mass of cells 1 .. 8s-1 of the leftover tracers' columns
  unsigned char kid[NLEV][16];     // kid(k) (generic column loop)
  int slow;                        // some column has kid(k) outside {k, k+1}
  // phase-1 scratch inside the coefficient arrays (dead before they are written)
  __device__ double (*pio())[16] { return reinterpret_cast<double (*)[16]>(&cd[0][0][0]); }   // [NLEV + 2][16]: index j-1, j = 1..NLEV+2
  __device__ double* dA() { return &ca[0][0][0]; }                                            // hyai(k+1)-hyai(k)
  __device__ double* dB() { return &ca[0][0][0] + NLEV; }                                     // hybi(k+1)-hybi(k)
};
static_assert(2 * NLEV <= (NLEV + 1) * 3 * 16, "phase-1 scratch fits the coefficient arrays");
static_assert(2 * sizeof(RemapLds) <= 160 * 1024, "two elements per CU");
// The column arithmetic below is compiled WITHOUT implicit FMA contraction and spells its fused operations out: what the
// compiler fuses on its own depends on which multiplies it happens to see in the same basic block -- in an unrolled block of
// levels the product a = m/dp of the previous level, across a block boundary not -- so the rounding of a level would depend
// on where in a block, a sweep or a segment task it is evaluated.  Written out, every level of every column is one fixed
// sequence of roundings, whatever loop produces it (tests/test_gpu_parity.py: segment tasks against whole sweeps, bit for bit).
#pragma clang fp contract(off)
__device__ __forceinline__ double ppm_dma(double e1, double e2, double am, double a0, double ap) {
  double da = fma(e1, ap - a0, e2 * (a0 - am));
  double m = fmin(fabs(da), 2. * fmin(fabs(a0 - am), fabs(ap - a0)));   // minval(|da|, 2|a(j)-a(j-1)|, 2|a(j+1)-a(j)|): the doubling is exact
  double r = copysign(m, da);
  if ((ap - a0) * (a0 - am) <= 0.) r = 0.;
  return r;
}
// interface value between cells j and j+1 (compute_ppm stage 2, :295-303) in the folded form: aj + f3*(ajp-aj) - f8*dma(j+1) + f9*dma(j)
__device__ __forceinline__ double ppm_ai(double f3, double f8, double f9, double aj, double ajp, double dmajp, double dmaj) {
  return fma(f9, dmaj, fma(-f8, dmajp, fma(f3, ajp - aj, aj)));
}
__device__ __forceinline__ double remap_dma_at(const RemapLds& S, int j, int p, double am, double a0, double ap) {
  return ppm_dma(S.cd[j][0][p], S.cd[j][1][p], am, a0, ap);
}
__device__ __forceinline__ double remap_ai_at(const RemapLds& S, int j, int p, double aj, double ajp, double dmajp, double dmaj) {
  return ppm_ai(S.ca[j][0][p], S.ca[j][1][p], S.ca[j][2][p], aj, ajp, dmajp, dmaj);
}
// limited parabola of one cell from its mean a0 and interface values (compute_ppm stage 3, :309-331)
__device__ __forceinline__ void remap_coefs(double al, double ar, double a0, double& c0, double& c1, double& c2) {
  if ((ar - a0) * (a0 - al) <= 0.) { al = a0; ar = a0; }
  if ((ar - al) * fma(-0.5, al + ar, a0) > (ar - al) * (ar - al) * (1.0 / 6.0)) al = fma(3., a0, -2. * ar);
  if ((ar - al) * fma(-0.5, al + ar, a0) < -((ar - al) * (ar - al)) * (1.0 / 6.0)) ar = fma(3., a0, -2. * al);
  c0 = fma(1.5, a0, -0.25 * (al + ar));
  c1 = ar - al;
  c2 = fma(3., al + ar, -6. * a0);
}
// integrate_parabola (:349-356) from x1 = -1/2 to x2: the level-only powers, then the integral
__device__ __forceinline__ void ppm_zterms(double x2, double& z1, double& zz2, double& z3) {
  z1 = x2 + 0.5; zz2 = fma(x2, x2, -0.25) * 0.5; z3 = fma(x2 * x2, x2, 0.125);
}
__device__ __forceinline__ double ppm_integ(double c0, double c1, double c2, double z1, double zz2, double z3) {
  return fma(c2 * z3, 1.0 / 3.0, fma(c1, zz2, c0 * z1));
}
// ---- DSS + time average ON READ (the remap that closes an rsplit cycle inside tse_prim_run_subcycle) ----------------------------
// The last tracer step of a cycle leaves its stage-3 result C pre-DSS in the scratch layout and launches no k_dss_patch<1>: the remap
// assembles Qdp(np1) = (Qdp(n0) + 2 * rspheremp * DSS(C)) / 3 (prim_advection_mod.F90:929-960 + :645-662) for its own columns while it
// reads them, and writes the REMAPPED field -- the write of the DSS pass and the read of the remap (2/3 of a field pass per step at
// rsplit = 3) never happen.  Same contributions, same order (S, E, N, W edges, then the corner), same roundings as gather_sum +
// k_dss_patch<1>: the same bits as the two-kernel route (tests/test_gpu_parity.py).
// A column thread (tracer q, point p) loads, per chunk of 4 levels, its own entry (32 bytes), two neighbour entries and the 4 values
// of Qdp(n0).  Of a slab's 20 neighbour values an interior point needs none, an edge point one, a corner three: the corner's diagonal
// term is fetched by the edge lane next to it (lanes 1, 2, 13, 14 of the 16 -- their second slot is free) and handed over with one
// quad_perm DPP move, as in gather_sum; absent contributions point at the all-zero slot.
struct RemapFuse {
  const double* C;                 // stage-3 scratch, halo columns filled (null: plain remap of Q in place)
  Scr S;
  const unsigned* etab;            // [e][16][3]: entry (within a chunk) of the DSS contributions of point p, in the reference's order; absent = the zero slot
  const int* slot_of;
  const unsigned long long* pperm;
  const double* Qn0;               // Qdp(n0), standard layout
  const double* rspheremp;
  double* var_out;                 // omega_p <- rspheremp * DSS(plane qsize of C)  (prim_advection_mod.F90:943-957), may be null
  unsigned zero0;                  // entry of the all-zero slot
};
struct FuseLane { unsigned oown, oa, ob; double rs, cm /* 1.0 on a corner point, else 0.0 */; };
// x / 3, correctly rounded, in three instructions: q = RN(x * RN(1/3)) is a faithful quotient, r = x - 3q is exact in an FMA, and
// RN(q + r * RN(1/3)) is then RN(x / 3) (Markstein's theorem; 3 is not one of its exceptional divisors) -- the bits of the IEEE
// division k_dss_patch<1> performs (`/ 3`: a dozen instructions with the range scaling our field values do not need), which the
// bit-for-bit comparison of the two routes checks on every value of every parity test.
__device__ __forceinline__ double div3(double x) {
  constexpr double c = 1.0 / 3.0;
  const double q = x * c;
  return fma(fma(-3.0, q, x), c, q);
}
struct FuseRaw { double2 o[2], a[2], b[2]; };
__device__ __forceinline__ FuseLane fuse_lane(const RemapFuse& F, int e, int p) {
  FuseLane L;
  const int s = F.slot_of[e];
  const unsigned* et = F.etab + (size_t)e * 48;
  L.oown = ((unsigned)s * 16 + ppos(F.pperm[s], p)) * (CL * 8u);
  L.oa = et[p * 3] * (CL * 8u);
  const bool corner = (p == 0) | (p == 3) | (p == 12) | (p == 15);
  L.cm = corner ? 1.0 : 0.0;
  const int pc = p == 1 ? 0 : p == 2 ? 3 : p == 13 ? 12 : p == 14 ? 15 : -1;   // the corner whose diagonal term this lane fetches
  L.ob = (corner ? et[p * 3 + 1] : pc >= 0 ? et[pc * 3 + 2] : F.zero0) * (CL * 8u);
  L.rs = F.rspheremp[(size_t)e * 16 + p];
  return L;
}
// the loads of chunk kc of one tracer column: plane = &C[q][0] as bytes, q0col = &Qn0[e][q][0][p].  The Qdp(n0) values are issued
// apart from the scratch entries: they are 4 registers a chunk against 12, so that stream is kept TSE_FUSE_QAHEAD chunks further ahead
#ifndef TSE_FUSE_QAHEAD
#define TSE_FUSE_QAHEAD 0   // (1, 2: measured, no gain -- profiles/r04_ab_remap_dss_on_read.txt)
#endif
__device__ __forceinline__ void fuse_issue(FuseRaw& r, const char* __restrict__ plane, unsigned cstride /* bytes per chunk */, int kc, const FuseLane& L) {
  const char* b = plane + (size_t)kc * cstride;
  r.o[0] = *reinterpret_cast<const double2*>(b + L.oown); r.o[1] = *reinterpret_cast<const double2*>(b + L.oown + 16);
  r.a[0] = *reinterpret_cast<const double2*>(b + L.oa);   r.a[1] = *reinterpret_cast<const double2*>(b + L.oa + 16);
  r.b[0] = *reinterpret_cast<const double2*>(b + L.ob);   r.b[1] = *reinterpret_cast<const double2*>(b + L.ob + 16);
}
__device__ __forceinline__ void fuse_issue_q(double q[CL], int kc, const double* __restrict__ q0col) {
#pragma unroll
  for (int i = 0; i < CL; i++) q[i] = q0col[(size_t)(kc * CL + i) * 16];
}
// (all four lanes of a quad call it together: the hand-over of the corner's diagonal term is a DPP move)
__device__ __forceinline__ void fuse_combine(const FuseRaw& r, const double q[CL], const FuseLane& L, double cur[CL]) {
  const double o[CL] = {r.o[0].x, r.o[0].y, r.o[1].x, r.o[1].y}, a[CL] = {r.a[0].x, r.a[0].y, r.a[1].x, r.a[1].y},
               b[CL] = {r.b[0].x, r.b[0].y, r.b[1].x, r.b[1].y};
#pragma unroll
  for (int i = 0; i < CL; i++) {
    const double d = dppq<0xA5>(b[i]);   // quad_perm [1,1,2,2]: lanes 0 and 3 of the quad receive what lanes 1 and 2 fetched for them
    // "this contribution belongs to my point" as an exact 1.0 / 0.0 factor folded into FMAs, as in gather_sum: fma(1, x, t) = t + x rounds
    // like the addition, fma(0, x, t) = t (x is some finite field value)
    double t = o[i] + a[i];
    t = fma(L.cm, b[i], t);
    t = fma(L.cm, d, t);
    cur[i] = div3(fma(2.0, L.rs * t, q[i]));   // (Qdp(n0) + (rkstage-1)*Qdp(np1))/rkstage, rkstage = 3
  }
}
// Qout[e][q][.][.] = (Qdp(n0) + 2*rspheremp*DSS(C))/3 for the tracers q0 <= q < q1 of element e, and var_out <- rspheremp*DSS(plane qsize)
// if asked for: all threads of the block; thread = (plane, chunk, point), all three contributions loaded by the thread itself.  Used for
// what the sweeps do not assemble on read: the tracers that go through segment tasks, the extra plane, and every tracer of an element
// that takes the generic column loop.
__device__ __forceinline__ void fuse_materialize(const RemapFuse& F, int e, int qsize, int q0, int q1, bool var, double* __restrict__ Qout, int tid, int nthreads) {
  const int s = F.slot_of[e];
  const unsigned long long perm = F.pperm[s];
  const unsigned cstride = F.S.cse * (CL * 8u);
  const int nplane = (q1 - q0) + (var && F.var_out ? 1 : 0), nitem = nplane * NCHUNK * 16;
  constexpr int U = 5;   // items per thread in flight: the 3 + 1 planes of the usual case (35 tracers, 16 tracer slots) are 4.5 items per thread -- one memory round trip
  for (int w0 = tid; w0 < nitem; w0 += U * nthreads) {
    double2 v[U][4][2]; double qn[U][CL];
    int pl[U], kc[U], pt[U]; bool on[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int w = w0 + u * nthreads;
      on[u] = w < nitem;
      const int ww = on[u] ? w : tid;
      pt[u] = ww & 15; kc[u] = (ww >> 4) % NCHUNK; pl[u] = (ww >> 4) / NCHUNK;
      const bool isvar = q0 + pl[u] >= q1;
      const int q = isvar ? qsize : q0 + pl[u];
      const char* b = reinterpret_cast<const char*>(F.C + (size_t)q * F.S.tps) + (size_t)kc[u] * cstride;
      const unsigned* et = F.etab + ((size_t)e * 16 + pt[u]) * 3;
      const unsigned ent[4] = {(unsigned)s * 16 + ppos(perm, pt[u]), et[0], et[1], et[2]};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        v[u][c][0] = *reinterpret_cast<const double2*>(b + ent[c] * (CL * 8u));
        v[u][c][1] = *reinterpret_cast<const double2*>(b + ent[c] * (CL * 8u) + 16);
      }
#pragma unroll
      for (int i = 0; i < CL; i++) qn[u][i] = isvar ? 0.0 : F.Qn0[(((size_t)e * qsize + q) * NLEV + kc[u] * CL + i) * 16 + pt[u]];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!on[u]) continue;
      const bool isvar = q0 + pl[u] >= q1;
      const double rs = F.rspheremp[(size_t)e * 16 + pt[u]];
#pragma unroll
      for (int i = 0; i < CL; i++) {
        const int h = i >> 1;
        const double o = (i & 1) ? v[u][0][h].y : v[u][0][h].x, c0 = (i & 1) ? v[u][1][h].y : v[u][1][h].x,
                     c1 = (i & 1) ? v[u][2][h].y : v[u][2][h].x, c2 = (i & 1) ? v[u][3][h].y : v[u][3][h].x;
        const double x = rs * (((o + c0) + c1) + c2);
        if (isvar) F.var_out[((size_t)e * NLEV + kc[u] * CL + i) * 16 + pt[u]] = x;
        else Qout[(((size_t)e * qsize + q0 + pl[u]) * NLEV + kc[u] * CL + i) * 16 + pt[u]] = fma(2.0, x, qn[u][i]) / 3.0;
      }
    }
  }
}

// Generic column loop (any kid(k) >= k-1): thread = (tracer, column) walks down the column, advancing a 5-cell window by a
// data-dependent number of cells per level (compute_ppm :267-342, integrate_parabola :349-356, mass differencing :203-209).
// Only taken by elements whose Lagrangian interfaces moved by more than one layer somewhere (see k_remap).
// Plain straight-line code on purpose: lambdas/arrays passed by pointer ended up in scratch memory inside this loop.
// ALG2 (vert_remap_q_alg = 2, control_mod.F90:61-66): no mirrored ghost cells -- the two cells at either end of a column are
// piecewise constant (prim_advection_mod.F90:336-341); the parabolas of cells 3 .. nlev-2 see no ghost value either way
// (dma(2..nlev-1), ai(2..nlev-2): :283-314), so they are those of the mirrored form.
template <bool ALG2>
__device__ __forceinline__ void ppm_alg2(int cell, double a0, double& c0, double& c1, double& c2) {
  if (ALG2) { const bool pc = cell <= 2 || cell >= NLEV - 1; c0 = pc ? a0 : c0; c1 = pc ? 0. : c1; c2 = pc ? 0. : c2; }
}
template <bool ALG2>
__device__ __forceinline__ void remap_columns_generic(const RemapLds& S, double* __restrict__ Q, int e, int qsize, int tid, int nthreads,
                                                   double* __restrict__ mn_out, double* __restrict__ mx_out, const double* __restrict__ rdpg /* [k][p]: 1/dp of the next step */) {
  const int p = tid & 15;
  for (int q = tid >> 4; q < qsize; q += nthreads >> 4) {
    double* col = Q + ((size_t)e * qsize + q) * NLEV * 16 + p;
    // Cells are consumed strictly in order 1,2,3,...; a register FIFO keeps REMAP_PF column loads in flight per thread.
    int R = 0;                                   // highest cell consumed so far (ghost cells continue past NLEV)
    double pf[REMAP_PF];
#pragma unroll
    for (int t = 0; t < REMAP_PF; t++) pf[t] = col[(size_t)t * 16];
    double pm1 = 0, pa1 = 0, pm2 = 0, pa2 = 0;   // the last two real cells (mass, mean) for the bottom mirror
#define TSE_READ_NEXT(m_, a_)                                                            \
    do {                                                                                 \
      R++;                                                                               \
      if (R <= NLEV) {                                                                   \
        m_ = pf[0];                                                                      \
        _Pragma("unroll") for (int t = 0; t + 1 < REMAP_PF; t++) pf[t] = pf[t + 1];      \
        if (R + REMAP_PF <= NLEV) pf[REMAP_PF - 1] = col[(size_t)(R + REMAP_PF - 1) * 16]; \
        a_ = m_ * S.rdpo[R + 1][p];                                                      \
        pm2 = pm1; pa2 = pa1; pm1 = m_; pa1 = a_;                                        \
      } else if (R == NLEV + 1) { m_ = pm1; a_ = pa1; } /* a(nlev+1) = a(nlev)   */      \
      else { m_ = pm2; a_ = pa2; }                      /* a(nlev+2) = a(nlev-1) */      \
    } while (0)
    double m1, a1, m2, a2, m3, a3;
    TSE_READ_NEXT(m1, a1); TSE_READ_NEXT(m2, a2); TSE_READ_NEXT(m3, a3);
    int kk = 1;
    double am2 = a2, am1 = a1, a0 = a1, ap1 = a2, ap2 = a3;  // a(-1)=a(2), a(0)=a(1)
    double m0 = m1, mp1 = m2, mp2 = m3;
    double dma_m1 = remap_dma_at(S, 0, p, am2, am1, a0), dma_0 = remap_dma_at(S, 1, p, am1, a0, ap1),
           dma_p1 = remap_dma_at(S, 2, p, a0, ap1, ap2);
    double ai_m1 = remap_ai_at(S, 0, p, am1, a0, dma_0, dma_m1), ai_0 = remap_ai_at(S, 1, p, a0, ap1, dma_p1, dma_0);
    double masso_kk = 0.0, massn1 = 0.0;
    double c0, c1, c2;
    remap_coefs(ai_m1, ai_0, a0, c0, c1, c2);
    ppm_alg2<ALG2>(kk, a0, c0, c1, c2);
    for (int k = 1; k <= NLEV; k++) {
      const int kt = S.kid[k - 1][p];
      while (kk < kt) {
        masso_kk = masso_kk + m0;
        am2 = am1; am1 = a0; a0 = ap1; ap1 = ap2; m0 = mp1; mp1 = mp2;
        TSE_READ_NEXT(mp2, ap2);
        kk++;
        dma_m1 = dma_0; dma_0 = dma_p1; dma_p1 = remap_dma_at(S, kk + 1, p, a0, ap1, ap2);
        ai_m1 = ai_0; ai_0 = remap_ai_at(S, kk, p, a0, ap1, dma_p1, dma_0);
        remap_coefs(ai_m1, ai_0, a0, c0, c1, c2);
        ppm_alg2<ALG2>(kk, a0, c0, c1, c2);
      }
      double z1, zz2, z3;
      ppm_zterms(S.z2[k - 1][p], z1, zz2, z3);
      double massn2 = fma(ppm_integ(c0, c1, c2, z1, zz2, z3), S.dsel[kk + 1][p] /* = dpo(kk): this loop keeps phase 1's dpo there */, masso_kk);
      const double qnew = massn2 - massn1;
      col[(size_t)(k - 1) * 16] = qnew;
      massn1 = massn2;
      if (mn_out) {   // element min/max of Q = Qdp/dp over the 16 columns (one DPP row) for the next step's stage 1
        const double x = qnew * rdpg[(k - 1) * 16 + p];
        double mn = x, mx = x;
        mn = fmin(mn, dppq<0xB1>(mn)); mn = fmin(mn, dppq<0x4E>(mn)); mn = fmin(mn, dppq<0x124>(mn)); mn = fmin(mn, dppq<0x128>(mn));
        mx = fmax(mx, dppq<0xB1>(mx)); mx = fmax(mx, dppq<0x4E>(mx)); mx = fmax(mx, dppq<0x124>(mx)); mx = fmax(mx, dppq<0x128>(mx));
        if (p == 0) { mn_out[mm_idx(e, q, k - 1, qsize)] = mn; mx_out[mm_idx(e, q, k - 1, qsize)] = mx; }
      }
    }
#undef TSE_READ_NEXT
  }
}

// Lockstep column loop for the normal case kid(k) in {k, k+1} for every level of every column of the element (the
// interfaces moved by less than one layer: vertical CFL < 1).  All lanes then consume exactly one old cell per level, so
// the loop is branch-free and statically scheduled: at level k every lane reads cell k+3, forms dma(k+2) and ai(k+1)
// (compute_ppm stages 1-2), and picks the parabola of cell k or k+1 by its own offset o = kid(k)-k.  Loads sit in a
// register FIFO with compile-time slots (the level loop is unrolled by REMAP_PF), NT tracers per thread share every LDS
// read and the level-only part of integrate_parabola.  Arithmetic per value is the same, in the same order, as in the
// generic loop.
//
// Work split (NT = 1).  The loop is bound by VALU issue, so what counts is the number of wave-sweeps on the busiest SIMD.  A
// block has 8 waves = 32 tracer slots of 16 columns; tracers are swept 32 at a time, and the remainder (3 of 35) is NOT
// given a ninth wave -- that would put 3 waves on one SIMD and 2 on the others -- but cut into SEGMENTS of REMAP_PF levels
// that all waves share: a segment task starts from the column's serial mass prefix (k_remap phase 1a left it in LDS, summed
// in the order of the sweep), primes the 5-cell window from the cells around its first level with the formulas of the
// sweep, runs the level before its first one with the stores turned into a dump (that yields the running new-grid mass), and
// then its REMAP_PF levels: the same values in the same order as a whole sweep produces.
template <int NT, bool ALG2, bool FUSED = false>
__device__ __forceinline__ void remap_columns_fast(const RemapLds& S, double* __restrict__ Q, int e, int qsize, int tid, int nthreads,
                                                   double* __restrict__ mn_out, double* __restrict__ mx_out, double* __restrict__ sink,
                                                   const double* __restrict__ rdpg /* [k][p]: 1/dp of the next step (bounds emission) */,
                                                   const RemapFuse& F = RemapFuse{}) {
  static_assert(!FUSED || NT == 1, "DSS on read: one tracer per thread");
  const int p = tid & 15, slots = (nthreads >> 4) * NT;
  // FUSED: the whole sweeps read (Qdp(n0) + 2*rspheremp*DSS(C))/3 assembled on the fly (fuse_issue / fuse_combine: one chunk of 4 levels
  // in flight while the one before is consumed) and write Q = Qdp(np1); the segment tasks run in place on tracers the kernel has
  // materialized in Q beforehand (k_remap)
  FuseLane FL{};
  FuseRaw fraw;
  double fq[TSE_FUSE_QAHEAD + 1][CL];   // Qdp(n0) of the chunk in fraw ([0]) and of the TSE_FUSE_QAHEAD chunks behind it
  double fcur[CL] = {0, 0, 0, 0};
  const char* fplane = nullptr;
  const unsigned fcstride = FUSED ? F.S.cse * (CL * 8u) : 0u;
  if (FUSED) FL = fuse_lane(F, e, p);
  // The level body below is free of branches and predicated stores, so that the 8 levels of an unrolled block form one
  // basic block and the scheduler can overlap the dependency chains of neighbouring levels.  A surplus tracer slot of the
  // NT = 2 form (q >= qsize) therefore reads the last tracer and writes into `sink`; the element min/max is stored by all 16
  // lanes of the row (same value).
  const double* col[NT];
  double *colw[NT], *mnp[NT], *mxp[NT];
  double pf[NT][REMAP_PF];
  double ak[NT], ak1[NT], ak2[NT], mk[NT], mk1[NT], mk2[NT], dmak1[NT], aikm1[NT], aik[NT], masso[NT], massn1[NT];
  double xq[NT][CL];   // Qdp of the levels of the current chunk (fused bounds emission: times dnq = 1/dp of the next step, then min/max)
  double dnq[CL];
  double* const dump_mn = sink + NLEV * 16;
  double* const dump_mx = sink + NLEV * 16 + (size_t)NLEV * mm_qpad(qsize);
  auto aim = [&](int t, int qq, bool on, auto fused_tag) __attribute__((always_inline)) {
    col[t] = (decltype(fused_tag)::value ? F.Qn0 : Q) + ((size_t)e * qsize + qq) * NLEV * 16 + p;   // fused: the column of Qdp(n0)
    colw[t] = on ? Q + ((size_t)e * qsize + qq) * NLEV * 16 + p : sink + p;
    if (decltype(fused_tag)::value) fplane = reinterpret_cast<const char*>(F.C + (size_t)qq * F.S.tps);
    // bounds of tracer qq at level k: base[(k / CL) * mm_qpad(qsize) * CL + k % CL] (mm_idx); the dump areas of `sink` take the same offsets
    mnp[t] = on && mn_out ? mn_out + mm_idx(e, qq, 0, qsize) : dump_mn;
    mxp[t] = on && mn_out ? mx_out + mm_idx(e, qq, 0, qsize) : dump_mx;
  };
  // window at the top of the column: a(0) = a(1), a(-1) = a(2)
  auto prime_top = [&](auto fused_tag) __attribute__((always_inline)) {
    if constexpr (decltype(fused_tag)::value) {   // chunk 0 -> cells 1 .. 4 (the fourth waits in fcur[3]); chunk 1 on its way
      fuse_issue(fraw, fplane, fcstride, 0, FL);
      fuse_issue_q(fq[0], 0, col[0]);
      fuse_combine(fraw, fq[0], FL, fcur);
      mk[0] = fcur[0]; mk1[0] = fcur[1]; mk2[0] = fcur[2];
      asm volatile("" : "+v"(fcur[3]) : : "memory");   // (the values have left fraw before the next loads are issued into it)
      fuse_issue(fraw, fplane, fcstride, 1, FL);
#pragma unroll
      for (int u = 0; u <= TSE_FUSE_QAHEAD; u++) fuse_issue_q(fq[u], 1 + u, col[0]);
    } else {
#pragma unroll
    for (int t = 0; t < NT; t++) {
      mk[t] = col[t][0]; mk1[t] = col[t][16]; mk2[t] = col[t][32];
#pragma unroll
      for (int r = 0; r < REMAP_PF; r++) pf[t][r] = col[t][(size_t)(r + 3) * 16];   // cells 4 .. 3+REMAP_PF
    }
    }
    const double r1 = S.rdpo[2][p], r2 = S.rdpo[3][p], r3 = S.rdpo[4][p];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      ak[t] = mk[t] * r1; ak1[t] = mk1[t] * r2; ak2[t] = mk2[t] * r3;
      const double dma0 = remap_dma_at(S, 0, p, ak1[t], ak[t], ak[t]);
      const double dma1 = remap_dma_at(S, 1, p, ak[t], ak[t], ak1[t]);
      dmak1[t] = remap_dma_at(S, 2, p, ak[t], ak1[t], ak2[t]);
      aikm1[t] = remap_ai_at(S, 0, p, ak[t], ak[t], dma1, dma0);
      aik[t] = remap_ai_at(S, 1, p, ak[t], ak1[t], dmak1[t], dma1);
      masso[t] = 0.0; massn1[t] = 0.0;
    }
  };
  // one level; TAIL: the window may run into the mirrored ghost cells and the FIFO may run dry (the last two blocks of 8, and
  // every block of a segment task, whose kb differs from lane to lane)
  const double *bcd, *bca, *brdpo, *bds, *bz2;   // per-block LDS bases
  auto bases = [&](int kb) __attribute__((always_inline)) {
    bcd = &S.cd[kb][0][p]; bca = &S.ca[kb][0][p]; brdpo = &S.rdpo[kb][p]; bds = &S.dsel[kb][p]; bz2 = &S.z2[kb][p];
  };
  // The level-only LDS operands of a level (two dma coefficients, three interface coefficients, dpo(kid) with the kid bit, z2,
  // 1/dpo of the cell entering the window) are fetched ONE LEVEL AHEAD: a level starts with its operands in registers and
  // issues the next level's reads before it computes, so no wave parks behind an LDS latency at the top of every level (with two
  // waves per SIMD nothing else covered those: 47 % of the wave cycles were waits).  Rows are addressed relative to the
  // per-block bases (bcd = &cd[kb][0][p] etc., set up once per block): the offsets are small compile-time immediates, also for
  // the first level of the NEXT block (sl + 1 == REMAP_PF), which is just the next row.
  struct LevelOps { double e1, e2, f3, f8, f9, dss, z2, rr; };
  LevelOps nxt;
  auto fetch = [&](auto tail_tag, int kb, int sl) __attribute__((always_inline)) -> LevelOps {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const int k = kb + sl + 1;
    LevelOps v{0., 0., 0., 0., 0., 0., 0., 0.};
    if (TAIL && k > NLEV) return v;   // (the level after the last one)
    const int jdo = !TAIL || k + 2 <= NLEV + 1 ? sl + 3 : NLEV + 1 - kb, jao = !TAIL || k + 1 <= NLEV ? sl + 2 : NLEV - kb;   // last level: unused
    v.e1 = bcd[(jdo * 2 + 0) * 16]; v.e2 = bcd[(jdo * 2 + 1) * 16];
    v.f3 = bca[(jao * 3 + 0) * 16]; v.f8 = bca[(jao * 3 + 1) * 16]; v.f9 = bca[(jao * 3 + 2) * 16];
    v.dss = bds[sl * 16];
    v.z2 = bz2[sl * 16];
    if (!TAIL || k + 3 <= NLEV) v.rr = brdpo[(sl + 5) * 16];   // rdpo[r + 1], r = k + 3
    return v;
  };
  auto level = [&](auto tail_tag, auto emit_tag, int kb, int sl, auto reload_tag, auto fused_tag) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_tag)::value, EMIT = decltype(emit_tag)::value, RELOAD = decltype(reload_tag)::value;   // RELOAD: keep the FIFO filled
    constexpr bool FZ = decltype(fused_tag)::value;
    const int k = kb + sl + 1, r = k + 3;   // level being written, cell entering the window
    // (FUSED: the level fetches its own operands -- the one-level lookahead costs 16 registers the chunk in flight needs: with it the
    // kernel spills inside the level loop, without it 9 registers are parked once per sweep)
    LevelOps c;
    if (FZ) c = fetch(tail_tag, kb, sl);
    else { c = nxt; nxt = fetch(tail_tag, kb, sl + 1); }
    double ak3[NT], mk3[NT];
    if (!TAIL || r <= NLEV) {
      if constexpr (FZ) {
        // cell r is level (sl + 3) & 3 of chunk (r - 1) / CL (kb is a multiple of REMAP_PF): a new chunk starts at sl = 1, 5 -- its loads were
        // issued four levels ago; combine them and send for the next chunk
        const int ci = (sl + 3) & (CL - 1);
        if (ci == 0) {
          fuse_combine(fraw, fq[0], FL, fcur);
          asm volatile("" : "+v"(fcur[0]), "+v"(fcur[1]), "+v"(fcur[2]), "+v"(fcur[3]) : : "memory");
          const int kn = (r - 1) / CL + 1;
          if (!TAIL || kn < NCHUNK) fuse_issue(fraw, fplane, fcstride, kn, FL);
#pragma unroll
          for (int u = 0; u < TSE_FUSE_QAHEAD; u++)
#pragma unroll
            for (int i = 0; i < CL; i++) fq[u][i] = fq[u + 1][i];   // (renamed away inside the unrolled block)
          if (!TAIL || kn + TSE_FUSE_QAHEAD < NCHUNK) fuse_issue_q(fq[TSE_FUSE_QAHEAD], kn + TSE_FUSE_QAHEAD, col[0]);
        }
        mk3[0] = fcur[ci];
        ak3[0] = mk3[0] * c.rr;
      } else {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        mk3[t] = pf[t][sl];
        if (RELOAD && (!TAIL || r + REMAP_PF <= NLEV)) pf[t][sl] = col[t][(size_t)(r + REMAP_PF - 1) * 16];
        ak3[t] = mk3[t] * c.rr;
      }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; t++) {   // a(nlev+1) = a(nlev), a(nlev+2) = a(nlev-1); nothing beyond is used
        mk3[t] = 0.0;
        ak3[t] = r == NLEV + 1 ? ak2[t] : ak[t];
      }
    }
    const bool o = __double2hiint(c.dss) < 0;   // kid(k) == k+1 rides in the sign bit of dpo(kid(k))
    const double dsel = fabs(c.dss);
    double z1, zz2, z3;
    ppm_zterms(c.z2, z1, zz2, z3);
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const double dmak2 = ppm_dma(c.e1, c.e2, ak1[t], ak2[t], ak3[t]);
      const double aik1 = ppm_ai(c.f3, c.f8, c.f9, ak1[t], ak2[t], dmak2, dmak1[t]);
      const double mo1 = masso[t] + mk[t];
      const double al = o ? aik[t] : aikm1[t], ar = o ? aik1 : aik[t], a0 = o ? ak1[t] : ak[t], ms = o ? mo1 : masso[t];
      double c0, c1, c2;
      remap_coefs(al, ar, a0, c0, c1, c2);
      ppm_alg2<ALG2>(o ? k + 1 : k, a0, c0, c1, c2);
      const double massn2 = fma(ppm_integ(c0, c1, c2, z1, zz2, z3), dsel, ms);
      const double qnew = massn2 - massn1[t];
      colw[t][(size_t)(k - 1) * 16] = qnew;
      massn1[t] = massn2;
      if (EMIT) xq[t][sl % CL] = qnew;        // times 1/dp of the next step's stage 1 and reduced over the columns by emit4
      masso[t] = mo1;
      ak[t] = ak1[t]; ak1[t] = ak2[t]; ak2[t] = ak3[t];
      mk[t] = mk1[t]; mk1[t] = mk2[t]; mk2[t] = mk3[t];
      dmak1[t] = dmak2; aikm1[t] = aik[t]; aik[t] = aik1;
    }
  };
  // Element min/max of Q over the 16 columns (one DPP row) for the 4 levels of a chunk at once, as a halving butterfly: lane
  // pairs (xor 1) split the 4 levels between them -- each keeps two and hands the other two over, the moves shared by min and
  // max --, lane pairs (xor 2) split again, and two row rotations finish across the quads: 42 instead of 96 DPP moves and
  // min/max per chunk, and the lane that ends up with level 2*(p&1) + ((p>>1)&1) of the chunk stores it (4 lanes: one line).
  auto emit4 = [&](int kb, int sl0) __attribute__((always_inline)) {
    const bool b0 = p & 1, b1 = p & 2;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const double x[CL] = {xq[t][0] * dnq[0], xq[t][1] * dnq[1], xq[t][2] * dnq[2], xq[t][3] * dnq[3]};   // Q = Qdp * (1/dp)
      const double s0 = b0 ? x[0] : x[2], s1 = b0 ? x[1] : x[3], k0 = b0 ? x[2] : x[0], k1 = b0 ? x[3] : x[1];
      const double r0 = dppq<0xB1>(s0), r1 = dppq<0xB1>(s1);
      const double n0 = fmin(k0, r0), n1 = fmin(k1, r1), m0 = fmax(k0, r0), m1 = fmax(k1, r1);
      double mn = fmin(b1 ? n1 : n0, dppq<0x4E>(b1 ? n0 : n1));
      double mx = fmax(b1 ? m1 : m0, dppq<0x4E>(b1 ? m0 : m1));
      mn = fmin(mn, dppq<0x124>(mn)); mn = fmin(mn, dppq<0x128>(mn));
      mx = fmax(mx, dppq<0x124>(mx)); mx = fmax(mx, dppq<0x128>(mx));
      const size_t mo = (size_t)((kb + sl0) / CL) * mm_qpad(qsize) * CL + (b0 ? 2 : 0) + (b1 ? 1 : 0);   // kb, sl0 are multiples of CL
      mnp[t][mo] = mn; mxp[t][mo] = mx;   // (the four quads of the row hold and store the same four values)
    }
  };
  // scheduler fence every 4 levels (fences every 2 or 8 levels: same time; 244 instead of 212 registers without)
  auto block = [&](auto tail_tag, auto emit_tag, int kb, auto reload_tag, auto fused_tag) __attribute__((always_inline)) {
    bases(kb);
#pragma unroll
    for (int sl = 0; sl < REMAP_PF; sl++) {
      if (decltype(emit_tag)::value && sl % CL == 0) {   // the chunk's 1/dp: loaded here, used four levels later
#pragma unroll
        for (int i = 0; i < CL; i++) dnq[i] = rdpg[(kb + sl + i) * 16 + p];
      }
      level(tail_tag, emit_tag, kb, sl, reload_tag, fused_tag);
      if (sl % CL == CL - 1) { if (decltype(emit_tag)::value) emit4(kb, sl - (CL - 1)); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  constexpr int CLEAN = NLEV - 2 * REMAP_PF;   // blocks starting below this never see a ghost cell or an empty FIFO slot
  using FusedTag = std::integral_constant<bool, FUSED>;
  auto column = [&](auto emit_tag) __attribute__((always_inline)) {
    for (int kb = 0; kb < CLEAN; kb += REMAP_PF) block(std::false_type{}, emit_tag, kb, std::true_type{}, FusedTag{});
    for (int kb = CLEAN; kb < NLEV; kb += REMAP_PF) block(std::true_type{}, emit_tag, kb, std::true_type{}, FusedTag{});
  };

  // ---- whole sweeps, `slots` tracers at a time
  const int left = remap_left(qsize, slots, NT);          // tracers that go through segment tasks instead
  const int qsweep = qsize - left;
  for (int q0 = (tid >> 4) * NT; q0 < qsweep; q0 += slots) {
#pragma unroll
    for (int t = 0; t < NT; t++) { const bool on = q0 + t < qsize; aim(t, on ? q0 + t : qsize - 1, on, FusedTag{}); }
    prime_top(FusedTag{});
    bases(0);
    nxt = fetch(std::false_type{}, 0, 0);   // the first level's operands
    if (mn_out) column(std::true_type{});
    else column(std::false_type{});
  }
  // ---- the remaining tracers as segment tasks: (tracer, segment) pairs, 16 columns each, dealt to the tracer slots -- the
  //      segments that start inside the column first, the top segments (plain start) last, so that a wave holds one kind.
  //      The column is updated in place: a task's window reaches into its neighbours' levels, so all tasks of a tracer run in
  //      the same round, and a workgroup barrier separates the loads of a round (everything a task reads is in registers
  //      after its run-in level) from its stores.
  if constexpr (NT == 1) {
    constexpr int NSEG = NLEV / REMAP_PF;
    const int per = slots / NSEG;                                  // tracers per round
    for (int tb = 0; tb < left; tb += per) {                       // (uniform over the block)
      const int ntr = min(per, left - tb), nwarm = ntr * (NSEG - 1), g = tid >> 4;
      const bool act = g < ntr * NSEG, warm = g < nwarm;
      const int tr = warm ? g / (NSEG - 1) : g - nwarm, kb0 = warm ? (g - tr * (NSEG - 1) + 1) * REMAP_PF : 0;
      if (act) {
        aim(0, qsweep + tb + tr, true, std::false_type{});
        if (warm) {
          // state on entry of level kb0 (the one before the segment): cells kb0-2 .. kb0+2 (cell j = col[(j-1)*16], a = m*rdpo[j+1])
          const double* c = col[0] + (size_t)(kb0 - 3) * 16;
          const double m_2 = c[0], m_1 = c[16];
          mk[0] = c[32]; mk1[0] = c[48]; mk2[0] = c[64];
          pf[0][REMAP_PF - 1] = c[80];                                              // cell kb0+3, consumed by the run-in level
#pragma unroll
          for (int r = 0; r + 1 < REMAP_PF; r++) pf[0][r] = kb0 + r + 4 <= NLEV ? c[(size_t)(r + 6) * 16] : 0.0;   // cells kb0+4 .. kb0+10
          const double* rd = &S.rdpo[kb0 - 1][p];
          const double a_2 = m_2 * rd[0], a_1 = m_1 * rd[16];
          ak[0] = mk[0] * rd[32]; ak1[0] = mk1[0] * rd[48]; ak2[0] = mk2[0] * rd[64];
          const double dma_1 = remap_dma_at(S, kb0 - 1, p, a_2, a_1, ak[0]);
          const double dma0 = remap_dma_at(S, kb0, p, a_1, ak[0], ak1[0]);
          dmak1[0] = remap_dma_at(S, kb0 + 1, p, ak[0], ak1[0], ak2[0]);
          aikm1[0] = remap_ai_at(S, kb0 - 1, p, a_1, ak[0], dma0, dma_1);
          aik[0] = remap_ai_at(S, kb0, p, ak[0], ak1[0], dmak1[0], dma0);
          masso[0] = S.mpre[tb + tr][kb0 / REMAP_PF - 1][p];                         // sum of cells 1 .. kb0-1 in sweep order
          massn1[0] = 0.0;
          // run-in: level kb0 itself (slot REMAP_PF-1 of the block before), its value dumped; it also fetches cell kb0+11
          double* const w = colw[0];
          colw[0] = sink + p;
          bases(kb0 - REMAP_PF);
          nxt = fetch(std::true_type{}, kb0 - REMAP_PF, REMAP_PF - 1);
          level(std::true_type{}, std::false_type{}, kb0 - REMAP_PF, REMAP_PF - 1, std::true_type{}, std::false_type{});
          colw[0] = w;
        } else {
          prime_top(std::false_type{});
          bases(0);
          nxt = fetch(std::false_type{}, 0, 0);
        }
      }
      __syncthreads();   // (waits for the loads, too)
      if (act) {
        if (mn_out) block(std::true_type{}, std::true_type{}, kb0, std::false_type{}, std::false_type{});
        else block(std::true_type{}, std::false_type{}, kb0, std::false_type{}, std::false_type{});
      }
    }
  }
}

#pragma clang fp contract(fast)   // (the default of the rest of the file)

template <int NT, bool ALG2 = false, bool FUSED = false>
__global__ __launch_bounds__(REMAP_THREADS / NT, NT == 1 ? 2 : 1 /* <= 256 registers: two blocks per CU */) void k_remap(
    int qsize, double dt, double ps0, const double* __restrict__ hyai, const double* __restrict__ hybi, const double* __restrict__ dp,
    const double* __restrict__ divdp_proj, double* __restrict__ dp3d, double* __restrict__ ps_v, double* __restrict__ Q, int* __restrict__ bad,
    double* __restrict__ mn_out, double* __restrict__ mx_out, int force_generic, double* __restrict__ sink,
    const double* __restrict__ dp2 /* null: the target grid of vertical_remap (:1313-1319); else remap_Q_ppm's dp2 argument [e][k][p] */,
    const int* __restrict__ elist /* elements of this launch (null: 0..nwork) */, int nwork,
    double* __restrict__ rdp_g /* [e][k][p] work field: 1/dp of the next step, for the bounds emission (written and read by the same block) */,
    RemapFuse F /* FUSED: Q is written only; its cell values are (Qdp(n0) + 2*rspheremp*DSS(F.C))/3, assembled on read */) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  RemapLds& S = *reinterpret_cast<RemapLds*>(smem_raw);
  // blocks are dealt round-robin to the 8 XCDs: logical block = (blockIdx % 8) * (gridDim / 8) + blockIdx / 8 gives every XCD a contiguous
  // range of the list (the lists are in slot order: an XCD works on whole patches, so that the neighbour entries the DSS on read
  // gathers were fetched into the same L2 by the neighbouring blocks)
  const int lb = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (lb >= nwork) return;   // (whole block, before any barrier)
  const int e = elist ? elist[lb] : lb, tid = threadIdx.x, nthreads = blockDim.x;
  double (*const pio)[16] = S.pio();
  double* const dA = S.dA();
  double* const dB = S.dB();
  double (*const dpo)[16] = S.dsel;   // phase 1 keeps dpo (index j+1) where the column loop finds dpo(kid(k)) afterwards
  double* const rdpg = rdp_g + (size_t)e * NLEV * 16;
  if (tid == 0) S.slow = force_generic;
  if (FUSED) {
    // what the sweeps do not assemble on read: the extra plane of C (omega_p), and the tracers that go through segment tasks -- those
    // are put into Q here and handled in place like in the plain kernel (their old-mass prefix below reads them after the barrier)
    const int left = remap_left(qsize, (nthreads >> 4) * NT, NT);
    fuse_materialize(F, e, qsize, qsize - left, qsize, true, Q, tid, nthreads);
  }
  // ---- phase 1a: dp3d = dp - dt*divdp_proj (all threads), then one thread per column for the scans
  for (int w = tid; w < NLEV * 16; w += nthreads) {
    size_t o = (size_t)e * NLEV * 16 + w;
    double d = dp[o] - dt * divdp_proj[o];
    dp3d[o] = d;
    dpo[(w >> 4) + 2][w & 15] = d;
    if (mn_out) rdpg[w] = 1.0 / dp[o];   // Q = Qdp * (1/dp), as every kernel forms its bounds (k_qminmax)
    if (d < 0) atomicOr(bad, 1);
  }
  if (tid < NLEV) { dA[tid] = hyai[tid + 1] - hyai[tid]; dB[tid] = hybi[tid + 1] - hybi[tid]; }
  __syncthreads();
  if (tid < 16) {
    // The column scans keep the reference's serial order (their roundings are part of the result), so 16 lanes do them; what
    // can be taken off the dependency chain is everything but the adds: the layers are fetched from LDS a dozen at a time
    // (one latency per batch instead of one per level), and sum(dp3d) of ps_v (:1313) IS the last old-grid interface
    // pressure pio(nlev+1) (:142-144: the same terms added in the same order from 0), so it is not summed a second time; the
    // hybrid coefficient differences of the target grid (:1316-1317) wait in LDS instead of behind one scalar load per level.
    const int p = tid;
    constexpr int SB = 12;
    static_assert(NLEV % SB == 0, "scan batches");
    double run = 0.0;
    pio[0][p] = 0.0;
    for (int kb = 0; kb < NLEV; kb += SB) {
      double d[SB];
#pragma unroll
      for (int i = 0; i < SB; i++) d[i] = dpo[kb + i + 2][p];
#pragma unroll
      for (int i = 0; i < SB; i++) { run = run + d[i]; pio[kb + i + 1][p] = run; }
    }
    const double pio_prev = run;
    const double ps = fma(hyai[0], ps0, run);
    ps_v[(size_t)e * 16 + p] = ps;
    pio[NLEV + 1][p] = pio_prev + 1.;
    for (int k = 1; k <= 2; k++) { dpo[2 - k][p] = dpo[k + 1][p]; dpo[NLEV + k + 1][p] = dpo[NLEV + 2 - k][p]; }
    // new-grid interface pressures pin(k+1) (serial sum, as the reference's :150-158), parked in z2's slot until the
    // bracket search below replaces them
    double pin = 0.0;
    for (int kb = 0; kb < NLEV; kb += SB) {
      double d[SB];
      if (dp2) {
#pragma unroll
        for (int i = 0; i < SB; i++) d[i] = dp2[((size_t)e * NLEV + kb + i) * 16 + p];
      } else {
#pragma unroll
        for (int i = 0; i < SB; i++) d[i] = fma(dA[kb + i], ps0, __dmul_rn(dB[kb + i], ps));   // dA*ps0 + dB*ps, one rounding of the sum
      }
#pragma unroll
      for (int i = 0; i < SB; i++) {
        const int k = kb + i + 1;
        pin = pin + d[i];
        S.z2[k - 1][p] = (k == NLEV) ? pio_prev : pin;  // pin(nlev+1) = pio(nlev+1)
      }
    }
  } else if (tid >= 64 && tid < 64 + 16 * remap_left(qsize, (nthreads >> 4) * NT, NT)) {
    // meanwhile the second wave sums the old masses of the tracers that phase 2 handles as segment tasks: sum of cells
    // 1 .. 8s-1 for s = 1..8, added from 0 in the order of the sweep (masso in remap_columns_fast)
    const int left = remap_left(qsize, (nthreads >> 4) * NT, NT), tr = (tid - 64) >> 4, p = tid & 15;
    const double* c = Q + ((size_t)e * qsize + (qsize - left + tr)) * NLEV * 16 + p;
    // all loads first (one memory round trip; the registers are free in this phase), then the serial adds
    constexpr int NPRE = NLEV - REMAP_PF;
    double m[NPRE];
#pragma unroll
    for (int i = 0; i < NPRE; i++) m[i] = c[(size_t)i * 16];
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < NPRE; i++) {
      if (i % REMAP_PF == REMAP_PF - 1) S.mpre[tr][i / REMAP_PF][p] = run;   // before cell 8s is added: cells 1 .. 8s-1
      run = run + m[i];
    }
  }
  __syncthreads();
  // bracket search (:160-172), one (k,p) per work item: every level starts its own search at kk = k
  for (int w = tid; w < NLEV * 16; w += nthreads) {
    const int k = (w >> 4) + 1, p = w & 15;
    const double pin_k1 = S.z2[k - 1][p];
    int kk = k;
    while (pio[kk - 1][p] <= pin_k1) kk++;
    kk--;
    if (kk == NLEV + 1) kk = NLEV;
    S.kid[k - 1][p] = (unsigned char)kk;
    if (kk != k && kk != k + 1) S.slow = 1;   // a displacement of more than one layer: this element takes the generic loop
    S.z2[k - 1][p] = (pin_k1 - (pio[kk - 1][p] + pio[kk][p]) * 0.5) / dpo[kk + 1][p];
  }
  __syncthreads();   // pio, dA, dB are dead: the coefficients go where they were
  // ---- phase 1b: grid coefficients (compute_ppm_grids, :221-260) in the folded form the column loop uses, one (j,p) per work item
  for (int w = tid; w < (NLEV + 4) * 16; w += nthreads) S.rdpo[w >> 4][w & 15] = 1.0 / dpo[w >> 4][w & 15];
  for (int w = tid; w < (NLEV + 2) * 16; w += nthreads) {
    const int jj = w >> 4, p = w & 15;  // jj = j, j = 0..NLEV+1
#define DX(j) dpo[(j) + 1][p]
    const double c1 = DX(jj) / (DX(jj - 1) + DX(jj) + DX(jj + 1));
    S.cd[jj][0][p] = c1 * ((2. * DX(jj - 1) + DX(jj)) / (DX(jj + 1) + DX(jj)));
    S.cd[jj][1][p] = c1 * ((DX(jj) + 2. * DX(jj + 1)) / (DX(jj - 1) + DX(jj)));
    if (jj <= NLEV) {
      const double c4 = DX(jj) / (DX(jj) + DX(jj + 1));
      const double c5 = 1. / (DX(jj - 1) + DX(jj) + DX(jj + 1) + DX(jj + 2));
      // (:255-257) c6*(c7 - c8), evaluated as the reference does (left to right)
      const double c678 = ((2. * DX(jj + 1) * DX(jj)) / (DX(jj) + DX(jj + 1))) *
                          ((DX(jj - 1) + DX(jj)) / (2. * DX(jj) + DX(jj + 1)) - (DX(jj + 2) + DX(jj + 1)) / (2. * DX(jj + 1) + DX(jj)));
      const double c9 = DX(jj) * (DX(jj - 1) + DX(jj)) / (2. * DX(jj) + DX(jj + 1));
      const double c10 = DX(jj + 1) * (DX(jj + 1) + DX(jj + 2)) / (DX(jj) + 2. * DX(jj + 1));
      S.ca[jj][0][p] = c4 + c5 * c678;
      S.ca[jj][1][p] = c5 * c9;
      S.ca[jj][2][p] = c5 * c10;
    }
#undef DX
  }
  __syncthreads();
  // ---- phase 2: data part
  if (S.slow) {
    if (FUSED) {   // an element that takes the generic loop: all its tracers first (the loop's windows advance by data-dependent steps)
      fuse_materialize(F, e, qsize, 0, qsize - remap_left(qsize, (nthreads >> 4) * NT, NT), false, Q, tid, nthreads);
      __syncthreads();
    }
    remap_columns_generic<ALG2>(S, Q, e, qsize, tid, nthreads, mn_out, mx_out, rdpg);
    return;
  }
  {
    // dpo -> dpo(kid(k)) per new level, in place: read first, write after a barrier (a row is read by the two levels above it)
    constexpr int PER = (NLEV * 16 + REMAP_THREADS / NT - 1) / (REMAP_THREADS / NT);
    double v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int w = tid + i * nthreads;
      v[i] = 0.0;
      if (w < NLEV * 16) {
        const int k = (w >> 4) + 1, p = w & 15, kk = S.kid[k - 1][p];
        const double d = dpo[kk + 1][p];
        v[i] = kk == k ? d : -d;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
      const int w = tid + i * nthreads;
      if (w < NLEV * 16) S.dsel[w >> 4][w & 15] = v[i];
    }
    __syncthreads();
  }
  remap_columns_fast<NT, ALG2, FUSED>(S, Q, e, qsize, tid, nthreads, mn_out, mx_out, sink, rdpg, F);
}

// ---------------------------------------------------------------------------------------------------
// Prescribed DCMIP 1-1 / 1-2 fields on the device: dcmip_123_mod.F90:85-260,262-409 evaluated
// with zcoords=1, as dcmip_wrapper_mod.F90:49-243 drives them.
struct DcmipPt { double u, v, w, p, rho, q[4]; };
__device__ inline DcmipPt dcmip_point(int test, double time, double lon, double lat, double z) {
  const double a = 6.376e6, pi = 3.141592653589793238462643383279, Rd = 287.04, g = 9.80616, T0 = 300.0, P0 = 100000.0;
  const double H = Rd * T0 / g;
  DcmipPt r;
  double p = P0 * exp(-z / H);
  if (test == 1) {
    const double tau = 12.0 * 86400.0, u0 = (2.0 * pi * a) / tau, k0 = (10.0 * a) / tau, omega0 = (23000.0 * pi) / tau;
    const double RR = 0.5, ZZ = 1000.0, z0 = 5000.0, lambda0 = 5.0 * pi / 6.0, lambda1 = 7.0 * pi / 6.0;
    double ptop = P0 * exp(-12000.0 / H);
    double lonp = lon - 2.0 * pi * time / tau;
    double plim = fmax(p, ptop);
    double bs = (double)0.2f;
    double s = 1.0 + exp((ptop - P0) / (bs * ptop)) - exp((plim - P0) / (bs * ptop)) - exp((ptop - plim) / (bs * ptop));
    double cl = cos(lat);
    double ud = (omega0 * a) / (bs * ptop) * cos(lonp) * (cl * cl) * cos(2.0 * pi * time / tau) *
                (-exp((plim - P0) / (bs * ptop)) + exp((ptop - plim) / (bs * ptop)));
    r.u = k0 * sin(lonp) * sin(lonp) * sin(2.0 * lat) * cos(pi * time / tau) + u0 * cos(lat) + ud;
    r.v = k0 * sin(2.0 * lonp) * cos(lat) * cos(pi * time / tau);
    r.w = -((Rd * T0) / (g * plim)) * omega0 * sin(lonp) * cos(lat) * cos(2.0 * pi * time / tau) * s;
    r.rho = p / (Rd * T0);
    double sin_tmp = sin(lat) * sin(0.0), cos_tmp = cos(lat) * cos(0.0);
    double rr1 = acos(sin_tmp + cos_tmp * cos(lon - lambda0));
    double rr2 = acos(sin_tmp + cos_tmp * cos(lon - lambda1));
    double hz = (z - z0) / ZZ;
    double d1 = fmin(1.0, (rr1 / RR) * (rr1 / RR) + hz * hz), d2 = fmin(1.0, (rr2 / RR) * (rr2 / RR) + hz * hz);
    r.q[0] = 0.5 * (1.0 + cos(pi * d1)) + 0.5 * (1.0 + cos(pi * d2));
    r.q[1] = 0.9 - 0.8 * (r.q[0] * r.q[0]);
    r.q[2] = (d1 <= RR || d2 <= RR) ? 1.0 : 0.1;
    if (z > z0 && fabs(lat) < 0.125) r.q[2] = 0.1;
    r.q[3] = 1.0 - 0.3 * (r.q[0] + r.q[1] + r.q[2]);
  } else {
    const double tau = 86400.0, u0 = 40.0, w0 = 0.15, K = 5.0, z1 = 2000.0, z2 = 5000.0, z0 = 0.5 * (z1 + z2), ztop = 12000.0;
    double ptop = P0 * exp(-ztop / H);
    double rho = fmax(p, ptop) / (Rd * T0), rho0 = P0 / (Rd * T0);
    r.u = u0 * cos(lat);
    double hstar = fmin(z / ztop, 1.0);
    r.v = -(rho0 / rho) * (a * w0 * pi) / (K * ztop) * cos(lat) * sin(K * lat) * cos(pi * hstar) * cos(pi * time / tau);
    r.w = (rho0 / rho) * (w0 / K) * (-2.0 * sin(K * lat) * sin(lat) + K * cos(lat) * cos(K * lat)) * sin(pi * hstar) * cos(pi * time / tau);
    r.rho = rho;
    r.q[0] = 0.0;
    r.q[1] = (z < z2 && z > z1) ? 0.5 * (1.0 + cos(2.0 * pi * (z - z0) / (z2 - z1))) : 0.0;
    r.q[2] = 0.0; r.q[3] = 0.0;
  }
  r.p = p;
  return r;
}

// per-step inputs: derived%dp, vn0 = u(t_wind)*dp, eta_dot_dpdn(t_now), omega_p = 0
// zm[72], zi[73], pint[73] are per-level constants prepared on the host (dcmip_wrapper_mod.F90:68-89,183).
// The prescribed fields factor into (level-only) x (column-only) x (time-only) terms; evaluating dcmip_point per grid point
// spends ~15 transcendental calls on every one of the 73 levels of a column.  k_dcmip_tables evaluates the level-only
// factors once (same device libm, same expression trees as dcmip_point, cut exactly where the left-to-right product order
// allows it, so the values are identical), k_dcmip_step evaluates the column-only factors once per column and step and
// then only multiplies.
struct DcmipTab { double m1[NLEV], m2[NLEV], i1[NLEVP], i2[NLEVP], i3[NLEVP]; };
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_dcmip_tables(int test, const double* __restrict__ zm, const double* __restrict__ zi, DcmipTab* __restrict__ T) {
  const int k = threadIdx.x;
  if (k >= NLEVP) return;
  const double a = 6.376e6, pi = 3.141592653589793238462643383279, Rd = 287.04, g = 9.80616, T0 = 300.0, P0 = 100000.0;
  const double H = Rd * T0 / g;
  for (int half = 0; half < 2; half++) {   // 0: interface zi[k], 1: mid level zm[k]
    if (half == 1 && k >= NLEV) break;
    const double z = half ? zm[k] : zi[k];
    const double p = P0 * exp(-z / H);
    if (test == 1) {
      const double tau = 12.0 * 86400.0, omega0 = (23000.0 * pi) / tau;
      const double ptop = P0 * exp(-12000.0 / H), plim = fmax(p, ptop), bs = (double)0.2f;
      if (half) {
        T->m1[k] = -exp((plim - P0) / (bs * ptop)) + exp((ptop - plim) / (bs * ptop));   // last factor of ud
        T->m2[k] = 0.0;
      } else {
        T->i1[k] = -((Rd * T0) / (g * plim)) * omega0;                                     // leading factors of w
        T->i2[k] = 1.0 + exp((ptop - P0) / (bs * ptop)) - exp((plim - P0) / (bs * ptop)) - exp((ptop - plim) / (bs * ptop));   // s
        T->i3[k] = -9.80616 * (p / (Rd * T0));                                             // -g*rho
      }
    } else {
      const double w0 = 0.15, K = 5.0, ztop = 12000.0;
      const double ptop = P0 * exp(-ztop / H), rho = fmax(p, ptop) / (Rd * T0), rho0 = P0 / (Rd * T0);
      const double hstar = fmin(z / ztop, 1.0);
      if (half) {
        T->m1[k] = -(rho0 / rho) * (a * w0 * pi) / (K * ztop);   // leading factors of v
        T->m2[k] = cos(pi * hstar);
      } else {
        T->i1[k] = (rho0 / rho) * (w0 / K);                      // leading factors of w
        T->i2[k] = sin(pi * hstar);
        T->i3[k] = -9.80616 * rho;
      }
    }
  }
}
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ __launch_bounds__(256) void k_dcmip_step(int nelemd, int test, double t_wind, double t_now, const double* __restrict__ lat,
                                                    const double* __restrict__ lon, const DcmipTab* __restrict__ T,
                                                    const double* __restrict__ pint, double* __restrict__ vn0, double* __restrict__ dp,
                                                    double* __restrict__ eta, double* __restrict__ omega_p) {
  // thread = one column (e,p); the 16 columns of an element are 16 consecutive lanes, so every level is a 128-B store
  const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= (size_t)nelemd * 16) return;
  const int p = (int)(col & 15), e = (int)(col >> 4);
  const double lo = lon[col], la = lat[col];
  const double a = 6.376e6, pi = 3.141592653589793238462643383279;
  double u_col, v_col, u_lev = 0.0, w_col, ct_w;   // u = u_col + u_lev*m1 ; v = v_col (test 1) | ((m1*v_col)... (test 2) ; w per test
  double cl = cos(la);
  if (test == 1) {
    const double tau = 12.0 * 86400.0, u0 = (2.0 * pi * a) / tau, k0 = (10.0 * a) / tau, omega0 = (23000.0 * pi) / tau;
    const double H = 287.04 * 300.0 / 9.80616, ptop = 100000.0 * exp(-12000.0 / H), bs = (double)0.2f;
    const double lonp = lo - 2.0 * pi * t_wind / tau;
    u_col = k0 * sin(lonp) * sin(lonp) * sin(2.0 * la) * cos(pi * t_wind / tau) + u0 * cos(la);
    u_lev = (omega0 * a) / (bs * ptop) * cos(lonp) * (cl * cl) * cos(2.0 * pi * t_wind / tau);
    v_col = k0 * sin(2.0 * lonp) * cos(la) * cos(pi * t_wind / tau);
    const double lonw = lo - 2.0 * pi * t_now / tau;
    w_col = sin(lonw); ct_w = cos(2.0 * pi * t_now / tau);
  } else {
    const double tau = 86400.0, u0 = 40.0, K = 5.0;
    u_col = u0 * cos(la);
    v_col = sin(K * la); ct_w = cos(pi * t_wind / tau);                 // v = (((m1*cos(lat))*sin(K lat))*m2)*cos(pi t/tau)
    w_col = -2.0 * sin(K * la) * sin(la) + K * cos(la) * cos(K * la);   // w = ((i1*w_col)*i2)*cos(pi t_now/tau)
    u_lev = cos(pi * t_now / tau);
  }
  for (int k = 0; k < NLEVP; k++) {
    double w;
    if (test == 1) w = T->i1[k] * w_col * cl * ct_w * T->i2[k];
    else w = T->i1[k] * w_col * T->i2[k] * u_lev;
    eta[((size_t)e * NLEVP + k) * 16 + p] = T->i3[k] * w;
    if (k < NLEV) {
      const double dpr = pint[k + 1] - pint[k];
      double u, v;
      if (test == 1) { u = u_col + u_lev * T->m1[k]; v = v_col; }
      else { u = u_col; v = T->m1[k] * cl * v_col * T->m2[k] * ct_w; }
      const size_t o = ((size_t)e * NLEV + k) * 16 + p;
      if (dp) dp[o] = dpr;                 // (null: still holding these time-independent values from an earlier step)
      vn0[(((size_t)e * NLEV + k) * 2 + 0) * 16 + p] = u * dpr;
      vn0[(((size_t)e * NLEV + k) * 2 + 1) * 16 + p] = v * dpr;
      if (omega_p) omega_p[o] = 0.0;
    }
  }
}

// initial tracers: Qdp(:,:,:,q,1:2) = Q*dp(hyai,hybi,ps_v=p0) (prim_driver_mod.F90:646-669); checkerboard for the extra
// tracers (dcmip_wrapper_mod.F90:215-243)
template <int = 0>   // (a template only so that two translation units can include this header: tse_stage3.hip)
__global__ void k_dcmip_init(int nelemd, int qsize, int test, const double* __restrict__ lat, const double* __restrict__ lon,
                             const double* __restrict__ zm, const double* __restrict__ pint, const double* __restrict__ dph,
                             double* __restrict__ qdp0, double* __restrict__ qdp1, double* __restrict__ dp3d,
                             double* __restrict__ ps_v) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)nelemd * NLEV * 16) return;
  const int p = (int)(t & 15), k = (int)((t >> 4) % NLEV), e = (int)(t / (NLEV * 16));
  const double lo = lon[(size_t)e * 16 + p], la = lat[(size_t)e * 16 + p];
  DcmipPt m = dcmip_point(test, 0.0, lo, la, zm[k]);
  double term = sin(9. * lo) * sin(9. * la);
  double checker = term < 0. ? 0.0 : 1.0;
  for (int q = 0; q < qsize; q++) {
    double Q = test == 1 ? (q < 4 ? m.q[q] : checker) : (q == 1 ? m.q[1] : checker);
    size_t o = (((size_t)e * qsize + q) * NLEV + k) * 16 + p;
    qdp0[o] = Q * dph[k];
    qdp1[o] = Q * dph[k];
  }
  dp3d[((size_t)e * NLEV + k) * 16 + p] = pint[k + 1] - pint[k];
  if (k == 0) ps_v[(size_t)e * 16 + p] = pint[NLEV];
}

}  // namespace tse
