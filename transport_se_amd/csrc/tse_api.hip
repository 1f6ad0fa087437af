// tse_api.hip -- host side of libtransport_se_hip.so: the C ABI declared in include/transport_se_hip.h.
//
// Owns all device memory for the run (as cuda_mod owns qdp_d etc., reference cuda_mod.F90:74-106), converts the
// reference's edge descriptors (putmapP/getmapP/reverse + Send/RecvCycle slots) into on-device gather tables once
// at init, and sequences the kernels of tse_kernels.h on two HIP streams: the compute stream and -- when the rank has
// neighbour ranks -- a communication stream that carries pack -> RCCL send/recv -> unpack of the rank-boundary columns
// while the compute stream works on the interior elements (the reference's own accelerator seam orders its work the
// same way: cuda_mod.F90:358-401,961-1005; the exchange it replaces is bndry_mod.F90:74-124).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/transport_se_hip.h"
#include "tse_kernels.h"

using namespace tse;

namespace tse {   // tse_stage3.hip: k_advance<2,3> lives in a translation unit of its own (another scheduler strategy)
void launch_advance23(int psz, unsigned blocks, hipStream_t stream, int nelemd, const Dvv_t& D, const GeoPtrs& G, int qsize, double dt, double nu_q,
                      const double* B, const double* lapT, double* C, const double* vn0, const double* dp, const double* divdp,
                      const double* divdp_proj, double* qmin, double* qmax, const double* dp0, const GatherArgs& ga);
}

// TSE_DSS_ON_READ=0 falls back to one DSS pass per stage in the whole-step call (the per-stage API always does that)
static bool dss_on_read() { const char* e = getenv("TSE_DSS_ON_READ"); return !(e && e[0] == '0'); }
// TSE_REMAP_FUSED=0: inside tse_prim_run_subcycle the last step of a cycle runs its own final DSS pass and the remap works in place
static bool remap_fused() { const char* e = getenv("TSE_REMAP_FUSED"); return !(e && e[0] == '0'); }

// Experiment and fault-injection switches (TSE_AB_*: A/B variants, some of them WRONG on purpose; TSE_TEST_*: failures on request for the
// tests; the tse_debug_* entry points at the end of this file) exist only in a build with -DTSE_AB_HOOKS
// (libtransport_se_hip_hooks.so: tools/ab_build.sh, _lib.build_hooks()); the product library does not read them at all.
#ifdef TSE_AB_HOOKS
static const char* hook_env(const char* name) { return getenv(name); }
#else
static const char* hook_env(const char*) { return nullptr; }
#endif

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return 1;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("%s:%d %s: %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); } while (0)
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail("%s:%d %s: %s", __FILE__, __LINE__, #x, ncclGetErrorString(r_)); } while (0)

struct KTimer { double ms = 0; long n = 0; };

// the tables of one patch tiling (tse_kernels.h: Patch<PSZ>): built for the 4 x 4 storage tiling and for every wider block
// shape some kernel was given
struct PatchSet {
  int psz = 0, npatch = 0, np_bnd = 0, np_int = 0;
  int *pslots = nullptr, *plist_bnd = nullptr, *plist_int = nullptr, *pering = nullptr;
  unsigned* pring = nullptr;
  unsigned short* plds = nullptr;
  unsigned char* pnb = nullptr;
};
// the DSS-on-read kernels of the whole-step path, as indices of tse_ctx::kshape
enum { K_ADV1 = 0, K_LAP = 1, K_ADV2 = 2, K_DSS = 3 };

struct tse_ctx {
  int nelemd = 0, qsize = 0, device = 0, rsplit = 3;
  bool remap_alg2 = false;   // control_mod vert_remap_q_alg == 2: piecewise-constant boundary cells in the PPM remap
  double nu_q = 0, ps0 = 0;
  Dvv_t D;
  hipStream_t stream = nullptr;
  // metric + tables
  double *Dinv = nullptr, *metdet = nullptr, *rmetdet = nullptr, *spheremp = nullptr, *rspheremp = nullptr;
  double *hyai = nullptr, *hybi = nullptr, *dp0 = nullptr;
  int2 *dss_tab = nullptr, *send_src = nullptr;
  int *nbr = nullptr, *order = nullptr;
  // state
  double *qlev[2] = {nullptr, nullptr};   // Qdp(:,:,:,:,1) and (:,:,:,:,2): two allocations (place_fields)
  double* q(int tl) const { return qlev[tl - 1]; }
  double *T = nullptr, *B = nullptr, *C = nullptr;   // C: third scratch field (stage-3 output of the whole-step path)
  int place_n = 0, place_sel[5] = {0, 1, 2, 3, 4};   // field placement (place_fields): chunks tried, which try became T, Qdp1, Qdp2, B, C
  double place_bw[32] = {0};                         // their streaming-write GB/s, in the order tried
  double* qorig[2] = {nullptr, nullptr};   // developer experiment: the tracer state's own allocations while it lives in pool chunks
  std::vector<double*> pool;   // developer experiment (tools/placement_probe.py): scratch-sized allocations T, B, C can be re-assigned to
  double *vn0 = nullptr, *dp = nullptr, *divdp = nullptr, *divdp_proj = nullptr, *eta = nullptr, *omega_p = nullptr;
  double *dp3d = nullptr, *ps_v = nullptr, *lvl_tmp = nullptr;
  double *qmin = nullptr, *qmax = nullptr, *qmin2 = nullptr, *qmax2 = nullptr;
  int* bad = nullptr;
  int* bad_host = nullptr; hipEvent_t bad_ev[2] = {nullptr, nullptr};   // page-locked copies of `bad`, one per cycle in flight (tse_prim_run_subcycle)
  int mm_valid = 0;   // time level (1|2) whose element min/max of Q sit in qmin2/qmax2 (emitted by the previous step), 0 = none
  int mm_halo = 0;    // == mm_valid: the neighbour ranks' share of those bounds is already in recvbuf_mm (or on its way: ev_mm)
  hipEvent_t ev_mm = nullptr;   // completion of that prefetched exchange on the communication stream
  // halo: one slot per neighbour rank (Schedule(1)%SendCycle/RecvCycle), entries per slot for the two exchange kinds
  int ncol_send = 0, ncol_recv = 0, nlyr_halo = 0;
  int nmm_send = 0, nmm_recv = 0;              // entries of the compact min/max exchange
  std::vector<int> send_peer, recv_peer, send_len, recv_len;   // kind 0: edge-buffer columns per slot
  std::vector<int> mm_send_len, mm_recv_len;                   // kind 1: (element, direction) pairs per slot
  int2* mm_send_src = nullptr;
  double *sendbuf = nullptr, *recvbuf = nullptr, *sendbuf_mm = nullptr, *recvbuf_mm = nullptr;
  tse_exchange_fn exchange = nullptr; void* exchange_user = nullptr;
  // in-library exchange: RCCL communicator + communication stream; elements that touch another rank / that do not
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  hipStream_t int_stream = nullptr;   // several ranks: the interior launch of a split stage runs here, beside the boundary launch's last blocks (split_stage)
  // the prescribed-wind generator of step n+1 runs beside the neighbour min/max pass that opens the step (tse_prim_run_subcycle): its own
  // stream, forked behind the last launch of step n, joined before the first kernel that reads vn0 / eta_dot_dpdn
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_inputs = nullptr;
  bool inputs_pending = false;
  std::vector<hipEvent_t> sync_events; size_t sync_next = 0;
  int *ord_bnd = nullptr, *ord_int = nullptr;
  int n_bnd = 0, n_int = 0;
  int *rl_all = nullptr, *rl_bnd = nullptr, *rl_int = nullptr;   // the same three element sets in SLOT order: the remap's block lists (k_remap)
  // element patches of the scratch layout (tse_kernels.h): slot = patch*16 + position
  int nslots = 0;
  int* slot_of = nullptr;
  PatchSet pset[3];                        // patch tilings of 16, 24, 32 slots (4x4, 6x4, 8x4 elements); [0] is also the storage order
  int kshape[4] = {16, 16, 16, 16};        // block shape of k_advance<1,1>, k_lap1<1>, k_advance<2,3>, k_dss_patch (patch_shape)
  const PatchSet& set_of(int k) const { return pset[kshape[k] == 32 ? 2 : kshape[k] == 24 ? 1 : 0]; }
  unsigned long long* pperm = nullptr;   // point order inside every slot of the scratch layout
  unsigned char* pexp = nullptr;         // [slot] lines of the slot that hold points read from outside its patch (k_lap1<1> stores only those)
  unsigned* etab = nullptr;              // [e][16][3] entry (within a chunk) of the DSS contributions of a point: the remap's DSS on read (RemapFuse)
  int dss_deferred_n0 = 0;               // != 0: the last tracer step left C pre-DSS for the remap to assemble; value = its n0_qdp (tse_prim_run_subcycle only)
  int2* send_src_s = nullptr;   // the send columns in slot space
  unsigned cse = 0;   // entries (points, halo columns) per chunk of a scratch plane
  bool halo() const { return ncol_send || ncol_recv; }
  // dcmip
  int dcmip_test = 0;
  bool dcmip_static = false;   // dp and omega_p hold the prescribed case's time-independent values (p_i(k+1) - p_i(k); 0, which its DSS leaves 0)
  double *lat = nullptr, *lon = nullptr, *zm = nullptr, *zi = nullptr, *pint = nullptr, *dph = nullptr;
  // page-locked range of the host's elem(:) (host_pin) + timing
  uintptr_t pin_lo = 0, pin_hi = 0;            // host range registered with tse_host_register
  double* stage[2] = {nullptr, nullptr};       // page-locked staging buffers for every other host pointer
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  bool timing = false;
  std::map<std::string, KTimer> timers;
  struct Pending { const char* name; hipEvent_t a, b; };
  std::vector<Pending> pending;       // event pairs recorded on `stream`, resolved lazily (no sync inside the step)
  std::vector<hipEvent_t> free_events;
  double *sink = nullptr;   // write-only dump of k_remap (run-in levels of its segment tasks, surplus tracer slots): 16 columns x 72 levels, then two bounds areas
  double *eta2 = nullptr;   // with lvl_tmp: twin buffers of the level fields (k_dss_lvl writes out of place, then swap)
  bool t_zero_dirty = false;   // the per-stage stage-3 path used T as a plain [e][q][k][p] field (overwrites its zero elements)
  size_t tps = 0;   // plane stride (doubles) of the scratch fields T and B: NCHUNK chunks of (slots, a zero slot, the halo columns)
  Scr scr() const { return Scr{tps, cse}; }
  unsigned zero0() const { return (unsigned)nslots * 16; }          // entry index of the zero slot within a chunk
  unsigned halo0() const { return (unsigned)(nslots + 1) * 16; }    // entry index of halo column 0
  GatherArgs gargs(const PatchSet& P, const int* order_, int nwork_, const int* plist_, int npwork_, const double* var_in = nullptr, int var_in_lev = 0,
                   double* var_out = nullptr, int var_out_lev = 0) const {
    return GatherArgs{scr(), slot_of, order_, nwork_, rspheremp, P.pslots, P.pring, P.plds, plist_, npwork_, var_in, var_in_lev, var_out, var_out_lev, nullptr,
                      P.pering, P.pnb, pperm, nullptr};
  }
  int mm_m() const { return mm_qpad(qsize) * NLEV; }   // entries per element of the bounds arrays (tse_kernels.h: mm_idx)
  size_t lev() const { return (size_t)nelemd * NLEV * 16; }
  size_t trc() const { return lev() * qsize; }
  DcmipTab* dcmip_tab = nullptr;   // level-only factors of the prescribed fields
  double* dvv_d = nullptr;   // device copy of Dvv
  GeoPtrs geo() const { return GeoPtrs{Dinv, metdet, rmetdet, spheremp, rspheremp, dvv_d}; }
};

const char* tse_last_error(void) { return g_err; }

// the element bounds of Qdp(tl)/dp now sit in qmin2/qmax2 (tl = 0: nothing cached); any halo of older bounds is stale
static void set_bounds_cache(tse_ctx* c, int tl) { c->mm_valid = tl; c->mm_halo = 0; }

template <class T>
static int dalloc(T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T) > 0 ? n * sizeof(T) : sizeof(T)));
  return 0;
}
template <class T>
static int upload(T** p, const std::vector<T>& h) {
  if (dalloc(p, h.size())) return 1;
  if (!h.empty()) HIPCHK(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// time a group of launches with HIP events on the launch stream (only when timing is enabled)
static hipEvent_t get_event(tse_ctx* c) {
  if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
struct Scope {
  tse_ctx* c; const char* name; hipEvent_t a = nullptr; hipStream_t st;
  Scope(tse_ctx* c_, const char* n, hipStream_t st_ = nullptr) : c(c_), name(n), st(st_ ? st_ : c_->stream) { if (c->timing) { a = get_event(c); (void)hipEventRecord(a, st); } }
  ~Scope() {
    if (!a) return;
    hipEvent_t b = get_event(c);
    (void)hipEventRecord(b, st);
    c->pending.push_back({name, a, b});
  }
};
static void resolve_timers(tse_ctx* c) {
  if (c->pending.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) {
    float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
    KTimer& t = c->timers[p.name]; t.ms += ms; t.n += 1;
    c->free_events.push_back(p.a); c->free_events.push_back(p.b);
  }
  c->pending.clear();
}

static inline int edge_point(int d, int k) {  // d: 0 W, 1 E, 2 S, 3 N  (edge_mod.F90:407-422)
  switch (d) { case 0: return k * 4 + 0; case 1: return k * 4 + 3; case 2: return k; default: return 12 + k; }
}
static inline int corner_point(int d) {  // d: 4 SW, 5 SE, 6 NW, 7 NE
  switch (d) { case 4: return 0; case 5: return 3; case 6: return 12; default: return 15; }
}

static void gather_strided(std::vector<double>& out, const double* base, size_t stride_bytes, int n, int cnt) {
  out.resize((size_t)n * cnt);
  for (int e = 0; e < n; e++)
    memcpy(&out[(size_t)e * cnt], (const char*)base + (size_t)e * stride_bytes, sizeof(double) * cnt);
}

// streaming pass over n double2: in -> out (in null: write only; out null: read only); every block works through contiguous 16 KB pieces
__global__ __launch_bounds__(256) void k_probe_copy(size_t n, const double2* __restrict__ in, double2* __restrict__ out) {
  const size_t per = 1024;
  for (size_t p = blockIdx.x; p * per < n; p += gridDim.x)
#pragma unroll
    for (int u = 0; u < 4; u++) { const size_t i = p * per + u * 256 + threadIdx.x; if (i < n) { double2 v = in ? in[i] : make_double2(1., 2.); if (out) out[i] = v; else if (v.x == 1.2345e300) ((double2*)in)[i] = v; } }
}

// Where the five tracer-sized fields live.  The 288 GB are not one uniform memory: streaming WRITES into separately allocated 28 GB
// chunks run at 5.5 to 6.9 TB/s depending on the chunk (reads: 6.35 everywhere; tools/region_probe.hip); the rate is a property of the
// allocation (of how its physical memory happens to be put together: the same virtual range freed and allocated again behind a small
// pad writes at another rate; in short bursts all chunks are alike), it is fixed for the life of the allocation -- and the kernels
// follow it:
//  * with a 5.5 TB/s chunk as T (written by stages 1 and 3a) a tracer step takes 2-3 ms longer than with any faster one (k_lap1 14.8
//    instead of 13.4-13.7 ms, k_advance<0,0> 12.6 instead of 12.0; tools/placement_probe.py timed all 60-120 assignments of three out
//    of five or six chunks with the real kernels);
//  * k_dss_patch, which writes Qdp(np1), takes 16.3-17.1 ms into a chunk that streams at 5.9 TB/s or more and 18.5-19.3 ms into one
//    at 5.5-5.7 (tools/dss_probe.py); with both time levels in one allocation it alternated between the two step by step, the
//    first allocation of a process being a slow one nearly always (tools/step_probe.py).
// This was the unexplained "run-to-run modes" of rounds 1 and 2.  So tse_init places the five fields by trial: it allocates field-sized
// chunks (up to 8 held at a time, memory allowing) and times a streaming write into each (3 x 5 ms); as long as fewer than three of the
// held chunks reach TSE_PLACEMENT_GOOD (6000 GB/s) it frees the slowest, allocates a 256 MB pad and tries again -- at most
// TSE_PLACEMENT tries in all (default 20; 0 = off).  The five fastest become T, Qdp(1), Qdp(2), B, C in that order (the first three are
// the roles that matter), the rest and the pads are freed before anything else is allocated.  Pure placement: no bit of any result moves.
// (profiles/r03_ab_placement.txt)
static int place_fields(tse_ctx* c, size_t scr_n, size_t trc) {
  c->place_n = 0;
  const int budget = std::min(32, getenv("TSE_PLACEMENT") ? atoi(getenv("TSE_PLACEMENT")) : 20);   // (place_bw / tse_placement report 32 tries)
  const double good = getenv("TSE_PLACEMENT_GOOD") ? atof(getenv("TSE_PLACEMENT_GOOD")) : 6000.0;
  const size_t chunk = std::max(scr_n, trc);
  double** role[5] = {&c->T, &c->qlev[0], &c->qlev[1], &c->B, &c->C};
  if (budget <= 5 || chunk * 8 < ((size_t)1 << 30)) {   // small fields live in the caches: nothing to choose
    if (dalloc(&c->qlev[0], trc) || dalloc(&c->qlev[1], trc) || dalloc(&c->T, scr_n) || dalloc(&c->B, scr_n) || dalloc(&c->C, scr_n)) return 1;
    return 0;
  }
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  struct Held { double gbs; double* p; int tryno; };
  std::vector<Held> held;
  std::vector<void*> pads;
  const size_t reserve = (size_t)24 << 30;   // what the rest of tse_init allocates (level fields, bounds, halo) and a margin
  auto ngood = [&]() { int n = 0; for (const Held& h : held) n += h.gbs >= good; return n; };
  int rc = 0;
  while (c->place_n < budget) {
    if (held.size() >= 5 && ngood() >= 3) break;
    size_t fr = 0, tot = 0;
    const bool room = held.size() < 5 || (held.size() < 8 && hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= chunk * 8 + reserve);
    if (!room) {   // give the slowest one back and shift what the next allocation gets
      size_t w = 0;
      for (size_t i = 1; i < held.size(); i++) if (held[i].gbs < held[w].gbs) w = i;
      (void)hipFree(held[w].p); held.erase(held.begin() + w);
      void* pad = nullptr;
      if (hipMalloc(&pad, (size_t)256 << 20) == hipSuccess) pads.push_back(pad); else (void)hipGetLastError();
    }
    double* p = nullptr;
    if (hipMalloc((void**)&p, chunk * 8) != hipSuccess) { (void)hipGetLastError(); if (held.size() < 5) rc = 1; break; }
    hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, c->stream, chunk / 2, (const double2*)nullptr, (double2*)p);
    (void)hipEventRecord(a, c->stream);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, c->stream, chunk / 2, (const double2*)nullptr, (double2*)p);
    float ms = 0;
    if (hipEventRecord(b, c->stream) != hipSuccess || hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipFree(p); rc = 1; break; }
    const double gbs = (double)chunk * 8 / (ms / 2) / 1e6;
    if (c->place_n < 32) c->place_bw[c->place_n] = gbs;
    held.push_back({gbs, p, c->place_n++});
  }
  if (held.size() < 5) rc = 1;
  std::stable_sort(held.begin(), held.end(), [](const Held& x, const Held& y) { return x.gbs > y.gbs; });
  for (size_t i = 0; i < held.size(); i++) {
    if (!rc && i < 5) { *role[i] = held[i].p; c->place_sel[i] = held[i].tryno; }
    else (void)hipFree(held[i].p);
  }
  for (void* p : pads) (void)hipFree(p);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  if (rc) {
    // the trial ran out of memory or of luck part-way (a chunk given back and its replacement refused: fragmentation, another
    // process on the device): everything it held is free again -- take the fields as they come, as with TSE_PLACEMENT=0
    c->place_n = 0;
    for (double** r : role) *r = nullptr;
    if (dalloc(&c->qlev[0], trc) || dalloc(&c->qlev[1], trc) || dalloc(&c->T, scr_n) || dalloc(&c->B, scr_n) || dalloc(&c->C, scr_n)) return 1;
    return 0;
  }
  return 0;
}

static int init_impl(tse_ctx* c, const tse_init_args* a) {
  if (a->device >= 0) HIPCHK(hipSetDevice(a->device));
  HIPCHK(hipGetDevice(&c->device));
  c->nelemd = a->nelemd; c->qsize = a->qsize; c->nu_q = a->nu_q; c->ps0 = a->ps0; c->rsplit = a->rsplit;
  c->exchange = a->exchange; c->exchange_user = a->exchange_user;
  c->remap_alg2 = a->vert_remap_q_alg == 2;
  {
    // block shapes of the DSS-on-read kernels (measured defaults); TSE_PATCH_SHAPE="adv1,lap,adv2,dss" overrides them (A/B)
    const char* e = getenv("TSE_PATCH_SHAPE");
    int v[4] = {c->kshape[0], c->kshape[1], c->kshape[2], c->kshape[3]};
    if (e && sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) != 4) return fail("tse_init: TSE_PATCH_SHAPE=\"%s\" (expected four of 16|24|32)", e);
    for (int k = 0; k < 4; k++) {
      if (v[k] != 16 && v[k] != 24 && v[k] != 32) return fail("tse_init: patch shape %d (16, 24 or 32 element slots)", v[k]);
      c->kshape[k] = v[k];
    }
  }
  memcpy(c->D.d, a->Dvv, sizeof c->D.d);
  { std::vector<double> dv(a->Dvv, a->Dvv + 16); if (upload(&c->dvv_d, dv)) return 1; }
  HIPCHK(hipStreamCreate(&c->stream));   // blocking w.r.t. the legacy default stream: see the note above split_stage
  HIPCHK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_inputs, hipEventDisableTiming));
  const int n = a->nelemd;
  std::vector<double> h;
  gather_strided(h, a->Dinv, a->Dinv_stride, n, 64);
  // Dinv(a,b,i,j) in Fortran memory = [p][b][a] -> keep as is: [e][p][4] = {D11, D21, D12, D22}
  if (upload(&c->Dinv, h)) return 1;
  gather_strided(h, a->metdet, a->metdet_stride, n, 16);       if (upload(&c->metdet, h)) return 1;
  gather_strided(h, a->rmetdet, a->rmetdet_stride, n, 16);     if (upload(&c->rmetdet, h)) return 1;
  gather_strided(h, a->spheremp, a->spheremp_stride, n, 16);   if (upload(&c->spheremp, h)) return 1;
  gather_strided(h, a->rspheremp, a->rspheremp_stride, n, 16); if (upload(&c->rspheremp, h)) return 1;
  {
    std::vector<double> v(a->hyai, a->hyai + NLEVP); if (upload(&c->hyai, v)) return 1;
    std::vector<double> w(a->hybi, a->hybi + NLEVP); if (upload(&c->hybi, w)) return 1;
    std::vector<double> d0(NLEV);  // dp0(k), prim_advection_mod.F90:818-819
    for (int k = 0; k < NLEV; k++) d0[k] = (a->hyai[k + 1] - a->hyai[k]) * a->ps0 + (a->hybi[k + 1] - a->hybi[k]) * a->ps0;
    if (upload(&c->dp0, d0)) return 1;
  }

  // ---- edge descriptors -> gather tables -------------------------------------------------------
  int maxcol = 0;
  for (int i = 0; i < n * 8; i++) { if (a->putmapP[i] + 4 > maxcol) maxcol = a->putmapP[i] + 4; if (a->getmapP[i] + 4 > maxcol) maxcol = a->getmapP[i] + 4; }
  for (int s = 0; s < a->nsend; s++) maxcol = std::max(maxcol, a->send_ptrP[s] - 1 + a->send_lengthP[s]);
  for (int s = 0; s < a->nrecv; s++) maxcol = std::max(maxcol, a->recv_ptrP[s] - 1 + a->recv_lengthP[s]);
  std::vector<int> own_e(maxcol, -1), own_p(maxcol, -1), send_idx(maxcol, -1), recv_idx(maxcol, -1);
  std::vector<int> put_start(maxcol, -1), get_start(maxcol, 0), mm_recv_idx(maxcol, -1);   // first column of an edge/corner -> element
  for (int e = 0; e < n; e++)
    for (int d = 0; d < 8; d++) {
      int pm = a->putmapP[e * 8 + d];
      if (pm < 0) continue;
      put_start[pm] = e;
      if (d < 4) {
        for (int k = 0; k < 4; k++) {  // reversal is applied at pack time (edge_mod.F90:443-485)
          int col = pm + (a->reverse[e * 8 + d] ? 3 - k : k);
          own_e[col] = e; own_p[col] = edge_point(d, k);
        }
      } else { own_e[pm] = e; own_p[pm] = corner_point(d); }
    }
  c->ncol_send = 0;
  for (int s = 0; s < a->nsend; s++) {
    c->send_peer.push_back(a->send_peer[s]); c->send_len.push_back(a->send_lengthP[s]);
    for (int i = 0; i < a->send_lengthP[s]; i++) send_idx[a->send_ptrP[s] - 1 + i] = c->ncol_send++;
  }
  c->ncol_recv = 0;
  for (int s = 0; s < a->nrecv; s++) {
    c->recv_peer.push_back(a->recv_peer[s]); c->recv_len.push_back(a->recv_lengthP[s]);
    for (int i = 0; i < a->recv_lengthP[s]; i++) recv_idx[a->recv_ptrP[s] - 1 + i] = c->ncol_recv++;
  }
  // compact min/max exchange: one entry per (element, direction) pair that crosses the rank boundary.  Sender and
  // receiver enumerate the edge/corner start columns of a slot in increasing column order, which is the same sequence
  // on both ranks because the two slots are mirror images (the sender writes where the receiver reads).
  for (int i = 0; i < n * 8; i++) if (a->getmapP[i] >= 0) get_start[a->getmapP[i]] = 1;
  std::vector<int2> mm_src;
  for (int s = 0; s < a->nsend; s++) {
    int cnt = 0;
    for (int i = 0; i < a->send_lengthP[s]; i++) {
      int col = a->send_ptrP[s] - 1 + i;
      if (put_start[col] >= 0) { mm_src.push_back(make_int2(put_start[col], 0)); cnt++; }
    }
    c->mm_send_len.push_back(cnt);
  }
  c->nmm_send = (int)mm_src.size();
  for (int s = 0; s < a->nrecv; s++) {
    int cnt = 0;
    for (int i = 0; i < a->recv_lengthP[s]; i++) {
      int col = a->recv_ptrP[s] - 1 + i;
      if (get_start[col]) { mm_recv_idx[col] = c->nmm_recv++; cnt++; }
    }
    c->mm_recv_len.push_back(cnt);
  }

  std::vector<int2> send_src(c->ncol_send);
  for (int col = 0; col < maxcol; col++)
    if (send_idx[col] >= 0) {
      if (own_e[col] < 0) return fail("tse_init: send column %d is written by no local element", col);
      send_src[send_idx[col]] = make_int2(own_e[col], own_p[col]);
    }
  auto source_of = [&](int col, int2& s) -> int {
    if (recv_idx[col] >= 0) { s = make_int2(-(recv_idx[col] + 2), 0); return 0; }
    if (own_e[col] < 0) return 1;
    s = make_int2(own_e[col], own_p[col]);
    return 0;
  };
  std::vector<int2> tab((size_t)n * 48, make_int2(-1, 0));
  std::vector<int> nbr((size_t)n * 8, -1);
  static const int eorder[4] = {2, 1, 3, 0};  // S, E, N, W  (edge_mod.F90:685-700)
  static const int corder[4] = {4, 5, 7, 6};  // SW, SE, NE, NW (:723-734)
  for (int e = 0; e < n; e++) {
    int cnt[16] = {0};
    for (int t = 0; t < 4; t++) {
      int d = eorder[t], gm = a->getmapP[e * 8 + d];
      if (gm < 0) return fail("tse_init: element %d has no neighbour across edge %d", e, d);
      for (int k = 0; k < 4; k++) {
        int2 s;
        if (source_of(gm + k, s)) return fail("tse_init: element %d edge %d reads column %d that nobody writes", e, d, gm + k);
        int p = edge_point(d, k);
        tab[((size_t)e * 16 + p) * 3 + cnt[p]++] = s;
        if (k == 0) nbr[e * 8 + d] = s.x >= 0 ? s.x : -(mm_recv_idx[gm] + 2);  // remote: entry of the compact min/max exchange
      }
    }
    for (int t = 0; t < 4; t++) {
      int d = corder[t], gm = a->getmapP[e * 8 + d];
      if (gm < 0) continue;
      int2 s;
      if (source_of(gm, s)) return fail("tse_init: element %d corner %d reads column %d that nobody writes", e, d, gm);
      int p = corner_point(d);
      tab[((size_t)e * 16 + p) * 3 + cnt[p]++] = s;
      nbr[e * 8 + d] = s.x >= 0 ? s.x : -(mm_recv_idx[gm] + 2);
    }
  }
  if (upload(&c->dss_tab, tab) || upload(&c->nbr, nbr) || upload(&c->send_src, send_src) || upload(&c->mm_send_src, mm_src)) return 1;
  {
    // Walk order for the DSS kernels.  Each XCD processes a contiguous range of elements (L2 is per XCD); inside
    // the range we follow a greedy neighbour walk over the local element graph (west/east/south/north links) in
    // strips, so that the elements whose edge values a block gathers were touched by the same XCD a few blocks
    // earlier instead of a whole row of the face earlier.  Pure scheduling: results do not depend on it.
    const int S8 = (n + 7) / 8;
    std::vector<int> order(n);
    const int W = getenv("TSE_DSS_STRIP") ? atoi(getenv("TSE_DSS_STRIP")) : 8;
    for (int x = 0; x < 8; x++) {
      const int lo = x * S8, hi = std::min(n, lo + S8);
      if (lo >= hi) continue;
      std::vector<char> used(hi - lo, 0);
      int pos = lo;
      auto in = [&](int e) { return e >= lo && e < hi && !used[e - lo]; };
      for (int seed = lo; seed < hi; seed++) {
        if (used[seed - lo]) continue;
        // strip: from `seed` go east up to W elements (row segment), then continue with the northern neighbours' segment
        int rowstart = seed;
        while (rowstart >= 0 && in(rowstart)) {
          int e = rowstart, cnt = 0, first = e;
          while (e >= 0 && in(e) && cnt < W) { used[e - lo] = 1; order[pos++] = e; cnt++; int ee = nbr[e * 8 + 1]; e = ee; }
          int nn = nbr[first * 8 + 3];   // north of the segment's first element
          rowstart = nn;
        }
      }
    }
    if (W <= 0) for (int e = 0; e < n; e++) order[e] = e;
    if (upload(&c->order, order)) return 1;
  }
  {
    // Boundary-first ordering (the reference's recv_external_indices / recv_internal_indices, cuda_mod.F90:358-401): the
    // elements that own a column of a send slot are computed first in every stage, so that their halo travels while
    // the remaining elements are computed.
    std::vector<char> isb(n, 0);
    for (const int2& s : send_src) isb[s.x] = 1;
    for (const int2& s : mm_src) isb[s.x] = 1;
    std::vector<int> ob, oi;
    for (int e = 0; e < n; e++) (isb[e] ? ob : oi).push_back(e);
    c->n_bnd = (int)ob.size(); c->n_int = (int)oi.size();
    if (upload(&c->ord_bnd, ob) || upload(&c->ord_int, oi)) return 1;
  }

  {
    // Patches: groups of neighbouring elements -- rows of up to PW elements joined by their east links, up to 4 rows joined by
    // the north link of each row's first element (no coordinates are needed and a patch may take any shape next to a cube seam or
    // a rank boundary).  A DSS-on-read block owns one patch: what its slabs need from inside the patch travels through LDS, only
    // the patch's halo ring comes from global memory.  Elements are taken in host order, so the patches of a full face tile it
    // from its south-west corner.  The 4 x 4 tiling (set 0) is also the STORAGE order of the scratch fields (slot = patch * 16 +
    // position); the wider shapes (6 x 4, 8 x 4: sets 1, 2) are built only for the kernels that were given them (patch_shape).
    auto ring_key = [](const int2& t) { return t.x >= 0 ? (long)t.x * 16 + t.y : -(long)(-(t.x + 2)) - 1; };
    std::vector<char> isb(n, 0);   // elements that touch another rank
    for (const int2& q : send_src) isb[q.x] = 1;
    for (const int2& q : mm_src) isb[q.x] = 1;
    // TSE_BOUNDARY_STRIPS=1: rank-boundary elements in patches of their own (below).  Off by default: on 8 ranks of ne120 it halves
    // the first launch of a stage (25.8 -> 11.8 % of the patches) but the ragged tiling behind the band costs 3 % of a rank's
    // step (12.6 -> 13.0 ms in the loopback rehearsal, profiles/r03_ab_boundary_bands.txt), and the first launch only matters
    // where an exchange outlasts the interior launch -- 0.7 ms of xGMI transfer against 1.3-2.2 ms of interior work here.
    const bool strips = getenv("TSE_BOUNDARY_STRIPS") && atoi(getenv("TSE_BOUNDARY_STRIPS")) != 0;
    auto build = [&](int psz, std::vector<std::vector<int>>& patches, std::vector<int>& pid) {
      const int pw = psz / 4, nrmax = patch_nrmax(psz);
      pid.assign(n, -1);
      auto ring_size = [&](const std::vector<int>& cand, int me) {
        std::vector<long> refs;
        for (int e : cand)
          for (int i = 0; i < 48; i++) {
            const int2 t = tab[(size_t)e * 48 + i];
            if (t.x == -1) continue;
            if (t.x >= 0 && pid[t.x] == me) continue;                       // inside the candidate (marked below)
            refs.push_back(ring_key(t));
          }
        std::sort(refs.begin(), refs.end());
        return (int)(std::unique(refs.begin(), refs.end()) - refs.begin());
      };
      auto ering_size = [&](const std::vector<int>& cand, int me) {   // distinct elements (local or received) around the candidate
        std::vector<long> refs;
        for (int e : cand)
          for (int d = 0; d < 8; d++) {
            const int nb = nbr[e * 8 + d];
            if (nb == -1 || (nb >= 0 && pid[nb] == me)) continue;
            refs.push_back(nb);
          }
        std::sort(refs.begin(), refs.end());
        return (int)(std::unique(refs.begin(), refs.end()) - refs.begin());
      };
      // Rank-boundary elements first, as patches of their own: a stage's first launch covers the patches that own a column of a
      // send slot (split_stage), and with the regular tiling a 4 x 4 patch is such a patch as soon as one of its elements is -- a
      // quarter of all patches on 8 ranks, for 6 % of the elements.  So up to half a patch of boundary elements is strung together
      // along the boundary (element by element over the 8-neighbourhood), and the patch is then filled with the elements right
      // behind them (edge neighbours of its members), as far as the halo ring and the element ring allow: a two-deep band along the
      // rank boundary, full patches (strips of boundary elements alone left a third of the lanes empty and cost 6 % of a step),
      // and a first launch of about twice the boundary elements' share.
      if (strips) {
        auto try_add = [&](std::vector<int>& cand, int me, int el) {
          pid[el] = me; cand.push_back(el);
          if (ring_size(cand, me) <= nrmax && ering_size(cand, me) <= NER) return true;
          pid[el] = -1; cand.pop_back();
          return false;
        };
        for (int seed = 0; seed < n; seed++) {
          if (!isb[seed] || pid[seed] >= 0) continue;
          const int me = (int)patches.size();
          std::vector<int> cand{seed};
          pid[seed] = me;
          bool grew = true;
          while (grew && (int)cand.size() < psz / 2) {   // the chain of boundary elements
            grew = false;
            for (int back = (int)cand.size() - 1; back >= 0 && !grew; back--)
              for (int d = 0; d < 8 && !grew; d++) {
                const int nb = nbr[cand[back] * 8 + d];
                if (nb >= 0 && isb[nb] && pid[nb] < 0) grew = try_add(cand, me, nb);
              }
          }
          grew = true;
          while (grew && (int)cand.size() < psz) {       // the elements behind it
            grew = false;
            for (size_t i = 0; i < cand.size() && !grew; i++)
              for (int d = 0; d < 4 && !grew; d++) {
                const int nb = nbr[cand[i] * 8 + d];
                if (nb >= 0 && pid[nb] < 0) grew = try_add(cand, me, nb);
              }
          }
          patches.push_back(cand);
        }
      }
      for (int seed = 0; seed < n; seed++) {
        if (pid[seed] >= 0) continue;
        const int me = (int)patches.size();
        // fewer rows, then narrower rows, until the halo ring and the element ring fit the tables (one element always does)
        for (int maxrows = 4, width = pw;; ) {
          std::vector<int> cand;
          int rowstart = seed;
          for (int r = 0; r < maxrows && rowstart >= 0 && pid[rowstart] < 0; r++) {
            int e = rowstart, cnt = 0;
            const int first = e;
            while (e >= 0 && pid[e] < 0 && cnt < width) { pid[e] = me; cand.push_back(e); cnt++; e = nbr[e * 8 + 1]; }   // east
            rowstart = nbr[first * 8 + 3];                                                                             // north
          }
          if ((ring_size(cand, me) <= nrmax && ering_size(cand, me) <= NER) || (maxrows == 1 && width == 1)) { patches.push_back(cand); break; }
          for (int e : cand) pid[e] = -1;
          if (maxrows > 1) maxrows--; else width--;
        }
      }
    };
    std::vector<std::vector<int>> patches[3];
    std::vector<int> pid[3];
    static const int shape_psz[3] = {16, 24, 32};
    bool want[3] = {true, false, false};
    for (int k = 0; k < 4; k++) for (int si = 0; si < 3; si++) if (c->kshape[k] == shape_psz[si]) want[si] = true;
    for (int si = 0; si < 3; si++) if (want[si]) build(shape_psz[si], patches[si], pid[si]);
    // ---- storage: the 4 x 4 tiling
    c->nslots = (int)patches[0].size() * PS;
    std::vector<int> slot_of(n, -1);
    for (size_t pi = 0; pi < patches[0].size(); pi++)
      for (size_t i = 0; i < patches[0][pi].size(); i++) slot_of[patches[0][pi][i]] = (int)pi * PS + (int)i;
    c->cse = (unsigned)(c->nslots + 1) * 16 + (unsigned)std::max(0, c->ncol_recv);
    // Point order inside every slot (tse_kernels.h: ppos).  An edge of an element is READ FROM OUTSIDE when the neighbour across it
    // belongs to another patch of some tiling in use (that patch's halo ring) or to another rank (the pack kernel); such an edge
    // gets a 128-byte line of its own, in the order S, N, W, E.  An edge that shares a corner point with an edge placed before
    // it (the corner elements of a patch export two edges) brings only its remaining points into a fresh line: it then costs
    // its reader two lines.  The points nobody reads from outside fill what is left.
    // TSE_AB_FIXED_PERM=1: the former fixed perimeter-first order (A/B).
    std::vector<unsigned long long> pperm((size_t)c->nslots, 0x67895FEA4DCB3210ULL);
    // lines of a slot that k_lap1<1> must store: 1 + the last line that holds a point some patch's halo ring (of any tiling in use) or a
    // send column reads -- taken from those tables themselves below, so that it covers corner-only readers and irregular patches too;
    // the per-slot order packs the exported edges into the first lines, so this is a quarter of the field on average
    std::vector<unsigned char> pexp((size_t)c->nslots, 0);
    if (!(hook_env("TSE_AB_FIXED_PERM") && atoi(hook_env("TSE_AB_FIXED_PERM")))) {
      static const int edge_dir[4] = {2, 3, 0, 1};   // S, N, W, E as direction indices (west, east, south, north = 0..3)
      for (int e = 0; e < n; e++) {
        int pos_of[16]; bool placed[16] = {false};
        int line = 0;
        for (int t = 0; t < 4; t++) {
          const int d = edge_dir[t], nb = nbr[e * 8 + d];
          bool outside = nb <= -2;
          for (int si = 0; si < 3; si++) if (want[si] && nb >= 0 && pid[si][nb] != pid[si][e]) outside = true;
          if (!outside) continue;
          int cnt = 0;
          for (int k = 0; k < 4; k++) { const int pt = edge_point(d, k); if (!placed[pt]) { placed[pt] = true; pos_of[pt] = line * 4 + cnt++; } }
          if (cnt) line++;
        }
        bool used[16] = {false};
        for (int pt = 0; pt < 16; pt++) if (placed[pt]) used[pos_of[pt]] = true;
        int f = 0;
        for (int pt = 0; pt < 16; pt++) if (!placed[pt]) { while (used[f]) f++; pos_of[pt] = f; used[f] = true; }
        unsigned long long w = 0;
        for (int pt = 0; pt < 16; pt++) w |= (unsigned long long)pos_of[pt] << (4 * pt);
        pperm[slot_of[e]] = w;
      }
    }
    const bool ab_noring = hook_env("TSE_AB_NORING") && atoi(hook_env("TSE_AB_NORING"));   // A/B: no halo-ring loads at all (WRONG results; bounds what the ring costs)
    // ---- the tables of every tiling in use
    for (int si = 0; si < 3; si++) {
      if (!want[si]) continue;
      PatchSet& P = c->pset[si];
      const int psz = shape_psz[si], nrmax = patch_nrmax(psz);
      const std::vector<std::vector<int>>& pt = patches[si];
      P.psz = psz; P.npatch = (int)pt.size();
      const size_t nts = (size_t)P.npatch * psz;   // table slots
      std::vector<int> pslots(nts, -1), tslot_of(n, -1);
      for (int pi = 0; pi < P.npatch; pi++)
        for (size_t i = 0; i < pt[pi].size(); i++) { pslots[(size_t)pi * psz + i] = pt[pi][i]; tslot_of[pt[pi][i]] = pi * psz + (int)i; }
      std::vector<unsigned> pring((size_t)P.npatch * nrmax, c->zero0());
      const int lds_ring = psz * 20, lds_zero = lds_ring + nrmax;   // Patch<psz>::LDS_RING, LDS_ZERO
      std::vector<unsigned short> plds(nts * 48, (unsigned short)lds_zero);
      for (int pi = 0; pi < P.npatch; pi++) {
        std::map<long, int> ring;   // source -> ring entry
        for (size_t i = 0; i < pt[pi].size(); i++) {
          const int e = pt[pi][i];
          for (int k = 0; k < 48; k++) {
            const int2 t = tab[(size_t)e * 48 + k];
            unsigned short ent = (unsigned short)lds_zero;
            if (t.x >= 0 && pid[si][t.x] == pi) ent = (unsigned short)lds_own_entry(tslot_of[t.x] - pi * psz, t.y);
            else if (t.x != -1) {
              const long key = ring_key(t);
              auto it = ring.find(key);
              if (it == ring.end()) {
                if ((int)ring.size() >= nrmax) return fail("tse_init: halo ring of patch %d exceeds %d entries", pi, nrmax);
                it = ring.emplace(key, (int)ring.size()).first;
                if (!ab_noring)
                  pring[(size_t)pi * nrmax + it->second] = t.x >= 0 ? (unsigned)slot_of[t.x] * 16 + ppos(pperm[slot_of[t.x]], t.y) : c->halo0() + (unsigned)(-(t.x + 2));
              }
              ent = (unsigned short)(lds_ring + it->second);
            }
            plds[((size_t)pi * psz + i) * 48 + k] = ent;
          }
        }
      }
      for (unsigned ent : pring)   // what this tiling's halo rings read from the slots
        if (ent < (unsigned)c->nslots * 16) pexp[ent / 16] = std::max<unsigned char>(pexp[ent / 16], (unsigned char)((ent % 16) / 4 + 1));
      // element ring and neighbour entries of every patch, for the bounds image of the stage-3 kernel (k_advance<2,3>)
      std::vector<int> pering((size_t)P.npatch * NER, 0);
      std::vector<unsigned char> pnb(nts * 8, 255);
      for (int pi = 0; pi < P.npatch; pi++) {
        std::map<int, int> ring;   // element (or -(received entry) - 2) -> ring entry
        for (int r = 0; r < NER; r++) pering[(size_t)pi * NER + r] = pt[pi][0];   // unused entries: any valid element
        for (size_t i = 0; i < pt[pi].size(); i++) {
          const int e = pt[pi][i];
          for (int d = 0; d < 8; d++) {
            const int nb = nbr[e * 8 + d];
            if (nb == -1) continue;
            if (nb >= 0 && pid[si][nb] == pi) { pnb[((size_t)pi * psz + i) * 8 + d] = (unsigned char)(tslot_of[nb] - pi * psz); continue; }
            auto it = ring.find(nb);
            if (it == ring.end()) {
              if ((int)ring.size() >= NER) return fail("tse_init: patch %d has more than %d elements around it", pi, NER);
              it = ring.emplace(nb, (int)ring.size()).first;
              pering[(size_t)pi * NER + it->second] = nb >= 0 ? nb : n + (-(nb + 2));
            }
            pnb[((size_t)pi * psz + i) * 8 + d] = (unsigned char)(psz + it->second);
          }
        }
      }
      // rank-boundary patches first, as the elements above
      std::vector<int> pb, pin;
      for (int pi = 0; pi < P.npatch; pi++) {
        bool b = false;
        for (int e : pt[pi]) b = b || isb[e];
        (b ? pb : pin).push_back(pi);
      }
      P.np_bnd = (int)pb.size(); P.np_int = (int)pin.size();
      if (upload(&P.pslots, pslots) || upload(&P.pring, pring) || upload(&P.plds, plds) || upload(&P.pering, pering) || upload(&P.pnb, pnb) ||
          upload(&P.plist_bnd, pb) || upload(&P.plist_int, pin)) return 1;
    }
    std::vector<int2> send_s(send_src);
    for (int2& t : send_s) { t.x = slot_of[t.x]; t.y = ppos(pperm[t.x], t.y); }   // {slot, position within the slot}
    for (const int2& t : send_s) pexp[t.x] = std::max<unsigned char>(pexp[t.x], (unsigned char)(t.y / 4 + 1));   // what the pack kernel reads
    if (upload(&c->slot_of, slot_of) || upload(&c->send_src_s, send_s) || upload(&c->pperm, pperm) || upload(&c->pexp, pexp)) return 1;
    // the same contributions per ELEMENT as global entries of a chunk, for the remap that assembles the last DSS of a cycle on read
    std::vector<unsigned> etab((size_t)n * 48);
    for (size_t i = 0; i < etab.size(); i++) {
      const int2 t = tab[i];
      etab[i] = t.x >= 0 ? (unsigned)slot_of[t.x] * 16 + ppos(pperm[slot_of[t.x]], t.y) : t.x == -1 ? c->zero0() : c->halo0() + (unsigned)(-(t.x + 2));
    }
    if (upload(&c->etab, etab)) return 1;
    // the remap's block lists: all / rank-boundary / interior elements in slot order (patch by patch)
    {
      std::vector<int> by_slot(n);
      for (int e = 0; e < n; e++) by_slot[e] = e;
      std::sort(by_slot.begin(), by_slot.end(), [&](int x, int y) { return slot_of[x] < slot_of[y]; });
      std::vector<int> rb, ri;
      for (int e : by_slot) (isb[e] ? rb : ri).push_back(e);
      if (upload(&c->rl_all, by_slot) || upload(&c->rl_bnd, rb) || upload(&c->rl_int, ri)) return 1;
    }
  }

  // ---- state -------------------------------------------------------------------------------------
  const size_t lev = c->lev(), trc = c->trc();
  // scratch fields T, B: one plane per tracer = NCHUNK chunks of (element slots + one all-zero slot, the target of empty DSS
  // contributions, + the received halo columns, which DSS-on-read reads from there); see tse_kernels.h
  c->tps = ((size_t)NCHUNK * c->cse * CL + 15) / 16 * 16;
  if (c->tps < (size_t)n * 16 * NLEV) return fail("tse_init: scratch plane smaller than a tracer plane");
  const size_t scr_n = (size_t)(c->qsize + 1) * c->tps;   // qsize tracer planes + the plane of the stage's extra DSS variable
  if (place_fields(c, scr_n, trc)) return fail("tse_init: out of device memory (%zu B per tracer field, 5 fields)", trc * 8);
  HIPCHK(hipMemset(c->T, 0, scr_n * 8)); HIPCHK(hipMemset(c->B, 0, scr_n * 8)); HIPCHK(hipMemset(c->C, 0, scr_n * 8));
  if (dalloc(&c->vn0, 2 * lev) || dalloc(&c->dp, lev) || dalloc(&c->divdp, lev) || dalloc(&c->divdp_proj, lev) ||
      dalloc(&c->eta, (size_t)n * NLEVP * 16) || dalloc(&c->omega_p, lev) || dalloc(&c->dp3d, lev) || dalloc(&c->ps_v, (size_t)n * 16) ||
      dalloc(&c->lvl_tmp, lev) || dalloc(&c->sink, (size_t)NLEV * 16 + 2 * (size_t)c->mm_m()) || dalloc(&c->eta2, (size_t)n * NLEVP * 16)) return 1;
  // (behind the local elements: room for the received bounds of the compact exchange, see k_unpack_minmax)
  const size_t mm = (size_t)(n + c->nmm_recv) * c->mm_m();
  if (dalloc(&c->qmin, mm) || dalloc(&c->qmax, mm) || dalloc(&c->qmin2, mm) || dalloc(&c->qmax2, mm) || dalloc(&c->bad, 1)) return 1;
  HIPCHK(hipMemset(c->qlev[0], 0, trc * 8)); HIPCHK(hipMemset(c->qlev[1], 0, trc * 8));
  HIPCHK(hipMemset(c->vn0, 0, 2 * lev * 8)); HIPCHK(hipMemset(c->dp, 0, lev * 8)); HIPCHK(hipMemset(c->divdp, 0, lev * 8));
  HIPCHK(hipMemset(c->divdp_proj, 0, lev * 8)); HIPCHK(hipMemset(c->eta, 0, (size_t)n * NLEVP * 16 * 8));
  HIPCHK(hipMemset(c->omega_p, 0, lev * 8)); HIPCHK(hipMemset(c->dp3d, 0, lev * 8)); HIPCHK(hipMemset(c->ps_v, 0, (size_t)n * 16 * 8));
  HIPCHK(hipMemset(c->qmin, 0, mm * 8)); HIPCHK(hipMemset(c->qmax, 0, mm * 8));
  HIPCHK(hipMemset(c->qmin2, 0, mm * 8)); HIPCHK(hipMemset(c->qmax2, 0, mm * 8));   // (the pad tracers of the layout are exchanged and reduced like the others)
  HIPCHK(hipMemset(c->bad, 0, sizeof(int)));
  HIPCHK(hipHostMalloc((void**)&c->bad_host, 2 * sizeof(int), hipHostMallocDefault));
  c->bad_host[0] = c->bad_host[1] = 0;
  for (hipEvent_t& e : c->bad_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // Halo buffers.  The reference sizes one aliased buffer for 3*qsize*nlev layers (prim_advection_mod.F90:488); here the
  // tracer DSS (qsize*nlev + nlev layers per column) and the element-constant min/max exchange (2*qsize*nlev per
  // (element, direction) pair) have their own buffers, so that the two exchanges of stage 3 can be in flight together.
  c->nlyr_halo = c->qsize * NLEV + NLEV;
  if (c->halo()) {
    const size_t m2 = (size_t)2 * c->mm_m();
    if (dalloc(&c->sendbuf, (size_t)std::max(1, c->ncol_send) * c->nlyr_halo) || dalloc(&c->recvbuf, (size_t)std::max(1, c->ncol_recv) * c->nlyr_halo) ||
        dalloc(&c->sendbuf_mm, (size_t)std::max(1, c->nmm_send) * m2) || dalloc(&c->recvbuf_mm, (size_t)std::max(1, c->nmm_recv) * m2)) return 1;
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically lowest = greatest priority
    HIPCHK(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, hi));
    if (!(hook_env("TSE_AB_SPLIT_STREAMS") && !atoi(hook_env("TSE_AB_SPLIT_STREAMS"))))   // A/B: 0 = both launches on the compute stream
      HIPCHK(hipStreamCreateWithPriority(&c->int_stream, hipStreamNonBlocking, lo));
    c->sync_events.assign(16, nullptr);
    for (hipEvent_t& e : c->sync_events) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_mm, hipEventDisableTiming));
  }
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipFuncSetAttribute((const void*)k_remap<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RemapLds)));
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

int tse_init(tse_ctx** out, const tse_init_args* a) {
  if (!out || !a) return fail("tse_init: null argument");
  *out = nullptr;
  if (a->limiter_option != 8) return fail("tse_init: only limiter_option=8 is supported (got %d)", a->limiter_option);
  if (a->nelemd <= 0 || a->qsize <= 0) return fail("tse_init: nelemd=%d qsize=%d", a->nelemd, a->qsize);
  if (a->vert_remap_q_alg < 0 || a->vert_remap_q_alg > 2)
    return fail("tse_init: vert_remap_q_alg=%d (0|1: mirrored ghost cells, 2: piecewise-constant boundary cells; control_mod.F90:61-66)", a->vert_remap_q_alg);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("tse_init: no HIP device (this library has no CPU fallback)");
#ifdef TSE_AB_HOOKS
  { static bool said = false; if (!said) { fprintf(stderr, "transport_se_hip: this is the TSE_AB_HOOKS build (A/B switches and fault injection enabled) -- not the product library\n"); said = true; } }
#endif
  tse_ctx* c = new tse_ctx();
  if (init_impl(c, a)) {   // release whatever was allocated before the failure (the message in g_err survives)
    tse_finalize(c);
    return 1;
  }
  *out = c;
  return 0;
}

void tse_finalize(tse_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
  if (c->pin_hi) (void)hipHostUnregister((void*)c->pin_lo);
  if (c->bad_host) (void)hipHostFree(c->bad_host);
  for (hipEvent_t e : c->bad_ev) if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; i++) { if (c->stage[i]) (void)hipHostFree(c->stage[i]); if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]); }
  void* ptrs[] = {c->dcmip_tab, c->dvv_d, c->Dinv, c->metdet, c->rmetdet, c->spheremp, c->rspheremp, c->hyai, c->hybi, c->dp0, c->dss_tab, c->send_src,
                  c->nbr, c->mm_send_src, c->qorig[0] ? c->qorig[0] : c->qlev[0], c->qorig[0] ? c->qorig[1] : c->qlev[1], c->vn0, c->dp, c->divdp, c->divdp_proj, c->eta, c->omega_p, c->dp3d, c->ps_v,
                  c->lvl_tmp, c->eta2, c->sink, c->order, c->qmin, c->qmax, c->qmin2, c->qmax2, c->bad, c->lat, c->lon, c->zm, c->zi, c->pint, c->dph,
                  c->sendbuf, c->recvbuf, c->sendbuf_mm, c->recvbuf_mm, c->ord_bnd, c->ord_int, c->slot_of, c->send_src_s, c->pperm, c->pexp, c->etab, c->rl_all, c->rl_bnd, c->rl_int};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  if (c->pool.empty()) { for (double* p : {c->T, c->B, c->C}) if (p) (void)hipFree(p); }
  else for (double* p : c->pool) if (p) (void)hipFree(p);
  for (PatchSet& P : c->pset) {
    void* tp[] = {P.pslots, P.plist_bnd, P.plist_int, P.pering, P.pring, P.plds, P.pnb};
    for (void* p : tp) if (p) (void)hipFree(p);
  }
  resolve_timers(c);
  for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->sync_events) if (e) (void)hipEventDestroy(e);
  if (c->ev_mm) (void)hipEventDestroy(c->ev_mm);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->int_stream) { (void)hipStreamSynchronize(c->int_stream); (void)hipStreamDestroy(c->int_stream); }
  if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_inputs) (void)hipEventDestroy(c->ev_inputs);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int tse_synchronize(tse_ctx* c) {
  if (c->comm_stream) HIPCHK(hipStreamSynchronize(c->comm_stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int tse_halo_layout(tse_ctx* c, int* ns, int* nr) { if (ns) *ns = c->ncol_send; if (nr) *nr = c->ncol_recv; return 0; }
int tse_halo_minmax_layout(tse_ctx* c, int* send_len, int* recv_len) {
  for (size_t i = 0; i < c->mm_send_len.size(); i++) send_len[i] = c->mm_send_len[i];
  for (size_t i = 0; i < c->mm_recv_len.size(); i++) recv_len[i] = c->mm_recv_len[i];
  return 0;
}
int tse_boundary_layout(tse_ctx* c, int* nb, int* ni) { if (nb) *nb = c->n_bnd; if (ni) *ni = c->n_int; return 0; }
// field placement: chunks tried at init (0: no choice was made), their streaming-write GB/s in the order tried (the first 32), and
// which try became T, Qdp(1), Qdp(2), B, C
int tse_placement(tse_ctx* c, int* ntried, double* write_gbs /* [32] */, int* chosen /* [5] */) {
  if (ntried) *ntried = std::min(c->place_n, 32);
  if (write_gbs) for (int i = 0; i < 32; i++) write_gbs[i] = i < c->place_n ? c->place_bw[i] : 0.0;
  if (chosen) for (int r = 0; r < 5; r++) chosen[r] = c->place_sel[r];
  return 0;
}
// patches of the storage tiling that touch another rank (launched first in every stage) and that do not
int tse_patch_layout(tse_ctx* c, int* np_boundary, int* np_interior) {
  if (np_boundary) *np_boundary = c->pset[0].np_bnd;
  if (np_interior) *np_interior = c->pset[0].np_int;
  return 0;
}
int tse_invalidate_cache(tse_ctx* c) { set_bounds_cache(c, 0); c->dcmip_static = false; return 0; }

// ---- RCCL communicator ----------------------------------------------------------------------------
static const char* rccl_path() {
  static char buf[512] = "";
  if (!buf[0]) {
    Dl_info info;
    if (dladdr((void*)&ncclGetVersion, &info) && info.dli_fname) strncpy(buf, info.dli_fname, sizeof buf - 1);
    else strcpy(buf, "?");
  }
  return buf;
}
int tse_comm_unique_id(void* id_out) {
  static_assert(sizeof(ncclUniqueId) == TSE_COMM_ID_BYTES, "ncclUniqueId size");
  if (!id_out) return fail("tse_comm_unique_id: null argument");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return 0;
}
// Everything tse_comm_init can find wrong WITHOUT talking to the other ranks.  ncclCommInitRank is a blocking collective: a rank
// that returned early from tse_comm_init would leave its peers inside the bootstrap for ever, so a host first calls this on
// every rank, agrees on the outcome over its own control plane (MPI_Allreduce, a gloo all_gather), and enters tse_comm_init
// only if every rank is ready (driver.PrimRun, cuda_mod_hip.F90).
int tse_comm_precheck(tse_ctx* c, int rank, int nranks) {
  if (!c) return fail("tse_comm_precheck: null context");
  if (c->comm) return fail("tse_comm_precheck: communicator already initialised");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail("tse_comm_precheck: rank %d of %d", rank, nranks);
  for (int p : c->send_peer) if (p < 0 || p >= nranks) return fail("tse_comm_precheck: send peer %d (this is rank %d of %d)", p, rank, nranks);
  for (int p : c->recv_peer) if (p < 0 || p >= nranks) return fail("tse_comm_precheck: recv peer %d (this is rank %d of %d)", p, rank, nranks);
  if (hook_env("TSE_TEST_FAIL_PRECHECK_RANK") && atoi(hook_env("TSE_TEST_FAIL_PRECHECK_RANK")) == rank)   // tests: one rank alone is not ready
    return fail("tse_comm_precheck: rank %d fails on request (TSE_TEST_FAIL_PRECHECK_RANK)", rank);
  if (c->halo() && !c->comm_stream) return fail("tse_comm_precheck: no communication stream");
  HIPCHK(hipSetDevice(c->device));
  int rt = 0;
  NCCLCHK(ncclGetVersion(&rt));
  // The library was compiled against rccl.h NCCL_MAJOR.NCCL_MINOR.NCCL_PATCH; the process runs whichever librccl.so.1 was loaded
  // first (under Python: the copy bundled with torch, which ships no rccl.h to build against -- see _lib.py: one RCCL per process).
  // What is used of RCCL is eleven entry points (ncclGetVersion, ncclGetErrorString, ncclGetUniqueId, ncclCommInitRank,
  // ncclCommUserRank, ncclCommCount, ncclGroupStart/End, ncclSend, ncclRecv, ncclCommAbort/Destroy), ncclDouble and the 128-byte
  // ncclUniqueId: unchanged through all of 2.x, and every one of them is EXECUTED against the runtime of the process by the loopback
  // tests (tests/test_gpu_rccl_exchange.py: RCCL 2.26.6 from torch under headers of 2.27.7).  Policy:
  //   * another major, or a runtime older than point-to-point send/recv (2.7): refused;
  //   * a runtime OLDER than the headers (the Python case): accepted, and said once on stderr with both versions and the path --
  //     refusing it would silently turn every multi-GPU run under Python into a host-staged one; TSE_RCCL_STRICT=1 refuses it
  //     (a host that wants header == runtime, e.g. the Fortran seam linking /opt/rocm/lib/librccl.so, gets exactly that);
  //   * a runtime newer than the headers: accepted silently (RCCL keeps old entry points).
  if (rt / 10000 != NCCL_MAJOR || rt < 20700)
    return fail("tse_comm_precheck: RCCL runtime %d.%d.%d (%s) cannot serve a library built with the headers of %d.%d.%d", rt / 10000, rt / 100 % 100,
                rt % 100, rccl_path(), NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
  if (rt < NCCL_VERSION_CODE) {
    const char* strict = getenv("TSE_RCCL_STRICT");
    if (strict && atoi(strict))
      return fail("tse_comm_precheck: RCCL runtime %d.%d.%d (%s) is older than the headers the library was built with (%d.%d.%d) and TSE_RCCL_STRICT is set",
                  rt / 10000, rt / 100 % 100, rt % 100, rccl_path(), NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
    static bool said = false;
    if (!said && rank == 0) {
      fprintf(stderr, "transport_se_hip: note: RCCL runtime %d.%d.%d (%s) is older than the build headers %d.%d.%d; the entry points used are common to "
                      "both (TSE_RCCL_STRICT=1 refuses this)\n", rt / 10000, rt / 100 % 100, rt % 100, rccl_path(), NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
      said = true;
    }
  }
  return 0;
}
int tse_comm_init(tse_ctx* c, const void* id_in, int rank, int nranks) {
  if (!c || !id_in) return fail("tse_comm_init: null argument");
  if (tse_comm_precheck(c, rank, nranks)) return 1;
  ncclUniqueId id;
  memcpy(&id, id_in, sizeof id);
  ncclComm_t cm = nullptr;
  NCCLCHK(ncclCommInitRank(&cm, nranks, id, rank));   // on failure the context keeps its callback route (c->comm stays null)
  c->comm = cm;
  return 0;
}
// which RCCL this process resolved: version code of the runtime (ncclGetVersion), of the headers the library was built with, and
// the path of the shared object that provides ncclGetVersion (dladdr)
int tse_comm_version(int* runtime, int* built, char* path, size_t path_len) {
  int rt = 0;
  NCCLCHK(ncclGetVersion(&rt));
  if (runtime) *runtime = rt;
  if (built) *built = NCCL_VERSION_CODE;
  if (path && path_len) { strncpy(path, rccl_path(), path_len - 1); path[path_len - 1] = 0; }
  return 0;
}
int tse_comm_abort(tse_ctx* c) {
  if (!c) return fail("tse_comm_abort: null context");
  if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
  return 0;
}
int tse_comm_info(tse_ctx* c, int* rank, int* nranks) {
  int r = 0, n = 1;
  if (c->comm) { NCCLCHK(ncclCommUserRank(c->comm, &r)); NCCLCHK(ncclCommCount(c->comm, &n)); }
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  return 0;
}

// ---- host <-> device copies ---------------------------------------------------------------------
// The host keeps elem(:) as a strided array of structures.  A host that has declared that array with tse_host_register
// (page-locked once, for the life of the context) gets every field copy as ONE 2-D DMA straight between elem(:) and the dense
// device array (row = one element's field, pitch = sizeof(element_t)): no staging copy at all.  Any other host pointer --
// temporaries, numpy arrays -- goes through the library's own page-locked staging buffers, two of them, so that packing the
// next block of elements overlaps the DMA of the previous one.  (Page-locking memory the caller may free behind the
// library's back is not safe, so it is never done implicitly.)
int tse_host_register(tse_ctx* c, void* base, size_t bytes) {
  static const double limit_gb = getenv("TSE_PIN_LIMIT_GB") ? atof(getenv("TSE_PIN_LIMIT_GB")) : 64.0;
  if (!base || !bytes) return fail("tse_host_register: null range");
  if (c->pin_hi) return fail("tse_host_register: a host range is already registered");
  if ((double)bytes > limit_gb * 1073741824.0) return 0;   // staged copies (TSE_PIN_LIMIT_GB raises the limit)
  if (hipHostRegister(base, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }   // not fatal: staged copies
  c->pin_lo = (uintptr_t)base; c->pin_hi = c->pin_lo + bytes;
  return 0;
}
static const size_t STAGE_BYTES = (size_t)32 << 20;
// `cnt` doubles per element between host (element stride `stride` bytes) and a dense device array [nelemd][cnt_dev]; asynchronous
// on the compute stream for registered host memory, complete on return otherwise
static int copy_field(tse_ctx* c, double* dev, size_t cnt_dev, void* host, size_t stride, size_t cnt, bool to_device) {
  if (!host) return 0;
  const int n = c->nelemd;
  if (stride < cnt * 8 && n > 1) return fail("field copy: element stride %zu smaller than the field (%zu bytes)", stride, cnt * 8);
  const uintptr_t a = (uintptr_t)host, b = a + (size_t)(n - 1) * stride + cnt * 8;
  if (c->pin_hi && a >= c->pin_lo && b <= c->pin_hi) {
    const size_t pitch = n > 1 ? stride : cnt * 8;
    if (to_device) HIPCHK(hipMemcpy2DAsync(dev, cnt_dev * 8, host, pitch, cnt * 8, n, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(hipMemcpy2DAsync(host, pitch, dev, cnt_dev * 8, cnt * 8, n, hipMemcpyDeviceToHost, c->stream));
    return 0;
  }
  if (!c->stage[0]) {
    HIPCHK(hipHostMalloc((void**)&c->stage[0], STAGE_BYTES, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->stage[1], STAGE_BYTES, hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&c->stage_ev[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->stage_ev[1], hipEventDisableTiming));
  }
  const int per = (int)std::max<size_t>(1, STAGE_BYTES / (cnt * 8));   // elements per block
  if (to_device) {
    int bi = 0;
    for (int e0 = 0; e0 < n; e0 += per, bi ^= 1) {
      const int m = std::min(per, n - e0);
      HIPCHK(hipEventSynchronize(c->stage_ev[bi]));   // the DMA that last read this buffer
      for (int e = 0; e < m; e++) memcpy(c->stage[bi] + (size_t)e * cnt, (const char*)host + (size_t)(e0 + e) * stride, cnt * 8);
      HIPCHK(hipMemcpy2DAsync(dev + (size_t)e0 * cnt_dev, cnt_dev * 8, c->stage[bi], cnt * 8, cnt * 8, m, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipEventRecord(c->stage_ev[bi], c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
  } else {
    int bi = 0, pend_e0 = -1, pend_m = 0;
    auto drain = [&](int buf, int e0, int m) {
      (void)hipEventSynchronize(c->stage_ev[buf]);
      for (int e = 0; e < m; e++) memcpy((char*)host + (size_t)(e0 + e) * stride, c->stage[buf] + (size_t)e * cnt, cnt * 8);
    };
    for (int e0 = 0; e0 < n; e0 += per, bi ^= 1) {
      const int m = std::min(per, n - e0);
      HIPCHK(hipMemcpy2DAsync(c->stage[bi], cnt * 8, dev + (size_t)e0 * cnt_dev, cnt_dev * 8, cnt * 8, m, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipEventRecord(c->stage_ev[bi], c->stream));
      if (pend_e0 >= 0) drain(bi ^ 1, pend_e0, pend_m);   // unpack the previous block while this one travels
      pend_e0 = e0; pend_m = m;
    }
    if (pend_e0 >= 0) drain(bi ^ 1, pend_e0, pend_m);
  }
  return 0;
}
int tse_copy_qdp_h2d(tse_ctx* c, const double* q1, size_t stride, int qsize_d, int nt) {
  set_bounds_cache(c, 0);
  if (nt < 1 || nt > 2 || qsize_d < c->qsize) return fail("tse_copy_qdp_h2d: nt=%d qsize_d=%d", nt, qsize_d);
  const size_t per = (size_t)c->qsize * NLEV * 16;   // Qdp(np,np,nlev,qsize_d,2): time level nt starts qsize_d*nlev*16 doubles in
  if (copy_field(c, c->q(nt), per, (char*)q1 + (size_t)(nt - 1) * qsize_d * NLEV * 16 * 8, stride, per, true)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int tse_copy_qdp_d2h(tse_ctx* c, double* q1, size_t stride, int qsize_d, int nt) {
  if (nt < 1 || nt > 2 || qsize_d < c->qsize) return fail("tse_copy_qdp_d2h: nt=%d qsize_d=%d", nt, qsize_d);
  const size_t per = (size_t)c->qsize * NLEV * 16;
  if (copy_field(c, c->q(nt), per, (char*)q1 + (size_t)(nt - 1) * qsize_d * NLEV * 16 * 8, stride, per, false)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
static int put_level(tse_ctx* c, double* dev, const double* host, size_t stride, size_t cnt_dev, size_t cnt_host) {
  return copy_field(c, dev, cnt_dev, (void*)host, stride, std::min(cnt_dev, cnt_host), true);
}
static int get_level(tse_ctx* c, const double* dev, double* host, size_t stride, size_t cnt_dev, size_t cnt_host) {
  return copy_field(c, (double*)dev, cnt_dev, host, stride, std::min(cnt_dev, cnt_host), false);
}
int tse_set_derived(tse_ctx* c, const double* vn0, size_t s0, const double* dp, size_t s1, const double* eta, size_t s2,
                    const double* omega_p, size_t s3) {
  // vn0(np,np,2,nlev) in Fortran memory is [k][c][p]: the device layout
  if (put_level(c, c->vn0, vn0, s0, 2 * NLEV * 16, 2 * NLEV * 16)) return 1;
  if (dp) set_bounds_cache(c, 0);   // bounds were formed with the previous dp
  if (dp || omega_p) c->dcmip_static = false;
  if (put_level(c, c->dp, dp, s1, NLEV * 16, NLEV * 16)) return 1;
  if (put_level(c, c->eta, eta, s2, NLEVP * 16, NLEVP * 16)) return 1;
  if (put_level(c, c->omega_p, omega_p, s3, NLEV * 16, NLEV * 16)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));   // the caller may reuse its arrays
  return 0;
}
int tse_set_divdp(tse_ctx* c, const double* divdp, size_t s0, const double* divdp_proj, size_t s1) {
  if (put_level(c, c->divdp, divdp, s0, NLEV * 16, NLEV * 16)) return 1;
  if (put_level(c, c->divdp_proj, divdp_proj, s1, NLEV * 16, NLEV * 16)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int tse_get_derived(tse_ctx* c, double* divdp_proj, size_t s1, double* eta, size_t s2, double* omega_p, size_t s3, double* divdp,
                    size_t s4, double* dp3d, size_t s5, double* ps_v, size_t s6) {
  if (get_level(c, c->divdp_proj, divdp_proj, s1, NLEV * 16, NLEV * 16)) return 1;
  if (get_level(c, c->eta, eta, s2, NLEVP * 16, NLEV * 16)) return 1;  // levels 1:nlev only are DSS'd (:835-837)
  if (get_level(c, c->omega_p, omega_p, s3, NLEV * 16, NLEV * 16)) return 1;
  if (get_level(c, c->divdp, divdp, s4, NLEV * 16, NLEV * 16)) return 1;
  if (get_level(c, c->dp3d, dp3d, s5, NLEV * 16, NLEV * 16)) return 1;
  if (get_level(c, c->ps_v, ps_v, s6, 16, 16)) return 1;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}
int tse_get_qminmax(tse_ctx* c, double* qmin, double* qmax) {
  const size_t mm = (size_t)c->nelemd * c->mm_m();
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<double> h(mm);
  double* outs[2] = {qmin, qmax};
  const double* devs[2] = {c->qmin, c->qmax};
  for (int a = 0; a < 2; a++) {
    if (!outs[a]) continue;
    HIPCHK(hipMemcpy(h.data(), devs[a], mm * 8, hipMemcpyDeviceToHost));
    for (int e = 0; e < c->nelemd; e++)          // device layout [e][k / CL][q][k % CL] (tse_kernels.h: mm_idx) -> [e][q][k]
      for (int q = 0; q < c->qsize; q++)
        for (int k = 0; k < NLEV; k++)
          outs[a][((size_t)e * c->qsize + q) * NLEV + k] = h[(((size_t)e * NCHUNK + k / CL) * mm_qpad(c->qsize) + q) * CL + (k % CL)];
  }
  return 0;
}

// ---- the path -----------------------------------------------------------------------------------
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return fail("%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

int tse_compute_divdp(tse_ctx* c) {
  Scope s(c, "level");
  hipLaunchKernelGGL(k_divdp<>, dim3(flat_blocks(c->nelemd)), dim3(FLAT_THREADS), 0, c->stream, c->nelemd, c->D, c->geo(), c->vn0, c->divdp, c->divdp_proj);
  LAUNCH_CHECK();
  return 0;
}

// bndry_exchangeV on the packed rank-boundary columns, ordered on stream `st` (no-op on one rank).
// kind 0: sendbuf/recvbuf, one entry per edge-buffer column; kind 1: the compact min/max buffers.
// With a communicator: one grouped RCCL receive + send per neighbour slot enqueued on `st`, no host synchronisation
// (bndry_mod.F90:74-112 posts all sends and receives, then waits).  Otherwise the host's callback, after `st` has drained.
static int halo_exchange(tse_ctx* c, int nlyr, int kind, hipStream_t st) {
  if (!c->halo()) return 0;
  double* sb = kind ? c->sendbuf_mm : c->sendbuf;
  double* rb = kind ? c->recvbuf_mm : c->recvbuf;
  if (c->comm) {
    const std::vector<int>& ls = kind ? c->mm_send_len : c->send_len;
    const std::vector<int>& lr = kind ? c->mm_recv_len : c->recv_len;
    NCCLCHK(ncclGroupStart());
    size_t off = 0;
    for (size_t i = 0; i < lr.size(); i++) {
      if (lr[i]) NCCLCHK(ncclRecv(rb + off * nlyr, (size_t)lr[i] * nlyr, ncclDouble, c->recv_peer[i], c->comm, st));
      off += lr[i];
    }
    off = 0;
    for (size_t i = 0; i < ls.size(); i++) {
      if (ls[i]) NCCLCHK(ncclSend(sb + off * nlyr, (size_t)ls[i] * nlyr, ncclDouble, c->send_peer[i], c->comm, st));
      off += ls[i];
    }
    NCCLCHK(ncclGroupEnd());
    return 0;
  }
  if (!c->exchange) return fail("halo exchange: this rank has neighbour ranks but neither tse_comm_init was called nor an exchange callback given");
  HIPCHK(hipStreamSynchronize(st));
  if (c->exchange(c->exchange_user, sb, rb, nlyr, kind)) return fail("exchange callback failed");
  return 0;
}

// ---- pack / unpack launches (on stream st) ----
// scratch field (T, B or C) -> sendbuf layers [0, nlyr) of nlyr_halo; nlyr = qsize*nlev (tracers) or qsize*nlev + nlev (tracers and
// the plane of the stage's extra variable behind them: the reference's edgeAdv_p1 message, prim_advection_mod.F90:497,911-919)
// 32-bit work-item indices in the pack / unpack kernels
static int halo_items(size_t tot, unsigned* blocks) {
  if (tot >= ((size_t)1 << 32)) return fail("halo pack: %zu work items", tot);
  *blocks = (unsigned)((tot + 255) / 256);
  return 0;
}
static int pack_tracers(tse_ctx* c, hipStream_t st, const double* scratch, int nlyr_halo, int nlyr = 0) {
  const int nq = nlyr ? nlyr : c->qsize * NLEV;
  if (!c->ncol_send) return 0;
  unsigned nb;
  if (halo_items((size_t)c->ncol_send * (nq / CL), &nb)) return 1;
  hipLaunchKernelGGL(k_pack_scratch<>, dim3(nb), dim3(256), 0, st, c->ncol_send, nq / CL, c->send_src_s, scratch, c->sendbuf, nlyr_halo, c->scr());
  LAUNCH_CHECK();
  return 0;
}
// spheremp*var -> sendbuf layers [qsize*nlev, +nlev).  eta_dot_dpdn carries nlev+1 levels per element; levels 1:nlev are exchanged (:835-837)
static int pack_var(tse_ctx* c, hipStream_t st, const double* var, int var_levels) {
  const int nq = c->qsize * NLEV;
  if (!c->ncol_send || !var) return 0;
  unsigned nb;
  if (halo_items((size_t)c->ncol_send * NLEV, &nb)) return 1;
  hipLaunchKernelGGL(k_pack<>, dim3(nb), dim3(256), 0, st, c->ncol_send, NLEV, c->send_src, var, c->spheremp, c->sendbuf, nq + NLEV, nq, var_levels);
  LAUNCH_CHECK();
  return 0;
}
static int pack_minmax(tse_ctx* c, hipStream_t st, const double* qmin = nullptr, const double* qmax = nullptr) {
  const int m = c->mm_m();
  if (!c->nmm_send) return 0;
  unsigned nb;
  if (halo_items((size_t)c->nmm_send * (m / 2), &nb)) return 1;
  hipLaunchKernelGGL(k_pack_minmax<>, dim3(nb), dim3(256), 0, st, c->nmm_send, m, c->mm_send_src,
                     qmin ? qmin : (const double*)c->qmin, qmax ? qmax : (const double*)c->qmax, c->sendbuf_mm, 2 * m, 0);
  LAUNCH_CHECK();
  return 0;
}
// DSS on read: copy the received tracer halo behind the planes of the scratch field the next slab kernel gathers from
static int unpack_halo(tse_ctx* c, hipStream_t st, double* field, int nlyr_halo, int nlyr = 0) {
  if (!c->ncol_recv) return 0;
  const int nq = nlyr ? nlyr : c->qsize * NLEV;   // layers to copy: the tracer planes, or also the extra variable's plane
  unsigned nb;
  if (halo_items((size_t)c->ncol_recv * (nq / CL), &nb)) return 1;
  hipLaunchKernelGGL(k_unpack_halo<>, dim3(nb), dim3(256), 0, st, c->ncol_recv, nq / CL, c->recvbuf, nlyr_halo, field, c->scr(), c->halo0());
  LAUNCH_CHECK();
  return 0;
}

// received element bounds -> behind the local elements of qmin/qmax (read by the stage-3 kernel through its element ring)
static int unpack_minmax(tse_ctx* c, hipStream_t st) {
  const int m = c->mm_m();
  if (!c->nmm_recv) return 0;
  unsigned nb;
  if (halo_items((size_t)c->nmm_recv * (m / 2), &nb)) return 1;
  hipLaunchKernelGGL(k_unpack_minmax<>, dim3(nb), dim3(256), 0, st, c->nmm_recv, m, (const double*)c->recvbuf_mm, c->qmin + (size_t)c->nelemd * m,
                     c->qmax + (size_t)c->nelemd * m);
  LAUNCH_CHECK();
  return 0;
}

// min/max over the <= 8 neighbours of qmin/qmax (double-buffered on the device); the halo part must have arrived
static int nbr_minmax_kernel(tse_ctx* c) {
  const int m = c->mm_m();
  {
    Scope s(c, "minmax");
    hipLaunchKernelGGL(k_nbr_minmax_patch<8>, dim3(nbr_patch_blocks(c->pset[0].npatch, c->qsize)), dim3(512), 0, c->stream, c->pset[0].npatch, c->qsize, c->nbr,
                       c->pset[0].pslots, c->slot_of, c->qmin, c->qmax, c->qmin2, c->qmax2, c->recvbuf_mm, 2 * m);
    LAUNCH_CHECK();
  }
  std::swap(c->qmin, c->qmin2); std::swap(c->qmax, c->qmax2);
  return 0;
}
// neighbor_minmax (viscosity_mod.F90:748-816), everything on the compute stream (per-stage API)
static int neighbor_minmax(tse_ctx* c) {
  if (pack_minmax(c, c->stream)) return 1;
  if (halo_exchange(c, 2 * c->mm_m(), 1, c->stream)) return 1;
  return nbr_minmax_kernel(c);
}

// DSS of the extra level variable of a stage (divdp_proj / eta_dot_dpdn / omega_p): rspheremp*DSS(spheremp*var), remote
// contributions from recvbuf layers [qsize*nlev, +nlev) (prim_advection_mod.F90:911-919,943-957)
static int dss_level_var(tse_ctx* c, double** varp, int var_levels) {
  if (!varp) return 0;
  const int nq = c->qsize * NLEV;
  Scope s(c, "level");
  // out of place into the field's twin buffer (the source must stay intact while neighbours read it), then swap the two
  double** twin = var_levels == NLEV ? &c->lvl_tmp : &c->eta2;
  hipLaunchKernelGGL(k_dss_lvl<>, dim3(8 * dss_blocks_per_xcd<LVL_UNITS>(c->nelemd)), dim3(DSS_FLAT_THREADS), 0, c->stream, c->nelemd, c->dss_tab,
                     c->rspheremp, c->spheremp, *varp, var_levels, *twin, var_levels, c->recvbuf, nq + NLEV, nq, c->order);
  LAUNCH_CHECK();
  std::swap(*varp, *twin);
  return 0;
}
// tracer DSS pass src (scratch layout, halo columns filled) -> dst (standard layout), optionally fused with qdp_time_avg and the
// next step's bounds; over all patches (npwork < 0) or over the patch list of a split launch
// run f(std::integral_constant<int, PSZ>) for the block shape psz
template <class F>
static int with_shape(int psz, F f) {
  if (psz == 32) return f(std::integral_constant<int, 32>{});
  if (psz == 24) return f(std::integral_constant<int, 24>{});
  return f(std::integral_constant<int, 16>{});
}
static int dss_tracer_launch(tse_ctx* c, const double* src, double* dst, const double* Qn0_avg, const int* plist, int npwork,
                             double* var_out = nullptr, int var_out_lev = 0) {
  if (!npwork) return 0;
  const PatchSet& P = c->set_of(K_DSS);
  const GatherArgs ga = c->gargs(P, nullptr, c->nelemd, plist, npwork, nullptr, 0, var_out, var_out_lev);
  const dim3 grid(patch_blocks(npwork));
  with_shape(P.psz, [&](auto psz) {
    constexpr int Z = decltype(psz)::value;
    if (Qn0_avg)
      hipLaunchKernelGGL((k_dss_patch<1, Z>), grid, dim3(Patch<Z>::THREADS), 0, c->stream, c->qsize, src, dst, Qn0_avg, (const double*)c->dp, c->qmin2, c->qmax2, ga);
    else
      hipLaunchKernelGGL((k_dss_patch<0, Z>), grid, dim3(Patch<Z>::THREADS), 0, c->stream, c->qsize, src, dst, (const double*)nullptr, (const double*)nullptr,
                         (double*)nullptr, (double*)nullptr, ga);
    return 0;
  });
  LAUNCH_CHECK();
  return 0;
}
static int dss_tracer_pass(tse_ctx* c, const double* src, double* dst, const double* Qn0_avg, double* var_out = nullptr, int var_out_lev = 0) {
  Scope s(c, "dss");
  return dss_tracer_launch(c, src, dst, Qn0_avg, nullptr, c->set_of(K_DSS).npatch, var_out, var_out_lev);
}

// One euler_step of the per-stage API (prim_advection_mod.F90:667-970): every stage ends with a tracer DSS pass, because
// Qdp(np1) must exist after each call; all work and the halo exchanges are ordered on the compute stream.
static int euler_step_impl(tse_ctx* c, int np1_qdp, int n0_qdp, double dt, int DSSopt, int rhs, bool fuse_avg, int avg_n0, bool fused_mm) {
  if (np1_qdp < 1 || np1_qdp > 2 || n0_qdp < 1 || n0_qdp > 2) return fail("euler_step: bad time levels %d %d", np1_qdp, n0_qdp);
  if (rhs < 0 || rhs > 2) return fail("euler_step: rhs_multiplier=%d", rhs);
  double* Qn0 = c->q(n0_qdp);
  double* Qnp1 = c->q(np1_qdp);
  double** var = DSSopt == 1 ? &c->eta : DSSopt == 2 ? &c->omega_p : DSSopt == 3 ? &c->divdp_proj : nullptr;
  const int var_levels = DSSopt == 1 ? NLEVP : NLEV;
  const int nq = c->qsize * NLEV;
  const dim3 grid(flat_blocks(c->nelemd)), blk(FLAT_THREADS);
  const GatherArgs plain = c->gargs(c->pset[0], nullptr, c->nelemd, nullptr, 0);
  if (rhs == 0) {
    if (fused_mm && c->mm_valid == n0_qdp) {
      // the previous step's last kernel (final DSS or remap) already left the element min/max of Qdp(n0)/dp in qmin2/qmax2
      std::swap(c->qmin, c->qmin2); std::swap(c->qmax, c->qmax2);
    } else {
      Scope s(c, "minmax");
      hipLaunchKernelGGL(k_qminmax<>, grid, blk, 0, c->stream, c->nelemd, c->qsize, 0.0, Qn0, c->dp, c->divdp_proj, c->qmin, c->qmax);
      LAUNCH_CHECK();
    }
    set_bounds_cache(c, 0);
    if (neighbor_minmax(c)) return 1;
    Scope s(c, "advance0");
    hipLaunchKernelGGL(k_advance<0>, grid, blk, 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, dt, c->nu_q, Qn0, (const double*)nullptr, c->T, c->vn0,
                       c->dp, c->divdp, c->divdp_proj, c->qmin, c->qmax, c->dp0, plain);
    LAUNCH_CHECK();
  } else if (rhs == 1) {
    Scope s(c, "advance1");
    hipLaunchKernelGGL(k_advance<1>, grid, blk, 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, dt, c->nu_q, Qn0, (const double*)nullptr, c->T, c->vn0,
                       c->dp, c->divdp, c->divdp_proj, c->qmin, c->qmax, c->dp0, plain);
    LAUNCH_CHECK();
  } else {
    c->t_zero_dirty = true;   // below, T receives rspheremp*DSS(lap) in the plain tracer layout
    {
      Scope s(c, "lap");
      hipLaunchKernelGGL(k_lap1<0>, grid, blk, 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, 2 * dt, Qn0, c->B, c->dp, c->divdp_proj, c->qmin, c->qmax, plain);
      LAUNCH_CHECK();
    }
    // biharmonic_wk_scalar_minmax: DSS(lap1) (+ min/max exchange) -> T = rspheremp*DSS(lap1).  The reference's message is
    // (lap, Qmin, Qmax) = 3*qsize*nlev layers (viscosity_mod.F90:389-391); here the Laplacian and the bounds travel separately.
    if (pack_tracers(c, c->stream, c->B, nq)) return 1;
    if (halo_exchange(c, nq, 0, c->stream)) return 1;
    if (unpack_halo(c, c->stream, c->B, nq)) return 1;
    if (dss_tracer_pass(c, c->B, c->T, nullptr)) return 1;
    if (neighbor_minmax(c)) return 1;
    Scope s(c, "advance2");
    hipLaunchKernelGGL(k_advance<2>, grid, blk, 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, dt, c->nu_q, Qn0, c->T, c->B, c->vn0, c->dp, c->divdp,
                       c->divdp_proj, c->qmin, c->qmax, c->dp0, plain);
    LAUNCH_CHECK();
  }
  double* pre = rhs == 2 ? c->B : c->T;
  const double* avg = fuse_avg ? c->q(avg_n0) : nullptr;
  // edgeVpack(Qdp) + edgeVpack(spheremp*DSSvar) -> bndry_exchangeV -> edgeVunpack + rspheremp  (:911-960)
  if (pack_tracers(c, c->stream, pre, nq + NLEV)) return 1;
  if (var && pack_var(c, c->stream, *var, var_levels)) return 1;
  if (halo_exchange(c, nq + NLEV, 0, c->stream)) return 1;
  if (unpack_halo(c, c->stream, pre, nq + NLEV)) return 1;
  if (dss_tracer_pass(c, pre, Qnp1, avg)) return 1;
  return dss_level_var(c, var, var_levels);
}

int tse_euler_step(tse_ctx* c, int np1_qdp, int n0_qdp, double dt, int DSSopt, int rhs_multiplier) {
  set_bounds_cache(c, 0);
  return euler_step_impl(c, np1_qdp, n0_qdp, dt, DSSopt, rhs_multiplier, false, 0, false);
}

int tse_qdp_time_avg(tse_ctx* c, int rkstage, int n0_qdp, int np1_qdp) {
  set_bounds_cache(c, 0);
  Scope s(c, "avg");
  size_t n = c->trc();
  hipLaunchKernelGGL(k_time_avg<>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, c->stream, n, rkstage,
                     c->q(n0_qdp), c->q(np1_qdp));
  LAUNCH_CHECK();
  return 0;
}

// ---- whole-step path: DSS on read + boundary-first overlap ------------------------------------------
// Stages 1 and 2 leave their pre-DSS scratch (T, then B) un-DSS'd and the next stage's slab kernel assembles
// rspheremp*DSS(.) while reading it; the stage-3 Laplacian is handed over the same way.
//
// On several ranks every slab kernel is launched twice: over the elements that touch another rank (`ord_bnd`), then over
// the rest (`ord_int`).  Between the two launches the communication stream is handed the pack -> exchange -> unpack of
// the stage, so the halo travels over xGMI while the interior elements are computed; the compute stream waits for it
// only before the next stage's first launch.  With an RCCL communicator nothing blocks the host.  With the callback
// form of the seam the host has to drain the packs and run the callback; that is done after the interior launch has
// been queued.  Whether the exchange then overlaps the interior launch depends on the callback: one that moves the slots on
// streams of its own (or over MPI from host memory) does; the Python callbacks of driver.HaloExchange copy on the legacy
// default stream, which waits for this (blocking) compute stream -- their exchange runs behind the interior launch.  (The
// compute stream stays a blocking stream on purpose: the synchronous hipMemcpy / hipMemset calls of the set-up and of the
// operator-level entry points are ordered against it by the default-stream rule.)
static hipEvent_t next_sync_event(tse_ctx* c) {
  hipEvent_t e = c->sync_events[c->sync_next];
  c->sync_next = (c->sync_next + 1) % c->sync_events.size();
  return e;
}

// the part of the local mesh one launch covers: everything (single rank), the elements / patches that touch another rank, the rest
struct Work { const int* order; int nwork; const PatchSet* P; const int* plist; int npwork; };
static Work work_of(const tse_ctx* c, int part, int kidx) {
  const PatchSet& P = c->set_of(kidx);
  if (part == 0) return Work{nullptr, c->nelemd, &P, nullptr, P.npatch};
  if (part == 1) return Work{c->ord_bnd, c->n_bnd, &P, P.plist_bnd, P.np_bnd};
  return Work{c->ord_int, c->n_int, &P, P.plist_int, P.np_int};
}

// done_out == nullptr: the compute stream waits for the communication work before it goes on; otherwise the event that marks
// its completion is handed back and whoever consumes the halo waits for it (the prefetched bounds exchange)
template <class Launch, class CommWork>
static int split_stage(tse_ctx* c, const char* timer, int kidx /* whose patch tiling the launches walk */, Launch launch /* (Work) */,
                       CommWork comm_work /* () on c->comm_stream */, hipEvent_t* done_out = nullptr) {
  Scope s(c, timer);
  if (!c->halo()) return launch(work_of(c, 0, kidx));
  // The two launches touch disjoint elements and depend on the same predecessors, not on each other: the interior launch goes to a
  // stream of its own (lower priority) that waits for what the compute stream has seen so far, so its first blocks fill the CUs the
  // boundary launch's last, partial round of blocks leaves idle (a rank's boundary launch is 1-3 rounds of blocks); the compute stream
  // goes on when both are done.
  hipEvent_t evA = nullptr;
  if (c->int_stream) { evA = next_sync_event(c); HIPCHK(hipEventRecord(evA, c->stream)); }
  if (launch(work_of(c, 1, kidx))) return 1;
  hipEvent_t evB = next_sync_event(c), evC = done_out ? c->ev_mm : next_sync_event(c);
  HIPCHK(hipEventRecord(evB, c->stream));
  HIPCHK(hipStreamWaitEvent(c->comm_stream, evB, 0));
  if (c->comm) { if (comm_work()) return 1; HIPCHK(hipEventRecord(evC, c->comm_stream)); }
  if (c->int_stream) {
    HIPCHK(hipStreamWaitEvent(c->int_stream, evA, 0));
    std::swap(c->stream, c->int_stream);   // (the launch closures launch on c->stream)
    const int rc = launch(work_of(c, 2, kidx));
    std::swap(c->stream, c->int_stream);
    if (rc) return 1;
    hipEvent_t evI = next_sync_event(c);
    HIPCHK(hipEventRecord(evI, c->int_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, evI, 0));
  } else if (launch(work_of(c, 2, kidx))) return 1;
  if (!c->comm) { if (comm_work()) return 1; HIPCHK(hipEventRecord(evC, c->comm_stream)); }
  if (done_out) *done_out = evC;
  else HIPCHK(hipStreamWaitEvent(c->stream, evC, 0));
  return 0;
}

// prefetch: the caller knows that the next thing to happen to Qdp(np1) is the next tracer step (no remap in between), so the
// bounds exchange that step would start with is started here, under the interior part of the last kernel
// the step's inputs (vn0, eta_dot_dpdn; dp, omega_p on the first step) are being written on the auxiliary stream: wait for them
static int join_inputs(tse_ctx* c) {
  if (!c->inputs_pending) return 0;
  HIPCHK(hipStreamWaitEvent(c->stream, c->ev_inputs, 0));
  c->inputs_pending = false;
  return 0;
}
// defer_dss: the caller launches the remap next and lets IT assemble the final DSS + time average on read (remap_launch): no
// k_dss_patch<1> here, Qdp(np1) is not written by this call
static int advec_dss_on_read(tse_ctx* c, double dts /* stage dt = dt/2 */, int n0_qdp, int np1_qdp, bool prefetch, bool defer_dss = false) {
  double* Qn0 = c->q(n0_qdp);
  double* Qnp1 = c->q(np1_qdp);
  const int nq = c->qsize * NLEV;
  const dim3 blk(FLAT_THREADS);
  hipStream_t cs = c->comm_stream;
  // the stage's extra DSS variable travels as plane qsize of the stage's scratch output and is assembled on read by the next
  // kernel (var_in / var_out of GatherArgs): divdp_proj with stage 1, eta_dot_dpdn with stage 2, omega_p with stage 3
  auto gargs = [&](const Work& w, const double* vin = nullptr, int vin_lev = 0, double* vout = nullptr, int vout_lev = 0) {
    return c->gargs(*w.P, w.order, w.nwork, w.plist, w.npwork, vin, vin_lev, vout, vout_lev);
  };
  const int nqv = nq + NLEV;   // layers of a tracer halo message with the extra variable behind the tracers

  // ---- stage 1 (rhs_multiplier 0, DSS extra = divdp_proj): bounds, neighbour min/max, advance Qdp(n0) -> T
  const bool halo_ready = c->halo() && c->mm_valid == n0_qdp && c->mm_halo == n0_qdp;   // prefetched by the previous step / remap
  if (c->mm_valid != n0_qdp && join_inputs(c)) return 1;   // k_qminmax reads dp
  if (c->mm_valid == n0_qdp) {
    // the previous step's last kernel (final DSS or remap) already left the element min/max of Qdp(n0)/dp in qmin2/qmax2
    std::swap(c->qmin, c->qmin2); std::swap(c->qmax, c->qmax2);
  } else {
    Scope s(c, "minmax");
    hipLaunchKernelGGL(k_qminmax<>, dim3(flat_blocks(c->nelemd)), blk, 0, c->stream, c->nelemd, c->qsize, 0.0, Qn0, c->dp, c->divdp_proj, c->qmin, c->qmax);
    LAUNCH_CHECK();
  }
  set_bounds_cache(c, 0);
  // (divdp = div(vn0) has no pass of its own here: stage 1's kernel forms it for its own slab and stores it for the later stages)
  if (halo_ready) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_mm, 0));
  } else if (c->halo()) {
    hipEvent_t ev0 = next_sync_event(c), evM = next_sync_event(c);
    HIPCHK(hipEventRecord(ev0, c->stream));
    HIPCHK(hipStreamWaitEvent(cs, ev0, 0));
    if (pack_minmax(c, cs) || halo_exchange(c, 2 * c->mm_m(), 1, cs)) return 1;
    HIPCHK(hipEventRecord(evM, cs));
    HIPCHK(hipStreamWaitEvent(c->stream, evM, 0));
  }
  if (nbr_minmax_kernel(c)) return 1;
  if (join_inputs(c)) return 1;   // (the wind generator ran beside the neighbour min/max pass)
  if (split_stage(c, "advance0", K_ADV1,
        [&](Work w) -> int {
          if (!w.nwork) return 0;
          GatherArgs ga = gargs(w, c->divdp_proj, NLEV);
          ga.divdp_out = c->divdp;
          hipLaunchKernelGGL(k_advance<0>, dim3(flat_blocks(w.nwork)), blk, 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, dts, c->nu_q, (const double*)Qn0,
                             (const double*)nullptr, c->T, c->vn0, c->dp, c->divdp, c->divdp_proj, c->qmin, c->qmax, c->dp0, ga);
          LAUNCH_CHECK(); return 0; },
        [&]() -> int { return pack_tracers(c, cs, c->T, nqv, nqv) || halo_exchange(c, nqv, 0, cs) || unpack_halo(c, cs, c->T, nqv, nqv); })) return 1;

  // ---- stage 2 (rhs_multiplier 1, DSS extra = eta_dot_dpdn): T (+) edges -> B
  if (split_stage(c, "advance1", K_ADV1,
        [&](Work w) -> int {
          if (!w.npwork) return 0;
          with_shape(w.P->psz, [&](auto psz) {
            constexpr int Z = decltype(psz)::value;
            hipLaunchKernelGGL((k_advance<1, 1, true, Z>), dim3(patch_blocks(w.npwork)), dim3(Patch<Z>::THREADS), 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, dts,
                               c->nu_q, (const double*)c->T, (const double*)nullptr, c->B, c->vn0, c->dp, c->divdp, c->divdp_proj, c->qmin, c->qmax, c->dp0,
                               gargs(w, c->eta, NLEVP, c->divdp_proj, NLEV));
            return 0; });
          LAUNCH_CHECK(); return 0; },
        [&]() -> int { return pack_tracers(c, cs, c->B, nqv, nqv) || halo_exchange(c, nqv, 0, cs) || unpack_halo(c, cs, c->B, nqv, nqv); })) return 1;

  // ---- stage 3 (rhs_multiplier 2, DSS extra = omega_p)
  // 3a: B (+) edges -> first Laplacian (pre-DSS) of the stage-2 tracers in T, element min/max (the DSS'd tracers themselves are
  //     not stored: 3b assembles them again from B); the element bounds and the Laplacian halo travel together (biharmonic_wk_scalar_minmax packs lap, Qmin, Qmax into one message: viscosity_mod.F90:389-391)
  if (split_stage(c, "lap", K_LAP,
        [&](Work w) -> int {
          if (!w.npwork) return 0;
          with_shape(w.P->psz, [&](auto psz) {
            constexpr int Z = decltype(psz)::value;
            GatherArgs ga = gargs(w, nullptr, 0, c->eta, NLEVP);
            ga.pexp = c->pexp;   // only what other patches and ranks read of the first Laplacian is stored: stage 3b forms its own slots' itself
            hipLaunchKernelGGL((k_lap1<1, Z>), dim3(patch_blocks(w.npwork)), dim3(Patch<Z>::THREADS), 0, c->stream, c->nelemd, c->D, c->geo(), c->qsize, 2 * dts,
                               (const double*)c->B, c->T, c->dp, c->divdp_proj, c->qmin, c->qmax, ga);
            return 0; });
          LAUNCH_CHECK(); return 0; },
        [&]() -> int { return pack_minmax(c, cs) || halo_exchange(c, 2 * c->mm_m(), 1, cs) || unpack_minmax(c, cs) || pack_tracers(c, cs, c->T, nq) ||
                              halo_exchange(c, nq, 0, cs) || unpack_halo(c, cs, c->T, nq); })) return 1;
  // (no neighbour min/max pass here: 3b forms it from the element bounds -- its patch's and the element ring's -- while it runs)
  // 3b: B (+) edges, T (+) edges -> C (2nd Laplacian + biharmonic scaling + advance + limiter)
  if (split_stage(c, "advance2", K_ADV2,
        [&](Work w) -> int {
          if (!w.npwork) return 0;
          launch_advance23(w.P->psz, patch_blocks(w.npwork), c->stream, c->nelemd, c->D, c->geo(), c->qsize, dts, c->nu_q, c->B, c->T, c->C, c->vn0, c->dp,
                           c->divdp, c->divdp_proj, c->qmin, c->qmax, c->dp0, gargs(w, c->omega_p, NLEV));   // (tse_stage3.hip)
          LAUNCH_CHECK(); return 0; },
        [&]() -> int { return pack_tracers(c, cs, c->C, nqv, nqv) || halo_exchange(c, nqv, 0, cs) || unpack_halo(c, cs, c->C, nqv, nqv); })) return 1;
  if (defer_dss) {   // C (halo columns filled: the stage's exchange is ordered before the next launch on the compute stream) waits for the remap
    c->dss_deferred_n0 = n0_qdp;
    set_bounds_cache(c, 0);
    return 0;
  }
  // final DSS fused with qdp_time_avg (:645-662) and with the next step's element min/max
  if (prefetch && c->halo()) {
    hipEvent_t done = nullptr;
    if (split_stage(c, "dss", K_DSS,
          [&](Work w) -> int { return dss_tracer_launch(c, c->C, Qnp1, Qn0, w.plist, w.npwork, c->omega_p, NLEV); },
          [&]() -> int { return pack_minmax(c, cs, c->qmin2, c->qmax2) || halo_exchange(c, 2 * c->mm_m(), 1, cs); }, &done)) return 1;
    set_bounds_cache(c, np1_qdp);
    c->mm_halo = np1_qdp;
  } else {
    if (dss_tracer_pass(c, c->C, Qnp1, Qn0, c->omega_p, NLEV)) return 1;
    set_bounds_cache(c, np1_qdp);
  }
  return 0;
}

static int advec_step(tse_ctx* c, double dt, int n0_qdp, int np1_qdp, bool prefetch, bool defer_dss = false) {
  if (n0_qdp == np1_qdp || n0_qdp < 1 || n0_qdp > 2 || np1_qdp < 1 || np1_qdp > 2)
    return fail("advec_tracers_remap_rk2: time levels n0_qdp=%d np1_qdp=%d", n0_qdp, np1_qdp);
  bool gor = dss_on_read();
  if (!gor && join_inputs(c)) return 1;
  // gather offsets are 32-bit bytes within a plane; TSE_TEST_PLANE_LIMIT lowers the 4 GiB limit so that tests reach the fallback
  const size_t plane_limit = hook_env("TSE_TEST_PLANE_LIMIT") ? (size_t)strtoull(hook_env("TSE_TEST_PLANE_LIMIT"), nullptr, 10) : ((size_t)1 << 32);
  if (gor && c->tps * 8 >= plane_limit) {
    static bool said = false;
    if (!said) {
      fprintf(stderr, "transport_se_hip: a scratch plane is %zu bytes (>= %zu): DSS on read disabled, one DSS pass per stage\n", c->tps * 8, plane_limit);
      said = true;
    }
    gor = false;
    if (join_inputs(c)) return 1;
  }
  if (gor) {
    if (c->t_zero_dirty) {   // restore the all-zero slots of T
      const int tot = c->qsize * NCHUNK * 16 * CL;
      hipLaunchKernelGGL(k_zero_slot<>, dim3((tot + 255) / 256), dim3(256), 0, c->stream, c->qsize, c->T, c->scr(), c->zero0());
      LAUNCH_CHECK();
      c->t_zero_dirty = false;
    }
    return advec_dss_on_read(c, dt / 2, n0_qdp, np1_qdp, prefetch, defer_dss);   // (sets the bounds cache itself)
  } else {
    if (tse_compute_divdp(c)) return 1;
    if (euler_step_impl(c, np1_qdp, n0_qdp, dt / 2, 3, 0, false, 0, true)) return 1;
    if (euler_step_impl(c, np1_qdp, np1_qdp, dt / 2, 1, 1, false, 0, false)) return 1;
    if (euler_step_impl(c, np1_qdp, np1_qdp, dt / 2, 2, 2, true, n0_qdp, false)) return 1;
  }
  set_bounds_cache(c, np1_qdp);   // the final DSS emitted min/max of Qdp(np1)/dp for the next step
  return 0;
}

int tse_advec_tracers_remap_rk2(tse_ctx* c, double dt, int n0_qdp, int np1_qdp) { return advec_step(c, dt, n0_qdp, np1_qdp, false); }

// prefetch: as in advec_dss_on_read -- the next tracer step's bounds exchange starts under the interior elements of the remap
static int remap_launch(tse_ctx* c, double dt, int np1_qdp, bool prefetch) {
  if (np1_qdp < 1 || np1_qdp > 2) return fail("vertical_remap: np1_qdp=%d", np1_qdp);
  // TSE_REMAP_NT: tracers per thread in the lockstep column loop; TSE_REMAP_GENERIC=1: always take the generic loop (tests)
  const int nt = getenv("TSE_REMAP_NT") ? atoi(getenv("TSE_REMAP_NT")) : 1;
  const int generic = getenv("TSE_REMAP_GENERIC") ? atoi(getenv("TSE_REMAP_GENERIC")) : 0;
  double* Qr = c->q(np1_qdp);
  // the tracer step before left its final DSS + time average to this launch (advec_dss_on_read, defer_dss): Qr is written only
  const int fused_n0 = c->dss_deferred_n0;
  c->dss_deferred_n0 = 0;
  if (fused_n0 && (fused_n0 == np1_qdp || nt != 1)) return fail("vertical_remap: deferred DSS of time level %d cannot be assembled here", fused_n0);
  RemapFuse F{};
  if (fused_n0) F = RemapFuse{c->C, c->scr(), c->etab, c->slot_of, c->pperm, c->q(fused_n0), c->rspheremp, c->omega_p, c->zero0()};
  auto launch = [&](Work w) -> int {   // block = element
    if (!w.nwork) return 0;
    const int* list = w.order == c->ord_bnd ? c->rl_bnd : w.order == c->ord_int ? c->rl_int : c->rl_all;   // the same elements, in slot order
    auto go = [&](auto kern, int threads) {
      hipLaunchKernelGGL(kern, dim3(8 * ((w.nwork + 7) / 8)), dim3(threads), sizeof(RemapLds), c->stream, c->qsize, dt, c->ps0, c->hyai, c->hybi,
                         c->dp, c->divdp_proj, c->dp3d, c->ps_v, Qr, c->bad, c->qmin2, c->qmax2, generic, c->sink, (const double*)nullptr, list, w.nwork, c->lvl_tmp, F);
    };
    if (fused_n0) { if (c->remap_alg2) go(k_remap<1, true, true>, REMAP_THREADS); else go(k_remap<1, false, true>, REMAP_THREADS); }
    else if (nt == 1) { if (c->remap_alg2) go(k_remap<1, true>, REMAP_THREADS); else go(k_remap<1, false>, REMAP_THREADS); }
    else { if (c->remap_alg2) go(k_remap<2, true>, REMAP_THREADS / 2); else go(k_remap<2, false>, REMAP_THREADS / 2); }
    LAUNCH_CHECK();
    return 0;
  };
  if (hook_env("TSE_TEST_FAIL_REMAP")) {   // tests: this process's remap reports a negative layer thickness (one rank of several failing alone)
    static const int one = 1;
    HIPCHK(hipMemcpyAsync(c->bad, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  if (prefetch && c->halo()) {
    const int nq = c->qsize * NLEV;
    hipStream_t cs = c->comm_stream;
    hipEvent_t done = nullptr;
    if (split_stage(c, "remap", K_DSS, launch, [&]() -> int { return pack_minmax(c, cs, c->qmin2, c->qmax2) || halo_exchange(c, 2 * c->mm_m(), 1, cs); }, &done))
      return 1;
    set_bounds_cache(c, np1_qdp);   // k_remap emitted the element min/max of the remapped field
    c->mm_halo = np1_qdp;
  } else {
    Scope s(c, "remap");
    if (launch(work_of(c, 0, K_DSS))) return 1;
    set_bounds_cache(c, np1_qdp);
  }
  return 0;
}
// the reference's abort condition (prim_advection_mod.F90:1323): the device flag every remap since the last check ORs into
static int remap_check(tse_ctx* c) {
  int bad = 0;
  HIPCHK(hipMemcpyAsync(&bad, c->bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (bad) {
    set_bounds_cache(c, 0);
    HIPCHK(hipMemset(c->bad, 0, sizeof(int)));
    fail("negative layer thickness.  timestep or remap time too large");
    return 2;
  }
  return 0;
}
int tse_vertical_remap(tse_ctx* c, double dt, int np1_qdp) {
  c->dss_deferred_n0 = 0;   // (only tse_prim_run_subcycle defers a final DSS to its remap; a call of its that failed in between must not leave the request behind)
  if (remap_launch(c, dt, np1_qdp, false)) return 1;
  return remap_check(c);
}

// ---- single calls of the public operators the path is built from (operator-level parity; host arrays in, host arrays out) ----
template <int OP>
static int elem_op(tse_ctx* c, const double* in, size_t in_per_elem, double* out) {
  if (!in || !out) return fail("element operator: null argument");
  double *din = nullptr, *dout = nullptr;
  if (dalloc(&din, (size_t)c->nelemd * in_per_elem) || dalloc(&dout, (size_t)c->nelemd * 16)) { if (din) (void)hipFree(din); return 1; }
  int rc = 0;
  do {
    if (hipMemcpyAsync(din, in, (size_t)c->nelemd * in_per_elem * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail("element operator: H2D copy failed"); break; }
    hipLaunchKernelGGL(k_elem_op<OP>, dim3((c->nelemd * 4 + 255) / 256), dim3(256), 0, c->stream, c->nelemd, c->D, c->geo(), (const double*)din, dout);
    if (hipGetLastError() != hipSuccess) { rc = fail("element operator: kernel launch failed"); break; }
    if (hipMemcpyAsync(out, dout, (size_t)c->nelemd * 16 * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail("element operator: D2H copy failed"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail("element operator: stream synchronisation failed"); break; }
  } while (0);
  (void)hipFree(din); (void)hipFree(dout);
  return rc;
}
int tse_divergence_sphere(tse_ctx* c, const double* v, double* div) { return elem_op<0>(c, v, 32, div); }
int tse_laplace_sphere_wk(tse_ctx* c, const double* s, double* lap) { return elem_op<1>(c, s, 16, lap); }

// remap_Q_ppm(Qdp,np,qsize,dp1,dp2) (prim_advection_mod.F90:98-214) for every local element: the column kernel of
// vertical_remap with the caller's source and target thicknesses.  Uses time level 1 of the device tracer state and the
// level fields dp/divdp_proj/dp3d as work space (a test/utility call, not part of the time loop).
int tse_remap_q_ppm(tse_ctx* c, double* Qdp, const double* dp1, const double* dp2) {
  if (!Qdp || !dp1 || !dp2) return fail("tse_remap_q_ppm: null argument");
  set_bounds_cache(c, 0);
  c->dcmip_static = false;   // dp is overwritten below: the next tse_dcmip_step_inputs must write the prescribed values again
  const size_t lev = c->lev();
  double* d2 = nullptr;
  if (dalloc(&d2, lev)) return 1;
  int rc = 0;
  do {
    if (hipMemcpy(c->q(1), Qdp, c->trc() * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(c->dp, dp1, lev * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d2, dp2, lev * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemset(c->divdp_proj, 0, lev * 8) != hipSuccess) { rc = fail("tse_remap_q_ppm: upload failed"); break; }
    const int generic = getenv("TSE_REMAP_GENERIC") ? atoi(getenv("TSE_REMAP_GENERIC")) : 0;
    auto go = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3(8 * ((c->nelemd + 7) / 8)), dim3(REMAP_THREADS), sizeof(RemapLds), c->stream, c->qsize, 0.0, c->ps0, c->hyai, c->hybi,
                         c->dp, c->divdp_proj, c->dp3d, c->ps_v, c->q(1), c->bad, (double*)nullptr, (double*)nullptr, generic, c->sink, (const double*)d2,
                         (const int*)nullptr, c->nelemd, c->lvl_tmp, RemapFuse{});
    };
    if (c->remap_alg2) go(k_remap<1, true>); else go(k_remap<1, false>);
    if (hipGetLastError() != hipSuccess) { rc = fail("tse_remap_q_ppm: kernel launch failed"); break; }
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(Qdp, c->q(1), c->trc() * 8, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail("tse_remap_q_ppm: download failed"); break; }
    rc = remap_check(c);
  } while (0);
  (void)hipFree(d2);
  return rc;
}

// per-element tracer mass of time level nt -> host out[nelemd][qsize] (fixed summation order inside an element; the caller adds
// the elements up reproducibly: transport_se_amd/diagnostics.py)
int tse_element_mass(tse_ctx* c, int nt, double* out) {
  if (nt < 1 || nt > 2 || !out) return fail("tse_element_mass: nt=%d", nt);
  const size_t n = (size_t)c->nelemd * c->qsize;
  double* d = nullptr;
  if (dalloc(&d, n)) return 1;
  hipLaunchKernelGGL(k_elem_mass<>, dim3((unsigned)n), dim3(128), 0, c->stream, c->qsize, (const double*)(c->q(nt)),
                     (const double*)c->spheremp, d);
  int rc = 0;
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, d, n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) rc = fail("tse_element_mass: launch or copy failed");
  (void)hipFree(d);
  return rc;
}

// per-element shares of Qmass = integral(sum_k Qdp) and Qvar = integral(sum_k Qdp*Q) of time level nt (prim_diag_scalars,
// prim_state_mod.F90:604-655, with the ps_v the last remap left) -> host mass_out, var_out [nelemd][qsize]
int tse_element_qdiag(tse_ctx* c, int nt, double* mass_out, double* var_out, double* min_out, double* max_out) {
  if (nt < 1 || nt > 2 || !mass_out || !var_out) return fail("tse_element_qdiag: nt=%d", nt);
  const size_t n = (size_t)c->nelemd * c->qsize;
  double* d = nullptr;
  if (dalloc(&d, 4 * n)) return 1;
  hipLaunchKernelGGL(k_elem_qdiag<>, dim3((unsigned)n), dim3(64), 0, c->stream, c->qsize, (const double*)(c->q(nt)), (const double*)c->spheremp,
                     (const double*)c->ps_v, (const double*)c->hyai, (const double*)c->hybi, c->ps0, d, d + n, d + 2 * n, d + 3 * n);
  int rc = hipGetLastError() != hipSuccess;
  double* outs[4] = {mass_out, var_out, min_out, max_out};
  for (int i = 0; i < 4 && !rc; i++)
    if (outs[i]) rc = hipMemcpyAsync(outs[i], d + i * n, n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess;
  if (!rc) rc = hipStreamSynchronize(c->stream) != hipSuccess;
  if (rc) rc = fail("tse_element_qdiag: launch or copy failed");
  (void)hipFree(d);
  return rc;
}

// ---- prescribed fields + device-resident driver ---------------------------------------------------
int tse_dcmip_init(tse_ctx* c, int test, const double* lat, const double* lon, const double* hyam, const double* hybm) {
  if (test != 1 && test != 2) return fail("tse_dcmip_init: test_case=%d", test);
  c->dcmip_test = test; c->dcmip_static = false;
  const double H = 287.04 * 300.0 / 9.80616, P0 = 100000.0;
  std::vector<double> hyai(NLEVP), hybi(NLEVP);
  HIPCHK(hipMemcpy(hyai.data(), c->hyai, NLEVP * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hybi.data(), c->hybi, NLEVP * 8, hipMemcpyDeviceToHost));
  std::vector<double> zm(NLEV), zi(NLEVP), pint(NLEVP), dph(NLEV);
  for (int k = 0; k < NLEVP; k++) { zi[k] = H * log(1.0 / (hyai[k] + hybi[k])); pint[k] = P0 * exp(-zi[k] / H); }
  for (int k = 0; k < NLEV; k++) {
    zm[k] = H * log(1.0 / (hyam[k] + hybm[k]));
    dph[k] = (hyai[k + 1] - hyai[k]) * c->ps0 + (hybi[k + 1] - hybi[k]) * pint[NLEV];
  }
  std::vector<double> la(lat, lat + (size_t)c->nelemd * 16), lo(lon, lon + (size_t)c->nelemd * 16);
  void* old[] = {c->lat, c->lon, c->zm, c->zi, c->pint, c->dph};
  for (void* p : old) if (p) (void)hipFree(p);
  c->lat = c->lon = c->zm = c->zi = c->pint = c->dph = nullptr;
  if (upload(&c->lat, la) || upload(&c->lon, lo) || upload(&c->zm, zm) || upload(&c->zi, zi) || upload(&c->pint, pint) || upload(&c->dph, dph)) return 1;
  if (!c->dcmip_tab && dalloc(&c->dcmip_tab, 1)) return 1;
  hipLaunchKernelGGL(k_dcmip_tables<>, dim3(1), dim3(128), 0, c->stream, test, c->zm, c->zi, c->dcmip_tab);   // level-only factors
  LAUNCH_CHECK();
  return 0;
}
int tse_dcmip_set_initial(tse_ctx* c) {
  set_bounds_cache(c, 0);
  if (!c->dcmip_test) return fail("tse_dcmip_set_initial: call tse_dcmip_init first");
  Scope s(c, "dcmip");
  size_t tot = c->lev();
  hipLaunchKernelGGL(k_dcmip_init<>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->nelemd, c->qsize, c->dcmip_test, c->lat,
                     c->lon, c->zm, c->pint, c->dph, c->q(1), c->q(2), c->dp3d, c->ps_v);
  LAUNCH_CHECK();
  return 0;
}
static int dcmip_step_launch(tse_ctx* c, int nstep, double tstep, hipStream_t st) {
  if (!c->dcmip_test) return fail("tse_dcmip_step_inputs: call tse_dcmip_init first");
  Scope s(c, "dcmip", st);
  size_t tot = (size_t)c->nelemd * 16;   // one thread per column
  double t_wind = (nstep > 0 ? nstep - 1 : 0) * tstep, t_now = nstep * tstep;
  // derived%dp is rewritten with the same time-independent p_i(k+1) - p_i(k) on every step (dcmip_wrapper_mod.F90:183,199), so
  // the cached next-step bounds (formed with that dp) stay valid; every other writer of dp drops them (tse_set_derived)
  hipLaunchKernelGGL(k_dcmip_step<>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, c->nelemd, c->dcmip_test, t_wind, t_now,
                     c->lat, c->lon, c->dcmip_tab, c->pint, c->vn0, c->dcmip_static ? nullptr : c->dp, c->eta, c->dcmip_static ? nullptr : c->omega_p);
  LAUNCH_CHECK();
  c->dcmip_static = true;   // until someone else writes dp or omega_p (tse_set_derived, tse_invalidate_cache)
  return 0;
}
int tse_dcmip_step_inputs(tse_ctx* c, int nstep, double tstep) { return join_inputs(c) || dcmip_step_launch(c, nstep, tstep, c->stream); }
int tse_prim_run_subcycle(tse_ctx* c, double tstep, int nsub, int* nstep_io) {
  c->dss_deferred_n0 = 0;
  const int nstep0 = *nstep_io;
  int nstep = nstep0;
  // The reference aborts in the first remap that meets a negative layer thickness (prim_advection_mod.F90:1323).  Here the
  // device flag is copied to page-locked memory behind every cycle's remap and looked at TWO cycles later -- by then the copy
  // has long landed, so the host never waits for the device while it keeps a full cycle queued ahead -- and the call returns 2
  // with *nstep = the step count at the end of the failing cycle (at most two more cycles were started on the bad state).
  // (With the callback form of the halo exchange every stage synchronises with the host anyway.)
  // (*nstep may lie inside an rsplit cycle -- a caller that stepped part of it through the step-by-step entries: the first of the nsub
  // cycles is then the REST of that cycle, rsplit - nstep % rsplit steps and its remap)
  const int r0 = ((nstep0 % c->rsplit) + c->rsplit) % c->rsplit;
  auto failed = [&](int cyc) -> int {
    *nstep_io = nstep0 - r0 + (cyc + 1) * c->rsplit;
    set_bounds_cache(c, 0);
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemset(c->bad, 0, sizeof(int));
    fail("negative layer thickness.  timestep or remap time too large");
    return 2;
  };
  const bool overlap_inputs = !(getenv("TSE_INPUT_OVERLAP") && getenv("TSE_INPUT_OVERLAP")[0] == '0');
  // the remap that closes the cycle assembles the last step's final DSS + time average on read (TSE_REMAP_FUSED=0: two kernels)
  const bool fuse_remap = remap_fused() && dss_on_read() && c->tps * 8 < ((size_t)1 << 32) && !(getenv("TSE_REMAP_NT") && atoi(getenv("TSE_REMAP_NT")) != 1);
  for (int s = 0; s < nsub; s++) {
    if (s >= 2) {
      HIPCHK(hipEventSynchronize(c->bad_ev[s & 1]));
      if (c->bad_host[s & 1]) return failed(s - 2);
    }
    int n0 = 1, np1 = 2;
    for (int r = (s == 0 ? r0 : 0); r < c->rsplit; r++) {
      if (overlap_inputs) {   // fork: behind everything launched so far (the previous step / remap read what this writes)
        HIPCHK(hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
        if (dcmip_step_launch(c, nstep, tstep, c->aux_stream)) return 1;
        HIPCHK(hipEventRecord(c->ev_inputs, c->aux_stream));
        c->inputs_pending = true;   // joined by the step before its first kernel that reads them (join_inputs)
      } else if (tse_dcmip_step_inputs(c, nstep, tstep)) return 1;
      if (nstep % 2 == 0) { n0 = 1; np1 = 2; } else { n0 = 2; np1 = 1; }  // TimeLevel_Qdp, time_mod.F90:85-109
      if (advec_step(c, tstep, n0, np1, r + 1 < c->rsplit, fuse_remap && r + 1 == c->rsplit)) return 1;   // (the remap follows the last one: its bounds would be stale)
      nstep++;
    }
    if (remap_launch(c, tstep * c->rsplit, np1, true)) { *nstep_io = nstep; return 1; }
    HIPCHK(hipMemcpyAsync(&c->bad_host[s & 1], c->bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->bad_ev[s & 1], c->stream));
  }
  *nstep_io = nstep;
  for (int s = std::max(0, nsub - 2); s < nsub; s++) {
    HIPCHK(hipEventSynchronize(c->bad_ev[s & 1]));
    if (c->bad_host[s & 1]) return failed(s);
  }
  return 0;
}

// ---- introspection ------------------------------------------------------------------------------
void* tse_device_ptr(tse_ctx* c, const char* name, size_t* nbytes) {
  struct Ent { const char* n; void* p; size_t b; };
  const size_t lev = c->lev() * 8, trc = c->trc() * 8, mm = (size_t)c->nelemd * c->mm_m() * 8, scr = (size_t)c->qsize * c->tps * 8;
  const size_t m2 = (size_t)2 * c->mm_m() * 8;
  Ent ents[] = {{"qdp1", c->q(1), trc}, {"qdp2", c->q(2), trc}, {"T", c->T, scr + c->tps * 8}, {"B", c->B, scr + c->tps * 8}, {"C", c->C, scr + c->tps * 8}, {"vn0", c->vn0, 2 * lev}, {"dp", c->dp, lev},
                {"divdp", c->divdp, lev}, {"divdp_proj", c->divdp_proj, lev}, {"eta_dot_dpdn", c->eta, (size_t)c->nelemd * NLEVP * 16 * 8},
                {"omega_p", c->omega_p, lev}, {"dp3d", c->dp3d, lev}, {"ps_v", c->ps_v, (size_t)c->nelemd * 16 * 8}, {"qmin", c->qmin, mm},
                {"qmax", c->qmax, mm}, {"sendbuf", c->sendbuf, (size_t)c->ncol_send * c->nlyr_halo * 8},
                {"recvbuf", c->recvbuf, (size_t)c->ncol_recv * c->nlyr_halo * 8}, {"sendbuf_mm", c->sendbuf_mm, (size_t)c->nmm_send * m2},
                {"recvbuf_mm", c->recvbuf_mm, (size_t)c->nmm_recv * m2}};
  for (auto& e : ents) if (!strcmp(e.n, name)) { if (nbytes) *nbytes = e.p ? e.b : 0; return e.p; }
  if (nbytes) *nbytes = 0;
  return nullptr;
}
#ifdef TSE_AB_HOOKS
// ---- developer experiment: where the scratch fields live (tools/placement_probe.py) ------------------------------------------
// make the pool K scratch-sized allocations (the first three are T, B, C as allocated)
extern "C" int tse_debug_scratch_pool(tse_ctx* c, int K) {
  const size_t scr_n = (size_t)(c->qsize + 1) * c->tps;
  if (c->pool.empty()) { c->pool = {c->T, c->B, c->C}; }
  while ((int)c->pool.size() < K) {
    double* p = nullptr;
    if (hipMalloc((void**)&p, scr_n * 8) != hipSuccess) { (void)hipGetLastError(); return fail("tse_debug_scratch_pool: out of memory at %zu chunks", c->pool.size()); }
    HIPCHK(hipMemset(p, 0, scr_n * 8));
    c->pool.push_back(p);
  }
  HIPCHK(hipDeviceSynchronize());
  return 0;
}
extern "C" int tse_debug_zero_pool(tse_ctx* c) {
  const size_t scr_n = (size_t)(c->qsize + 1) * c->tps;
  for (double* p : c->pool) HIPCHK(hipMemset(p, 0, scr_n * 8));
  HIPCHK(hipDeviceSynchronize());
  return 0;
}
extern "C" int tse_debug_assign_scratch(tse_ctx* c, int iT, int iB, int iC) {
  const int K = (int)c->pool.size();
  if (iT < 0 || iB < 0 || iC < 0 || iT >= K || iB >= K || iC >= K || iT == iB || iT == iC || iB == iC) return fail("tse_debug_assign_scratch: %d %d %d of %d", iT, iB, iC, K);
  HIPCHK(hipStreamSynchronize(c->stream));
  c->T = c->pool[iT]; c->B = c->pool[iB]; c->C = c->pool[iC];
  return 0;
}
// the tracer state's two time levels in pool chunks i1, i2 (the state must be set again afterwards)
extern "C" int tse_debug_assign_qdp(tse_ctx* c, int i1, int i2) {
  const int K = (int)c->pool.size();
  if (i1 < 0 || i2 < 0 || i1 >= K || i2 >= K || i1 == i2) return fail("tse_debug_assign_qdp: %d %d of %d", i1, i2, K);
  for (double* p : {c->T, c->B, c->C}) if (p == c->pool[i1] || p == c->pool[i2]) return fail("tse_debug_assign_qdp: chunk in use as scratch");
  HIPCHK(hipStreamSynchronize(c->stream));
  if (!c->qorig[0]) { c->qorig[0] = c->qlev[0]; c->qorig[1] = c->qlev[1]; }
  c->qlev[0] = c->pool[i1]; c->qlev[1] = c->pool[i2];
  c->mm_valid = 0;
  return 0;
}
// GB/s of a streaming pass: src/dst = -1 none (write-only / read-only), 0..K-1 pool chunk, 100 | 101 = Qdp time level 1 | 2
extern "C" int tse_debug_probe(tse_ctx* c, int src, int dst, double* gbps) {
  const size_t n = std::min((size_t)c->qsize * c->tps, c->trc()) / 2;   // double2 units: the smaller of a scratch field and a tracer field
  auto ptr = [&](int i) -> double2* { return i < 0 ? nullptr : i >= 100 ? (double2*)c->q(i - 99) : (double2*)c->pool[i]; };
  hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, c->stream, n, (const double2*)ptr(src), ptr(dst));
  HIPCHK(hipEventRecord(a, c->stream));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_probe_copy, dim3(2048), dim3(256), 0, c->stream, n, (const double2*)ptr(src), ptr(dst));
  HIPCHK(hipEventRecord(b, c->stream)); HIPCHK(hipEventSynchronize(b));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
  *gbps = ((src >= 0) + (dst >= 0)) * (double)n * 16 / (ms / 2) / 1e6;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return 0;
}
#endif   // TSE_AB_HOOKS
int tse_timing(tse_ctx* c, int enable) { resolve_timers(c); c->timing = enable != 0; c->timers.clear(); return 0; }
int tse_kernel_time(tse_ctx* c, const char* name, double* ms, long* launches) {
  resolve_timers(c);
  double t = 0; long n = 0;
  const size_t len = strlen(name);
  for (auto& kv : c->timers)   // prefix match: "advance" = advance0 + advance1 + advance2
    if (kv.first.compare(0, len, name) == 0) { t += kv.second.ms; n += kv.second.n; }
  if (ms) *ms = t;
  if (launches) *launches = n;
  return 0;
}

#ifdef TSE_LIMITER_STATS
extern "C" int tse_debug_limiter_hist(unsigned long long* out, int reset) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(tse::g_lim_hist), sizeof(unsigned long long) * 20));
  if (reset) { unsigned long long z[20] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(tse::g_lim_hist), z, sizeof z)); }
  return 0;
}
#endif
