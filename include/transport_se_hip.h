/* transport_se_hip.h -- C ABI of the MI355X-native tracer-advection engine (libtransport_se_hip.so).
 *
 * Drop-in boundary: these entry points are what a Fortran `hip_mod` binds with ISO_C_BINDING in place of
 * the reference's CUDA-Fortran seam `cuda_mod` (reference src/share/cuda_mod.F90:57-64), which the reference
 * calls through `#if USE_CUDA_FORTRAN` hooks:
 *   cuda_mod_init(elem,hybrid,deriv,hvcoord)            prim_driver_mod.F90:686-689   -> tse_init
 *   copy_qdp_h2d(elem,nt) / copy_qdp_d2h(elem,nt)       prim_driver_mod.F90:781-784,798-801 -> tse_copy_qdp_h2d/_d2h
 *   euler_step_cuda(np1_qdp,n0_qdp,dt,elem,...,DSSopt,rhs_multiplier)
 *                                                       prim_advection_mod.F90:715-718 -> tse_euler_step
 *   qdp_time_avg_cuda(elem,rkstage,n0_qdp,np1_qdp,...)  prim_advection_mod.F90:646-656 -> tse_qdp_time_avg
 *   vertical_remap_cuda(elem,hvcoord,dt,np1,np1_qdp,..) prim_advection_mod.F90:1279-1282 -> tse_vertical_remap
 * plus the whole-step call Prim_Advec_Tracers_remap_rk2 (prim_advection_mod.F90:579-640) -> tse_advec_tracers_remap_rk2,
 * which is the fast path (one call per tracer step, everything stays in HBM).
 * The Fortran-side binding is shown in INTEGRATION.md and shipped as transport_se_amd/fortran/cuda_mod_hip.F90.
 *
 * Conventions
 *   - All functions return 0 on success, nonzero on error (the Fortran side then calls abortmp, as the
 *     reference does on every failure: parallel_mod.F90:274-287); tse_last_error() gives the message.
 *   - np = 4, nlev = 72 are compile-time constants exactly as in the reference (dimensions_mod.F90:19,27).
 *   - Host arrays are passed as the address of element 1's field plus the byte stride between consecutive
 *     elements (element_t is a fixed-size derived type, element_mod.F90:112-221, so `elem(:)` is strided AoS);
 *     inside one element the reference's own memory order is assumed (i fastest, then j, k, q, time level).
 *   - Time-level arguments (n0_qdp, np1_qdp, nt) are the reference's 1-based values (time_mod.F90:85-109).
 *   - DSSopt: 1 = eta_dot_dpdn, 2 = omega_p, 3 = divdp_proj (prim_advection_mod.F90:454-457).
 *   - Edge descriptors are the reference's own: putmapP/getmapP are 0-based column offsets into the edge
 *     buffer, -1 for a missing corner neighbour, reverse is a logical (edge_mod.F90:36-43, schedule_mod.F90:905,929);
 *     direction index order west,east,south,north,swest,seast,nwest,neast (control_mod.F90:173-181).
 *   - Neighbour-rank message slots are the reference's Schedule(1)%SendCycle/RecvCycle entries
 *     (schedtype_mod.F90:7-29): peer rank (0-based), ptrP (1-based first column, as stored), lengthP (columns).
 */
#ifndef TRANSPORT_SE_HIP_H
#define TRANSPORT_SE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TSE_NP 4
#define TSE_NLEV 72
#define TSE_NLEVP 73

typedef struct tse_ctx tse_ctx;

/* Exchange callback = the body of bndry_exchangeV (bndry_mod.F90:21-126): send `sendbuf` slot s
 * (len[s]*nlyr doubles at entry offset sum(len[:s]), layer index fastest) to rank send_peer[s], receive the
 * mirror-image slot into `recvbuf`.  Both pointers are DEVICE pointers.  The library has finished writing
 * sendbuf (stream-synchronised) when it calls this, and reads recvbuf only after it returns.
 *   kind 0: one entry per edge-buffer column, len[s] = lengthP of the slot (the reference's message layout);
 *   kind 1: the neighbour min/max exchange (viscosity_mod.F90:748-816): the packed fields are element constants, so one
 *           entry per neighbouring (element, direction) pair is sent instead of one per column; len[s] = the number of
 *           shared edges + shared corners with that rank (tse_halo_layout gives the totals); nlyr = 2 * nlev * (qsize rounded
 *           up to a multiple of 4): a minimum set and a maximum set in the library's bounds layout, pad tracers included -- a
 *           callback sizes its buffers by the nlyr it is given. */
typedef int (*tse_exchange_fn)(void *user, double *sendbuf, double *recvbuf, int nlyr, int kind);
/* The callback is the portable form of the seam (an MPI host keeps its own communicator: cuda_mod_hip.F90 passes
 * MPI_Isend/Irecv).  The native form is tse_comm_init below: the library then performs the exchange itself with RCCL
 * send/recv on its own streams and the callback is not used. */

typedef struct {
  int nelemd;          /* elements on this rank */
  int qsize;           /* active tracers */
  int device;          /* HIP device ordinal, -1 = current */
  double nu_q;         /* control_mod nu_q */
  int limiter_option;  /* must be 8 (the only limiter wired in the reference: prim_advection_mod.F90:858,880) */
  int rsplit;          /* control_mod rsplit (vertical remap frequency), informational */
  const double *Dvv;   /* deriv%Dvv(np,np), Fortran order */
  const double *hyai;  /* hvcoord%hyai(nlevp) */
  const double *hybi;  /* hvcoord%hybi(nlevp) */
  double ps0;          /* hvcoord%ps0 */
  /* per-element metric terms: address of elem(1)%X and byte stride to elem(2)%X */
  const double *Dinv;      size_t Dinv_stride;      /* Dinv(2,2,np,np) */
  const double *metdet;    size_t metdet_stride;    /* (np,np) */
  const double *rmetdet;   size_t rmetdet_stride;
  const double *spheremp;  size_t spheremp_stride;
  const double *rspheremp; size_t rspheremp_stride;
  /* edge descriptors, dense copies [nelemd][8] */
  const int *putmapP; const int *getmapP; const int *reverse;
  /* neighbour-rank slots (0 for a single rank) */
  int nsend; const int *send_peer; const int *send_ptrP; const int *send_lengthP;
  int nrecv; const int *recv_peer; const int *recv_ptrP; const int *recv_lengthP;
  tse_exchange_fn exchange; void *exchange_user;   /* may be NULL when nsend == nrecv == 0, or when tse_comm_init follows */
  int vert_remap_q_alg; /* control_mod vert_remap_q_alg (control_mod.F90:61-66): 0|1 = PPM with mirrored ghost cells, 2 = PPM with
                         * piecewise-constant boundary cells (prim_advection_mod.F90:230-250,283-341); anything else is refused */
} tse_init_args;

int  tse_init(tse_ctx **ctx, const tse_init_args *args);
void tse_finalize(tse_ctx *ctx);
const char *tse_last_error(void);
int  tse_synchronize(tse_ctx *ctx);

/* ---- bndry_exchangeV inside the library: RCCL neighbour send/recv over xGMI (bndry_mod.F90:74-124) ----
 * One communicator rank per tse_ctx/GPU.  Rank 0 obtains an id with tse_comm_unique_id and hands it to the other ranks
 * by whatever the host has (MPI_Bcast in a Fortran/MPI host, a torch.distributed/gloo broadcast in the Python driver);
 * every rank then calls tse_comm_init (collective).  send_peer/recv_peer of tse_init_args are ranks of this
 * communicator.  Afterwards every DSS halo is one `ncclGroupStart; ncclRecv/ncclSend per neighbour slot; ncclGroupEnd`
 * on the library's communication stream, with no host synchronisation; in the whole-step call the rank-boundary
 * elements are computed first and the exchange runs under the interior elements (the reference's
 * cuda_mod.F90:358-401,961-1005 ordering). */
#define TSE_COMM_ID_BYTES 128
int tse_comm_unique_id(void *id_out /* TSE_COMM_ID_BYTES */);
int tse_comm_init(tse_ctx *ctx, const void *id /* TSE_COMM_ID_BYTES */, int rank, int nranks);
/* Everything tse_comm_init can find wrong WITHOUT the other ranks: rank/peer ranges, device, streams, and that the RCCL this
 * process resolved can serve the headers the library was built with (same major version, point-to-point capable).  ncclCommInitRank is a blocking collective, so
 * a host calls this on every rank first, agrees on the outcome over its own control plane (MPI_Allreduce, gloo) and enters
 * tse_comm_init only when every rank is ready -- a rank that failed alone inside tse_comm_init would strand its peers. */
int tse_comm_precheck(tse_ctx *ctx, int rank, int nranks);
/* the RCCL this process runs: version code of the runtime (ncclGetVersion: major*10000 + minor*100 + patch), of the headers
 * the library was built with, and the path of the shared object that provides it (bench.py prints all three) */
int tse_comm_version(int *runtime, int *built, char *path, size_t path_len);
/* rank / size as the communicator itself reports them (ncclCommUserRank/ncclCommCount); 0/1 without a communicator */
int tse_comm_info(tse_ctx *ctx, int *rank, int *nranks);
/* give the communicator up (ncclCommAbort; no-op without one): afterwards the halo goes through the exchange callback of
 * tse_init_args again.  For a host that found tse_comm_init failing on some rank and falls back to its own transport. */
int tse_comm_abort(tse_ctx *ctx);
/* number of local elements that touch another rank (computed first in every stage) and that do not */
int tse_boundary_layout(tse_ctx *ctx, int *n_boundary, int *n_interior);
/* the same in patches (the blocks of the DSS-on-read kernels): the first launch of every stage covers the boundary patches.
 * With TSE_BOUNDARY_STRIPS=1 the rank-boundary elements are grouped into patches of their own (a two-deep band), which halves it. */
int tse_patch_layout(tse_ctx *ctx, int *np_boundary, int *np_interior);

/* Optional: declare the host's element array (elem(1) .. elem(nelemd), contiguous, alive until tse_finalize).  It is page-locked
 * once and every later field copy whose host side lies inside it is a single 2-D DMA with pitch = sizeof(element_t) straight
 * from/to elem(:); host pointers outside a registered range are staged through the library's own pinned buffers.  Returns 0 also
 * when the range is not registered (larger than TSE_PIN_LIMIT_GB, default 64, or refused by the driver). */
int tse_host_register(tse_ctx *ctx, void *base, size_t bytes);
/* elem(ie)%state%Qdp(np,np,nlev,qsize_d,2) <-> device, time level nt (1|2); qsize_d = host array extent */
int tse_copy_qdp_h2d(tse_ctx *ctx, const double *qdp_elem1, size_t elem_stride, int qsize_d, int nt);
int tse_copy_qdp_d2h(tse_ctx *ctx, double *qdp_elem1, size_t elem_stride, int qsize_d, int nt);

/* per-step inputs the caller (prim_step/prim_advance_exp) leaves in elem%derived:
 * vn0(np,np,2,nlev), dp(np,np,nlev), eta_dot_dpdn(np,np,nlevp), omega_p(np,np,nlev); one common element stride per array.
 * NULL pointers are skipped. */
int tse_set_derived(tse_ctx *ctx, const double *vn0, size_t vn0_stride, const double *dp, size_t dp_stride,
                    const double *eta_dot_dpdn, size_t eta_stride, const double *omega_p, size_t omega_stride);
/* derived%divdp / derived%divdp_proj as the host computed them (Prim_Advec_Tracers_remap_rk2 does this outside the
 * hooked routines, prim_advection_mod.F90:614-623); tse_compute_divdp is the on-device equivalent. */
int tse_set_divdp(tse_ctx *ctx, const double *divdp, size_t s0, const double *divdp_proj, size_t s1);
/* outputs the path writes back into elem: derived%divdp_proj, derived%eta_dot_dpdn (DSS'd, levels 1:nlev),
 * derived%omega_p (DSS'd), derived%divdp, state%dp3d(:,:,:,np1), state%ps_v(:,:,np1).  NULL pointers are skipped. */
int tse_get_derived(tse_ctx *ctx, double *divdp_proj, size_t s1, double *eta_dot_dpdn, size_t s2, double *omega_p, size_t s3,
                    double *divdp, size_t s4, double *dp3d, size_t s5, double *ps_v, size_t s6);

/* Prim_Advec_Tracers_remap_rk2: divdp = div(vn0); 3 x euler_step(dt/2); qdp_time_avg */
int tse_advec_tracers_remap_rk2(tse_ctx *ctx, double dt, int n0_qdp, int np1_qdp);
/* the pieces, with the reference's argument meaning */
int tse_compute_divdp(tse_ctx *ctx);
int tse_euler_step(tse_ctx *ctx, int np1_qdp, int n0_qdp, double dt, int DSSopt, int rhs_multiplier);
int tse_qdp_time_avg(tse_ctx *ctx, int rkstage, int n0_qdp, int np1_qdp);
/* vertical_remap; returns 2 on "negative layer thickness" (prim_advection_mod.F90:1323).  The reference aborts the whole
 * job there (abortmp = MPI_Abort): a caller with several ranks must escalate a return of 2 the same way (every rank
 * exits non-zero), otherwise the healthy ranks block in the next halo exchange. */
int tse_vertical_remap(tse_ctx *ctx, double dt, int np1_qdp);
/* The library caches the element min/max of Qdp(np1)/dp that the last kernel of a tracer step emits and reuses them as
 * the next step's first-stage bounds (prim_advection_mod.F90:765-779 recomputes them).  Every entry point that changes
 * Qdp or dp drops the cache; a caller that writes state through tse_device_ptr must call this itself. */
int tse_invalidate_cache(tse_ctx *ctx);

/* Single calls of the public routines the path is built from, on host arrays (one call covers every local element):
 *   divergence_sphere(v,deriv,elem)        derivative_mod.F90:2364-2414   v[ie][2][np*np] -> div[ie][np*np]
 *   laplace_sphere_wk(s,deriv,elem,.true.) derivative_mod.F90:2418-2460   s[ie][np*np]    -> lap[ie][np*np]
 *   remap_Q_ppm(Qdp,np,qsize,dp1,dp2)      prim_advection_mod.F90:98-214  Qdp[ie][qsize][nlev][np*np] in place,
 *                                                                          dp1, dp2[ie][nlev][np*np]
 * They run the same device routines as the fused kernels (operator-level parity checks; not used by the time loop;
 * tse_remap_q_ppm overwrites time level 1 of the device tracer state and the dp/divdp_proj/dp3d level fields). */
int tse_divergence_sphere(tse_ctx *ctx, const double *v, double *div);
int tse_laplace_sphere_wk(tse_ctx *ctx, const double *s, double *lap);
int tse_remap_q_ppm(tse_ctx *ctx, double *Qdp, const double *dp1, const double *dp2);

/* qmin/qmax(nlev,qsize,nelemd) module state of prim_advection_mod (:459), for inspection: out[ie][q][k].  Maintained as in the
 * reference by tse_euler_step (every stage leaves the bounds its limiter used); tse_advec_tracers_remap_rk2 keeps only what a later
 * stage reads (nothing reads them after stage 2, stage 3 and the next step recompute theirs) */
int tse_get_qminmax(tse_ctx *ctx, double *qmin, double *qmax);

/* ---- "next" rows: on-device prescribed fields + device-resident prim_run loop ---- */
/* lat/lon: elem(ie)%spherep(np,np)%{lat,lon} dense [nelemd][np*np]; hyam/hybm(nlev) */
int tse_dcmip_init(tse_ctx *ctx, int test_case /*1: dcmip1-1, 2: dcmip1-2*/, const double *lat, const double *lon,
                   const double *hyam, const double *hybm);
/* set_dcmip_*_fields(time=0) + Qdp = Q*dp for both time levels (prim_driver_mod.F90:548-556,646-669) */
int tse_dcmip_set_initial(tse_ctx *ctx);
/* what prim_step + prim_advance_exp produce for the step that starts at tl%nstep = nstep */
int tse_dcmip_step_inputs(tse_ctx *ctx, int nstep, double tstep);
/* prim_run_subcycle x nsub: rsplit x (step inputs + tracer step) + vertical_remap; *nstep is tl%nstep in/out.  *nstep may lie inside
 * an rsplit cycle (steps taken through tse_advec_tracers_remap_rk2): the first of the nsub cycles then completes that cycle.
 * Returns 2 on "negative layer thickness" (prim_advection_mod.F90:1323) with *nstep = the step count at the end of the FIRST
 * failing cycle; the flag is polled two cycles behind the launches (the host never waits for the device), so up to two
 * further cycles may have been started on the bad state.  Several ranks: escalate as for tse_vertical_remap. */
int tse_prim_run_subcycle(tse_ctx *ctx, double tstep, int nsub, int *nstep);

/* ---- diagnostics ---- */
/* out[ie][q] = sum_k sum_ij spheremp(i,j,ie) * Qdp(i,j,k,q,nt): the element's share of global_integral of the tracer mass
 * (global_norms_mod.F90:39-86, prim_state_mod.F90:352-385), summed in a fixed order inside the element so that it is
 * bit-identical however the elements are distributed; add the elements up with an exact / fixed-order sum for the
 * reference's task-count-independent result (repro_sum's purpose, global_norms_mod.F90:66-68). */
int tse_element_mass(tse_ctx *ctx, int nt, double *out);
/* The element shares of the two integrals prim_printstate's "Q<q>,Q diss, dQ^2/dt:" line is made of (prim_state_mod.F90:352-385):
 * mass[ie][q] = sum_ij spheremp * (sum_k Qdp) and var[ie][q] = sum_ij spheremp * (sum_k Qdp*Q), Q = Qdp/dp with
 * dp = dhyai*ps0 + dhybi*ps_v(np1) (prim_diag_scalars, :604-655; prim_driver_mod.F90:810-815), operations in the reference's order
 * (levels inside a point, then the points i fastest as global_integral adds them, global_norms_mod.F90:74-80).
 * qmin/qmax[ie][q] (may be null) = the element's minimum / maximum of Q, for the "qv=" line (:184-192). */
int tse_element_qdiag(tse_ctx *ctx, int nt, double *mass, double *var, double *qmin, double *qmax);

/* ---- introspection for tests and the benchmark harness ---- */
/* device pointers of internal fields: "qdp1", "qdp2" [nelemd][qsize][nlev][16] (the two time levels are two allocations), "vn0", "dp", "divdp", "divdp_proj",
 * "eta_dot_dpdn", "omega_p", "dp3d", "ps_v", "qmin", "qmax", "sendbuf", "recvbuf" */
void *tse_device_ptr(tse_ctx *ctx, const char *name, size_t *nbytes);
/* accumulated HIP-event time (ms) and launch count of a named kernel group since the last reset; names:
 * "advance" (= "advance0" + "advance1" + "advance2", the three RK stages), "dss", "lap", "minmax", "remap", "level", "dcmip", "avg" */
int tse_kernel_time(tse_ctx *ctx, const char *name, double *ms, long *launches);
int tse_timing(tse_ctx *ctx, int enable); /* enable/disable + reset per-kernel event timing */
/* where the five tracer-sized fields were placed (device memory is not uniform for writes and the rate is a property of the
 * allocation: tse_init tries field-sized chunks -- up to TSE_PLACEMENT, default 20, in all, up to 8 held at a time, the slowest given
 * back and the next one allocated behind a small pad -- until three of them take a streaming write at TSE_PLACEMENT_GOOD, default
 * 6000 GB/s, and keeps the five fastest): number of chunks tried (0: no choice was made), their write rates in
 * GB/s in the order tried, and which try became T, Qdp(1), Qdp(2), B, C */
int tse_placement(tse_ctx *ctx, int *ntried, double *write_gbs /* [32] */, int *chosen /* [5] */);
int tse_halo_layout(tse_ctx *ctx, int *ncol_send, int *ncol_recv);
/* per-slot entry counts of the kind-1 (min/max) exchange, in send-slot / recv-slot order */
int tse_halo_minmax_layout(tse_ctx *ctx, int *send_len, int *recv_len);

#ifdef __cplusplus
}
#endif
#endif
