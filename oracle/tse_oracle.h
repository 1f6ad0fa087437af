/* tse_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's tracer hot path, used as the parity checker by tests/,
 * by __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing under transport_se_amd/ may
 * include, link or call this.  Pinned against outputs of the reference itself (oracle/_ref, built by
 * oracle/ref/Makefile) through the fixtures under tests/golden/ -- see oracle/README.md.
 *
 * Array layouts (C order, last index fastest), np=4, NLEV=72, point p = j*4+i:
 *   tracer field   Qdp[tl][ie][q][k][p]          (reference: elem(ie)%state%Qdp(i,j,k,q,tl))
 *   level field    dp[ie][k][p]                  (elem(ie)%derived%dp(i,j,k))
 *   vn0[ie][k][c][p]                             (elem(ie)%derived%vn0(i,j,c,k))
 *   eta_dot_dpdn[ie][k][p], k < NLEV+1
 *   Dinv[ie][p][b][a]  = Dinv(a,b,i,j)  (a fastest, as in Fortran memory)
 *   Dvv[l*4+i] = Dvv(i,l)
 *   qmin/qmax[ie][q][k]
 * Directions (0-based; reference control_mod.F90:173-181 minus 1): W=0 E=1 S=2 N=3 SW=4 SE=5 NW=6 NE=7.
 */
#ifndef TSE_ORACLE_H
#define TSE_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NP 4
#define ORC_NPSQ 16
#define ORC_NLEV 72
#define ORC_NLEVP 73

typedef struct orc_s orc_t;

/* GLL points/weights and differentiation matrix, computed in __float128 then rounded
 * (quadrature_mod.F90:305-470, derivative_mod.F90:116-192,451-486). */
void orc_gll(double pts[4], double wts[4]);
void orc_dvv(double dvv[16]);

/* create a context for the uniform cubed sphere with ne x ne elements per face, qsize tracers.
 * hya/hyb arrays follow hybvcoord_mod.F90:36-171 (hyai,hybi: 73; hyam,hybm: 72). */
orc_t *orc_create(int ne, int qsize, const double *hyai, const double *hybi,
                  const double *hyam, const double *hybm);
void orc_destroy(orc_t *o);

/* named access to internal arrays: returns pointer, writes element count to *n (0 if unknown name).
 * doubles: lat lon D Dinv metdet rmetdet mp spheremp rspheremp Dvv hyai hybi alpha
 *          qdp(2 time levels) vn0 dp divdp divdp_proj eta_dot_dpdn omega_p dp3d ps_v qmin qmax qtens
 * ints:    nbr_elem nbr_dir nbr_rev face putmap getmap reverse */
double *orc_dptr(orc_t *o, const char *name, long *n);
int    *orc_iptr(orc_t *o, const char *name, long *n);

/* run-time parameters (control_mod: nu_q, rsplit; time_mod: tstep) */
void orc_set_params(orc_t *o, double nu_q, int rsplit);
void orc_set_threads(int nthreads);

/* element-local operators on one 16-point slab of element ie */
void orc_divergence_sphere(const orc_t *o, int ie, const double v[32], double div[16]);
void orc_gradient_sphere(const orc_t *o, int ie, const double s[16], double ds[32]);
void orc_divergence_sphere_wk(const orc_t *o, int ie, const double v[32], double div[16]);
void orc_laplace_sphere_wk(const orc_t *o, int ie, const double s[16], double lap[16]);

/* limiter_optim_iter_full (prim_advection_mod.F90:976-1094); returns the iteration count used */
int orc_limiter8(double ptens[16], const double sphweights[16], double *minp, double *maxp,
                 const double dpmass[16]);

/* remap_Q_ppm for one element: Qdp[q][k][p] in/out, dp1/dp2[k][p] (prim_advection_mod.F90:98-214) */
void orc_remap_q_ppm(double *Qdp, int qsize, const double *dp1, const double *dp2);
/* control_mod's vert_remap_q_alg (control_mod.F90:61-66) for every later remap of this process: 0|1 mirrored ghost cells
 * (default), 2 piecewise-constant boundary cells (prim_advection_mod.F90:230-250,283-341) */
void orc_set_vert_remap_q_alg(int alg);

/* DSS of an arbitrary nlyr-layer field f[ie][lyr][p] (edgeVpack + bndry_exchangeV + edgeVunpack,
 * edge_mod.F90:366-511,648-742), op: 0 sum, 1 min, 2 max */
void orc_dss(const orc_t *o, double *f, int nlyr, int op);

/* hot path entry points (prim_advection_mod.F90:579-640, 667-970, 645-662, 1242-1330) */
void orc_euler_step(orc_t *o, int np1_qdp, int n0_qdp, double dt, int dssopt, int rhs_multiplier);
void orc_advec_tracers_remap_rk2(orc_t *o, double dt, int nstep);
int  orc_vertical_remap(orc_t *o, double dt, int np1_qdp);
void orc_qdp_levels(int nstep, int *n0_qdp, int *np1_qdp);

/* prescribed fields (dcmip_wrapper_mod.F90:49-243, dcmip_123_mod.F90:85-409) */
void orc_dcmip_point(int test, double time, double lon, double lat, double z,
                     double *u, double *v, double *w, double *p, double *rho, double q[4]);
void orc_dcmip_init(orc_t *o, int test);                 /* prim_init2: state at t=0 */
void orc_dcmip_step_inputs(orc_t *o, int test, int nstep, double tstep);  /* prim_step + prim_advance_exp */

/* prim_run_subcycle loop: nsub remap cycles of rsplit tracer steps; returns tracer steps done (<0 on
 * negative layer thickness) */
int orc_prim_run(orc_t *o, int test, double tstep, int nsub, int *nstep_io);

#ifdef __cplusplus
}
#endif
#endif
