"""pyoracle.py -- TEST INFRASTRUCTURE ONLY: ctypes front end of oracle/liboracle.so (the CPU restatement of
the reference hot path) plus readers for the raw dumps written by oracle/_ref/ref_harness.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under transport_se_amd/ does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NLEV, NLEVP, NP = 72, 73, 4
VCOORD_DIR = os.path.join(os.path.dirname(HERE), "transport_se_amd", "data", "vcoord")   # the one copy of the ACME 72-level tables (data)


def read_vcoord(vdir=VCOORD_DIR):
    """acme-72{i,m}.ascii as hvcoord_init reads them (reference src/share/hybvcoord_mod.F90:57-105):
    a count line then values, A block then B block."""
    def rd(path):
        toks = []
        for line in open(path):
            line = line.split("!")[0].strip()
            if line:
                toks += line.split()
        n = int(toks[0]); a = np.array(toks[1:1 + n], dtype=np.float64)
        assert int(toks[1 + n]) == n
        b = np.array(toks[2 + n:2 + 2 * n], dtype=np.float64)
        return a, b
    hyai, hybi = rd(os.path.join(vdir, "acme-72i.ascii"))
    hyam, hybm = rd(os.path.join(vdir, "acme-72m.ascii"))
    assert hyai.size == NLEVP and hyam.size == NLEV
    return hyai, hybi, hyam, hybm


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "tse_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_dptr.restype = dp
        L.orc_dptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
        L.orc_iptr.restype = C.POINTER(C.c_int)
        L.orc_iptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
        L.orc_set_params.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_set_threads.argtypes = [C.c_int]
        for f in ("orc_divergence_sphere", "orc_gradient_sphere", "orc_divergence_sphere_wk", "orc_laplace_sphere_wk"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_limiter8.restype = C.c_int
        L.orc_limiter8.argtypes = [dp, dp, dp, dp, dp]
        L.orc_remap_q_ppm.argtypes = [dp, C.c_int, dp, dp]
        L.orc_set_vert_remap_q_alg.argtypes = [C.c_int]
        L.orc_dss.argtypes = [C.c_void_p, dp, C.c_int, C.c_int]
        L.orc_euler_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_advec_tracers_remap_rk2.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_vertical_remap.restype = C.c_int
        L.orc_vertical_remap.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.orc_dcmip_init.argtypes = [C.c_void_p, C.c_int]
        L.orc_dcmip_step_inputs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.orc_prim_run.restype = C.c_int
        L.orc_prim_run.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.orc_gll.argtypes = [dp, dp]
        L.orc_dvv.argtypes = [dp]
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One cubed-sphere context of the CPU restatement; arrays are numpy views of the C memory."""

    def __init__(self, ne, qsize, nu_q=0.0, rsplit=3, threads=1, vcoord=None):
        L = lib()
        self.ne, self.qsize, self.nelem = ne, qsize, 6 * ne * ne
        self.hyai, self.hybi, self.hyam, self.hybm = vcoord if vcoord is not None else read_vcoord()
        L.orc_set_threads(threads)
        self.h = L.orc_create(ne, qsize, _p(self.hyai), _p(self.hybi), _p(self.hyam), _p(self.hybm))
        L.orc_set_params(self.h, nu_q, rsplit)
        self.nu_q, self.rsplit = nu_q, rsplit
        n, q = self.nelem, qsize
        shapes = dict(lat=(n, 4, 4), lon=(n, 4, 4), D=(n, 4, 4, 2, 2), Dinv=(n, 4, 4, 2, 2), metdet=(n, 4, 4),
                      rmetdet=(n, 4, 4), mp=(n, 4, 4), spheremp=(n, 4, 4), rspheremp=(n, 4, 4), Dvv=(4, 4),
                      gll=(4,), gllw=(4,), alpha=(1,),
                      qdp=(2, n, q, NLEV, 4, 4), qtens=(n, q, NLEV, 4, 4), vn0=(n, NLEV, 2, 4, 4),
                      dp=(n, NLEV, 4, 4), divdp=(n, NLEV, 4, 4), divdp_proj=(n, NLEV, 4, 4),
                      eta_dot_dpdn=(n, NLEVP, 4, 4), omega_p=(n, NLEV, 4, 4), dp3d=(n, NLEV, 4, 4),
                      ps_v=(n, 4, 4), qmin=(n, q, NLEV), qmax=(n, q, NLEV))
        for name, shp in shapes.items():
            cnt = C.c_long()
            ptr = L.orc_dptr(self.h, name.encode(), C.byref(cnt))
            assert cnt.value == int(np.prod(shp)), (name, cnt.value, shp)
            setattr(self, name, np.ctypeslib.as_array(ptr, shape=shp))
        for name in ("nbr_elem", "nbr_dir", "nbr_rev", "putmap", "getmap", "reverse"):
            cnt = C.c_long()
            ptr = L.orc_iptr(self.h, name.encode(), C.byref(cnt))
            setattr(self, name, np.ctypeslib.as_array(ptr, shape=(n, 8)))
        cnt = C.c_long()
        self.face = np.ctypeslib.as_array(L.orc_iptr(self.h, b"face", C.byref(cnt)), shape=(n,))

    def close(self):
        if self.h:
            lib().orc_destroy(self.h)
            self.h = None

    # element-local operators (slab = [4 j][4 i])
    def divergence_sphere(self, ie, v):
        out = np.empty((4, 4)); lib().orc_divergence_sphere(self.h, ie, _p(np.ascontiguousarray(v)), _p(out)); return out

    def gradient_sphere(self, ie, s):
        out = np.empty((2, 4, 4)); lib().orc_gradient_sphere(self.h, ie, _p(np.ascontiguousarray(s)), _p(out)); return out

    def divergence_sphere_wk(self, ie, v):
        out = np.empty((4, 4)); lib().orc_divergence_sphere_wk(self.h, ie, _p(np.ascontiguousarray(v)), _p(out)); return out

    def laplace_sphere_wk(self, ie, s):
        out = np.empty((4, 4)); lib().orc_laplace_sphere_wk(self.h, ie, _p(np.ascontiguousarray(s)), _p(out)); return out

    def dss(self, f, op=0):
        f = np.ascontiguousarray(f, dtype=np.float64)
        nlyr = int(f.size // (self.nelem * 16))
        lib().orc_dss(self.h, _p(f), nlyr, op)
        return f

    def euler_step(self, np1_qdp, n0_qdp, dt, dssopt, rhs_multiplier):
        lib().orc_euler_step(self.h, np1_qdp, n0_qdp, dt, dssopt, rhs_multiplier)

    def advec_tracers_remap_rk2(self, dt, nstep):
        lib().orc_advec_tracers_remap_rk2(self.h, dt, nstep)

    def vertical_remap(self, dt, np1_qdp):
        return lib().orc_vertical_remap(self.h, dt, np1_qdp)

    def dcmip_init(self, test=1):
        lib().orc_dcmip_init(self.h, test)

    def dcmip_step_inputs(self, test, nstep, tstep):
        lib().orc_dcmip_step_inputs(self.h, test, nstep, tstep)

    def prim_run(self, test, tstep, nsub, nstep=0):
        ns = C.c_int(nstep)
        done = lib().orc_prim_run(self.h, test, tstep, nsub, C.byref(ns))
        return done, ns.value


def limiter8(ptens, sphweights, minp, maxp, dpmass):
    pt = np.array(ptens, dtype=np.float64).reshape(16).copy()
    mn, mx = C.c_double(minp), C.c_double(maxp)
    it = lib().orc_limiter8(_p(pt), _p(np.ascontiguousarray(sphweights, dtype=np.float64).reshape(16)),
                            C.byref(mn), C.byref(mx), _p(np.ascontiguousarray(dpmass, dtype=np.float64).reshape(16)))
    return pt.reshape(4, 4), mn.value, mx.value, it


def set_vert_remap_q_alg(alg):
    """control_mod's vert_remap_q_alg for every later remap of this process (0|1 mirrored ghosts, 2 piecewise-constant ends)"""
    lib().orc_set_vert_remap_q_alg(int(alg))


def remap_q_ppm(Qdp, dp1, dp2):
    """Qdp[q][k][4][4], dp1/dp2[k][4][4] -> remapped copy"""
    Q = np.array(Qdp, dtype=np.float64, order="C")
    lib().orc_remap_q_ppm(_p(Q), Q.shape[0], _p(np.ascontiguousarray(dp1, dtype=np.float64)),
                          _p(np.ascontiguousarray(dp2, dtype=np.float64)))
    return Q


# ---------------------------------------------------------------------------------------------------
# readers for oracle/_ref/ref_harness dumps (Fortran stream files; arrays come out in C order with the
# Fortran index order reversed, e.g. Qdp(i,j,k,q) -> [q][k][j][i])
class _Rd:
    def __init__(self, path):
        self.b = open(path, "rb").read(); self.o = 0

    def i4(self, n=1):
        a = np.frombuffer(self.b, dtype="<i4", count=n, offset=self.o); self.o += 4 * n; return a.copy()

    def f8(self, *shape):
        n = int(np.prod(shape)) if shape else 1
        a = np.frombuffer(self.b, dtype="<f8", count=n, offset=self.o); self.o += 8 * n
        return a.reshape(shape).copy() if shape else float(a[0])

    def done(self):
        return self.o == len(self.b)


def read_static(path):
    r = _Rd(path)
    ne, nelem, nelemd, qsize, nlev, np_, rank, nprocs = r.i4(8)
    d = dict(ne=int(ne), nelem=int(nelem), nelemd=int(nelemd), qsize=int(qsize), rank=int(rank), nprocs=int(nprocs))
    d["alpha"] = r.f8(); d["Dvv"] = r.f8(4, 4); d["gll"] = r.f8(4); d["gllw"] = r.f8(4)
    d["hyai"] = r.f8(NLEVP); d["hybi"] = r.f8(NLEVP); d["hyam"] = r.f8(NLEV); d["hybm"] = r.f8(NLEV); d["ps0"] = r.f8()
    keys = ("gid", "putmap", "getmap", "reverse", "face", "lon", "lat", "D", "Dinv", "metdet", "rmetdet", "mp",
            "spheremp", "rspheremp")
    acc = {k: [] for k in keys}
    for _ in range(nelemd):
        acc["gid"].append(r.i4(1)[0]); acc["putmap"].append(r.i4(8)); acc["getmap"].append(r.i4(8))
        acc["reverse"].append(r.i4(8)); acc["face"].append(r.i4(1)[0])
        acc["lon"].append(r.f8(4, 4)); acc["lat"].append(r.f8(4, 4))
        acc["D"].append(r.f8(4, 4, 2, 2)); acc["Dinv"].append(r.f8(4, 4, 2, 2))
        for k in ("metdet", "rmetdet", "mp", "spheremp", "rspheremp"):
            acc[k].append(r.f8(4, 4))
    for k in keys:
        d[k] = np.array(acc[k])
    ncyc, nsend, nrecv = r.i4(3)
    d["send_cycles"] = np.array([r.i4(3) for _ in range(nsend)]).reshape(-1, 3)  # dest(1-based rank), ptrP, lengthP
    d["move_cycle"] = r.i4(2)
    assert r.done()
    return d


def read_ops(path):
    r = _Rd(path)
    ne_, nq = r.i4(2)
    ops = []
    for _ in range(ne_):
        ie = int(r.i4(1)[0]) - 1
        ops.append(dict(ie=ie, s=r.f8(4, 4), v=r.f8(2, 4, 4), div=r.f8(4, 4), grad=r.f8(2, 4, 4),
                        div_wk=r.f8(4, 4), lap_wk=r.f8(4, 4)))
    remaps = []
    for _ in range(2):
        remaps.append(dict(dp1=r.f8(NLEV, 4, 4), dp2=r.f8(NLEV, 4, 4), Qin=r.f8(nq, NLEV, 4, 4), Qout=r.f8(nq, NLEV, 4, 4)))
    assert r.done()
    return ops, remaps


def read_state(path):
    r = _Rd(path)
    istep, nq, nelemd, qsize = [int(x) for x in r.i4(4)]
    keys = ("qdp", "vn0", "dp", "divdp", "divdp_proj", "eta_dot_dpdn", "omega_p", "dp3d", "ps_v")
    acc = {k: [] for k in keys}
    for _ in range(nelemd):
        acc["qdp"].append(r.f8(qsize, NLEV, 4, 4)); acc["vn0"].append(r.f8(NLEV, 2, 4, 4))
        for k in ("dp", "divdp", "divdp_proj", "eta_dot_dpdn", "omega_p", "dp3d"):
            acc[k].append(r.f8(NLEV, 4, 4))
        acc["ps_v"].append(r.f8(4, 4))
    assert r.done()
    d = {k: np.array(v) for k, v in acc.items()}
    d.update(istep=istep, nq=nq)
    return d
