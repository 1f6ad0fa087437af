"""hip_mod -- Python mirror of the Fortran host seam.

The reference plugs its accelerator path in through `cuda_mod` (src/share/cuda_mod.F90:57-64):
    cuda_mod_init(elem,hybrid,deriv,hvcoord), copy_qdp_h2d(elem,nt), copy_qdp_d2h(elem,nt),
    euler_step_cuda(np1_qdp,n0_qdp,dt,elem,...,DSSopt,rhs_multiplier), qdp_time_avg_cuda(...), vertical_remap_cuda(...)
`HipMod` exposes the same operations with the same argument meaning over the C ABI of
include/transport_se_hip.h; `transport_se_amd/fortran/cuda_mod_hip.F90` is the ISO_C_BINDING twin a Fortran host uses.

`elem` here is a dict of dense numpy arrays holding the element_t fields the path touches, in the reference's
own index order reversed to C order (element index first):
    Qdp[ie][tl][q][k][j][i]  (state%Qdp(np,np,nlev,qsize_d,2)),  vn0[ie][k][c][j][i], dp/divdp/divdp_proj/omega_p[ie][k][j][i],
    eta_dot_dpdn[ie][nlevp][j][i], Dinv[ie][j][i][b][a], metdet/rmetdet/spheremp/rspheremp[ie][j][i],
    putmapP/getmapP/reverse[ie][8]
"""
import ctypes as C

import numpy as np

from . import _lib

NP, NLEV, NLEVP = 4, 72, 73
DSSeta, DSSomega, DSSdiv_vdp_ave = 1, 2, 3  # prim_advection_mod.F90:454-456


class TseError(RuntimeError):
    """what the Fortran side turns into abortmp(msg) (parallel_mod.F90:274-287)"""


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class HipMod:
    def __init__(self, elem, deriv_Dvv, hvcoord, qsize, nu_q, limiter_option=8, rsplit=3, device=-1,
                 schedule=None, exchange=None, vert_remap_q_alg=0, lib_path=None):
        """cuda_mod_init.  hvcoord = (hyai, hybi, ps0).  schedule = dict(send=[(peer, ptrP, lengthP)...],
        recv=[...]) as in Schedule(1)%SendCycle/RecvCycle; exchange(sendbuf_ptr, recvbuf_ptr, nlyr, kind) -> 0.
        lib_path: another build of the same sources (_lib.HOOKS_SO: the tests' fault injection) instead of the product library."""
        L = _lib.lib(lib_path)
        self.L = L
        self.qsize = int(qsize)
        self.nelemd = int(elem["metdet"].shape[0])
        self._keep = []
        a = _lib.InitArgs()
        a.nelemd, a.qsize, a.device, a.nu_q = self.nelemd, self.qsize, device, float(nu_q)
        a.limiter_option, a.rsplit = int(limiter_option), int(rsplit)
        a.vert_remap_q_alg = int(vert_remap_q_alg)   # control_mod.F90:61-66 (0|1 mirrored ghost cells, 2 piecewise-constant ends)

        def keep(x, dtype):
            x = np.ascontiguousarray(x, dtype=dtype); self._keep.append(x); return x
        hyai, hybi, ps0 = hvcoord
        a.Dvv = _vp(keep(deriv_Dvv, np.float64)); a.hyai = _vp(keep(hyai, np.float64)); a.hybi = _vp(keep(hybi, np.float64))
        a.ps0 = float(ps0)
        for name, cnt in (("Dinv", 64), ("metdet", 16), ("rmetdet", 16), ("spheremp", 16), ("rspheremp", 16)):
            arr = keep(elem[name], np.float64)
            assert arr.size == self.nelemd * cnt, name
            setattr(a, name, _vp(arr)); setattr(a, name + "_stride", cnt * 8)
        a.putmapP = _vp(keep(elem["putmapP"], np.int32)); a.getmapP = _vp(keep(elem["getmapP"], np.int32))
        a.reverse = _vp(keep(elem["reverse"], np.int32))
        sched = schedule or dict(send=[], recv=[])
        for side in ("send", "recv"):
            cyc = np.array(sched[side], dtype=np.int32).reshape(-1, 3)
            setattr(a, "n" + side, cyc.shape[0])
            for col, nm in enumerate(("peer", "ptrP", "lengthP")):
                setattr(a, "%s_%s" % (side, nm), _vp(keep(cyc[:, col], np.int32)))
        self.schedule = sched
        if exchange is not None:
            def _cb(user, sbuf, rbuf, nlyr, kind):
                try:
                    return int(exchange(sbuf, rbuf, nlyr, kind) or 0)
                except Exception as ex:  # noqa: BLE001
                    print("exchange callback failed:", ex)
                    return 1
            self._cb = _lib.EXCHANGE_FN(_cb)
            a.exchange = self._cb
        h = C.c_void_p()
        self._chk(L.tse_init(C.byref(h), C.byref(a)))
        self.h = h
        # per-slot entry counts of the compact (kind 1) min/max exchange, for the exchange object
        ns, nr = len(sched["send"]), len(sched["recv"])
        sl = np.zeros(max(ns, 1), dtype=np.int32); rl = np.zeros(max(nr, 1), dtype=np.int32)
        L.tse_halo_minmax_layout(self.h, _vp(sl), _vp(rl))
        self.minmax_send_len, self.minmax_recv_len = sl[:ns].copy(), rl[:nr].copy()
        if exchange is not None and hasattr(exchange, "set_minmax_layout"):
            exchange.set_minmax_layout(self.minmax_send_len, self.minmax_recv_len)

    def _chk(self, rc):
        if rc:
            raise TseError(self.L.tse_last_error().decode())

    # ---- in-library bndry_exchangeV: RCCL communicator (include/transport_se_hip.h, tse_comm_*) ----
    @staticmethod
    def comm_unique_id():
        """bytes of a fresh communicator id (rank 0 calls this and hands it to every rank)"""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        L = _lib.lib()
        if L.tse_comm_unique_id(buf):
            raise TseError(L.tse_last_error().decode())
        return buf.raw

    def comm_precheck(self, rank, nranks):
        """everything comm_init can find wrong without the other ranks (raises TseError); hosts agree on the outcome over
        their control plane BEFORE anyone enters the blocking collective comm_init"""
        self._chk(self.L.tse_comm_precheck(self.h, int(rank), int(nranks)))

    @staticmethod
    def comm_version():
        """dict(runtime=, built=, path=): the RCCL this process resolved vs. the headers the library was built with"""
        rt, bl = C.c_int(), C.c_int()
        buf = C.create_string_buffer(512)
        L = _lib.lib()
        if L.tse_comm_version(C.byref(rt), C.byref(bl), buf, 512):
            raise TseError(L.tse_last_error().decode())
        fmt = lambda v: "%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100)   # noqa: E731
        return dict(runtime=fmt(rt.value), built=fmt(bl.value), path=buf.value.decode())

    def comm_init(self, comm_id, rank, nranks):
        """collective over all ranks; afterwards the library exchanges the halo itself (no callback)"""
        assert len(comm_id) == _lib.COMM_ID_BYTES
        self._chk(self.L.tse_comm_init(self.h, C.c_char_p(comm_id), int(rank), int(nranks)))

    def comm_abort(self):
        """drop the communicator (if any): the halo goes through the exchange callback again"""
        self._chk(self.L.tse_comm_abort(self.h))

    def comm_info(self):
        r, n = C.c_int(), C.c_int()
        self._chk(self.L.tse_comm_info(self.h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def boundary_layout(self):
        a, b = C.c_int(), C.c_int()
        self.L.tse_boundary_layout(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def patch_layout(self):
        """(patches that touch another rank, patches that do not) of the storage tiling"""
        a, b = C.c_int(), C.c_int()
        self.L.tse_patch_layout(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def placement(self):
        """placement of the five tracer-sized fields by tse_init: dict(tried=n, write_GBs=[... in the order tried], chosen=[T, Qdp1, Qdp2, B, C]
        as indices into the tries); tried 0 = no choice made"""
        n = C.c_int(); bw = (C.c_double * 32)(); sel = (C.c_int * 5)()
        self.L.tse_placement(self.h, C.byref(n), bw, sel)
        return dict(tried=n.value, write_GBs=[bw[i] for i in range(n.value)], chosen=[sel[i] for i in range(5)])

    def invalidate_cache(self):
        self.L.tse_invalidate_cache(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.tse_finalize(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ---- copy_qdp_h2d / copy_qdp_d2h (cuda_mod.F90:429-469) ----
    def copy_qdp_h2d(self, elem, nt):
        q = elem["Qdp"]; assert q.dtype == np.float64 and q.flags.c_contiguous
        self._chk(self.L.tse_copy_qdp_h2d(self.h, _vp(q), q.strides[0], q.shape[2], nt))

    def copy_qdp_d2h(self, elem, nt):
        q = elem["Qdp"]; assert q.dtype == np.float64 and q.flags.c_contiguous
        self._chk(self.L.tse_copy_qdp_d2h(self.h, _vp(q), q.strides[0], q.shape[2], nt))

    # ---- per-step derived inputs/outputs (what euler_step_cuda stages from elem%derived, cuda_mod.F90:535-547) ----
    def set_derived(self, elem):
        def arg(name):
            x = elem.get(name)
            if x is None:
                return None, 0
            assert x.dtype == np.float64 and x.flags.c_contiguous
            return _vp(x), x.strides[0]
        v, sv = arg("vn0"); d, sd = arg("dp"); e, se = arg("eta_dot_dpdn"); o, so = arg("omega_p")
        self._chk(self.L.tse_set_derived(self.h, v, sv, d, sd, e, se, o, so))

    def set_divdp(self, elem):
        """derived%divdp / divdp_proj as the host computed them (prim_advection_mod.F90:614-623)"""
        args = []
        for name in ("divdp", "divdp_proj"):
            x = elem.get(name)
            args += [_vp(x), x.strides[0]] if x is not None else [None, 0]
        self._chk(self.L.tse_set_divdp(self.h, *args))

    def get_derived(self, elem):
        args = []
        for name in ("divdp_proj", "eta_dot_dpdn", "omega_p", "divdp", "dp3d", "ps_v"):
            x = elem.get(name)
            args += [_vp(x), x.strides[0]] if x is not None else [None, 0]
        self._chk(self.L.tse_get_derived(self.h, *args))

    # ---- the path ----
    def compute_divdp(self):
        self._chk(self.L.tse_compute_divdp(self.h))

    def euler_step(self, np1_qdp, n0_qdp, dt, DSSopt, rhs_multiplier):
        self._chk(self.L.tse_euler_step(self.h, np1_qdp, n0_qdp, dt, DSSopt, rhs_multiplier))

    def qdp_time_avg(self, rkstage, n0_qdp, np1_qdp):
        self._chk(self.L.tse_qdp_time_avg(self.h, rkstage, n0_qdp, np1_qdp))

    def advec_tracers_remap_rk2(self, dt, n0_qdp, np1_qdp):
        self._chk(self.L.tse_advec_tracers_remap_rk2(self.h, dt, n0_qdp, np1_qdp))

    def vertical_remap(self, dt, np1_qdp):
        self._chk(self.L.tse_vertical_remap(self.h, dt, np1_qdp))

    # ---- single calls of the public operators (derivative_mod / vertremap_mod), host arrays in and out ----
    def divergence_sphere(self, v):
        """v[ie][2][np][np] -> div[ie][np][np]  (derivative_mod.F90:2364)"""
        v = np.ascontiguousarray(v, np.float64); assert v.shape == (self.nelemd, 2, NP, NP)
        out = np.empty((self.nelemd, NP, NP))
        self._chk(self.L.tse_divergence_sphere(self.h, _vp(v), _vp(out)))
        return out

    def laplace_sphere_wk(self, s):
        """s[ie][np][np] -> laplace_sphere_wk(s)  (derivative_mod.F90:2418)"""
        s = np.ascontiguousarray(s, np.float64); assert s.shape == (self.nelemd, NP, NP)
        out = np.empty((self.nelemd, NP, NP))
        self._chk(self.L.tse_laplace_sphere_wk(self.h, _vp(s), _vp(out)))
        return out

    def remap_q_ppm(self, qdp, dp1, dp2):
        """remap_Q_ppm(Qdp,np,qsize,dp1,dp2) (prim_advection_mod.F90:98): qdp[ie][q][k][np][np], dp1/dp2[ie][k][np][np] -> new qdp"""
        q = np.ascontiguousarray(qdp, np.float64).copy(); assert q.shape == (self.nelemd, self.qsize, NLEV, NP, NP)
        d1 = np.ascontiguousarray(dp1, np.float64); d2 = np.ascontiguousarray(dp2, np.float64)
        assert d1.shape == d2.shape == (self.nelemd, NLEV, NP, NP)
        self._chk(self.L.tse_remap_q_ppm(self.h, _vp(q), _vp(d1), _vp(d2)))
        return q

    def element_mass(self, nt):
        """[nelemd][qsize] per-element tracer mass of time level nt (fixed summation order; see diagnostics.global_sum)"""
        out = np.empty((self.nelemd, self.qsize))
        self._chk(self.L.tse_element_mass(self.h, int(nt), _vp(out)))
        return out

    def element_qdiag(self, nt):
        """[nelemd][qsize] each: (mass, variance, min Q, max Q) -- the element shares of prim_diag_scalars' Qmass / Qvar integrals of time
        level nt (prim_state_mod.F90:604-655, in the reference's order of operations) and the element extrema of Q = Qdp/dp(ps_v)"""
        out = [np.empty((self.nelemd, self.qsize)) for _ in range(4)]
        self._chk(self.L.tse_element_qdiag(self.h, int(nt), *[_vp(x) for x in out]))
        return tuple(out)

    def get_qminmax(self):
        qmin = np.empty((self.nelemd, self.qsize, NLEV)); qmax = np.empty_like(qmin)
        self._chk(self.L.tse_get_qminmax(self.h, _vp(qmin), _vp(qmax)))
        return qmin, qmax

    # ---- prescribed fields / device-resident loop ----
    def dcmip_init(self, test_case, lat, lon, hyam, hybm):
        self._chk(self.L.tse_dcmip_init(self.h, test_case, _vp(np.ascontiguousarray(lat, np.float64)),
                                        _vp(np.ascontiguousarray(lon, np.float64)),
                                        _vp(np.ascontiguousarray(hyam, np.float64)), _vp(np.ascontiguousarray(hybm, np.float64))))

    def dcmip_set_initial(self):
        self._chk(self.L.tse_dcmip_set_initial(self.h))

    def dcmip_step_inputs(self, nstep, tstep):
        self._chk(self.L.tse_dcmip_step_inputs(self.h, nstep, tstep))

    def prim_run_subcycle(self, tstep, nsub, nstep):
        """nsub remap cycles from step count nstep; returns the new step count.  On "negative layer thickness" the TseError carries
        .rc = 2 and .nstep = the step count at the end of the first failing cycle (prim_advection_mod.F90:1323 aborts there)."""
        ns = C.c_int(nstep)
        rc = self.L.tse_prim_run_subcycle(self.h, tstep, nsub, C.byref(ns))
        if rc:
            err = TseError(self.L.tse_last_error().decode())
            err.rc, err.nstep = rc, ns.value
            raise err
        return ns.value

    def synchronize(self):
        self._chk(self.L.tse_synchronize(self.h))

    # ---- introspection ----
    def device_ptr(self, name):
        n = C.c_size_t()
        p = self.L.tse_device_ptr(self.h, name.encode(), C.byref(n))
        return p, n.value

    def timing(self, enable=True):
        self.L.tse_timing(self.h, int(enable))

    def kernel_time(self, name):
        ms = C.c_double(); n = C.c_long()
        self.L.tse_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def halo_layout(self):
        a, b = C.c_int(), C.c_int()
        self.L.tse_halo_layout(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def fetch(self, name, shape):
        """debug/test helper: copy an internal device array to the host (hipMemcpy through torch-free ctypes)"""
        if name == "qdp":   # the two time levels are two allocations (tse_placement): (2, ...) = [Qdp(..,1), Qdp(..,2)]
            assert shape[0] == 2, shape
            return np.stack([self.fetch("qdp1", shape[1:]), self.fetch("qdp2", shape[1:])])
        p, nbytes = self.device_ptr(name)
        out = np.empty(shape, dtype=np.float64)
        assert out.nbytes <= nbytes, (name, out.nbytes, nbytes)
        self.synchronize()
        hip = C.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(p), C.c_size_t(out.nbytes), C.c_int(2))
        if rc:
            raise TseError("hipMemcpy D2H failed: %d" % rc)
        return out
