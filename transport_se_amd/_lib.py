"""Loader/builder of libtransport_se_hip.so (in-tree; built by __graft_entry__.build())."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libtransport_se_hip.so")
SRC = [os.path.join(HERE, "csrc", f) for f in ("tse_api.hip", "tse_stage3.hip", "tse_kernels.h", "tse_device.h")]
# the translation units and the extra compiler flags of each (tse_stage3.hip says why it has a scheduler strategy of its own)
UNITS = [("tse_api.hip", []), ("tse_stage3.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])]
# -fno-honor-nans: value-preserving (no reassociation, no reciprocal tricks); it only lets the compiler drop the
# v_max_f64 x,x "canonicalize" it otherwise puts in front of every fmin/fmax operand (6 per limiter iteration)
CFLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-honor-nans", "-fPIC"]
HDR = os.path.join(os.path.dirname(HERE), "include", "transport_se_hip.h")

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int)


class InitArgs(C.Structure):
    _fields_ = [
        ("nelemd", C.c_int), ("qsize", C.c_int), ("device", C.c_int), ("nu_q", C.c_double),
        ("limiter_option", C.c_int), ("rsplit", C.c_int),
        ("Dvv", C.c_void_p), ("hyai", C.c_void_p), ("hybi", C.c_void_p), ("ps0", C.c_double),
        ("Dinv", C.c_void_p), ("Dinv_stride", C.c_size_t), ("metdet", C.c_void_p), ("metdet_stride", C.c_size_t),
        ("rmetdet", C.c_void_p), ("rmetdet_stride", C.c_size_t), ("spheremp", C.c_void_p), ("spheremp_stride", C.c_size_t),
        ("rspheremp", C.c_void_p), ("rspheremp_stride", C.c_size_t),
        ("putmapP", C.c_void_p), ("getmapP", C.c_void_p), ("reverse", C.c_void_p),
        ("nsend", C.c_int), ("send_peer", C.c_void_p), ("send_ptrP", C.c_void_p), ("send_lengthP", C.c_void_p),
        ("nrecv", C.c_int), ("recv_peer", C.c_void_p), ("recv_ptrP", C.c_void_p), ("recv_lengthP", C.c_void_p),
        ("exchange", EXCHANGE_FN), ("exchange_user", C.c_void_p),
        ("vert_remap_q_alg", C.c_int),
    ]


HOOKS_SO = os.path.join(HERE, "libtransport_se_hip_hooks.so")
HOOKS_FLAGS = ["-DTSE_AB_HOOKS"]


def _compile_units(tmp, tag, flags, verbose):
    """start the compiler on every translation unit (they run side by side); returns (objs, [(cmd, Popen)])"""
    objs, procs = [], []
    for name, extra in UNITS:
        obj = os.path.join(tmp, tag + name.replace(".hip", ".o"))
        cmd = ["hipcc"] + CFLAGS + extra + list(flags) + ["-c", "-o", obj, os.path.join(HERE, "csrc", name)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd))); objs.append(obj)
    return objs, procs


def _finish(procs):
    """wait for EVERY compile before raising (a compiler left running would write into the temporary directory being removed)"""
    failed = None
    for cmd, p in procs:
        if p.wait() and failed is None:
            failed = subprocess.CalledProcessError(p.returncode, cmd)
    if failed:
        raise failed


def _link(out, objs, verbose):
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]   # RCCL: the in-library bndry_exchangeV (tse_comm_init)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def _stale(so):
    newest = max(os.path.getmtime(p) for p in SRC + [HDR])
    return not (os.path.exists(so) and os.path.getmtime(so) >= newest)


def build(force=False, verbose=False, out=None, flags=(), hooks=True):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Builds the product library and, with hooks=True, its twin with
    -DTSE_AB_HOOKS (libtransport_se_hip_hooks.so: the A/B switches, the fault injection the tests use, the tse_debug_* entry
    points -- none of which the product library contains); all translation units compile side by side.
    out/flags: one A/B variant of the same sources instead (tools/ab_build.sh; always with the hooks)."""
    import tempfile
    if out is not None:
        with tempfile.TemporaryDirectory() as tmp:
            objs, procs = _compile_units(tmp, "ab_", HOOKS_FLAGS + list(flags), verbose)
            _finish(procs)
            _link(out, objs, verbose)
        return out
    todo = [(so, fl) for so, fl in ((SO, []), (HOOKS_SO, HOOKS_FLAGS)) if (so == SO or hooks) and (force or _stale(so))]
    if not todo:
        return SO
    with tempfile.TemporaryDirectory() as tmp:
        started = [(so,) + _compile_units(tmp, "h_" if fl else "p_", fl, verbose) for so, fl in todo]
        _finish([pr for _, _, procs in started for pr in procs])
        for so, objs, _ in started:
            _link(so, objs, verbose)
    return SO


def source_hash():
    """identifies the kernel sources a committed measurement (profiles/*pmc_traffic*, *l2_dcmip11*) was taken with"""
    import hashlib
    h = hashlib.sha256()
    for f in SRC:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_libs = {}

# every symbol include/transport_se_hip.h declares
SYMBOLS = ["tse_init", "tse_finalize", "tse_last_error", "tse_synchronize", "tse_copy_qdp_h2d", "tse_copy_qdp_d2h",
           "tse_set_derived", "tse_set_divdp", "tse_get_derived", "tse_advec_tracers_remap_rk2", "tse_compute_divdp", "tse_euler_step",
           "tse_qdp_time_avg", "tse_vertical_remap", "tse_get_qminmax", "tse_dcmip_init", "tse_dcmip_set_initial",
           "tse_dcmip_step_inputs", "tse_prim_run_subcycle", "tse_device_ptr", "tse_kernel_time", "tse_timing",
           "tse_halo_layout", "tse_halo_minmax_layout", "tse_comm_unique_id", "tse_comm_init", "tse_comm_precheck", "tse_comm_version", "tse_comm_info", "tse_comm_abort",
           "tse_boundary_layout", "tse_patch_layout", "tse_placement", "tse_invalidate_cache", "tse_divergence_sphere", "tse_laplace_sphere_wk", "tse_remap_q_ppm", "tse_host_register", "tse_element_mass", "tse_element_qdiag"]
COMM_ID_BYTES = 128


def lib(path=None):
    """Load the HIP library; raises (never falls back) if it has not been built.  path (or TSE_LIB): another build of the SAME
    sources -- the -DTSE_AB_HOOKS twin (HOOKS_SO: fault injection for the tests, A/B switches) or a tools/ab variant -- never a fallback."""
    path = path or os.environ.get("TSE_LIB", SO)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the product path)" % os.path.basename(path))
    # One HIP runtime and one RCCL per process.  PyTorch (the control plane of the multi-rank driver, and what the benchmark
    # contract synchronises with) loads the ROCm libraries bundled in its wheel by absolute path; a libamdhip64 / librccl that is
    # already in the process under the same SONAME does not stop it, so "library first, torch later" ends with TWO runtimes
    # (device pointers of one unknown to the other: hipErrorNoDevice on the first copy).  The other order is safe: this
    # library's DT_NEEDED entries (libamdhip64.so.7, librccl.so.1) resolve to the copies torch has loaded.  So torch goes first
    # whenever it is installed; tse_comm_version() reports which RCCL that is (bench.py prints it).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    L.tse_init.argtypes = [C.POINTER(vp), C.POINTER(InitArgs)]
    L.tse_finalize.argtypes = [vp]; L.tse_finalize.restype = None
    L.tse_last_error.restype = C.c_char_p
    L.tse_synchronize.argtypes = [vp]
    L.tse_copy_qdp_h2d.argtypes = [vp, vp, sz, i, i]
    L.tse_copy_qdp_d2h.argtypes = [vp, vp, sz, i, i]
    L.tse_set_derived.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz]
    L.tse_set_divdp.argtypes = [vp, vp, sz, vp, sz]
    L.tse_get_derived.argtypes = [vp] + [vp, sz] * 6
    L.tse_advec_tracers_remap_rk2.argtypes = [vp, d, i, i]
    L.tse_compute_divdp.argtypes = [vp]
    L.tse_euler_step.argtypes = [vp, i, i, d, i, i]
    L.tse_qdp_time_avg.argtypes = [vp, i, i, i]
    L.tse_vertical_remap.argtypes = [vp, d, i]
    L.tse_get_qminmax.argtypes = [vp, vp, vp]
    L.tse_dcmip_init.argtypes = [vp, i, vp, vp, vp, vp]
    L.tse_dcmip_set_initial.argtypes = [vp]
    L.tse_dcmip_step_inputs.argtypes = [vp, i, d]
    L.tse_prim_run_subcycle.argtypes = [vp, d, i, C.POINTER(i)]
    L.tse_device_ptr.argtypes = [vp, C.c_char_p, C.POINTER(sz)]; L.tse_device_ptr.restype = vp
    L.tse_kernel_time.argtypes = [vp, C.c_char_p, C.POINTER(d), C.POINTER(C.c_long)]
    L.tse_timing.argtypes = [vp, i]
    L.tse_halo_layout.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.tse_halo_minmax_layout.argtypes = [vp, vp, vp]
    L.tse_comm_unique_id.argtypes = [vp]
    L.tse_comm_init.argtypes = [vp, vp, i, i]
    L.tse_comm_precheck.argtypes = [vp, i, i]
    L.tse_comm_version.argtypes = [C.POINTER(i), C.POINTER(i), C.c_char_p, sz]
    L.tse_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.tse_comm_abort.argtypes = [vp]
    L.tse_boundary_layout.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.tse_patch_layout.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.tse_placement.argtypes = [vp, C.POINTER(i), C.POINTER(d), C.POINTER(i)]
    L.tse_invalidate_cache.argtypes = [vp]
    L.tse_divergence_sphere.argtypes = [vp, vp, vp]
    L.tse_laplace_sphere_wk.argtypes = [vp, vp, vp]
    L.tse_remap_q_ppm.argtypes = [vp, vp, vp, vp]
    L.tse_host_register.argtypes = [vp, vp, sz]
    L.tse_element_mass.argtypes = [vp, i, vp]
    L.tse_element_qdiag.argtypes = [vp, i, vp, vp, vp, vp]
    _libs[path] = L
    return L
