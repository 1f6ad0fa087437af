"""-m gpu: the drop-in boundary exercised from the reference's own language.

tests/fortran_dropin/_build/hip_harness (tests/fortran_dropin/Makefile) is the reference's UNMODIFIED prim_advection_mod.F90 compiled with
-DUSE_CUDA_FORTRAN=1 and linked against transport_se_amd/fortran/cuda_mod_hip.F90 (a Fortran module named `cuda_mod`
bound to libtransport_se_hip.so with ISO_C_BINDING): the reference's own `call euler_step_cuda(...)`,
`call qdp_time_avg_cuda(...)`, `call vertical_remap_cuda(...)` hooks run on the MI355X.  Its output must equal the
plain-Fortran reference's (tests/golden/ref_ne2_dcmip11.npz) to the step tolerance.  Built in the build container
(needs /root/reference + amdflang); the binary travels to the GPU box like the .so."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "fortran_dropin", "_build", "hip_harness")
MPIEXEC = "/opt/conda/bin/mpiexec"
from conftest import record_margin  # noqa: E402
# Tolerance against the reference's own output: TOL_STEP = 5e-13 of the field maximum per tracer step (test_gpu_parity.py: reordered
# 16-point sums, FMA contraction, reciprocal multiplies) times the number of steps the error has accumulated over, + one step's worth
# for the remap (a relative 1e-16 per level through the folded PPM coefficients).  gpurun_out/test_margins.jsonl records what each run
# measured (1e-14 ... 2e-13).
TOL_STEP = 5e-13
TOL_DROPIN = {"qdp_step1": 1 * TOL_STEP, "qdp_step3": 4 * TOL_STEP, "qdp_step6": 8 * TOL_STEP}


def _env(whole_step):
    """whole_step: the harness calls cuda_mod_hip's one-call entry advec_tracers_remap_rk2_hip (the hook a maintainer adds to
    Prim_Advec_Tracers_remap_rk2) instead of going through the reference's per-stage hooks"""
    return dict(os.environ, TSE_HARNESS_WHOLE_STEP="1" if whole_step else "0")


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(MPIEXEC)), reason="Fortran drop-in harness not built")
@pytest.mark.parametrize("whole_step", [False, True])
def test_reference_hooks_drive_the_hip_library(gold, whole_step):
    g = gold("ref_ne2_dcmip11.npz")
    cfg = json.loads(str(g["config"]))
    out = tempfile.mkdtemp(prefix="tse_f90_")
    stdin = "%d %d %d %r %r %d 1\n'%s'\n'%s'\n" % (cfg["ne"], cfg["qsize"], cfg["nsteps"], cfg["tstep"], cfg["nu_q"], cfg["test"],
                                                 out, os.path.join(ROOT, "transport_se_amd", "data", "vcoord"))
    res = subprocess.run([MPIEXEC, "-n", "1", HARNESS], input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                         env=_env(whole_step))
    log = res.stdout.decode()
    assert "ref_harness done" in log, log[-2000:]
    for tag, key in (("000001", "qdp_step1"), ("000003", "qdp_step3"), ("000006", "qdp_step6")):
        st = po.read_state(os.path.join(out, "state_%s_r0000.bin" % tag))
        err = np.abs(st["qdp"] - g[key]).max() / np.abs(g[key]).max()
        record_margin("fortran_dropin[%s] %s" % ("whole_step" if whole_step else "per_stage", key), err, TOL_DROPIN[key])
        assert err < TOL_DROPIN[key], (tag, err)
    s3 = po.read_state(os.path.join(out, "state_000003_r0000.bin"))
    assert np.abs(s3["dp3d"] - g["dp3d_step3"]).max() / np.abs(g["dp3d_step3"]).max() < 1e-13
    assert np.abs(s3["ps_v"] - g["ps_v_step3"]).max() / 1e5 < 1e-13


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(MPIEXEC)), reason="Fortran drop-in harness not built")
@pytest.mark.parametrize("nranks,whole_step", [(2, False), (3, False), (2, True)])
def test_reference_hooks_multirank_mpi_exchange(gold, nranks, whole_step):
    """same drop-in on 2 and 3 MPI ranks (all sharing the one GPU): the reference's own genEdgeSched schedule
    (SendCycle/RecvCycle slots) feeds tse_init, and the bndry_exchangeV replacement is the Fortran MPI callback of
    cuda_mod_hip.F90 (host-staged, MPICH is not GPU-aware).  Output must match the single-rank plain-Fortran reference."""
    g = gold("ref_ne2_dcmip11.npz")
    cfg = json.loads(str(g["config"]))
    out = tempfile.mkdtemp(prefix="tse_f90mr_")
    stdin = "%d %d %d %r %r %d 0\n'%s'\n'%s'\n" % (cfg["ne"], cfg["qsize"], cfg["nsteps"], cfg["tstep"], cfg["nu_q"], cfg["test"],
                                                 out, os.path.join(ROOT, "transport_se_amd", "data", "vcoord"))
    res = subprocess.run([MPIEXEC, "-n", str(nranks), HARNESS], input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                         env=_env(whole_step))
    log = res.stdout.decode()
    assert "ref_harness done" in log, log[-3000:]
    q = np.empty_like(g["qdp_step6"])
    seen = 0
    for r in range(nranks):
        st = po.read_static(os.path.join(out, "static_000000_r%04d.bin" % r))
        sd = po.read_state(os.path.join(out, "state_000006_r%04d.bin" % r))
        q[st["gid"] - 1] = sd["qdp"]; seen += st["gid"].size
    assert seen == q.shape[0]
    err = np.abs(q - g["qdp_step6"]).max() / np.abs(g["qdp_step6"]).max()
    record_margin("fortran_dropin[%d ranks, %s] qdp_step6" % (nranks, "whole_step" if whole_step else "per_stage"), err, TOL_DROPIN["qdp_step6"])
    assert err < TOL_DROPIN["qdp_step6"], err


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(MPIEXEC)), reason="Fortran drop-in harness not built")
def test_rccl_mode_falls_back_to_the_mpi_exchange_when_the_communicator_cannot_be_built(gold):
    """TSE_EXCHANGE=rccl on 2 MPI ranks that share the one GPU: ncclCommInitRank refuses the duplicate device on both ranks;
    cuda_mod_init must agree on that over MPI, drop the communicators and carry on with the MPI body of bndry_exchangeV --
    same output as the plain run -- instead of aborting or hanging."""
    g = gold("ref_ne2_dcmip11.npz")
    cfg = json.loads(str(g["config"]))
    out = tempfile.mkdtemp(prefix="tse_f90fb_")
    stdin = "%d %d %d %r %r %d 0\n'%s'\n'%s'\n" % (cfg["ne"], cfg["qsize"], cfg["nsteps"], cfg["tstep"], cfg["nu_q"], cfg["test"],
                                                 out, os.path.join(ROOT, "transport_se_amd", "data", "vcoord"))
    res = subprocess.run([MPIEXEC, "-n", "2", HARNESS], input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300,
                         env=dict(_env(True), TSE_EXCHANGE="rccl"))
    log = res.stdout.decode()
    assert "RCCL communicator could not be initialised" in log, log[-3000:]
    assert "ref_harness done" in log, log[-3000:]
    q = np.empty_like(g["qdp_step6"])
    for r in range(2):
        st = po.read_static(os.path.join(out, "static_000000_r%04d.bin" % r))
        q[st["gid"] - 1] = po.read_state(os.path.join(out, "state_000006_r%04d.bin" % r))["qdp"]
    assert np.abs(q - g["qdp_step6"]).max() / np.abs(g["qdp_step6"]).max() < TOL_DROPIN["qdp_step6"]
