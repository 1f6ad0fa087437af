"""-m gpu: vert_remap_q_alg = 2 (control_mod.F90:61-66) -- the PPM remap without mirrored ghost cells: the two cells at
either end of a column are piecewise constant (prim_advection_mod.F90:230-250,283-341).  Checked against the reference's own
outputs with that namelist value (tests/golden/ref_ne2_alg2.npz: oracle/ref/make_golden.py --alg2 = the reference harness run with
TSE_NL_VERT_REMAP_Q_ALG=2), through the single operator call, the device-resident time loop, every column-loop variant of
k_remap, and the Fortran seam.  Tolerance: 5e-13 of the field maximum per call / tracer step, as for the default algorithm."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import pyoracle as po
from gpu_common import elem_from_oracle, relerr
from transport_se_amd import HipMod

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "fortran_dropin", "_build", "hip_harness")
MPIEXEC = "/opt/conda/bin/mpiexec"


def _hip(o, elem, alg):
    return HipMod(elem, o.Dvv, (o.hyai, o.hybi, 1.0e5), o.qsize, o.nu_q, rsplit=o.rsplit, vert_remap_q_alg=alg)


@pytest.mark.parametrize("generic", ["0", "1"])
def test_remap_q_ppm_alg2_single_call_vs_reference(gold, monkeypatch, generic):
    g = gold("ref_ne2_alg2.npz")
    o = po.Oracle(2, int(g["remap_Qin"].shape[1]), nu_q=1e19)
    elem = elem_from_oracle(o)
    monkeypatch.setenv("TSE_REMAP_GENERIC", generic)      # the lockstep and the generic column loop
    hip = _hip(o, elem, 2)
    which = np.arange(o.nelem) % 2
    q = g["remap_Qin"][which]; dp1 = g["remap_dp1"][which]; dp2 = g["remap_dp2"][which]
    out = hip.remap_q_ppm(q, dp1, dp2)
    ref = g["remap_Qout"][which]
    from conftest import record_margin
    record_margin("remap_q_ppm alg 2 single call vs reference (generic=%s)" % generic, relerr(out, ref), 1e-13)
    assert relerr(out, ref) <= 1e-13, relerr(out, ref)   # measured 1.8e-15
    np.testing.assert_allclose(out.sum(2), q.sum(2), rtol=1e-13)    # column mass
    # and it IS a different algorithm: the default context does not reproduce these outputs
    hip0 = _hip(o, elem, 0)
    assert relerr(hip0.remap_q_ppm(q, dp1, dp2), ref) > 1e-6
    hip.close(); hip0.close(); o.close()


def test_six_steps_two_remaps_alg2_vs_reference(gold):
    g = gold("ref_ne2_alg2.npz")
    cfg = json.loads(str(g["config"]))
    assert cfg["vert_remap_q_alg"] == 2
    o = po.Oracle(cfg["ne"], cfg["qsize"], nu_q=cfg["nu_q"])
    elem = elem_from_oracle(o)
    hip = _hip(o, elem, 2)
    hip.dcmip_init(1, o.lat, o.lon, o.hyam, o.hybm)
    hip.dcmip_set_initial()
    assert hip.prim_run_subcycle(cfg["tstep"], 2, 0) == 6
    q = hip.fetch("qdp", (2, o.nelem, cfg["qsize"], 72, 4, 4))[0]
    assert relerr(q, g["qdp_step6"]) < 10 * 5e-13, relerr(q, g["qdp_step6"])
    assert relerr(hip.fetch("dp3d", (o.nelem, 72, 4, 4)), g["dp3d_step6"]) < 1e-13
    # against the checker with the same switch
    po.set_vert_remap_q_alg(2)
    try:
        o.dcmip_init(1)
        done, _ = o.prim_run(1, cfg["tstep"], 2)
        assert done == 6 and np.array_equal(o.qdp[0], g["qdp_step6"])      # (the oracle is bit-exact with the reference here too)
    finally:
        po.set_vert_remap_q_alg(0)
    hip.close(); o.close()


def test_unknown_remap_algorithm_is_refused():
    from transport_se_amd.hip_mod import TseError
    o = po.Oracle(2, 1)
    elem = elem_from_oracle(o)
    with pytest.raises(TseError, match="vert_remap_q_alg=3"):
        _hip(o, elem, 3)
    o.close()


def _harness(env, nranks=1, whole_step="0"):
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_ne2_alg2.npz"), allow_pickle=False)
    cfg = json.loads(str(g["config"]))
    out = tempfile.mkdtemp(prefix="tse_f90a2_")
    stdin = "%d %d %d %r %r %d 0\n'%s'\n'%s'\n" % (cfg["ne"], cfg["qsize"], cfg["nsteps"], cfg["tstep"], cfg["nu_q"], cfg["test"],
                                                 out, os.path.join(ROOT, "transport_se_amd", "data", "vcoord"))
    res = subprocess.run([MPIEXEC, "-n", str(nranks), HARNESS], input=stdin.encode(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=300, env=dict(os.environ, TSE_HARNESS_WHOLE_STEP=whole_step, **env))
    return res, out, g


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(MPIEXEC)), reason="Fortran drop-in harness not built")
def test_fortran_seam_passes_vert_remap_q_alg(gold):
    """the reference's own vertical_remap hook with control_mod's vert_remap_q_alg = 2: cuda_mod_init hands the value to tse_init"""
    res, out, g = _harness({"TSE_NL_VERT_REMAP_Q_ALG": "2"})
    log = res.stdout.decode()
    assert "ref_harness done" in log, log[-2000:]
    st = po.read_state(os.path.join(out, "state_000006_r0000.bin"))
    assert relerr(st["qdp"], g["qdp_step6"]) < 5e-12
    # ... and with the default value the same run does NOT give the alg-2 result (the switch is live, not ignored)
    res0, out0, _ = _harness({})
    st0 = po.read_state(os.path.join(out0, "state_000006_r0000.bin"))
    assert relerr(st0["qdp"], g["qdp_step6"]) > 1e-6


@pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(MPIEXEC)), reason="Fortran drop-in harness not built")
@pytest.mark.parametrize("env,text", [({"TSE_NL_QSPLIT": "2"}, "qsplit must be 1"),
                                      ({"TSE_NL_HYPERVIS_SUBCYCLE_Q": "2"}, "hypervis_subcycle_q must be 1"),
                                      ({"TSE_NL_HYPERVIS_POWER": "3.2"}, "hypervis_power and hypervis_scaling must be 0"),
                                      ({"TSE_NL_HYPERVIS_SCALING": "3.2"}, "hypervis_power and hypervis_scaling must be 0"),
                                      ({"TSE_NL_VERT_REMAP_Q_ALG": "7"}, "vert_remap_q_alg must be 0, 1 or 2")])
def test_fortran_seam_aborts_on_what_the_device_path_does_not_implement(env, text):
    """cuda_mod_init guards every control_mod switch the device path would otherwise silently ignore: abortmp with a message"""
    res, _, _ = _harness(env)
    log = res.stdout.decode()
    assert res.returncode != 0 and "ref_harness done" not in log
    assert text in log, log[-1500:]
