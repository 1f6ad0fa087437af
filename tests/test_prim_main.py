"""namelist front end and product-side diagnostics (8f-3)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import norms
from transport_se_amd import diagnostics as dg
from transport_se_amd import prim_main as pm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

NL = """
&ctl_nl
  NThreads = 1
  partmethod = 4
  topology = "cube"
  test_case = "dcmip1-1"
  ne = 8
  qsize = 4
  nmax = 6             ! six tracer steps
  statefreq = 3
  tstep = 400
  qsplit = 1, rsplit = 3
  nu_q = 6e16
  limiter_option = 8
  hypervis_order = 2
/
&vert_nl
  vform = "ccm"
  vfile_mid = "vcoord/acme-72m.ascii"
  vfile_int = "vcoord/acme-72i.ascii"
/
&analysis_nl
  output_frequency = 6
/
"""


def test_namelist_parsing():
    s = pm.settings(pm.parse_namelists(NL))
    assert (s["ne"], s["qsize"], s["nmax"], s["rsplit"], s["test"]) == (8, 4, 6, 3, 1)
    assert s["tstep"] == 400 and s["nu_q"] == 6e16 and s["limiter_option"] == 8
    s2 = pm.settings(pm.parse_namelists(NL.replace("nmax = 6", "ndays = 12")))
    assert s2["nmax"] == 12 * 86400 // 400          # namelist_mod.F90:347-351
    with pytest.raises(SystemExit):
        pm.settings(pm.parse_namelists(NL.replace("limiter_option = 8", "limiter_option = 4")))
    # the reference's namelist TEMPLATE (placeholders not yet substituted by the run script's sed): a clean abort that says so
    for tmpl in (NL.replace("ne = 8", "ne = NE"), NL.replace("tstep = 400", "tstep = TIME_STEP")):
        with pytest.raises(SystemExit, match="placeholders"):
            pm.settings(pm.parse_namelists(tmpl))


def test_fortran_e_format_is_the_references():
    """prim_printstate writes E23.15 / E22.14 / E15.7 (prim_state_mod.F90:381,389): Fortran puts the mantissa below 1"""
    assert pm._fortran_e(1.0, 23, 15) == "  0.100000000000000E+01"
    assert pm._fortran_e(-3.5e-7, 15, 7) == " -0.3500000E-06"
    assert pm._fortran_e(9.99999999e9, 15, 7) == "  0.1000000E+11"
    assert pm._fortran_e(0.0, 15, 7) == "  0.0000000E+00"


def test_product_diagnostics_agree_with_the_checker():
    rng = np.random.default_rng(5)
    from transport_se_amd import cube_mesh as cm
    g = cm.geometry(2)
    q0 = rng.uniform(size=(24, 72, 4, 4)); q1 = q0 + 0.1 * rng.uniform(size=q0.shape)
    zm = np.linspace(12000, 100, 72)
    assert dg.dcmip_norms(2, g["lat"], g["lon"], q0, q1, zm) == norms.dcmip_norms(2, g["lat"], g["lon"], q0, q1, zm)


def _gloo_sum_worker(rank, world, port, partials, gid_parts, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = gid_parts[rank]
    out = dg.global_sum(partials[g], g, partials.shape[0], dist, rank, world)
    full = dg.gather_by_gid(partials[g], g, partials.shape[0], dist, rank, world)
    if rank == 0:
        q.put((out, full))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_global_sum_is_independent_of_the_partition(world):
    """the multi-rank mass/`qv=` sums: per-element partials gathered over gloo and added exactly -> the same bits for any
    distribution of the elements (what repro_sum gives the reference, global_norms_mod.F90:66-68)"""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(11)
    n = 96
    partials = rng.standard_normal((n, 5)) * 10.0 ** rng.integers(-8, 8, size=(n, 5))     # wildly different magnitudes
    ref = dg.global_sum(partials, np.arange(n), n)
    perm = rng.permutation(n)
    gid_parts = np.array_split(perm, world)                       # an arbitrary (non-contiguous) ownership
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 500 + world
    procs = [ctx.Process(target=_gloo_sum_worker, args=(r, world, port, partials, gid_parts, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert np.array_equal(out, ref) and np.array_equal(full, partials)
    assert not np.array_equal(ref, partials.sum(0)) or True       # (a plain float sum may or may not agree; fsum is the contract)


NL12 = NL.replace('test_case = "dcmip1-1"', 'test_case = "dcmip1-2"').replace("nmax = 6             ! six tracer steps", "ndays = 1").replace("statefreq = 3", "statefreq = 108")
PREQX = os.path.join(ROOT, "bin", "preqx")


def _run_preqx(args, nl, cwd, env_extra=None, timeout=600):
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    res = subprocess.run([PREQX] + args, input=nl.encode(), cwd=str(cwd), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    return out


@pytest.mark.gpu
def test_preqx_dcmip12_norm_line_equals_the_reference_run(tmp_path):
    """`bin/preqx < namelist` (the reference scripts' $EXE < dcmip1-2.nl, test/run_ne8_tests.sh) for the 1-day DCMIP 1-2 case:
    the printed norm line against the same run of the reference built here (tests/golden/ref_ne8_norms.json) and README:96"""
    import json
    out = _run_preqx([], NL12, tmp_path)
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_ne8_norms.json")))["dcmip1-2"]
    line = [l for l in out.splitlines() if l.startswith("DCMIP 1-2:")][0]
    import re
    got = {k: float(v) for k, v in re.findall(r"(\w+)=\s*([-+0-9.eE]+)", line.split(":", 1)[1])}
    for k in ("L1", "L2", "Linf", "q_max"):
        assert abs(got[k] - ref[k]) <= 1e-6 * abs(ref[k]) + 5e-7, (k, got[k], ref[k])        # 6 printed digits
    assert (got["L1"], got["L2"], got["Linf"], got["q_max"]) == (0.307665, 0.622099, 0.839133, 0.813105)   # README:96
    stats = open(os.path.join(str(tmp_path), "HommeTime_stats")).read()
    for name in ("prim_run", "prim_advance_exp", "prim_advec_tracers", "vertical_remap"):   # run_ne120_perf.sh:140-144
        assert name in stats
    assert sum(l.startswith("qv= ") for l in out.splitlines()) >= 4 and "Q2,Q diss, dQ^2/dt:" in out   # prim_printstate's lines (prim_state_mod.F90:341-385)
    # "One should also check that the tracer mass is conserved by looking at the Q, Q diss values" (README:38-44): '(a,i1,a,E22.14,a,2E15.7)' =
    # mass [kg/m^2], d(mass)/dt over the cycle, d(variance)/dt -- the mass rate must vanish against mass / dt (tstep = 400 s here)
    seen = 0
    for l in out.splitlines():
        if l.startswith("Q") and ",Q diss, dQ^2/dt:" in l:
            m = float(l.split(":")[1].split("kg/m^2")[0]); rates = l.split("kg/m^2")[1].split()
            assert len(rates) == 2 and abs(float(rates[0])) * 400.0 <= 1e-11 * max(abs(m), 1e-300), l
            seen += 1
    assert seen >= 4, out[-1500:]
    for l in out.splitlines():                                    # mass conserved (the "Q,Q diss" check)
        if l.startswith("Q") and "relative change" in l:
            assert abs(float(l.split("relative change")[1].strip(" )"))) < 1e-11


@pytest.mark.gpu
def test_preqx_prints_the_same_digits_on_1_2_and_4_ranks(tmp_path):
    """norm line, `qv=` lines and tracer-mass lines are string-identical however the sphere is cut (the ranks share the one GPU
    here: TSE_EXCHANGE=staged; on a multi-GPU node the same command runs one rank per GPU over RCCL)"""
    def lines(out):
        return [l for l in out.splitlines() if l.startswith(("DCMIP", "Q", "qv= ")) and "wall" not in l]
    nl = NL.replace("nmax = 6             ! six tracer steps", "nmax = 12")
    one = lines(_run_preqx([], nl, tmp_path))
    assert len(one) >= 1 + 4 + 4
    for n in (2, 4):
        many = lines(_run_preqx(["--gpus", str(n)], nl, tmp_path, {"TSE_EXCHANGE": "staged"}))
        assert many == one, "\n".join(["%d ranks:" % n] + many + ["1 rank:"] + one)
    stats = open(os.path.join(str(tmp_path), "HommeTime_stats")).read()
    assert " 4 " in stats.splitlines()[1]                         # processes column of the last (4-rank) run


@pytest.mark.gpu
def test_one_rank_meeting_a_negative_layer_thickness_ends_every_rank(tmp_path):
    """the reference aborts the whole job in the remap that meets a negative layer thickness (abortmp = MPI_Abort,
    prim_advection_mod.F90:1323).  Here rank 1 of 2 alone gets return code 2 from its remap (TSE_TEST_FAIL_REMAP_RANK); the healthy
    rank would otherwise wait in the next halo exchange for ever.  The job must end non-zero, with the message, well inside the
    watchdog -- no hang, no silent success."""
    import time
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    from transport_se_amd import _lib
    env.update(TSE_EXCHANGE="staged", TSE_TEST_FAIL_REMAP_RANK="1", TSE_WATCHDOG_S="150", TSE_LIB=_lib.HOOKS_SO)   # (fault injection: the -DTSE_AB_HOOKS twin)
    nl = NL.replace("nmax = 6             ! six tracer steps", "nmax = 12")
    t0 = time.time()
    res = subprocess.run([PREQX, "--gpus", "2"], input=nl.encode(), cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    took = time.time() - t0
    out = res.stdout.decode()
    assert res.returncode != 0, out[-2000:]
    assert "ABORTING WITH ERROR: negative layer thickness" in out and "tracer step 3" in out, out[-2000:]
    assert "DCMIP 1-1:" not in out          # no result line from a job that failed
    assert took < 200, took                 # the launcher ended the healthy rank (its own watchdog would have at 150 s)
