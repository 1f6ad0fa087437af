"""-m gpu: BASELINE configs[4] (ne120, qsize=200 on 8 GPUs) at its PER-GPU shape on the one GPU of the box: 10 800 elements x 200
tracers, 20 GB per tracer field, about 100 GB of device memory.

* a complete sphere of (almost) that size -- ne42 = 10 584 elements -- with 200 tracers, checked through the size-independent
  properties of the path (as tests/test_gpu_baseline_configs.py does for ne120/q35): consistent advection (Q == 1), limiter
  bounds on 0/1 fields, tracer mass through a full rsplit cycle incl. hyperviscosity, DSS and PPM remap, and DSS-on-read
  against one DSS pass per stage;
* rank 3 of the reference's 8-rank space-filling-curve partition of ne120 with 200 tracers, alone, every neighbour slot in RCCL
  loopback (tools/rank_rehearsal.py): real split launches, pack / unpack and message sizes; finite results, the two DSS routes
  agree, timing table and device memory in use -> gpurun_out/q200_shape.json (committed as profiles/rNN_q200_shape.json);
* the fallback of the whole-step call when a scratch plane would not fit 32-bit offsets (tse_api.hip: advec_step), reached with the
  test-only TSE_TEST_PLANE_LIMIT.
Tolerance: TOL_STEP = 5e-13 of the field maximum per tracer step (tests/test_gpu_parity.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_STEP = 5e-13
OUT = os.path.join(ROOT, "gpurun_out", "q200_shape.json")


def _record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    d = json.load(open(OUT)) if os.path.exists(OUT) else {}
    d[key] = value
    json.dump(d, open(OUT, "w"), indent=1)


def _dev_tensor(torch, ptr, shape, dtype="<f8"):
    iface = {"shape": tuple(int(x) for x in shape), "typestr": dtype, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(type("DevArr", (), {"__cuda_array_interface__": iface})(), device="cuda:0")


def test_ne42_q200_full_sphere_properties(monkeypatch):
    import torch
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.driver import PrimRun
    ne, q = 42, 200
    run = PrimRun(ne, q, test_case=1)
    hip, n = run.hip, run.nelem
    dt = run.tstep
    Q = [_dev_tensor(torch, hip.device_ptr("qdp%d" % tl)[0], (n, q, 72, 16)) for tl in (1, 2)]   # the two time levels: two allocations
    sph = torch.as_tensor(run.elem["spheremp"].reshape(n, 16), device="cuda:0")
    dp = _dev_tensor(torch, hip.device_ptr("dp")[0], (n, 72, 16))

    def mass(tl):
        return torch.einsum("ep,eqkp->q", sph, Q[tl - 1]).cpu().numpy()

    topo = cm.topology(ne)
    chk = (np.sin(9 * run.lon) * np.sin(9 * run.lat) >= 0).astype(np.float64).reshape(n, 16)
    cons = (cm.dss_sum(chk, topo) / cm.dss_sum(np.ones_like(chk), topo) >= 0.5).astype(np.float64)   # single-valued at shared nodes
    hip.dcmip_step_inputs(0, dt); hip.synchronize()
    for Qt in Q:
        Qt[:, 4:q - 1] = (torch.as_tensor(cons, device="cuda:0")[:, None, :] * dp).unsqueeze(1)
        Qt[:, q - 1] = dp
    torch.cuda.synchronize(); hip.invalidate_cache()
    m0 = mass(1)
    saved0 = Q[0].clone()

    hip.advec_tracers_remap_rk2(dt, 1, 2); hip.synchronize()
    dvp = _dev_tensor(torch, hip.device_ptr("divdp_proj")[0], (n, 72, 16))
    dp_new = dp - dt * dvp
    one = Q[1][:, q - 1] / dp_new
    assert float((one - 1).abs().max()) <= 1e-12, float((one - 1).abs().max())          # consistent advection
    c01 = Q[1][:, 4:q - 1] / dp_new.unsqueeze(1)
    assert float(c01.min()) >= -1e-12 and float(c01.max()) <= 1 + 1e-12, (float(c01.min()), float(c01.max()))   # limiter bounds
    del one, c01
    np.testing.assert_allclose(mass(2), m0, rtol=1e-12)
    step1 = Q[1].clone()

    Q[0].copy_(saved0); Q[1].copy_(saved0)
    torch.cuda.synchronize(); hip.invalidate_cache()
    hip.dcmip_step_inputs(0, dt)
    monkeypatch.setenv("TSE_DSS_ON_READ", "0")
    hip.advec_tracers_remap_rk2(dt, 1, 2); hip.synchronize()
    monkeypatch.delenv("TSE_DSS_ON_READ")
    diff, scale = float((Q[1] - step1).abs().max()), float(step1.abs().max())
    assert diff <= TOL_STEP * scale, (diff, scale)
    free, total = torch.cuda.mem_get_info(0)
    high = (total - free) / 1e9
    del step1, saved0
    torch.cuda.empty_cache()

    run.nstep = 1
    np1 = run.run(2)                                   # two more steps + the remap
    hip.synchronize()
    np.testing.assert_allclose(mass(np1), m0, rtol=1e-12)
    # timing of one more cycle with the event timers on (what a rank of configs[4] costs per step, links apart)
    hip.timing(True)
    import time
    t0 = time.perf_counter()
    run.run(3); hip.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    kt = {k: round(hip.kernel_time(k)[0] / 3, 3) for k in ("advance0", "advance1", "advance2", "lap", "dss", "minmax", "remap", "dcmip")}
    _record("ne42_q200_full_sphere", dict(elements=n, qsize=q, field_GB=n * q * 72 * 16 * 8 / 1e9, device_memory_high_water_GB=round(high, 1),
                                          ms_per_step=round(ms, 2), tracer_dof_steps_per_s=n * 16 * 72 * q / (ms / 1e3), kernel_ms_per_step=kt))
    run.close()


def test_rank_of_the_8_rank_ne120_partition_with_200_tracers_in_loopback():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_rehearsal.py"), "--ne", "120", "--qsize", "200", "--world", "8", "--rank", "3",
                        "--cycles", "1", "--verify"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["elements"] == 10800 and d["qsize"] == 200 and d["neighbour_ranks"] >= 2 and d["boundary_elements"] > 0
    assert d["finite"] and d["dss_on_read_vs_per_stage_rel"] <= TOL_STEP, d
    assert d["dss_passes_of_the_two_routes"][0] < d["dss_passes_of_the_two_routes"][1], d      # the two runs did take different routes
    assert 60 < d["device_memory_used_GB"] < 288, d
    _record("ne120_q200_rank3_of_8_loopback", d)


@pytest.mark.parametrize("ne,q", [(4, 3)])
def test_whole_step_falls_back_when_a_plane_exceeds_32bit_offsets(monkeypatch, capfd, ne, q):
    """DSS on read addresses a scratch plane with 32-bit byte offsets; a plane of 4 GiB or more (466 000 elements on one rank) makes
    the whole-step call take one DSS pass per stage instead.  TSE_TEST_PLANE_LIMIT lowers the limit so that the branch runs here:
    same bits as TSE_DSS_ON_READ=0, the step tolerance against the default route, and the library says what it did."""
    import pyoracle as po
    from gpu_common import elem_from_oracle, make_hip, relerr
    from transport_se_amd import _lib
    o = po.Oracle(ne, q, nu_q=5e17)
    elem = elem_from_oracle(o)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        # (TSE_TEST_* exist only in the -DTSE_AB_HOOKS twin of the library: same sources, same kernels)
        hip = make_hip(o, elem, lib_path=_lib.HOOKS_SO)
        hip.dcmip_init(1, o.lat, o.lon, o.hyam, o.hybm); hip.dcmip_set_initial()
        assert hip.prim_run_subcycle(900.0, 1, 0) == 3
        out = hip.fetch("qdp", (2, o.nelem, q, 72, 4, 4)).copy()
        hip.close()
        for k in env:
            monkeypatch.delenv(k)
        return out
    base = run({})
    capfd.readouterr()
    lim = run({"TSE_TEST_PLANE_LIMIT": "1"})
    assert "DSS on read disabled" in capfd.readouterr().err
    off = run({"TSE_DSS_ON_READ": "0"})
    assert np.array_equal(lim, off)
    assert relerr(lim, base) <= 3 * TOL_STEP
    o.close()
