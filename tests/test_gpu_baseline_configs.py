"""-m gpu: the BASELINE.json configurations themselves.

* configs[1] (ne30, 72L, qsize=35, one GPU): three tracer steps + vertical remap of DCMIP 1-1 against the C oracle (OpenMP on
  the box's host cores) on the same inputs -- every tracer, every point, TOL_STEP per step.
* configs[2] (ne120, 72L, qsize=35, one GPU; the headline): the mesh is too large for a CPU oracle run in a test, so
  size-independent properties of the path: tracer mass conserved (euler_step conserves sum(spheremp*Qdp); the
  hyperviscosity and the limiter are mass-neutral; remap_Q_ppm conserves column mass), "consistent advection" (a tracer
  with Q == 1 leaves one tracer step as Qdp == dp - dt*divdp_proj, prim_advection_mod.F90:23-33), limiter-8 bounds
  (the 0/1 checkerboard tracers stay inside [0,1]), and the two implementations of the step (DSS on read / one DSS pass per
  stage) agreeing to TOL_STEP.
  Note on the inputs: the reference's 0/1 checkerboard tracers (dcmip_wrapper_mod.F90:215-243: sin(9 lon) sin(9 lat) >= 0,
  evaluated per element) are NOT continuous at ne30/ne120 -- 4120 element-local nodes on the lines lat = 0 and
  lon = k*pi/9 get different values from the elements that share them (their lat/lon differ in the last bit; none at
  ne8) -- and the initial state is never DSS'd, so the first steps do not conserve their mass (4e-5 relative in step 1 at
  ne120, a third of that in every further step, because qdp_time_avg keeps a third of Qdp(n0); the oracle does the
  same, see the ne30 test).  For the mass check the test therefore replaces them by the same checkerboard made
  single-valued (majority vote of the sharers of a node).
The 12-day README norm lines at ne30 and ne120 are in test_gpu_dcmip_norms.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL_STEP = 5e-13


def test_ne30_q35_three_steps_and_remap_vs_oracle():
    import pyoracle as po
    from gpu_common import elem_from_oracle, make_hip, relerr
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    o = po.Oracle(30, 35, nu_q=1e15, threads=threads)
    elem = elem_from_oracle(o)
    del elem["Qdp"]                                   # 3.5 GB host copy not needed: the state is generated on the device
    hip = make_hip(o, elem, device=0)
    hip.dcmip_init(1, o.lat, o.lon, o.hyam, o.hybm)
    hip.dcmip_set_initial()
    o.dcmip_init(1)
    assert hip.prim_run_subcycle(300.0, 1, 0) == 3
    done, _ = o.prim_run(1, 300.0, 1)
    assert done == 3
    got = hip.fetch("qdp", (2, o.nelem, 35, 72, 4, 4))[1]          # steps 0,1,2 -> np1 = 2,1,2; the remap acted on level 2
    ref = o.qdp[1]
    scale = np.abs(ref).reshape(o.nelem, 35, -1).max(2).max(0)       # per-tracer field maximum
    err = np.abs(got - ref).reshape(o.nelem, 35, -1).max(2).max(0) / scale
    assert err.max() <= 4 * TOL_STEP, err                            # 3 steps + remap
    assert relerr(hip.fetch("dp3d", (o.nelem, 72, 4, 4)), o.dp3d) <= 1e-14
    hip.close(); o.close()


def test_qsize_200_three_steps_and_remap_vs_oracle():
    """BASELINE configs[4] is the qsize=200 tracer stress at ne120 on 8 GPUs; its tracer count (beyond the reference's compile-time
    QSIZE_D=35) is exercised here on a mesh the oracle handles in seconds: ne8, 200 tracers, one rsplit cycle, every tracer"""
    import pyoracle as po
    from gpu_common import elem_from_oracle, make_hip
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    o = po.Oracle(8, 200, nu_q=6e16, threads=threads)
    elem = elem_from_oracle(o)
    del elem["Qdp"]
    hip = make_hip(o, elem, device=0)
    hip.dcmip_init(1, o.lat, o.lon, o.hyam, o.hybm)
    hip.dcmip_set_initial()
    o.dcmip_init(1)
    assert hip.prim_run_subcycle(400.0, 1, 0) == 3
    done, _ = o.prim_run(1, 400.0, 1)
    assert done == 3
    got = hip.fetch("qdp", (2, o.nelem, 200, 72, 4, 4))[1]
    ref = o.qdp[1]
    scale = np.abs(ref).reshape(o.nelem, 200, -1).max(2).max(0)
    err = np.abs(got - ref).reshape(o.nelem, 200, -1).max(2).max(0) / scale
    assert err.max() <= 4 * TOL_STEP, err.max()
    hip.close(); o.close()


def _dev_tensor(torch, ptr, shape, dtype="<f8"):
    iface = {"shape": tuple(int(x) for x in shape), "typestr": dtype, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(type("DevArr", (), {"__cuda_array_interface__": iface})(), device="cuda:0")


def test_ne120_q35_properties(monkeypatch):
    import torch
    from transport_se_amd.driver import PrimRun
    ne, q = 120, 35
    run = PrimRun(ne, q, test_case=1)
    hip, n = run.hip, run.nelem
    dt = run.tstep
    Q = [_dev_tensor(torch, hip.device_ptr("qdp%d" % tl)[0], (n, q, 72, 16)) for tl in (1, 2)]   # the two time levels: two allocations
    sph = torch.as_tensor(run.elem["spheremp"].reshape(n, 16), device="cuda:0")
    dp = _dev_tensor(torch, hip.device_ptr("dp")[0], (n, 72, 16))

    def mass(tl):
        return torch.einsum("ep,eqkp->q", sph, Q[tl - 1]).cpu().numpy()

    # tracers 5..34 := the single-valued 0/1 checkerboard, tracer 35 := Q == 1 (Qdp = Q*dp of the initial state, both time
    # levels); writing through the device pointer drops the bounds cache
    from transport_se_amd import cube_mesh as cm
    topo = cm.topology(ne)
    chk = (np.sin(9 * run.lon) * np.sin(9 * run.lat) >= 0).astype(np.float64).reshape(n, 16)
    cons = (cm.dss_sum(chk, topo) / cm.dss_sum(np.ones_like(chk), topo) >= 0.5).astype(np.float64)
    assert 0 < np.abs(cons - chk).sum() < 5000                       # the reference's field is multi-valued at a few thousand nodes
    hip.dcmip_step_inputs(0, dt); hip.synchronize()
    for Qt in Q:
        Qt[:, 4:q - 1] = (torch.as_tensor(cons, device="cuda:0")[:, None, :] * dp).unsqueeze(1)
        Qt[:, q - 1] = dp
    torch.cuda.synchronize(); hip.invalidate_cache()
    m0 = mass(1)
    saved0 = Q[0].clone()                                            # 27.9 GB: the initial state, for the second implementation

    # ---- one tracer step (DSS on read): consistent advection + limiter bounds
    hip.advec_tracers_remap_rk2(dt, 1, 2); hip.synchronize()
    dvp = _dev_tensor(torch, hip.device_ptr("divdp_proj")[0], (n, 72, 16))
    dp_new = dp - dt * dvp
    one = Q[1][:, q - 1] / dp_new
    assert float((one - 1).abs().max()) <= 1e-12, float((one - 1).abs().max())
    chk = Q[1][:, 4:q - 1] / dp_new.unsqueeze(1)                     # the 0/1 checkerboard tracers 5..34
    assert float(chk.min()) >= -1e-12 and float(chk.max()) <= 1 + 1e-12, (float(chk.min()), float(chk.max()))
    del one, chk
    m1 = mass(2)
    np.testing.assert_allclose(m1, m0, rtol=1e-12)
    step1 = Q[1].clone()

    # ---- the same step through one DSS pass per stage must agree to TOL_STEP
    Q[0].copy_(saved0); Q[1].copy_(saved0)
    torch.cuda.synchronize(); hip.invalidate_cache()
    hip.dcmip_step_inputs(0, dt)
    monkeypatch.setenv("TSE_DSS_ON_READ", "0")
    hip.advec_tracers_remap_rk2(dt, 1, 2); hip.synchronize()
    monkeypatch.delenv("TSE_DSS_ON_READ")
    scale = float(step1.abs().max())
    diff = float((Q[1] - step1).abs().max())
    assert diff <= TOL_STEP * scale, (diff, scale)
    del step1, saved0
    torch.cuda.empty_cache()

    # ---- two more steps + remap (a full rsplit cycle): mass through hyperviscosity, limiter, DSS and PPM remap
    run.nstep = 1
    np1 = run.run(2)
    assert run.nstep == 3 and np1 == 2
    hip.synchronize()
    m3 = mass(np1)
    np.testing.assert_allclose(m3, m0, rtol=1e-12)
    # column mass is what the remap conserves; dp3d/ps_v are its by-products (prim_advection_mod.F90:1300-1319)
    ps = _dev_tensor(torch, hip.device_ptr("ps_v")[0], (n, 16))
    assert float((ps / 1.0e5 - 1).abs().max()) < 0.05
    run.close()


def test_ne120_headline_run_carries_the_validated_q4_run_inside_it():
    """Ties the headline configuration (ne120 / qsize = 35: BASELINE configs[2], what bench.py times) to the one whose 12-day norms
    reproduce README:152-153 (ne120 / qsize = 4, test_gpu_dcmip_norms.py).  Tracers are independent of each other and the
    reference initialises tracers 1-4 identically whatever qsize is (dcmip_wrapper_mod.F90:215-243), so tracers 1-4 of a qsize = 35
    run ARE the qsize = 4 run: bit for bit, here through two rsplit cycles (6 tracer steps + 2 remaps) of the device-resident loop --
    different bounds-array strides, different tracer slots of the remap sweeps, the same arithmetic per tracer.  (The reference's perf
    script makes the same identification: run_ne120_perf.sh:27-31 only raises qsize on the run_ne120_tests.sh configuration.)"""
    import torch
    from transport_se_amd.driver import PrimRun
    ne = 120
    keep = {}
    for q in (35, 4):
        run = PrimRun(ne, q, test_case=1)
        np1 = run.run(6)
        assert run.nstep == 6
        run.hip.synchronize()
        n = run.nelem
        both = []
        for tl in (np1, 3 - np1):
            t = _dev_tensor(torch, run.hip.device_ptr("qdp%d" % tl)[0], (n, q, 72 * 16), dtype="<i8")
            both.append(t[:, :4].clone())                      # 3.2 GB per time level
        keep[q] = both
        del t
        run.close()
        torch.cuda.empty_cache()
    for a, b, what in zip(keep[35], keep[4], ("Qdp(np1)", "Qdp(n0)")):
        per_tracer = [int((a[:, t] != b[:, t]).sum()) for t in range(4)]
        assert torch.equal(a, b), "%s: tracers 1-4 of the qsize=35 run differ from the qsize=4 run in %s values (per tracer)" % (what, per_tracer)
    assert int((keep[35][0] != keep[35][1]).sum()) > 0             # (the two time levels are different fields: the comparison is not vacuous)
