"""-m gpu: single calls of the HIP operators against the REFERENCE's own single-call outputs (tests/golden/ref_ops.npz:
divergence_sphere, laplace_sphere_wk -- derivative_mod.F90:2364,2418 -- and remap_Q_ppm -- prim_advection_mod.F90:98 --
called by oracle/ref/ref_harness.F90 on LCG inputs), with the reference's own metric terms (ref_ne2_static.npz).
Tolerances (fp64): the device routines contract a*b+c into FMAs and re-associate the metric products of the Laplacian
(tse_device.h), so agreement is to a few ulp of the slab's largest term: 1e-14 (divergence), 5e-14 (Laplacian) relative to
the slab maximum; remap: TOL_STEP = 5e-13 of the field maximum, column mass to 1e-13."""
import numpy as np
import pytest

from transport_se_amd.hip_mod import HipMod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_ctx(gold):
    st = gold("ref_ne2_static.npz")
    elem = dict(Dinv=st["Dinv"], metdet=st["metdet"], rmetdet=st["rmetdet"], spheremp=st["spheremp"], rspheremp=st["rspheremp"],
                putmapP=st["putmap"].astype(np.int32), getmapP=st["getmap"].astype(np.int32), reverse=st["reverse"].astype(np.int32))
    hip = HipMod(elem, st["Dvv"], (st["hyai"], st["hybi"], float(st["ps0"])), 3, 1e19, device=0)
    yield st, hip
    hip.close()


def test_divergence_sphere_single_call(ref_ctx, gold):
    st, hip = ref_ctx
    g = gold("ref_ops.npz")
    v = np.zeros((hip.nelemd, 2, 4, 4))
    for n, ie in enumerate(g["ie"]):
        v[ie] = g["v"][n]
    out = hip.divergence_sphere(v)
    for n, ie in enumerate(g["ie"]):
        ref = g["div"][n]
        assert np.abs(out[ie] - ref).max() <= 1e-14 * np.abs(ref).max(), (ie, np.abs(out[ie] - ref).max() / np.abs(ref).max())
    assert not out[len(g["ie"]):].any()          # zero flux -> zero divergence, exactly


def test_laplace_sphere_wk_single_call(ref_ctx, gold):
    st, hip = ref_ctx
    g = gold("ref_ops.npz")
    s = np.zeros((hip.nelemd, 4, 4))
    for n, ie in enumerate(g["ie"]):
        s[ie] = g["s"][n]
    out = hip.laplace_sphere_wk(s)
    for n, ie in enumerate(g["ie"]):
        ref = g["lap_wk"][n]
        assert np.abs(out[ie] - ref).max() <= 5e-14 * np.abs(ref).max(), (ie, np.abs(out[ie] - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("generic", [0, 1])
def test_remap_q_ppm_single_call(ref_ctx, gold, monkeypatch, generic):
    """both reference cases (2 % and 20 % thickness perturbation) through the lockstep and the generic column loop"""
    st, hip = ref_ctx
    monkeypatch.setenv("TSE_REMAP_GENERIC", str(generic))
    g = gold("ref_ops.npz")
    n = hip.nelemd
    which = np.arange(n) % 2                                        # the two fixture cases alternate over the elements
    q = g["remap_Qin"][which]; dp1 = g["remap_dp1"][which]; dp2 = g["remap_dp2"][which]
    out = hip.remap_q_ppm(q, dp1, dp2)
    ref = g["remap_Qout"][which]
    # the accepted bound of ONE remap_Q_ppm call against the reference's own output (the kernel folds the ten PPM grid coefficients into
    # five and multiplies by reciprocals: relative 1e-16 per level): 1e-13 of the field maximum
    from conftest import record_margin
    # measured 1.8e-15 (profiles/r04_test_margins.jsonl); asserted 1e-13 (round 3: 5e-13)
    record_margin("remap_q_ppm single call vs reference (generic=%d)" % generic, np.abs(out - ref).max() / np.abs(ref).max(), 1e-13)
    assert np.abs(out - ref).max() <= 1e-13 * np.abs(ref).max(), np.abs(out - ref).max() / np.abs(ref).max()
    np.testing.assert_allclose(out.sum(2), q.sum(2), rtol=1e-13)    # column mass (pin(nlev+1) = pio(nlev+1), :144)


def test_remap_segment_tasks_and_generic_loop_leave_the_bits_of_a_whole_sweep(monkeypatch):
    """k_remap sweeps 32 tracers per round and cuts the remainder (3 of 35) into 8-level segment tasks that start in the middle of
    a column; elements with a displacement of more than one layer take the generic loop.  All three must produce the same bits
    for the same column (the column arithmetic spells its fused operations out for that): tracers 33-35 carry copies of
    tracers 1-3 here, and the whole call is repeated through the generic loop."""
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.hip_mod import HipMod
    from transport_se_amd.hybvcoord import HvCoord
    hv = HvCoord()
    topo = cm.topology(2); geo = cm.geometry(2, topo)
    d = cm.edge_descriptors(topo, np.zeros(24, int), 0)
    elem = dict(Dinv=geo["Dinv"], metdet=geo["metdet"], rmetdet=geo["rmetdet"], spheremp=geo["spheremp"], rspheremp=geo["rspheremp"],
                putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
    rng = np.random.default_rng(5)
    dp1 = 1000.0 * (1 + 0.2 * rng.random((24, 72, 4, 4)))
    dp2 = dp1 * (1 + 0.05 * (rng.random((24, 72, 4, 4)) - 0.5)); dp2 *= dp1.sum(1, keepdims=True) / dp2.sum(1, keepdims=True)
    # 32 + 3 segments; segments only (two rounds of them); 32 + 8; one tracer; 13 left over: a second, partly idle round of sweeps
    for qsize, nseg in ((35, 3), (5, 5), (40, 8), (1, 1), (45, 0)):
        hip = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), qsize, 1e15, device=0)
        Q = rng.random((24, qsize, 72, 4, 4)) * dp1[:, None]
        if qsize >= 32 and nseg:
            Q[:, qsize - nseg:] = Q[:, :nseg]
        monkeypatch.setenv("TSE_REMAP_GENERIC", "0")
        out = hip.remap_q_ppm(Q, dp1, dp2)
        if qsize >= 32 and nseg:
            assert np.array_equal(out[:, qsize - nseg:], out[:, :nseg]), "segment tasks differ from whole sweeps"
        monkeypatch.setenv("TSE_REMAP_GENERIC", "1")
        gen = hip.remap_q_ppm(Q, dp1, dp2)
        assert np.array_equal(out, gen), "generic column loop differs from the lockstep loop (qsize %d)" % qsize
        np.testing.assert_allclose(out.sum(2), Q.sum(2), rtol=1e-13)
        hip.close()


def test_element_qdiag_is_prim_diag_scalars_and_global_integral():
    """tse_element_qdiag: the element shares of prim_diag_scalars' Qmass / Qvar (prim_state_mod.F90:604-655: sum_k Qdp and sum_k Qdp*Q per
    point, Q = Qdp/dp with dp = dhyai*ps0 + dhybi*ps_v, prim_driver_mod.F90:810-815) integrated as global_integral does
    (global_norms_mod.F90:74-80: sum over the points of spheremp*h, i fastest) and the element extrema of Q -- against the same
    operations in numpy, in the same order (sequential sums), to the last bits a different FMA choice can move."""
    import pyoracle as po
    from gpu_common import elem_from_oracle, make_hip
    o = po.Oracle(3, 5, nu_q=2e18)
    elem = elem_from_oracle(o)
    hip = make_hip(o, elem)
    hip.dcmip_init(1, o.lat, o.lon, o.hyam, o.hybm); hip.dcmip_set_initial()
    assert hip.prim_run_subcycle(900.0, 1, 0) == 3
    qdp = hip.fetch("qdp", (2, o.nelem, 5, 72, 4, 4))[1]
    ps_v = hip.fetch("ps_v", (o.nelem, 4, 4))
    mass, var, qmn, qmx = hip.element_qdiag(2)
    dp = (np.diff(o.hyai) * 1.0e5)[None, :, None, None] + np.diff(o.hybi)[None, :, None, None] * ps_v[:, None]
    q = qdp / dp[:, None]
    hm = np.zeros((o.nelem, 5, 4, 4)); hv = np.zeros_like(hm)
    for k in range(72):                                   # levels inside a point, in order
        hm = hm + qdp[:, :, k]; hv = hv + qdp[:, :, k] * q[:, :, k]
    jm = np.zeros((o.nelem, 5)); jv = np.zeros_like(jm)
    for p in range(16):                                   # then the points, i fastest
        w = o.spheremp.reshape(o.nelem, 16)[:, p][:, None]
        jm = jm + w * hm.reshape(o.nelem, 5, 16)[:, :, p]; jv = jv + w * hv.reshape(o.nelem, 5, 16)[:, :, p]
    assert np.array_equal(mass, jm) and np.array_equal(var, jv)          # no contraction in the kernel: the very same roundings
    assert np.array_equal(qmn, q.reshape(o.nelem, 5, -1).min(2)) and np.array_equal(qmx, q.reshape(o.nelem, 5, -1).max(2))
    hip.close(); o.close()
