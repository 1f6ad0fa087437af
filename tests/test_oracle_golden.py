"""The CPU restatement (oracle/) against outputs of the reference itself (tests/golden/, produced by
oracle/ref/make_golden.py from the unmodified reference built with amdflang).  Every comparison here is
bit-for-bit (fp64 on the same host arithmetic) unless a tolerance is written next to it."""
import json

import numpy as np
import pytest

import pyoracle as po
import norms


@pytest.fixture(scope="module")
def o5():
    o = po.Oracle(2, 5, nu_q=1e19)
    yield o
    o.close()


def test_gll_dvv_vcoord(gold, o5):
    st = gold("ref_ne2_static.npz")
    assert np.array_equal(st["Dvv"], o5.Dvv)          # derivative_mod.F90:116-192
    assert np.array_equal(st["gll"], o5.gll) and np.array_equal(st["gllw"], o5.gllw)
    assert np.array_equal(st["hyai"], o5.hyai) and np.array_equal(st["hybi"], o5.hybi)
    assert st["alpha"] == o5.alpha[0]                  # prim_driver_mod.F90:265-272


def test_metric_terms(gold, o5):
    st = gold("ref_ne2_static.npz")
    for k in ("lon", "lat", "D", "Dinv", "metdet", "rmetdet", "mp", "spheremp", "rspheremp"):
        assert np.array_equal(st[k], getattr(o5, k)), k


def _nbr_from_maps(put, get, rev):
    owner = {int(put[e, d]): (e, d) for e in range(put.shape[0]) for d in range(8) if put[e, d] >= 0}
    nbr = -np.ones(put.shape + (3,), dtype=int)
    for e in range(put.shape[0]):
        for d in range(8):
            if get[e, d] >= 0:
                n, nd = owner[int(get[e, d])]
                nbr[e, d] = (n, nd, int(rev[n, nd]) if d < 4 else 0)
    return nbr


def test_topology_matches_reference_descriptors(gold, o5):
    """putmapP/getmapP/reverse of the reference (schedule_mod.F90:905,929, cube_mod.F90:2371-2381) imply the
    same neighbour/orientation table as the oracle's geometric construction."""
    st = gold("ref_ne2_static.npz")
    nbr = _nbr_from_maps(st["putmap"], st["getmap"], st["reverse"])
    assert np.array_equal(nbr[..., 0], o5.nbr_elem)
    assert np.array_equal(nbr[..., 1][nbr[..., 0] >= 0], o5.nbr_dir[o5.nbr_elem >= 0])
    assert np.array_equal(nbr[..., 2][:, :4], o5.nbr_rev[:, :4])
    assert (o5.nbr_elem < 0).sum() == 24                # 3 elements at each of the 8 cube vertices
    # the oracle's own descriptors obey the same contract
    own = _nbr_from_maps(o5.putmap, o5.getmap, o5.reverse)
    assert np.array_equal(own[..., 0], o5.nbr_elem)


def test_element_operators(gold, o5):
    g = gold("ref_ops.npz")
    for n, ie in enumerate(g["ie"]):
        assert np.array_equal(o5.divergence_sphere(ie, g["v"][n]), g["div"][n])        # derivative_mod.F90:2364
        assert np.array_equal(o5.gradient_sphere(ie, g["s"][n]), g["grad"][n])          # :1660
        assert np.array_equal(o5.divergence_sphere_wk(ie, g["v"][n]), g["div_wk"][n])   # :2027
        assert np.array_equal(o5.laplace_sphere_wk(ie, g["s"][n]), g["lap_wk"][n])      # :2418


def test_remap_q_ppm(gold):
    g = gold("ref_ops.npz")
    for n in range(2):
        out = po.remap_q_ppm(g["remap_Qin"][n], g["remap_dp1"][n], g["remap_dp2"][n])
        assert np.array_equal(out, g["remap_Qout"][n])                                  # prim_advection_mod.F90:98
        # column mass is conserved by construction (pin(nlev+1)=pio(nlev+1), :144)
        np.testing.assert_allclose(out.sum(1), g["remap_Qin"][n].sum(1), rtol=1e-13)


def test_dcmip11_six_steps_two_remaps(gold, o5):
    """prim_run_subcycle x2 (3 RK-SSP tracer steps + vertical_remap each), limiter 8 + hyperviscosity, 5
    tracers (4 analytic + 1 checkerboard): every Qdp bit equal to the reference's."""
    g = gold("ref_ne2_dcmip11.npz")
    cfg = json.loads(str(g["config"]))
    o5.dcmip_init(1)
    dt = cfg["tstep"]
    o5.dcmip_step_inputs(1, 0, dt)
    o5.advec_tracers_remap_rk2(dt, 0)
    sub = g["sub"]
    assert np.array_equal(o5.qdp[1], g["qdp_step1"])
    assert np.array_equal(o5.vn0[sub], g["vn0_step1"]) and np.array_equal(o5.dp[sub], g["dp_step1"])
    assert np.array_equal(o5.divdp[sub], g["divdp_step1"]) and np.array_equal(o5.divdp_proj[sub], g["divdp_proj_step1"])
    assert np.array_equal(o5.eta_dot_dpdn[sub][:, :72], g["eta_dot_dpdn_step1"])
    for nstep in (1, 2):
        o5.dcmip_step_inputs(1, nstep, dt)
        o5.advec_tracers_remap_rk2(dt, nstep)
    assert o5.vertical_remap(3 * dt, 2) == 0
    assert np.array_equal(o5.qdp[1], g["qdp_step3"])
    assert np.array_equal(o5.dp3d, g["dp3d_step3"]) and np.array_equal(o5.ps_v, g["ps_v_step3"])
    done, nstep = o5.prim_run(1, dt, 1, nstep=3)
    assert (done, nstep) == (3, 6)
    assert np.array_equal(o5.qdp[0], g["qdp_step6"])
    # tracer mass: sum spheremp*Qdp conserved by the whole step to round-off
    m0 = norms.tracer_mass(o5.spheremp, g["qdp_step1"]); m6 = norms.tracer_mass(o5.spheremp, o5.qdp[0])
    np.testing.assert_allclose(m6, m0, rtol=2e-13)


def test_dcmip12_three_steps(gold):
    g = gold("ref_ne2_dcmip12.npz")
    cfg = json.loads(str(g["config"]))
    o = po.Oracle(cfg["ne"], cfg["qsize"], nu_q=cfg["nu_q"])
    o.dcmip_init(2)
    done, nstep = o.prim_run(2, cfg["tstep"], 1)
    assert done == 3
    assert np.array_equal(o.qdp[1], g["qdp_step3"])
    o.close()


def test_reference_is_bfb_across_ranks(gold):
    g = gold("ref_ne2_2rank_sched.npz")
    assert bool(g["bfb_1v2"])  # recorded when the fixture was generated (README:46-47 "should be BFB")


def test_limiter8_properties():
    """limiter_optim_iter_full (prim_advection_mod.F90:976-1094): bounds, mass, relaxed-bound write-back."""
    rng = np.random.default_rng(12345)
    w = rng.uniform(0.5, 1.5, (4, 4)); dpm = rng.uniform(0.9, 1.1, (4, 4)) * 100
    x = rng.uniform(-0.5, 1.5, (4, 4))
    out, mn, mx, it = po.limiter8(x * dpm, w, 0.0, 1.0, dpm)
    q = out / dpm
    assert q.min() >= -1e-15 and q.max() <= 1 + 1e-15 and it > 1
    np.testing.assert_allclose((out * w).sum(), (x * dpm * w).sum(), rtol=1e-13)
    # infeasible bounds: mean above maxp -> maxp relaxed to the mean and written back
    out, mn, mx, it = po.limiter8(np.full((4, 4), 2.0) * dpm, w, 0.0, 1.0, dpm)
    assert abs(mx - 2.0) < 1e-14
    np.testing.assert_allclose(out / dpm, 2.0, rtol=1e-14)
    # sumc <= 0: untouched (:1016)
    out, mn, mx, it = po.limiter8(x * dpm, -w, 0.0, 1.0, dpm)
    assert np.array_equal(out, x * dpm) and it == 0


def test_dss_continuity_and_q1_preserved(o5):
    """DSS'd field is continuous across elements up to the corner summation order; Q==1 is preserved by a
    tracer step ("consistent advection", prim_advection_mod.F90:23-33)."""
    rng = np.random.default_rng(1)
    f = rng.uniform(size=(o5.nelem, 3, 4, 4))
    g = o5.dss(f * o5.spheremp[:, None], 0) * o5.rspheremp[:, None]
    # value seen from both sides of every shared edge point agrees
    for e in range(o5.nelem):
        n, nd, rev = o5.nbr_elem[e, 2], o5.nbr_dir[e, 2], o5.nbr_rev[e, 2]   # my south edge
        mine = g[e, :, 0, :]
        edge = {0: g[n][:, :, 0], 1: g[n][:, :, 3], 2: g[n][:, 0, :], 3: g[n][:, 3, :]}[int(nd)]
        np.testing.assert_allclose(mine, edge[:, ::-1] if rev else edge, rtol=1e-14)
    o5.dcmip_init(1)
    o5.dcmip_step_inputs(1, 0, 1800.0)
    o5.qdp[0][:, 4] = o5.dp          # tracer 5: Q = 1 w.r.t. the reference thickness the step uses
    o5.advec_tracers_remap_rk2(1800.0, 0)
    dp_end = o5.dp - 1800.0 * o5.divdp_proj          # dp after the three half-steps averaged: (dp + 2*(dp-1.5dt div))/3
    np.testing.assert_allclose(o5.qdp[1][:, 4] / dp_end, 1.0, rtol=5e-12)


def test_remap_alg2_bit_exact_vs_reference(gold):
    """vert_remap_q_alg = 2 (piecewise-constant boundary cells, prim_advection_mod.F90:230-250,283-341): the restatement against the
    reference's own outputs with that namelist value (oracle/ref/make_golden.py --alg2): single calls and a 6-step run with two remaps"""
    import json
    g = gold("ref_ne2_alg2.npz")
    cfg = json.loads(str(g["config"]))
    po.set_vert_remap_q_alg(2)
    try:
        for i in range(2):
            out = po.remap_q_ppm(g["remap_Qin"][i], g["remap_dp1"][i], g["remap_dp2"][i])
            assert np.array_equal(out, g["remap_Qout"][i])
        o = po.Oracle(cfg["ne"], cfg["qsize"], nu_q=cfg["nu_q"])
        o.dcmip_init(1)
        done, _ = o.prim_run(1, cfg["tstep"], 2)
        assert done == 6 and np.array_equal(o.qdp[0], g["qdp_step6"])
        o.close()
    finally:
        po.set_vert_remap_q_alg(0)
    # the default algorithm does not reproduce them
    assert not np.array_equal(po.remap_q_ppm(g["remap_Qin"][0], g["remap_dp1"][0], g["remap_dp2"][0]), g["remap_Qout"][0])
