"""norms.py -- TEST INFRASTRUCTURE: the error norms the reference's tests print, restated in numpy.

Follows test/dcmip1-1/dcmip1-1_error_norm_ng.ncl:39-77 (and the dcmip1-2 twin): on the unique-column native
grid, dq = Q(t_end) - Q(t0); dV = R cos(lat) dlon * R dlat * dh(lev) with dh rebuilt from the level heights
by the "2*(h-base)" recursion; L1, L2, Linf relative to (q0 - avg(q0)) with avg() the unweighted mean over
all (lev, ncol); q_max/q_min of the final field.  NCL is not in this image, so this file is the norm tool;
it is checked against the README's published lines for ne8 (tests/golden/ref_ne8_norms.json).
"""
import numpy as np

G = 9.80616
RGAS = 287.04
P0 = 100000.0


def unique_columns(lat, lon):
    """index (into the flattened [e][j][i] points) of one owner per distinct GLL node: the first point in
    (element, j, i) order, i.e. the smallest (ig-1)*np^2+(j-1)*np+i (dof_mod.F90:43-57,95-116)."""
    lat = np.asarray(lat).reshape(-1); lon = np.asarray(lon).reshape(-1)
    xyz = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], 1)
    key = np.round(xyz * 1e8).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    return np.sort(first)


def level_heights(hyam, hybm):
    H = RGAS * 300.0 / G
    return H * np.log(1.0 / (np.asarray(hyam) + np.asarray(hybm)))


def dcmip_norms(ne, lat, lon, q_i, q_f, zm):
    """q_i, q_f: [nelem][nlev][4][4] mixing ratios; returns dict L1 L2 Linf q_max q_min"""
    nlev = q_i.shape[1]
    cols = unique_columns(lat, lon)
    assert cols.size == 6 * ne * ne * 9 + 2, cols.size
    qi = np.moveaxis(q_i, 1, 0).reshape(nlev, -1)[:, cols]
    qf = np.moveaxis(q_f, 1, 0).reshape(nlev, -1)[:, cols]
    latc = np.asarray(lat).reshape(-1)[cols]
    dh = np.zeros(nlev); base = 0.0
    for i in range(1, nlev + 1):
        dh[nlev - i] = 2.0 * (zm[nlev - i] - base)
        base = base + dh[nlev - i]
    R = 6.37122e6
    dlat = 0.5 * np.pi / (ne * 3)
    dV = (R * np.cos(latc) * dlat)[None, :] * (R * dlat) * dh[:, None]
    dq = qf - qi
    dev = np.abs(qi - qi.mean())
    return dict(L1=float((np.abs(dq) * dV).sum() / (dev * dV).sum()),
                L2=float(np.sqrt((dq * dq * dV).sum()) / np.sqrt((dev * dev * dV).sum())),
                Linf=float((np.abs(dq) * dV).max() / (dev * dV).max()),
                q_max=float(qf.max()), q_min=float(qf.min()))


def hybrid_dp(hyai, hybi, ps_v):
    """dp(k) = dhyai*ps0 + dhybi*ps_v  (prim_driver_mod.F90:810-812): [nelem][nlev][4][4]"""
    da = np.diff(hyai) * P0; db = np.diff(hybi)
    return da[None, :, None, None] + db[None, :, None, None] * ps_v[:, None, :, :]


def dcmip_norms_from_qdp(ne, lat, lon, qdp0, qdp1, ps_v1, hyai, hybi, hyam, hybm):
    """Q(t0) = Qdp/dp(ps0) as prim_init2 built it; Q(t_end) = Qdp(np1)/dp(ps_v(np1)) (prim_driver_mod.F90:807-822)"""
    q0 = qdp0 / hybrid_dp(hyai, hybi, np.full_like(ps_v1, P0))
    q1 = qdp1 / hybrid_dp(hyai, hybi, ps_v1)
    return dcmip_norms(ne, lat, lon, q0, q1, level_heights(hyam, hybm))


def tracer_mass(spheremp, qdp):
    """sum_e sum_ij spheremp*sum_k Qdp per tracer: the conserved quantity of the "Q,Q diss" line
    (prim_state_mod.F90:352-385 via global_integral, global_norms_mod.F90:39-86, up to the 1/4pi factor)"""
    return np.einsum("eji,eqkji->q", spheremp, qdp)
