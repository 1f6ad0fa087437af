"""diagnostics -- the numbers the reference's tests look at, as product code.

* `dcmip_norms`: L1/L2/Linf/q_max/q_min exactly as test/dcmip1-1/dcmip1-1_error_norm_ng.ncl:39-77 (and the dcmip1-2 twin)
  computes them from the history file: unique-column native grid, dV = R cos(lat) dlon * R dlat * dh(lev) with dh rebuilt
  from the level heights by the "2*(h-base)" recursion, relative to (q0 - avg(q0)).
* `tracer_mass`: the conserved integral behind the "Q,Q diss" line (prim_state_mod.F90:352-385 through
  global_integral, global_norms_mod.F90:39-86).
* `printstate_lines`: the `qv=  min max sum` lines of prim_printstate (prim_state_mod.F90:341-347).
* `global_sum` / `gather_by_gid`: the multi-rank forms.  The reference makes its global integrals independent of the task
  count with a fixed-point reproducible sum (repro_sum_mod.F90:216-632; "independent of thread count and task count",
  global_norms_mod.F90:66-68).  Here every ELEMENT's partial sum is formed in a fixed order (on the device:
  tse_element_mass), the partials travel to rank 0, and rank 0 adds them with math.fsum -- the correctly rounded exact sum,
  which does not depend on the order, hence not on how the elements were distributed.  Fields needed whole (the norm
  calculator works on the unique-column grid) are reassembled on rank 0 in global element order, after which every number
  is computed exactly as on one rank.
"""
import math

import numpy as np

G = 9.80616      # physical_constants.F90:23
RGAS = 287.04    # :25
P0 = 100000.0    # :27


def unique_columns(lat, lon):
    """one owner per distinct GLL node: the first point in (element, j, i) order, i.e. the smallest
    (ig-1)*np^2+(j-1)*np+i among the sharers (dof_mod.F90:43-57,95-116)"""
    lat = np.asarray(lat).reshape(-1); lon = np.asarray(lon).reshape(-1)
    xyz = np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], 1)
    key = np.round(xyz * 1e8).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    return np.sort(first)


def level_heights(hyam, hybm):
    """z_m = H ln(1/eta_m), H = Rd*T0/g with T0 = 300 K (dcmip_wrapper_mod.F90:24-25,68)"""
    return (RGAS * 300.0 / G) * np.log(1.0 / (np.asarray(hyam) + np.asarray(hybm)))


def hybrid_dp(hyai, hybi, ps_v):
    """dp(k) = dhyai*ps0 + dhybi*ps_v (prim_driver_mod.F90:810-812) -> [nelem][nlev][4][4]"""
    da = np.diff(hyai) * P0; db = np.diff(hybi)
    return da[None, :, None, None] + db[None, :, None, None] * np.asarray(ps_v)[:, None, :, :]


def dcmip_norms(ne, lat, lon, q_i, q_f, zm):
    nlev = q_i.shape[1]
    cols = unique_columns(lat, lon)
    if cols.size != 6 * ne * ne * 9 + 2:
        raise ValueError("expected %d unique columns, found %d" % (6 * ne * ne * 9 + 2, cols.size))
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


def tracer_mass(spheremp, qdp):
    return np.einsum("eji,eqkji->q", spheremp, qdp)


def printstate_lines(qdp, dp):
    """(min, max, sum) of Q = Qdp/dp per tracer, as the `qv=` lines print them"""
    q = qdp / dp[:, None]
    return [(float(q[:, t].min()), float(q[:, t].max()), float(q[:, t].sum())) for t in range(q.shape[1])]


def gather_by_gid(local, gid, nelem, dist_mod, rank, world):
    """local[n_local, ...] on every rank -> full[nelem, ...] in global element order on rank 0 (None elsewhere)"""
    if world == 1:
        full = np.empty((nelem,) + local.shape[1:], dtype=local.dtype)
        full[gid] = local
        return full
    parts = [None] * world if rank == 0 else None
    dist_mod.gather_object((np.asarray(gid), np.ascontiguousarray(local)), parts, dst=0)
    if rank != 0:
        return None
    full = np.empty((nelem,) + local.shape[1:], dtype=local.dtype)
    for g, x in parts:
        full[g] = x
    return full


def global_sum(partials, gid, nelem, dist_mod=None, rank=0, world=1):
    """partials[n_local][m] (one row per local element, formed in a fixed order) -> [m] exact global sums on rank 0"""
    full = gather_by_gid(np.asarray(partials, dtype=np.float64), gid, nelem, dist_mod, rank, world)
    if full is None:
        return None
    return np.array([math.fsum(full[:, j].tolist()) for j in range(full.shape[1])])


def element_q_partials(qdp, dp):
    """per element and tracer: (min, max, sum) of Q = Qdp/dp, the sum over the element's points in a fixed (row) order"""
    q = (qdp / dp[:, None]).reshape(qdp.shape[0], qdp.shape[1], -1)
    return q.min(2), q.max(2), q.sum(2)
