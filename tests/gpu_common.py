"""helpers shared by the -m gpu parity tests: build the host-side `elem` dict from the oracle's geometry
(the checker supplies inputs; the product path under test is transport_se_amd + libtransport_se_hip.so)."""
import numpy as np

import pyoracle as po
from transport_se_amd import HipMod


def elem_from_oracle(o, qsize_d=None):
    q = qsize_d or o.qsize
    n = o.nelem
    elem = dict(
        Dinv=o.Dinv.copy(), metdet=o.metdet.copy(), rmetdet=o.rmetdet.copy(), spheremp=o.spheremp.copy(),
        rspheremp=o.rspheremp.copy(), putmapP=o.putmap.astype(np.int32), getmapP=o.getmap.astype(np.int32),
        reverse=o.reverse.astype(np.int32),
        Qdp=np.zeros((n, 2, q, 72, 4, 4)), vn0=np.zeros((n, 72, 2, 4, 4)), dp=np.zeros((n, 72, 4, 4)),
        divdp=np.zeros((n, 72, 4, 4)), divdp_proj=np.zeros((n, 72, 4, 4)), omega_p=np.zeros((n, 72, 4, 4)),
        eta_dot_dpdn=np.zeros((n, 73, 4, 4)), dp3d=np.zeros((n, 72, 4, 4)), ps_v=np.zeros((n, 4, 4)))
    return elem


def sync_inputs_from_oracle(o, elem):
    elem["vn0"][...] = o.vn0; elem["dp"][...] = o.dp
    elem["eta_dot_dpdn"][...] = o.eta_dot_dpdn; elem["omega_p"][...] = o.omega_p


def make_hip(o, elem, **kw):
    return HipMod(elem, o.Dvv, (o.hyai, o.hybi, 1.0e5), o.qsize, o.nu_q, rsplit=o.rsplit, **kw)


def relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
