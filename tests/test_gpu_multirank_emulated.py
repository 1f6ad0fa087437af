"""-m gpu: the library's multi-rank code path (k_pack/k_pack_minmax, remote columns in k_dss/k_nbr_minmax, the six
exchanges per tracer step) on ONE GPU: the sphere is cut into 2 and 3 "ranks", each a separate tse_ctx driven from its
own thread; the exchange callback swaps the packed slots between the contexts with hipMemcpy.  The result must equal the
single-context run BIT FOR BIT (north_star: BFB across GPU counts)."""
import ctypes as C
import threading

import numpy as np
import pytest

from transport_se_amd import cube_mesh as cm
from transport_se_amd.driver import partition
from transport_se_amd.hip_mod import HipMod
from transport_se_amd.hybvcoord import HvCoord

pytestmark = pytest.mark.gpu
NE, QSIZE, NU_Q, DT = 4, 3, 5e17, 900.0


def _run(world, ne=None):
    global NE
    NE = ne or 4
    return _run_ne(world)


def _run_ne(world):
    hv = HvCoord()
    topo = cm.topology(NE); geo = cm.geometry(NE, topo)
    nelem = 6 * NE * NE
    owner = partition(NE, world)
    descs = [cm.edge_descriptors(topo, owner, r) for r in range(world)]
    hip = C.CDLL("libamdhip64.so")
    barrier = threading.Barrier(world)
    bufs = [None] * world            # (sendbuf ptr, nlyr) published by each rank inside the callback
    ctxs = [None] * world
    result = [None] * world
    errors = []

    lens = [dict() for _ in range(world)]   # lens[r][kind] = (send_len per slot, recv_len per slot)

    class Exchange:
        """bndry_exchangeV between the emulated ranks: slot s of rank r <- the matching send slot of its peer"""

        def __init__(self, r):
            self.r = r
            d = descs[r]
            lens[r][0] = ([s[2] for s in d["send"]], [s[2] for s in d["recv"]])

        def set_minmax_layout(self, send_len, recv_len):       # called by HipMod after tse_init
            lens[self.r][1] = ([int(x) for x in send_len], [int(x) for x in recv_len])

        def __call__(self, sbuf, rbuf, nlyr, kind):
            r = self.r; d = descs[r]
            bufs[r] = (sbuf, nlyr)
            barrier.wait()
            roff = np.concatenate([[0], np.cumsum(lens[r][kind][1])]).astype(int)
            for i, (peer, _, _) in enumerate(d["recv"]):
                pd = descs[peer]
                j = [k for k, s in enumerate(pd["send"]) if s[0] == r][0]
                soff = np.concatenate([[0], np.cumsum(lens[peer][kind][0])]).astype(int)
                ln = lens[r][kind][1][i]
                assert lens[peer][kind][0][j] == ln and bufs[peer][1] == nlyr
                src = bufs[peer][0] + int(soff[j]) * nlyr * 8
                dst = rbuf + int(roff[i]) * nlyr * 8
                rc = hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(ln * nlyr * 8), C.c_int(3))
                assert rc == 0
            assert hip.hipDeviceSynchronize() == 0   # a D2D hipMemcpy may return early; recvbuf must be complete on return
            barrier.wait()
            return 0

    def make_exchange(r):
        return Exchange(r)

    def worker(r):
        try:
            d = descs[r]; mine = d["elems"]
            elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine],
                        spheremp=geo["spheremp"][mine], rspheremp=geo["rspheremp"][mine],
                        putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
            h = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), QSIZE, NU_Q, device=0,
                       schedule=dict(send=d["send"], recv=d["recv"]), exchange=make_exchange(r) if world > 1 else None)
            ctxs[r] = h
            h.dcmip_init(1, geo["lat"][mine], geo["lon"][mine], hv.hyam, hv.hybm)
            h.dcmip_set_initial()
            assert h.prim_run_subcycle(DT, 2, 0) == 6
            result[r] = (mine, h.fetch("qdp", (2, mine.size, QSIZE, 72, 4, 4))[0].copy(),
                         h.fetch("divdp_proj", (mine.size, 72, 4, 4)).copy())
            h.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    assert not errors, errors
    q = np.empty((nelem, QSIZE, 72, 4, 4)); dv = np.empty((nelem, 72, 4, 4))
    for mine, qq, dd in result:
        q[mine] = qq; dv[mine] = dd
    return q, dv


def test_bfb_on_an_irregular_partition():
    """ne5 on 4 ranks: SFC chunks of 37/38 elements that the 4 x 4 patches cannot tile (partial patches, holes, patches cut by the
    rank boundary), boundary-first split launches with few or no interior patches"""
    q1, d1 = _run(1, ne=5)
    qn, dn = _run(4, ne=5)
    assert np.isfinite(q1).all() and q1.max() > 0
    assert np.array_equal(qn, q1) and np.array_equal(dn, d1)


def test_bfb_across_rank_counts():
    q1, d1 = _run(1)
    assert np.isfinite(q1).all() and q1.max() > 0
    for world in (2, 3):
        qn, dn = _run(world)
        assert np.array_equal(qn, q1), "Qdp differs between 1 and %d ranks" % world
        assert np.array_equal(dn, d1)


@pytest.mark.parametrize("shape", ["16,16,16,16", "32,32,32,32", "24,16,32,24"])
def test_bfb_with_boundary_bands_and_other_block_shapes(monkeypatch, shape):
    """the tilings are pure scheduling: rank-boundary elements in patches of their own (TSE_BOUNDARY_STRIPS=1: chains along the boundary
    filled with the elements behind them, any shape the halo-ring and element-ring tables allow) and 6x4 / 8x4 blocks for the
    DSS-on-read kernels must leave every bit where the regular 4x4 tiling on one rank puts it"""
    ref5 = _run(1, ne=5)          # default tiling, one rank
    ref4 = _run(1)
    monkeypatch.setenv("TSE_BOUNDARY_STRIPS", "1")
    monkeypatch.setenv("TSE_PATCH_SHAPE", shape)
    for world, ne, ref in ((4, 5, ref5), (3, None, ref4)):
        qn, dn = _run(world, ne=ne)
        assert np.array_equal(qn, ref[0]) and np.array_equal(dn, ref[1]), (world, ne, shape)
