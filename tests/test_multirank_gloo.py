"""N>1 path on CPU: world_size-2 (and 3) torch.distributed/gloo runs of the bndry_exchangeV replacement
(transport_se_amd.driver.HaloExchange: one isend + one irecv per neighbour-rank slot) wrapped around a numpy emulation
of the library's pack (k_pack) and gather-DSS (k_dss) that uses the SAME host tables the library builds from the
reference-style descriptors.  Result must equal the single-rank DSS bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from transport_se_amd import cube_mesh as cm
from transport_se_amd.driver import HaloExchange, check_schedules_match, partition

NE, NLYR = 4, 5


def _tables(desc):
    """what tse_init derives from putmapP/getmapP/reverse + the Send/RecvCycle slots (tse_api.hip)"""
    put, get, rev = desc["putmapP"], desc["getmapP"], desc["reverse"]
    n = put.shape[0]
    own = {}
    for e in range(n):
        for d in range(8):
            if put[e, d] < 0:
                continue
            if d < 4:
                for k in range(4):
                    own[put[e, d] + (3 - k if rev[e, d] else k)] = (e, cm.edge_point(d, k))
            else:
                own[put[e, d]] = (e, cm.CORNER_POINT[d])
    send_idx, recv_idx = {}, {}
    for (_, ptr, ln) in desc["send"]:
        for i in range(ln):
            send_idx[ptr - 1 + i] = len(send_idx)
    for (_, ptr, ln) in desc["recv"]:
        for i in range(ln):
            recv_idx[ptr - 1 + i] = len(recv_idx)
    send_src = [own[c] for c in sorted(send_idx, key=send_idx.get)]
    return own, send_src, recv_idx


def _worker(rank, world, port, field, ref, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    topo = cm.topology(NE)
    owner = partition(NE, world)
    desc = cm.edge_descriptors(topo, owner, rank)
    mine = desc["elems"]
    own, send_src, recv_idx = _tables(desc)
    f = field[mine]                                               # [nelemd][NLYR][16]
    sendbuf = np.zeros((len(send_src), NLYR)); recvbuf = np.zeros((len(recv_idx), NLYR))
    for c, (e, p) in enumerate(send_src):                         # k_pack
        sendbuf[c] = f[e, :, p]
    # entries of the compact min/max exchange per slot = (element, direction) pairs whose first column lies in the slot (tse_init)
    starts = sorted(int(c) for c in desc["putmapP"].reshape(-1) if c >= 0)
    mm = [sum(1 for c in starts if ptr - 1 <= c < ptr - 1 + ln) for (_, ptr, ln) in desc["send"]]
    check_schedules_match(desc, (mm, mm), rank, dist)            # raises on any asymmetry between the ranks' slot lists
    ex = HaloExchange(desc, "cpu", dist, torch)
    assert ex(sendbuf.ctypes.data, recvbuf.ctypes.data, NLYR) == 0
    out = f.copy()                                                # k_dss: gather in the reference's order
    for e in range(mine.size):
        for d in (cm.S, cm.E, cm.N, cm.W):
            for k in range(4):
                col = desc["getmapP"][e, d] + k
                v = recvbuf[recv_idx[col]] if col in recv_idx else f[own[col][0], :, own[col][1]]
                out[e, :, cm.edge_point(d, k)] += v
        for d in (cm.SW, cm.SE, cm.NE, cm.NW):
            col = desc["getmapP"][e, d]
            if col >= 0:
                v = recvbuf[recv_idx[col]] if col in recv_idx else f[own[col][0], :, own[col][1]]
                out[e, :, cm.CORNER_POINT[d]] += v
    ok = np.array_equal(out, ref[mine])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_gloo(world):
    topo = cm.topology(NE)
    rng = np.random.default_rng(3)
    field = rng.uniform(size=(6 * NE * NE, NLYR, 16))
    ref = np.stack([cm.dss_sum(field[:, l], topo) for l in range(NLYR)], 1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, field, ref, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]
