"""-m gpu: the RCCL legs of the halo exchange on the one GPU we have.

RCCL refuses two ranks of one communicator on the same device ("Duplicate GPU detected"), so a real multi-rank run cannot
be staged on a 1-GPU box.  What can be checked here:
  * the in-library exchange (tse_comm_init + ncclGroupStart/ncclRecv/ncclSend/ncclGroupEnd on the library's communication
    stream, boundary-first split launches, event hand-over between the two streams) in LOOPBACK: a context that owns one
    rank's share of a 3-rank partition, with every neighbour slot pointed at rank 0 of a 1-rank communicator, so that each
    slot it sends comes back as the slot it receives.  The same loopback through the callback form of the seam (one
    hipMemcpy sendbuf -> recvbuf) must give the same bits: same slot offsets, same message sizes, same ordering of pack,
    exchange, unpack and compute -- everything except a second GPU.
  * torch.distributed backend "nccl" (= RCCL) with world_size 1 for the `exchange="torch"` callback (HaloExchange wrapping raw
    HIP device pointers).
The multi-rank slot logic itself is covered by tests/test_multirank_gloo.py (gloo, CPU) and
tests/test_gpu_multirank_emulated.py (several tse_ctx on one GPU, bit-for-bit against one context)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NE, QSIZE, NU_Q, DT = 4, 3, 5e17, 900.0


def _loopback_run(mode):
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.driver import partition
    from transport_se_amd.hip_mod import HipMod
    from transport_se_amd.hybvcoord import HvCoord
    hv = HvCoord()
    topo = cm.topology(NE); geo = cm.geometry(NE, topo)
    owner = partition(NE, 3)
    rank = max(range(3), key=lambda r: len(cm.edge_descriptors(topo, owner, r)["send"]))   # a rank with the most neighbour slots
    d = cm.edge_descriptors(topo, owner, rank)
    assert len(d["send"]) >= 1
    mine = d["elems"]
    sched = dict(send=[(0, p, l) for (_, p, l) in d["send"]], recv=[(0, p, l) for (_, p, l) in d["recv"]])   # every peer -> rank 0 (self)
    elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine], spheremp=geo["spheremp"][mine],
                rspheremp=geo["rspheremp"][mine], putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
    hip_rt = C.CDLL("libamdhip64.so")
    lens = {}

    class Loop:
        def set_minmax_layout(self, send_len, recv_len):
            lens[1] = int(sum(send_len)); assert list(send_len) == list(recv_len)

        def __call__(self, sbuf, rbuf, nlyr, kind):
            n = lens[kind] * nlyr * 8
            rc = hip_rt.hipMemcpy(C.c_void_p(rbuf), C.c_void_p(sbuf), C.c_size_t(n), C.c_int(3))
            # a device-to-device hipMemcpy may return before the copy has run; the seam's contract is that recvbuf is
            # complete when the callback returns (the library's streams do not synchronise with the null stream)
            return rc or hip_rt.hipDeviceSynchronize()

    lens[0] = sum(s[2] for s in sched["send"])
    h = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), QSIZE, NU_Q, device=0, schedule=sched,
               exchange=Loop() if mode == "callback" else None)
    if mode == "rccl":
        h.comm_init(HipMod.comm_unique_id(), 0, 1)
        assert h.comm_info() == (0, 1)
    else:
        assert h.comm_info() == (0, 1)      # no communicator: reported as a single rank
    nb, ni = h.boundary_layout()
    assert nb > 0 and nb + ni == mine.size
    h.dcmip_init(1, geo["lat"][mine], geo["lon"][mine], hv.hyam, hv.hybm)
    h.dcmip_set_initial()
    assert h.prim_run_subcycle(DT, 2, 0) == 6                      # 6 tracer steps, 2 remaps, whole-step (DSS on read) path
    q = h.fetch("qdp", (2, mine.size, QSIZE, 72, 4, 4)).copy()
    dv = h.fetch("divdp_proj", (mine.size, 72, 4, 4)).copy()
    # the per-stage API takes the other exchange route (everything on the compute stream); from the initial state again
    h.dcmip_set_initial(); h.dcmip_step_inputs(0, DT); h.compute_divdp()
    h.euler_step(2, 1, DT / 2, 3, 0); h.euler_step(2, 2, DT / 2, 1, 1); h.euler_step(2, 2, DT / 2, 2, 2); h.qdp_time_avg(3, 1, 2)
    q2 = h.fetch("qdp", (2, mine.size, QSIZE, 72, 4, 4)).copy()
    h.close()
    return q, dv, q2


def test_in_library_rccl_exchange_matches_callback_in_loopback():
    qa, da, q2a = _loopback_run("callback")
    qb, db, q2b = _loopback_run("rccl")
    assert np.isfinite(qa).all() and qa.max() > 0
    assert np.array_equal(qa, qb), "whole-step path: RCCL exchange differs from the callback exchange"
    assert np.array_equal(da, db)
    assert np.array_equal(q2a, q2b), "per-stage path: RCCL exchange differs from the callback exchange"


def test_halo_without_transport_fails_loudly():
    """neighbour slots, but neither a communicator nor a callback: the first step must fail with a message, not hang"""
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.driver import partition
    from transport_se_amd.hip_mod import HipMod, TseError
    from transport_se_amd.hybvcoord import HvCoord
    hv = HvCoord()
    topo = cm.topology(2); geo = cm.geometry(2, topo)
    d = cm.edge_descriptors(topo, partition(2, 2), 0)
    mine = d["elems"]
    elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine], spheremp=geo["spheremp"][mine],
                rspheremp=geo["rspheremp"][mine], putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
    h = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), 2, 1e19, device=0, schedule=dict(send=d["send"], recv=d["recv"]))
    h.dcmip_init(1, geo["lat"][mine], geo["lon"][mine], hv.hyam, hv.hybm)
    h.dcmip_set_initial()
    with pytest.raises(TseError, match="neither tse_comm_init"):
        h.prim_run_subcycle(1800.0, 1, 0)
    with pytest.raises(TseError, match=r"tse_comm_precheck: send peer 1 \(this is rank 0 of 1\)"):
        h.comm_init(HipMod.comm_unique_id(), 0, 1)     # peer rank 1 does not exist in a 1-rank communicator
    h.close()


def test_halo_exchange_on_rccl_self_peer():
    import torch
    import torch.distributed as dist
    from transport_se_amd.driver import HaloExchange
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nlyr, ncol = 7, 12
        send = torch.arange(ncol * nlyr, dtype=torch.float64, device="cuda:0") + 0.5
        recv = torch.zeros(ncol * nlyr, dtype=torch.float64, device="cuda:0")
        sched = dict(send=[(0, 1, 5), (0, 6, 7)], recv=[(0, 1, 5), (0, 6, 7)])   # two slots, both to/from rank 0
        ex = HaloExchange(sched, "cuda:0", dist, torch)
        # zero-copy wrap of a raw device pointer
        w = ex._wrap(send.data_ptr(), send.numel())
        assert w.data_ptr() == send.data_ptr() and torch.equal(w, send)
        torch.cuda.synchronize()
        assert ex(send.data_ptr(), recv.data_ptr(), nlyr) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv, send)
        # kind 1 (compact min/max exchange): other slot lengths over the same buffers
        ex.set_minmax_layout([2, 3], [2, 3])
        recv.zero_()
        assert ex(send.data_ptr(), recv.data_ptr(), nlyr, 1) == 0
        torch.cuda.synchronize()
        assert torch.equal(recv[:5 * nlyr], send[:5 * nlyr]) and not recv[5 * nlyr:].any()
    finally:
        dist.destroy_process_group()


_FALLBACK_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from transport_se_amd.driver import PrimRun
r = PrimRun(4, 2, rank=dist.get_rank(), world=dist.get_world_size(), device=0, dist_mod=dist, torch_mod=torch, exchange="rccl")
np1 = r.run(3)
cs = torch.tensor([r.state_checksum(np1, torch)], dtype=torch.int64)
dist.all_reduce(cs)
if dist.get_rank() == 0:
    print("RESULT " + json.dumps({"exchange": r.exchange_kind, "note": r.exchange_note, "checksum": int(cs.item()), "comm": r.hip.comm_info()}), flush=True)
r.close()
dist.destroy_process_group()
'''


def test_ranks_that_cannot_build_a_communicator_fall_back_together(tmp_path):
    """exchange="rccl" with two ranks on ONE device: ncclCommInitRank refuses the duplicate GPU on both.  The driver must not
    hang or leave half the ranks on RCCL: all ranks drop to the host-staged callback, say so, and the result has the bits of
    the single-rank run."""
    import json
    import subprocess
    import sys
    import torch
    from transport_se_amd import _lib
    from transport_se_amd.driver import PrimRun
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = PrimRun(4, 2, device=0, torch_mod=torch)
    ref = one.state_checksum(one.run(3), torch)
    one.close()
    w = tmp_path / "worker.py"
    w.write_text(_FALLBACK_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", str(w), root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out + r.stderr.decode()[-2000:]
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["exchange"] == "staged" and "RCCL communicator could not be initialised" in res["note"]
    assert res["comm"] == [0, 1]                      # no communicator left behind
    assert b"WARNING: RCCL communicator" in r.stderr
    assert res["checksum"] == ref


def test_one_rank_failing_its_precheck_takes_every_rank_off_rccl(tmp_path):
    """ncclCommInitRank is a blocking collective: a rank that cannot take part (bad peer table, no device, another librccl
    loaded ...) must be found BEFORE anyone enters it, or its peers wait in the bootstrap for ever.  Here rank 1 alone fails
    tse_comm_precheck (TSE_TEST_FAIL_PRECHECK_RANK); the ranks agree on that over the control plane, nobody calls
    tse_comm_init, all continue on the host-staged callback and the result has the bits of the single-rank run."""
    import json
    import subprocess
    import sys
    import torch
    from transport_se_amd import _lib
    from transport_se_amd.driver import PrimRun
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = PrimRun(4, 2, device=0, torch_mod=torch)
    ref = one.state_checksum(one.run(3), torch)
    one.close()
    w = tmp_path / "worker.py"
    w.write_text(_FALLBACK_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29635", str(w), root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, TSE_TEST_FAIL_PRECHECK_RANK="1", TSE_LIB=_lib.HOOKS_SO))   # (fault injection: the -DTSE_AB_HOOKS twin)
    out = r.stdout.decode()
    assert r.returncode == 0, out + r.stderr.decode()[-2000:]
    res = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["exchange"] == "staged" and "rank(s) 1 " in res["note"] and "fails on request" in res["note"], res["note"]
    assert res["comm"] == [0, 1] and res["checksum"] == ref


def test_rccl_version_is_reported_and_matches_the_headers():
    """tse_comm_version: the RCCL this process resolved (version + path of the shared object) against the headers the library was
    built with -- bench.py prints it; a major.minor mismatch is refused by tse_comm_precheck"""
    from transport_se_amd.hip_mod import HipMod
    v = HipMod.comm_version()
    assert v["path"].endswith(".so") or ".so." in v["path"], v
    assert v["runtime"].split(".")[0] == v["built"].split(".")[0], v      # same major version (tse_comm_precheck refuses anything else)
    import torch
    assert os.path.dirname(torch.__file__) in v["path"], v                # one RCCL per process: the copy torch loaded


def test_rank_rehearsal_tool_runs_one_rank_of_a_partition_in_loopback():
    """tools/rank_rehearsal.py (DESIGN section 4's per-rank timings): rank 1 of a 4-rank partition of ne8, alone, RCCL in loopback"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_rehearsal.py"), "--ne", "8", "--qsize", "4", "--world", "4", "--rank", "1",
                        "--cycles", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["world"] == 4 and d["rank"] == 1 and d["elements"] == 96 and 0 < d["boundary_elements"] < 96
    assert d["neighbour_ranks"] >= 1 and d["send_columns"] > 0 and d["ms_per_step"] > 0
    assert all(d["kernel_ms_per_step_timing_mode"][k] > 0 for k in ("advance0", "advance1", "advance2", "lap", "dss", "remap"))


def _rank_share_loopback(ne, qsize, world, rank, strips, cycles=1, env=None, lib_path=None):
    """rank `rank` of the `world`-rank partition alone on the GPU, every neighbour slot pointed at rank 0 of a ONE-rank RCCL communicator
    (tools/rank_rehearsal.py): the real split launches, pack / ncclSend / ncclRecv / unpack on the communication stream, prefetched
    bounds exchange -- with the rank-boundary patch tiling selected by TSE_BOUNDARY_STRIPS.  Returns (Qdp bits of both time levels
    as int64 tensors, (boundary patches, interior patches))."""
    import torch
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.driver import NU_Q, TSTEP, partition
    from transport_se_amd.hip_mod import HipMod
    from transport_se_amd.hybvcoord import HvCoord
    hv = HvCoord()
    topo = cm.topology(ne); geo = cm.geometry(ne, topo)
    d = cm.edge_descriptors(topo, partition(ne, world), rank)
    mine = d["elems"]
    sched = dict(send=[(0, p, l) for (_, p, l) in d["send"]], recv=[(0, p, l) for (_, p, l) in d["recv"]])
    elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine], spheremp=geo["spheremp"][mine],
                rspheremp=geo["rspheremp"][mine], putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
    os.environ["TSE_BOUNDARY_STRIPS"] = "1" if strips else "0"     # (read by tse_init)
    os.environ.update(env or {})
    try:
        h = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), qsize, NU_Q[ne], device=0, schedule=sched, lib_path=lib_path)
    finally:
        for k in ["TSE_BOUNDARY_STRIPS"] + list(env or {}):
            os.environ.pop(k, None)
    h.comm_init(HipMod.comm_unique_id(), 0, 1)
    assert h.comm_info() == (0, 1)
    h.dcmip_init(1, geo["lat"][mine], geo["lon"][mine], hv.hyam, hv.hybm)
    h.dcmip_set_initial()
    assert h.prim_run_subcycle(TSTEP[ne], cycles, 0) == 3 * cycles
    h.synchronize()
    out = []
    for tl in (1, 2):
        ptr, _ = h.device_ptr("qdp%d" % tl)
        iface = {"shape": (int(mine.size) * qsize * 72 * 16,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
        out.append(torch.as_tensor(type("DevArr", (), {"__cuda_array_interface__": iface})(), device="cuda:0").clone())
    layout = h.patch_layout()
    h.close()
    torch.cuda.empty_cache()
    return out, layout


def test_boundary_bands_under_rccl_split_launches_leave_the_bits_of_the_regular_tiling():
    """First-contact rehearsal for the 8-GPU run (BASELINE configs[3]) on the one GPU: rank 3 of the 8-rank partition of ne120 with 35
    tracers -- the per-rank shape of the headline multi-GPU configuration -- through the in-library RCCL exchange (c->comm set: every
    stage launched boundary-first, pack -> ncclSend/ncclRecv -> unpack on the communication stream under the interior launch, the next
    step's bounds exchange prefetched under the final DSS / the remap) once with the regular patch tiling and once with
    TSE_BOUNDARY_STRIPS=1 (rank-boundary elements in two-deep bands of their own: other patches, other launch split, other halo-ring
    tables).  The tiling is pure scheduling: the bits must not move.  (Loopback halo: the fields are not physical at the rank
    boundary, the comparison is between the two tilings.)"""
    import torch
    a, la = _rank_share_loopback(120, 35, 8, 3, strips=False)
    b, lb = _rank_share_loopback(120, 35, 8, 3, strips=True)
    assert la != lb and lb[0] < la[0], (la, lb)                     # the bands do shrink the first launch
    for x, y, tl in zip(a, b, (1, 2)):
        assert torch.equal(x, y), "time level %d: %d values differ between the two tilings" % (tl, int((x != y).sum()))
    assert int((a[0] != a[1]).sum()) > 0


def test_interior_launch_on_its_own_stream_leaves_the_bits_of_the_single_stream_order():
    """On several ranks the interior launch of every split stage runs on a stream of its own beside the boundary launch (split_stage,
    tse_api.hip); the -DTSE_AB_HOOKS library can switch that off (TSE_AB_SPLIT_STREAMS=0: both launches on the compute stream, the
    order up to round 4).  Pure scheduling: rank 3 of the 8-rank partition of ne120/q35 in RCCL loopback, two rsplit cycles (6 tracer
    steps, 2 fused remaps, prefetched bounds exchanges), both ways -- the bits must not move."""
    import torch
    from transport_se_amd import _lib
    a, _ = _rank_share_loopback(120, 35, 8, 3, strips=False, cycles=2, env={"TSE_AB_SPLIT_STREAMS": "1"}, lib_path=_lib.HOOKS_SO)
    b, _ = _rank_share_loopback(120, 35, 8, 3, strips=False, cycles=2, env={"TSE_AB_SPLIT_STREAMS": "0"}, lib_path=_lib.HOOKS_SO)
    for x, y, tl in zip(a, b, (1, 2)):
        assert torch.equal(x, y), "time level %d: %d values differ between the two launch orders" % (tl, int((x != y).sum()))
    assert int((a[0] != a[1]).sum()) > 0
