#!/usr/bin/env python3
"""One rank's share of a W-rank run, alone on one GPU: what a rank costs apart from the links.

Builds rank R of the W-rank space-filling-curve partition exactly as bench.py does, but points every neighbour slot at
rank 0 of a ONE-rank RCCL communicator (the loopback of tests/test_gpu_rccl_exchange.py): every slot the rank sends comes
back as the slot it receives.  The fields are therefore wrong at the rank boundary (nothing here is a parity check), but
the work is the real thing: boundary-first split launches, pack / unpack, ncclSend/ncclRecv of the real message sizes on the
communication stream (device-local copies instead of xGMI transfers), event hand-over, interior overlap.  ms/step here against
(1-GPU ms/step) / W is the part of the strong-scaling loss that does not depend on link bandwidth.

    python tools/rank_rehearsal.py --ne 120 --qsize 35 --world 8 [--rank 3] [--cycles 2]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ne", type=int, default=120); ap.add_argument("--qsize", type=int, default=35)
    ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--cycles", type=int, default=2)
    ap.add_argument("--verify", action="store_true", help="also compare one DSS-on-read step with the one-DSS-pass-per-stage route on the device")
    a = ap.parse_args()
    from transport_se_amd import cube_mesh as cm
    from transport_se_amd.driver import NU_Q, TSTEP, partition
    from transport_se_amd.hip_mod import HipMod
    from transport_se_amd.hybvcoord import HvCoord
    hv = HvCoord()
    topo = cm.topology(a.ne); geo = cm.geometry(a.ne, topo)
    owner = partition(a.ne, a.world)
    ranks = range(a.world) if a.rank < 0 else [a.rank]
    nu_q = NU_Q.get(a.ne, 1e15 * (30.0 / a.ne) ** 3.2); tstep = TSTEP.get(a.ne, 300.0 * 30.0 / a.ne)
    for rank in ranks:
        d = cm.edge_descriptors(topo, owner, rank)
        mine = d["elems"]
        sched = dict(send=[(0, p, l) for (_, p, l) in d["send"]], recv=[(0, p, l) for (_, p, l) in d["recv"]]) if a.world > 1 else None
        elem = dict(Dinv=geo["Dinv"][mine], metdet=geo["metdet"][mine], rmetdet=geo["rmetdet"][mine], spheremp=geo["spheremp"][mine],
                    rspheremp=geo["rspheremp"][mine], putmapP=d["putmapP"], getmapP=d["getmapP"], reverse=d["reverse"])
        h = HipMod(elem, cm.dvv(), (hv.hyai, hv.hybi, hv.ps0), a.qsize, nu_q, device=0, schedule=sched)
        if a.world > 1:
            h.comm_init(HipMod.comm_unique_id(), 0, 1)
        h.dcmip_init(1, geo["lat"][mine], geo["lon"][mine], hv.hyam, hv.hybm)
        h.dcmip_set_initial()
        n = h.prim_run_subcycle(tstep, 1, 0)          # warm-up: one rsplit cycle
        h.synchronize()
        t0 = time.perf_counter()
        n = h.prim_run_subcycle(tstep, a.cycles, n)
        h.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / (3 * a.cycles)
        h.timing(True)
        n = h.prim_run_subcycle(tstep, 1, n)
        h.synchronize()
        kt = {k: round(h.kernel_time(k)[0] / 3, 3) for k in ("advance0", "advance1", "advance2", "lap", "dss", "minmax", "remap", "level", "dcmip")}
        h.timing(False)
        nb, ni = h.boundary_layout()
        pb, pi = h.patch_layout()
        ncol = sum(s[2] for s in d["send"])
        out = {"rank": rank, "world": a.world, "ne": a.ne, "qsize": a.qsize, "elements": int(mine.size), "boundary_elements": nb,
               "boundary_patches": pb, "interior_patches": pi, "first_launch_fraction_of_patches": round(pb / max(pb + pi, 1), 4),
               "neighbour_ranks": len(set(s[0] for s in d["send"])), "send_columns": int(ncol),
               "main_message_MB": round(ncol * (a.qsize + 1) * 72 * 8 / 1e6, 1), "ms_per_step": round(ms, 3), "kernel_ms_per_step_timing_mode": kt}
        import torch   # (already in the process: _lib loads it first) -- device memory in use with every field of the rank allocated
        free, total = torch.cuda.mem_get_info(0)
        out["device_memory_used_GB"] = round((total - free) / 1e9, 2)
        if a.verify:
            # one tracer step from the initial state, DSS on read against one DSS pass per stage (same loopback halo in both):
            # the two routes must agree to the step tolerance at this shape, too
            def dev(name, shape):
                ptr, _ = h.device_ptr(name)
                iface = {"shape": shape, "typestr": "<f8", "data": (int(ptr), False), "version": 2}
                return torch.as_tensor(type("DevArr", (), {"__cuda_array_interface__": iface})(), device="cuda:0")
            Q = [dev("qdp%d" % tl, (int(mine.size), a.qsize, 72, 16)) for tl in (1, 2)]
            res, launches = [], []
            for env in ("1", "0"):
                os.environ["TSE_DSS_ON_READ"] = env
                h.dcmip_set_initial(); h.dcmip_step_inputs(0, tstep)
                h.timing(True)
                h.advec_tracers_remap_rk2(tstep, 1, 2); h.synchronize()
                launches.append(h.kernel_time("dss")[1])      # 1 tracer DSS pass with DSS on read, 4 with one pass per stage
                h.timing(False)
                res.append(Q[1].clone())
            os.environ.pop("TSE_DSS_ON_READ")
            out["dss_passes_of_the_two_routes"] = launches
            out["dss_on_read_vs_per_stage_rel"] = float((res[0] - res[1]).abs().max() / res[0].abs().max())
            out["finite"] = bool(torch.isfinite(res[0]).all())
            del res
        print(json.dumps(out), flush=True)
        h.close()


if __name__ == "__main__":
    main()
