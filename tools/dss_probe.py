#!/usr/bin/env python3
"""developer experiment: which property of a chunk makes k_dss_patch slow when it is the chunk written?  K pool chunks; T, B, C = chunks
0, 1, 2; for every other chunk i: Qdp(2) := chunk i (Qdp(1) another), the streaming-write rate of the chunk and the per-step kernel
times (k_dss_patch writes Qdp(2) on every second step).   python tools/dss_probe.py [K=7]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TSE_PLACEMENT", "0")   # the experiment assigns the roles itself
from transport_se_amd import _lib  # noqa: E402
os.environ.setdefault("TSE_LIB", _lib.HOOKS_SO)   # the tse_debug_* entry points exist only in the -DTSE_AB_HOOKS build
from transport_se_amd.driver import PrimRun  # noqa: E402
import torch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 7
run = PrimRun(120, 35, test_case=1, device=0, torch_mod=torch)
L, h = run.hip.L, run.hip.h
run.hip._chk(L.tse_debug_scratch_pool(h, C.c_int(K)))


def probe(src, dst):
    g = C.c_double()
    run.hip._chk(L.tse_debug_probe(h, C.c_int(src), C.c_int(dst), C.byref(g)))
    return round(g.value)


print(json.dumps({"write_GBs": [probe(-1, i) for i in range(K)], "copy_from_C_GBs": [probe(2, i) if i != 2 else 0 for i in range(K)]}), flush=True)
for i in range(3, K):
    j = 3 if i != 3 else 4
    run.hip._chk(L.tse_debug_assign_qdp(h, C.c_int(j), C.c_int(i)))
    run.hip._chk(L.tse_debug_zero_pool(h))
    run.hip.dcmip_set_initial()
    run.nstep = 0
    run.run(2)
    t = {1: [], 2: []}
    for s in range(4):
        run.hip.synchronize(); run.hip.timing(True)
        np1 = run.run(1)
        run.hip.synchronize()
        t[np1].append(round(run.hip.kernel_time("dss")[0], 2))
    run.hip.timing(False)
    print(json.dumps({"Qdp2": i, "Qdp1": j, "dss_writing_Qdp2": t[2], "dss_writing_Qdp1": t[1]}), flush=True)
run.close()
