#!/usr/bin/env python3
"""developer experiment: per-STEP kernel times (the tracer state's two time levels swap roles every step) next to the streaming-write
rates of the two levels.   python tools/step_probe.py [steps=8] [ne=120] [qsize=35]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_se_amd import _lib  # noqa: E402
os.environ.setdefault("TSE_LIB", _lib.HOOKS_SO)   # the tse_debug_* entry points exist only in the -DTSE_AB_HOOKS build
from transport_se_amd.driver import PrimRun  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 120
qs = int(sys.argv[3]) if len(sys.argv) > 3 else 35
run = PrimRun(ne, qs, test_case=1, device=0, torch_mod=torch)
L, h = run.hip.L, run.hip.h
run.hip._chk(L.tse_debug_scratch_pool(h, C.c_int(3)))   # (T, B, C as placed: probe indices 0..2)


def probe(src, dst):
    g = C.c_double()
    run.hip._chk(L.tse_debug_probe(h, C.c_int(src), C.c_int(dst), C.byref(g)))
    return round(g.value)


print(json.dumps({"write_q1": probe(-1, 100), "write_q2": probe(-1, 101), "read_q1": probe(100, -1), "read_q2": probe(101, -1),
                  "write_TBC": [probe(-1, i) for i in range(3)], "placement": run.hip.placement()}), flush=True)
run.hip.dcmip_set_initial()
run.nstep = 0
run.run(3)
for s in range(n):
    run.hip.synchronize(); run.hip.timing(True)
    run.run(1)
    run.hip.synchronize()
    kt = {k: round(run.hip.kernel_time(k)[0], 2) for k in ("advance0", "advance1", "advance2", "lap", "dss", "remap")}
    print(json.dumps({"step": 3 + s, "kernels": kt}), flush=True)
run.close()
