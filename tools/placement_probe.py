#!/usr/bin/env python3
"""Where the scratch fields live (developer experiment): K scratch-sized allocations in one context; streaming probes of every chunk
and pair, then the real kernels under every assignment of three of the chunks to the roles T, B, C.
    python tools/placement_probe.py [K=6] [ne=120] [qsize=35]"""
import ctypes as C
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TSE_PLACEMENT", "0")   # the experiment assigns the roles itself
from transport_se_amd import _lib  # noqa: E402
os.environ.setdefault("TSE_LIB", _lib.HOOKS_SO)   # the tse_debug_* entry points exist only in the -DTSE_AB_HOOKS build
from transport_se_amd.driver import PrimRun  # noqa: E402
import torch  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 120
q = int(sys.argv[3]) if len(sys.argv) > 3 else 35
run = PrimRun(ne, q, test_case=1, device=0, torch_mod=torch)
L, h = run.hip.L, run.hip.h
run.hip._chk(L.tse_debug_scratch_pool(h, C.c_int(K)))


def probe(src, dst):
    g = C.c_double()
    run.hip._chk(L.tse_debug_probe(h, C.c_int(src), C.c_int(dst), C.byref(g)))
    return round(g.value)


wr = [probe(-1, i) for i in range(K)]
rd = [probe(i, -1) for i in range(K)]
print(json.dumps({"write_GBs": wr, "read_GBs": rd}), flush=True)
pair = [[probe(i, j) if i != j else 0 for j in range(K)] for i in range(K)]
print(json.dumps({"copy_GBs[src][dst]": pair}), flush=True)
qd = {"q1_to": [probe(100, j) for j in range(K)], "q2_to": [probe(101, j) for j in range(K)], "to_q1": [probe(i, 100) for i in range(K)],
      "to_q2": [probe(i, 101) for i in range(K)], "q1_to_q2": probe(100, 101), "q2_to_q1": probe(101, 100)}
print(json.dumps(qd), flush=True)
run.hip._chk(L.tse_debug_zero_pool(h))
run.hip.dcmip_set_initial()
run.nstep = 0
res = []
for a in itertools.permutations(range(K), 3):
    run.hip._chk(L.tse_debug_assign_scratch(h, *[C.c_int(x) for x in a]))
    run.run(3)
    run.hip.synchronize(); run.hip.timing(True)
    run.run(6)
    run.hip.synchronize()
    kt = {k: round(run.hip.kernel_time(k)[0] / 6, 2) for k in ("advance0", "advance1", "advance2", "lap", "dss")}
    run.hip.timing(False)
    tot = round(sum(kt.values()), 2)
    res.append((tot, a, kt))
    print(json.dumps({"T,B,C": a, "sum": tot, "kernels": kt}), flush=True)
res.sort()
print("best", res[:5])
print("worst", res[-3:])
print("default (0,1,2)", [r for r in res if r[1] == (0, 1, 2)])
run.close()
