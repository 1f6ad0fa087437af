#!/usr/bin/env python3
"""What running the VALU-bound stage 3 of one half of the tracers beside the memory-bound kernels of the other half could buy
(DESIGN.md section 7): two contexts on ONE GPU, each with all elements and about half of the 35 tracers (tracers are independent),
driven from two threads on their own streams, against one context with all 35.  No code path of the library is changed: this only
prices the idea -- including its cost, the per-(element, level) prologue work done twice.
    python tools/overlap_probe.py [ne=120] [steps=12]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from transport_se_amd.driver import PrimRun  # noqa: E402

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 120
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.environ.setdefault("TSE_PLACEMENT", "0")   # (two contexts share the card: no placement trial)


def timed(runs, offset_steps=0):
    for r in runs:
        r.run(3); r.hip.synchronize()
    if offset_steps:                      # put the second pipeline out of phase with the first
        runs[1].run(offset_steps); runs[1].hip.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda r=r: (r.run(steps), r.hip.synchronize())) for r in runs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return 1e3 * (time.perf_counter() - t0) / steps


one = PrimRun(ne, 35, test_case=1, device=0, torch_mod=torch)
print("one context, 35 tracers: %.2f ms per step" % timed([one]))
one.close()
for split in ((18, 17), (24, 11)):
    a = PrimRun(ne, split[0], test_case=1, device=0, torch_mod=torch)
    b = PrimRun(ne, split[1], test_case=1, device=0, torch_mod=torch)
    ta = timed([a]); tb = timed([b])
    both = timed([a, b])
    print("two contexts, %d + %d tracers: alone %.2f + %.2f = %.2f ms per step; together on two streams %.2f ms per step" % (split[0], split[1], ta, tb, ta + tb, both))
    a.close(); b.close()
