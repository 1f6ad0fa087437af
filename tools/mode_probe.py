#!/usr/bin/env python3
"""Run-to-run "modes" of the tracer kernels: N contexts created one after the other in ONE process, the same 6 timed
steps in each; prints the per-kernel ms per step and the device addresses of the big arrays.  Developer tool.
    python tools/mode_probe.py [N] [extra env assignments applied before every context, e.g. TSE_QDP_SPLIT=1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_se_amd.driver import PrimRun  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1); os.environ[k] = v
import torch  # noqa: E402
shifts = [int(x) for x in os.environ.get("TSE_PROBE_SHIFTS_MB", "").split(",") if x]
for i in range(n):
    pad = None
    if shifts:   # a dummy allocation in front of the context's: moves every array of the context to other physical pages
        mb = shifts[i % len(shifts)]
        pad = torch.empty(mb << 20, dtype=torch.uint8, device="cuda:0") if mb else None
        print("shift %d MB" % mb, flush=True)
    run = PrimRun(120, 35, test_case=1, device=0, torch_mod=torch)
    run.run(3)
    run.hip.synchronize(); run.hip.timing(True)
    t0 = time.perf_counter()
    run.run(6)
    run.hip.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 6
    kt = {k: round(run.hip.kernel_time(k)[0] / 6, 2) for k in ("advance0", "advance1", "advance2", "lap", "dss", "minmax", "remap")}
    ptr = {k: hex(run.hip.device_ptr(k)[0] or 0) for k in ("qdp1", "qdp2", "T", "B", "C", "qmin")}
    print(json.dumps({"ctx": i, "ms_per_step": round(ms, 2), "kernels": kt, "ptr": ptr}), flush=True)
    run.close()
    del run, pad
    torch.cuda.empty_cache()
