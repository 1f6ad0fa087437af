"""developer tool: limiter-8 iteration statistics of a real run (library built with -DTSE_LIMITER_STATS)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transport_se_amd import _lib
os.environ["TSE_LIB"] = os.path.join(ROOT, "tools", "ab", "stats.so")   # tools/ab_build.sh stats -DTSE_LIMITER_STATS
from transport_se_amd.driver import PrimRun
ne = int(sys.argv[1]) if len(sys.argv) > 1 else 30
run = PrimRun(ne, 35)
L = _lib.lib()
hist = (C.c_ulonglong * 20)()
run.run(3)
L.tse_debug_limiter_hist(hist, 1)
for nsteps in (3, 30):
    run.run(nsteps)
    L.tse_debug_limiter_hist(hist, 1)
    h = list(hist)
    tot = sum(h[:17])
    print("after %d more steps: slabs %d" % (nsteps, tot))
    print("  converged at iteration: " + " ".join("%d:%.4f" % (i, h[i] / tot) for i in range(1, 17) if h[i]))
    print("  mean wave iterations %.3f" % (h[17] / max(h[18], 1)))
