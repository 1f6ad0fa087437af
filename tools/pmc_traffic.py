"""profiles/*_pmc_traffic_*.json from separate rocprofv3 --pmc passes (tools/pmc_passes.sh: FETCH_SIZE | WRITE_SIZE |
TCC_HIT_sum TCC_MISS_sum):   python tools/pmc_traffic.py gpurun_out/<dir> <ne> <qsize> <n_gpus> <out.json>

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of coalesced reads (128-B requests tallied at 64 B), so it is doubled.  Cross-check kept in the file: TCC_MISS_sum x 128 B
equals fetch (doubled) + write within a few per cent for every tracer kernel, e.g. k_advance<1,1>: 7.6e8 x 128 B = 97 GB
vs 63.8 + 34.0 GB -- with the raw FETCH_SIZE the misses would not even cover the field the kernel must read."""
import csv, glob, json, os, re, sys, collections

FETCH_FACTOR = 2


def short(n):
    m = re.match(r"(?:void )?tse::(k_\w+)(<[^>]*>)?", n)
    if not m:
        return None
    t = (m.group(2) or "").replace(" ", "").replace("true", "1").replace("false", "0")
    if m.group(1) == "k_advance" and t.count(",") >= 2:
        parts = t[1:-1].split(",")            # <RHS, GIN, DB[, PSZ]>: drop the DB flag, keep the block shape
        t = "<" + ",".join(parts[:2] + parts[3:]) + ">"
    return m.group(1) + t


d, ne, qsize, ngpu, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = k
    for (disp, ctr), v in per_dispatch.items():
        acc[names[disp]][ctr].append(v)
kern = {}
for k, c in sorted(acc.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fac = FETCH_FACTOR
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * fac
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    e = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write, "fetch_size_factor": fac,
         "launches_sampled": len(c["FETCH_SIZE"])}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        h, m = sum(c["TCC_HIT_sum"]), sum(c["TCC_MISS_sum"])
        e["l2_hit_rate"] = h / (h + m) if h + m else None
        e["tcc_miss_x128B_per_launch"] = m / len(c["TCC_MISS_sum"]) * 128
    kern[k] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_se_amd import _lib   # the same hash bench.py prints: ties the counters to a build
json.dump({"config": {"ne": ne, "nlev": 72, "qsize": qsize, "n_gpus": ngpu}, "kernel_source_hash": _lib.source_hash(),
           "command": "tools/pmc_passes.sh: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 3 --warmup 0 "
                      "--no-cpu-baseline, C in {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (separate passes)",
           "corrections": __doc__.split("Corrections", 1)[1].strip(), "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in kern.items()}))
