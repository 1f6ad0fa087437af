"""profiles/*_pmc_traffic_*.json from separate rocprofv3 --pmc passes (tools/pmc_passes.sh: FETCH_SIZE | WRITE_SIZE |
TCC_HIT_sum TCC_MISS_sum):   python tools/pmc_traffic.py gpurun_out/<dir> <ne> <qsize> <n_gpus> <out.json>

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of wide coalesced streaming reads (16 B per lane), so it is doubled for the kernels whose tracer reads are 16-B-per-lane
loads.  The DSS-on-read kernels and the remap read the tracers with 8-B-per-lane loads, which FETCH_SIZE counts in full
(checked against k_lap1<1>, whose raw count equals the field it must read), so their factor is 1 (their 16-B level-field
reads, <= 6 % of the bytes, are then under-counted)."""
import csv, glob, json, os, re, sys, collections

FETCH_FACTOR = {"k_advance<0,0>": 2, "k_advance<1,0>": 2, "k_advance<2,0>": 2, "k_advance<3,0>": 2, "k_lap1<0>": 2, "k_dss_t2<0>": 2,
                "k_dss_t2<1>": 2, "k_dss<0>": 2, "k_divdp": 2, "k_qminmax": 2, "k_advance<1,1>": 1, "k_advance<2,2>": 1, "k_lap1<1>": 1,
                "k_remap<1>": 1, "k_remap<2>": 1, "k_nbr_minmax": 1, "k_dcmip_step": 1, "k_dcmip_init": 1}


def short(n):
    m = re.match(r"(?:void )?tse::(k_\w+)(<[^>]*>)?", n)
    if not m:
        return None
    t = (m.group(2) or "").replace(" ", "").replace("true", "1").replace("false", "0")
    if m.group(1) == "k_advance" and t.count(",") == 2:
        t = t[:t.rindex(",")] + ">"           # drop the DB flag
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
    fac = FETCH_FACTOR.get(k, 1)
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * fac
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024
    e = {"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write, "fetch_size_factor": fac,
         "launches_sampled": len(c["FETCH_SIZE"])}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        h, m = sum(c["TCC_HIT_sum"]), sum(c["TCC_MISS_sum"])
        e["l2_hit_rate"] = h / (h + m) if h + m else None
    kern[k] = e
json.dump({"config": {"ne": ne, "nlev": 72, "qsize": qsize, "n_gpus": ngpu},
           "command": "tools/pmc_passes.sh: rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python3 bench.py --steps 3 --warmup 0 "
                      "--no-cpu-baseline, C in {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (separate passes)",
           "corrections": __doc__.split("Corrections", 1)[1].strip(), "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in kern.items()}))
