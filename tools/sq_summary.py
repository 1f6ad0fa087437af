"""profiles/*_sq_counters_*.json from rocprofv3 --pmc passes with SQ counters (tools/pmc_passes.sh):
    python tools/sq_summary.py gpurun_out/<dir> <ne> <qsize> <n_gpus> <out.json>
Per kernel: waves, instructions per wave by class, and how the wave cycles split (SQ_WAIT_ANY = parked at s_waitcnt/barrier,
SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_VALU = a VALU instruction in flight; MI355X_MICROARCH.md "rocprofv3 PMC slots")."""
import collections, csv, glob, json, os, re, sys

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
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = k
    for (disp, ctr), v in per.items():
        acc[names[disp]][ctr].append(v)
res = {}
for k, c in sorted(acc.items()):
    mean = {n: sum(v) / len(v) for n, v in c.items()}
    w = mean.get("SQ_WAVES")
    if not w:
        continue
    e = {"waves": w}
    for n, key in (("SQ_INSTS_VALU", "valu_insts_per_wave"), ("SQ_INSTS_SALU", "salu_insts_per_wave"), ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave"),
                   ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave"), ("SQ_INSTS_LDS", "lds_insts_per_wave")):
        if n in mean:
            e[key] = mean[n] / w
    wc = mean.get("SQ_WAVE_CYCLES")
    if wc:
        for n, key in (("SQ_WAIT_ANY", "wait_any_frac_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac_of_wave_cycles"),
                       ("SQ_ACTIVE_INST_VALU", "active_inst_valu_frac_of_wave_cycles"), ("SQ_ACTIVE_INST_ANY", "active_inst_any_frac_of_wave_cycles")):
            if n in mean:
                e[key] = mean[n] / wc
    res[k] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transport_se_amd import _lib   # the same hash bench.py prints: ties the counters to a build
json.dump({"config": {"ne": ne, "qsize": qsize, "n_gpus": ngpu}, "kernel_source_hash": _lib.source_hash(),
           "command": "tools/pmc_passes.sh (separate rocprofv3 --pmc passes over bench.py --steps 3 --warmup 0): "
                      "{SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS} | "
                      "{SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY}", "kernels": res}, open(out, "w"), indent=1)
for k, e in res.items():
    print(k, {a: (round(b, 3) if b < 100 else round(b)) for a, b in e.items()})
