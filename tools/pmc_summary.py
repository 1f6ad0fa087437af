"""per-kernel means of rocprofv3 --pmc csv passes: python tools/pmc_summary.py gpurun_out/<dir>  (prints counter per launch)"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
def short(n):
    m = re.match(r"void tse::(k_\w+)(<[^>]*>)?", n) or re.match(r"tse::(k_\w+)(<[^>]*>)?", n)
    if m: return m.group(1) + (m.group(2) or "")
    n = re.sub(r"^_ZN3tse\d+", "", n)
    m = re.match(r"(k_[a-z_0-9]+?)(I.*?E)?Ev", n)
    if m:
        t = re.findall(r"L[ib](\d+)E", m.group(2) or "")
        return m.group(1) + ("<" + ",".join(t) + ">" if t else "")
    return n[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "p*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
    if not k.startswith("k_"): continue
    # a launch appears once per counter (and per dimension instance: sum instances of the same dispatch)
    print(k)
    for c in names:
        v = acc[k].get(c)
        if v: print("   %-32s n=%-4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
