"""per-kernel summary of a rocprofv3 --kernel-trace result database (rocpd sqlite): python tools/kstats.py <results.db> [csv-out]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, sum(d.end-d.start)/1e6, min(d.end-d.start)/1e6, max(d.end-d.start)/1e6 "
                       f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"))
def short(n):
    n = re.sub(r"^_ZN3tse\d+", "", n); n = re.sub(r"\.kd$", "", n)
    m = re.match(r"(k_[a-z_0-9]+?)(I.*?E)?Ev", n)
    if m:
        t = re.findall(r"L[ib](\d+)E", m.group(2) or "")
        if m.group(1) == "k_advance":
            t = t[:2] + t[3:]  # <RHS, GIN[, PSZ]>; the third flag (double-buffered stores) follows from GIN
        return m.group(1) + ("<" + ",".join(t) + ">" if t else "")
    return n[:60]
tot = sum(r[3] for r in rows)
out = ["kernel,calls,avg_ms,total_ms,min_ms,max_ms,percent"]
for r in rows:
    out.append("%s,%d,%.4f,%.3f,%.4f,%.4f,%.2f" % (short(r[0]), r[1], r[2], r[3], r[4], r[5], 100 * r[3] / tot))
print("\n".join(out))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
