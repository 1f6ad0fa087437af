"""kernel timeline of a rocprofv3 --kernel-trace database: python tools/ktimeline.py <results.db> [first-dispatch] [count]
(runtime fill/copy kernels left out; a negative first-dispatch counts from the end)
prints start (ms, relative), duration (ms), queue/stream id and kernel name in start order -- to read stream overlap by eye"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(db.execute(f"select d.start, d.end, d.{qcol}, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
rows = [r for r in rows if "rocclr" not in r[3]]
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if first < 0: first = max(0, len(rows) + first)   # negative: counted from the end
t0 = rows[first][0]
for a, b, q, n in rows[first:first + count]:
    n = re.sub(r"^_ZN3tse\d*", "", n)[:40]
    print("%9.3f %8.3f  q%-3s %s" % ((a - t0) / 1e6, (b - a) / 1e6, q, n))
