"""Dev: per-kernel totals out of a rocprofv3 results database (kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
for r in cur.execute(q):
    print(f"{r[0][:90]:90s} {r[1]:6d} {r[2]:10.1f} us {r[3]:9.2f} ms")
