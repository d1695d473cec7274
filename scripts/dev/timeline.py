"""Dev: a rocprofv3 --kernel-trace csv as a timeline: the dispatches behind the LAST dispatch of a marker kernel (default zl_gather_kernel:
the end of the last compression), grouped by kernel and queue: first start, last end, busy time, count.
usage: python scripts/dev/timeline.py <dir with *_kernel_trace.csv> [marker substring] [--all]"""
import csv, glob, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "zl_gather_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Queue_Id", "?")))
rows.sort()
last = max((i for i, r in enumerate(rows) if marker in r[2]), default=-1)
tail = rows[last + 1:]
if not tail: sys.exit("nothing behind the marker")
t0 = tail[0][0]
print(f"{len(tail)} dispatches behind the last {marker}; span {(max(r[1] for r in tail) - t0) / 1e6:.3f} ms")
groups = {}
for s, e, name, q in tail:
    g = groups.setdefault((name, q), [s, e, 0, 0])
    g[0] = min(g[0], s); g[1] = max(g[1], e); g[2] += e - s; g[3] += 1
for (name, q), (s, e, busy, cnt) in sorted(groups.items(), key=lambda kv: kv[1][0]):
    print(f"{(s - t0) / 1e6:8.3f} .. {(e - t0) / 1e6:8.3f} ms  busy {busy / 1e6:7.3f} ms  x{cnt:<4d} q{q}  {name}")
if "--all" in sys.argv:
    for s, e, name, q in tail:
        print(f"   {(s - t0) / 1e6:8.3f} +{(e - s) / 1e3:8.1f} us q{q} {name}")
