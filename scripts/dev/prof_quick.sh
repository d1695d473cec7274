#!/bin/bash
# kernel stats of a short bench run under rocprofv3 (averages of the kernels named in $1, a grep pattern)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; rm -rf /tmp/pq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-c2 --no-extra > /tmp/pq_line.json 2> /tmp/pq_err.log
python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pq/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in sys.argv[1].split('|')):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
