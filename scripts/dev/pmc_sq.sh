#!/bin/bash
# Dev: SQ counters of the encoder kernels on one input kind (two passes of 8 counters).  usage: pmc_sq.sh [kind] [MiB] [layout]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sq1 gpurun_out/sq2
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/sq1 -- python3 scripts/dev/pmc_run.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM -d gpurun_out/sq2 -- python3 scripts/dev/pmc_run.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/sq1", "gpurun_out/sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); cnt[k] += 1
        for k, v in acc.items():
            if "zl_" in k:
                print(k, "dispatches", cnt[k])
                for c, x in sorted(v.items()): print(f"    {c:24s} {x / cnt[k]:16.0f}")
PY
