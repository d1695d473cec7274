"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter value per launch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    n = max(1, len(cnt[k]))
    print(k, "launches", n)
    for c in sorted(acc[k]):
        print(f"    {c:28s} {acc[k][c] / n:16.0f}")
