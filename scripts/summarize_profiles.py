"""Turns the rocprofv3 outputs of a bench.py run (gpurun_out/prof_bench, pmc_fetch, pmc_write) into the
small summaries committed under profiles/:  <round>_kernel_stats.csv (rocprofv3 --stats as is),
<round>_kernel_summary.json (per kernel, launches of the full 1 GiB grid only: average duration,
FETCH_SIZE / WRITE_SIZE per launch) and <round>_pmc_summary.json (what bench.py reads for
roofline.traffic).  Usage: python scripts/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern)), key=os.path.getmtime)
    return f[-1] if f else None  # newest run


stats = one("prof_bench/*/*kernel_stats.csv")
trace = one("prof_bench/*/*kernel_trace.csv")
if stats:
    shutil.copy(stats, os.path.join(P, f"{rnd}_kernel_stats.csv"))
n1 = os.path.join(G, "n1_kernel_stats.csv")
if os.path.exists(n1):  # the large-frame pipeline's kernels (scripts/n1_prof.py 1024: three decodes of the 1 GiB single frame)
    rows = [r for r in csv.DictReader(open(n1)) if "zl" in r["Name"]]
    with open(os.path.join(P, f"{rnd}_n1_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
        w.writeheader()
        w.writerows(rows)
line = os.path.join(G, "prof_bench_line.json")
if os.path.exists(line):
    shutil.copy(line, os.path.join(P, f"{rnd}_bench_under_rocprof.json"))

# full-grid launches: the largest grid each kernel was launched with (the 1 GiB steps; the smaller
# ones are the per-input 256 MiB calls and the C2 leg)
dur = collections.defaultdict(list)
if trace:
    rows = list(csv.DictReader(open(trace)))
    for r in rows:
        r["Grid_Size"] = r.get("Grid_Size") or r["Grid_Size_X"]
    big = collections.defaultdict(int)
    for r in rows:
        if ("zl_" in r["Kernel_Name"] or "zl2_" in r["Kernel_Name"]):
            big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"]
        if ("zl_" in k or "zl2_" in k) and int(r["Grid_Size"]) == big[k]:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)


def pmc(path):
    d = collections.defaultdict(list)
    if not path:
        return d
    rows = list(csv.DictReader(open(path)))
    big = collections.defaultdict(int)
    for r in rows:
        if ("zl_" in r["Kernel_Name"] or "zl2_" in r["Kernel_Name"]):
            big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        k = r["Kernel_Name"]
        if ("zl_" in k or "zl2_" in k) and int(r["Grid_Size"]) == big[k]:
            d[k].append(float(r["Counter_Value"]))
    return d


fetch = pmc(one("pmc_fetch/*/*counter_collection.csv"))
write = pmc(one("pmc_write/*/*counter_collection.csv"))
summary, short = {}, {}
names = {"zl_match_auto_kernel<false, false, true, true, false>": "match_windowed", "zl_match_auto_kernel": "match", "zl_match_auto_kernel<false, false, false, false, false>": "match", "zl_match_auto_kernel<false, false, true, true, true>": "match_windowed", "zl_match_auto_kernel<false, false, false, true, false>": "match_small", "zl_match_auto_kernel<false, false, false, false, true>": "match_level5", "zl_match_auto_kernel<false, false, false, false>": "match", "zl_match_auto_kernel<false, false, true, true>": "match_windowed", "zl_match_auto_kernel<false, false, false, true>": "match_small", "zl_far_build_kernel": "far_build", "zl_far_probe_kernel": "far_probe", "zl_match_auto_kernel<false, false>": "match", "zl_match_auto_kernel<false, true>": "match", "zl_match_auto_kernel<true, false>": "match_dict", "zl_entropy_kernel": "entropy", "zl_scan_sizes_kernel": "scan", "zl_gather_kernel": "gather",
         "zl_decode_kernel": "d_general", "zl2_index_kernel": "d_index", "zl2_entropy_kernel": "d_entropy", "zl2_seq_kernel": "d_entropy", "zl2_entropy1_kernel": "d_entropy1", "zl2_lit_kernel": "d_entropy_lit", "zl2_lit_kernel<8>": "d_entropy_lit", "zl2_lit_kernel<1>": "d_entropy_lit1", "zl2_order_kernel": "d_order", "zl2_exec_kernel": "d_exec", "zl2_exec_kernel<false>": "d_exec", "zl2_exec_kernel<true>": "d_exec_large", "zl2_exec_kernel<false, false>": "d_exec", "zl2_exec_kernel<false, true>": "d_exec", "zl2_exec_kernel<true, false>": "d_exec_large"}
for k in sorted(set(dur) | set(fetch) | set(write)):
    base = k.split("(")[0].replace("void ", "")
    f_kb = sum(fetch[k]) / len(fetch[k]) if fetch[k] else None
    w_kb = sum(write[k]) / len(write[k]) if write[k] else None
    e = {"full_grid_launches": len(dur[k]), "avg_ms": sum(dur[k]) / len(dur[k]) if dur[k] else None,
         "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb}
    if f_kb is not None and w_kb is not None:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950;
        # WRITE_SIZE is exact.  Both bounds are given; these kernels read in 1-16 byte pieces per lane.
        e["hbm_bytes_per_launch_raw"] = (f_kb + w_kb) * 1024
        e["hbm_bytes_per_launch_fetch_doubled"] = (2 * f_kb + w_kb) * 1024
        short[names.get(base, base)] = {"hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024, "kernel": base}
    summary[base] = e
# which kernels these numbers describe: bench.py reports roofline.traffic only while the sources are the same
import hashlib
h = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(ROOT, "zstdlite_amd", "csrc", "*.h*"))):  # the product kernels (csrc/experiments/ is not part of the library)
    h.update(open(f, "rb").read())
short["_kernel_sources_sha256"] = h.hexdigest()
json.dump(summary, open(os.path.join(P, f"{rnd}_kernel_summary.json"), "w"), indent=1)
json.dump(short, open(os.path.join(P, f"{rnd}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))

# ---- the DEFAULT (one-frame) layout: scripts/one_frame_prof.py under the profiler (kernel stats, FETCH_SIZE / WRITE_SIZE passes)
of_stats = one("prof_one_frame/*/*kernel_stats.csv")
if of_stats:
    rows = [r for r in csv.DictReader(open(of_stats)) if "zl" in r["Name"]]
    with open(os.path.join(P, f"{rnd}_one_frame_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
        w.writeheader()
        w.writerows(rows)
    of_fetch = pmc(one("pmc_of_fetch/*/*counter_collection.csv"))
    of_write = pmc(one("pmc_of_write/*/*counter_collection.csv"))
    avg = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e6 for r in rows}
    of = {}
    for k in sorted(set(of_fetch) | set(of_write)):
        base = k.split("(")[0].replace("void ", "")
        f_kb = sum(of_fetch[k]) / len(of_fetch[k]) if of_fetch.get(k) else None
        w_kb = sum(of_write[k]) / len(of_write[k]) if of_write.get(k) else None
        of[base] = {"avg_ms_all_launches": avg.get(base), "FETCH_SIZE_KB_per_full_launch": f_kb, "WRITE_SIZE_KB_per_full_launch": w_kb,
                    "hbm_bytes_per_launch_raw": (f_kb + w_kb) * 1024 if f_kb is not None and w_kb is not None else None,
                    "hbm_bytes_per_launch_fetch_doubled": (2 * f_kb + w_kb) * 1024 if f_kb is not None and w_kb is not None else None}
    of["_note"] = "compress + decompress of the 1 GiB bench mix in the one-frame layout, device-resident (scripts/one_frame_prof.py); FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes, largest-grid launches only; MI355X_MICROARCH.md: FETCH_SIZE halves wide streaming reads on gfx950 (both bounds given)"
    json.dump(of, open(os.path.join(P, f"{rnd}_one_frame_pmc.json"), "w"), indent=1)

# ---- L2 and fabric counters of the headline command (what the match kernel's candidate gathers do to the memory system)
l2csv = one("pmc_l2/*/*counter_collection.csv")
if l2csv:
    rows = list(csv.DictReader(open(l2csv)))
    for r in rows:
        r["Grid_Size"] = r.get("Grid_Size") or r.get("Grid_Size_X") or "0"
    big = collections.defaultdict(int)
    for r in rows:
        if "zl" in r["Kernel_Name"]:
            big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        if "zl" in k and int(r["Grid_Size"]) == big[k]:
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    l2 = {}
    for k, cs in acc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        l2[k] = e
    l2["_note"] = ("per full-grid launch of the headline command.  TCC_EA0_RDREQ counts the L2's read requests to the fabric; whether the fabric serves them from the "
                   "Infinity Cache or from HBM is decided behind the TCC block and is not visible to these counters (MI355X_MICROARCH.md: Infinity-Cache hits appear to be counted, not excluded)")
    json.dump(l2, open(os.path.join(P, f"{rnd}_l2_counters.json"), "w"), indent=1)

# ---- the deep match finder of levels >= 6 (scripts/dev/deep_prof.py under rocprofv3 --kernel-trace --stats): its kernels and the radix sort's
for name in ("deep9_kernel_stats.csv", "deep12_text_kernel_stats.csv", "opt16_kernel_stats.csv", "c4_kernel_stats.csv"):
    src = os.path.join(G, name)
    if os.path.exists(src):
        rows = [r for r in csv.DictReader(open(src)) if "zl" in r["Name"] or "radix" in r["Name"]]
        for r in rows:
            if "radix" in r["Name"]:
                r["Name"] = "rocprim radix sort: " + ("onesweep" if "onesweep" in r["Name"] else "") + " " + r["Name"][-60:]
        with open(os.path.join(P, f"{rnd}_{name}"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
            w.writeheader()
            w.writerows(rows)
