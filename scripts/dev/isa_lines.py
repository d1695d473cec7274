#!/usr/bin/env python3
"""Dev: static instruction counts per source line of one kernel, from an assembly listing compiled with
-gline-tables-only (hipcc -S --cuda-device-only).  usage: isa_lines.py listing.s mangled_kernel_name [file_no=0] [bucket]"""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
want_file = int(sys.argv[3]) if len(sys.argv) > 3 else 0
on = False
cur = (None, None)
valu = collections.Counter(); salu = collections.Counter(); lds = collections.Counter(); mem = collections.Counter(); tot = collections.Counter()
for line in open(path):
    if line.startswith(name + ":"):
        on = True
        continue
    if not on:
        continue
    s = line.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    key = cur
    tot[key] += 1
    if op.startswith("v_"): valu[key] += 1
    elif op.startswith("s_"): salu[key] += 1
    elif op.startswith("ds_"): lds[key] += 1
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): mem[key] += 1
    if op == "s_endpgm":
        break
print("total", sum(tot.values()), "valu", sum(valu.values()), "salu", sum(salu.values()), "lds", sum(lds.values()), "mem", sum(mem.values()))
rows = sorted(tot.items(), key=lambda kv: (kv[0][0], kv[0][1]))
for (f, l), n in rows:
    if n >= (int(sys.argv[4]) if len(sys.argv) > 4 else 8):
        print(f"file {f} line {l:5d}: {n:4d}  valu {valu[(f,l)]:4d} salu {salu[(f,l)]:4d} lds {lds[(f,l)]:3d} mem {mem[(f,l)]:3d}")
