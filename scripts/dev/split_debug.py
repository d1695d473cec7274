#!/usr/bin/env python3
"""Dev: which input hangs?  Every golden large frame (and a few live ones) in a process of its own, with a timeout."""
import subprocess, sys
if len(sys.argv) > 1:
    sys.path.insert(0, "."); sys.path.insert(0, "tests")
    import zstdlite_amd as z
    from zl_testlib import load_golden_large, GENERATORS, Ref
    k = int(sys.argv[1])
    if k < 3:
        c = load_golden_large()[k]
        out = z.zstd_decompress(c["comp"])
        print("golden", k, c["key"], len(c["comp"]), "ok" if out == c["plain"] else "MISMATCH", flush=True)
    else:
        n = [1200000, 2 << 20, 140000 * 9][k - 3]
        data = GENERATORS["text"](n, seed=k)
        comp = Ref().compress(data, level=3)
        print("live", k, n, "ok" if z.zstd_decompress(comp) == data else "MISMATCH", flush=True)
else:
    for k in range(6):
        try:
            r = subprocess.run([sys.executable, __file__, str(k)], capture_output=True, text=True, timeout=90)
            print(k, r.stdout.strip()[-200:], r.stderr.strip()[-300:])
        except subprocess.TimeoutExpired:
            print(k, "TIMEOUT")
