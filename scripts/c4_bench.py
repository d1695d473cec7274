#!/usr/bin/env python3
"""Config C4 alone (bench.py's c4_leg): 100k small objects with a trained dictionary through the C-ABI batch entries."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    out = bench.c4_leg(lambda m: print(m, file=sys.stderr), n)
    print(json.dumps(out, indent=1))
