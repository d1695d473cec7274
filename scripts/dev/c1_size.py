import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import zstdlite_amd as z
buf = open("tests/golden/c1_mtcars_head.bin", "rb").read()
print(len(buf), len(z.zstd_compress(buf)), len(z.zstd_compress(buf, level=5)))
