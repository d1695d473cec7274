import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l})
if order == "lib_first":
    from zstdlite_amd import _lib
    L = _lib.lib(); print("devcount(lib first)", L.zl_deviceCount()); print(maps())
    import torch; print("torch avail", torch.cuda.is_available()); torch.cuda.set_device(0)
else:
    import torch; print("torch avail", torch.cuda.is_available()); torch.cuda.set_device(0)
    from zstdlite_amd import _lib
    L = _lib.lib()
print(maps())
print("devcount", L.zl_deviceCount())
import zstdlite_amd as z
try:
    print(len(z.zstd_decompress(open("tests/golden/data.json.zst", "rb").read())))
except Exception as e:
    print("ERR", e)
x = torch.zeros(10, device="cuda"); print(x.sum().item())
