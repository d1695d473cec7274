"""End-to-end timings of the host entries (what the reference's R glue calls: zl_compress2 / zl_decompressDCtx with
host buffers), next to the PCIe rate measured on this box.  usage: python scripts/e2e_bench.py [MiB] [layout]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from zstdlite_amd import _lib
from zstdlite_amd.device import gen_mixed

def measure(n, layout="auto", pipeline=1, reps=3):
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    host = gen_mixed(n, dev)[0].cpu().numpy()
    cap = L.zl_compressBound(n)
    comp = np.empty(cap, dtype=np.uint8); comp.fill(1)
    back = np.empty(n, dtype=np.uint8); back.fill(1)
    cctx = L.zl_createCCtx(); dctx = L.zl_createDCtx()
    L.zl_CCtx_setParameter(cctx, 9003, {"auto": 0, "frames": 1, "one_frame": 2}[layout])
    L.zl_CCtx_setParameter(cctx, 9004, pipeline)
    L.zl_DCtx_setParameter(dctx, 9005, pipeline)
    tc = td = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = L.zl_compress2(cctx, comp.ctypes.data, cap, host.ctypes.data, n); t1 = time.perf_counter()
        assert not _lib.is_error(r), _lib.error_name(r)
        r2 = L.zl_decompressDCtx(dctx, back.ctypes.data, n, comp.ctypes.data, r); t2 = time.perf_counter()
        assert r2 == n, _lib.error_name(r2)
        tc = min(tc, t1 - t0); td = min(td, t2 - t1)
    assert np.array_equal(back, host)
    # PCIe: pageable host <-> device, as the library's own copies
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    th = tdn = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(torch.from_numpy(host)); torch.cuda.synchronize(); th = min(th, time.perf_counter() - t0)
        torch.cuda.synchronize(); t0 = time.perf_counter(); back2 = d.cpu(); torch.cuda.synchronize(); tdn = min(tdn, time.perf_counter() - t0)
    L.zl_freeCCtx(cctx); L.zl_freeDCtx(dctx)
    return {"bytes": n, "layout": layout, "pipeline": pipeline, "compressed": int(r), "compress_GBps": n / tc / 1e9, "decompress_GBps": n / td / 1e9,
            "pcie_h2d_GBps": n / th / 1e9, "pcie_d2h_GBps": n / tdn / 1e9, "compress_ms": tc * 1e3, "decompress_ms": td * 1e3}

if __name__ == "__main__":
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for layout in (sys.argv[2:] or ["auto", "frames"]):
        for pipe in (0, 1):
            print(measure(mib << 20, layout, pipe), flush=True)
