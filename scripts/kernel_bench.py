"""Per-input kernel timings (HIP events inside the library): the optimisation loop's tool."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from zstdlite_amd.device import DeviceCodec, gen_device, gen_mixed
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (256 << 20)
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["rand", "iid10", "text", "int20"]
import os
codec = DeviceCodec(profile=int(os.environ.get("ZL_PROF", "2")), frame_layout=os.environ.get("ZL_LAYOUT", "frames"))  # ZL_LAYOUT=one_frame: the windowed layout's kernels
for kind in kinds:
    src = gen_mixed(n, dev)[0] if kind == "mix" else gen_device(kind, n, dev)
    comp = torch.empty(codec.bound(n), dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8, device=dev)
    best = {}
    for it in range(4):
        _, total, fs = codec.compress(src, comp)
        one = os.environ.get("ZL_LAYOUT") == "one_frame"
        tbl = codec.frame_table([total], [n]) if one else codec.frame_table(fs, codec.chunk_content_sizes(n))
        codec.decompress(comp, tbl, back, chunk_sizes=fs if one else None)
        for k, v in codec.kernel_ms().items():
            best[k] = min(best.get(k, 1e9), v)
    assert torch.equal(back, src)
    gb = n / 1e9
    import ctypes as C
    st = (C.c_ulonglong * 16)()
    if hasattr(codec.L, "zl_dctx_phase_stats") and codec.L.zl_dctx_phase_stats(C.c_void_p(codec.dctx), st) == 0 and st[5]:
        nb = st[5]
        print(f"        decode phases, kcycles/block: H {st[0]/nb/1e3:.0f}  B {st[1]/nb/1e3:.0f}  A {st[2]/nb/1e3:.0f}  A||B {st[3]/nb/1e3:.0f}  C {st[4]/nb/1e3:.0f}   seq/block {st[6]/nb:.0f}  huflits/block {st[7]/nb:.0f}")
    ct = (C.c_ulonglong * 16)()
    if hasattr(codec.L, "zl_cctx_phase_stats") and codec.L.zl_cctx_phase_stats(C.c_void_p(codec.cctx), ct) == 0 and ct[4]:
        it = ct[4]
        print(f"        match phases, cycles/iteration: find {ct[0]/it:.0f}  parse {ct[1]/it:.0f}  emit {ct[2]/it:.0f}  insert {ct[3]/it:.0f}   sequences/iteration {ct[5]/it:.1f}")
    if ct[14]:
        nc = ct[14]
        print(f"        entropy phases, kcycles/chunk: statistics {ct[8]/nc/1e3:.0f}  tree|tables {ct[9]/nc/1e3:.0f}  codes+bits {ct[10]/nc/1e3:.0f}  chains|tree description (+tail) {ct[13]/nc/1e3:.0f}  blk literals {ct[11]/nc/1e3:.0f}  blk sequences {ct[12]/nc/1e3:.0f}")
    if st[13]:
        nb = st[13]
        print(f"        exec phases, kcycles/block: scan {st[8]/nb/1e3:.0f}  lit {st[9]/nb/1e3:.0f}  match {st[10]/nb/1e3:.0f}  flush {st[11]/nb/1e3:.0f}  batches/block {st[12]/nb:.1f}")
    if st[14]:
        nwg = (n + (128 << 10) - 1) // (128 << 10) / 16
        print(f"        entropy roles, kcycles/workgroup: literals {st[14]/nwg/1e3:.0f}  sequences {st[15]/nwg/1e3:.0f}")
    print(f"{kind:6s} ratio {n/total:6.3f} | " + " ".join(f"{k} {v:7.3f}ms ({gb/(v/1e3):7.1f} GB/s)" for k, v in best.items()), flush=True)
