import sys, time, warnings
sys.path.insert(0, "."); sys.path.insert(0, "tests")
warnings.simplefilter("ignore")
import torch
from zstdlite_amd.device import DeviceCodec, gen_device
dev = torch.device("cuda:0")
for mib in (255, 256):
    src = gen_device("text", mib << 20, dev)
    for lvl in (3, 9):
        codec = DeviceCodec(level=lvl, profile=True, frame_layout="frames")
        dst = torch.empty(codec.bound(src.numel()), dtype=torch.uint8, device=dev)
        for _ in range(3): codec.compress(src, dst)
        ms = codec.kernel_ms()
        print(mib, "MiB level", lvl, "match/parse %.3f ms = %.4f ms/MiB" % (ms["match"], ms["match"] / mib), "entropy %.3f" % ms["entropy"], flush=True)
        codec.close()
