"""Builds libzl_hip.so (HIP kernels + C-ABI) in-tree for gfx950.

`python -m zstdlite_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzl_hip.so")
SOURCES = ["zl_decode.hip", "zl_decode2.hip", "zl_decode3.hip", "zl_encode.hip", "zl_deep.hip", "zl_opt.hip", "zl_api.hip", "zl_api_compress.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines, verbose=False):
    """A/B builds for kernel experiments: the whole library compiled with extra -D flags into
    zstdlite_amd/build/variants/libzl_hip.<name>.so (select it with ZL_HIP_LIB=...)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(HERE, "build", "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, f"libzl_hip.{name}.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [hipcc] + FLAGS + ["-I" + CSRC] + list(defines) + ["-shared", "-Wl,-z,nodelete", "-o", out] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "zl_hip.h"))
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(HERE, "build", s + ".o")
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(OUT, objs):
        # -z nodelete: the library keeps worker threads parked in its own code (zl_host.h ZlPool); unloading it would unmap them
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,nodelete", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
