cd "${GRAFT_REPO_ROOT:-.}"
ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.ref0.so python3 scripts/dev/same_bytes.py write /tmp/ref0.json 2>&1 | grep -v amdgpu
python3 scripts/dev/same_bytes.py check /tmp/ref0.json 2>&1 | grep -v amdgpu
python3 scripts/kernel_bench.py 268435456 text,iid10,int20,rand 2>&1 | grep "entropy phases\|ratio"  | sed 's/scan.*//'
python3 scripts/layout_bench.py mix 1024 2>&1 | grep -v amdgpu
