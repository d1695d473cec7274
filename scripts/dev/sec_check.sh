cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 200 python3 scripts/layout_bench.py mix 1024 2>&1 | grep -v amdgpu
timeout 100 python3 scripts/layout_bench.py corpus 10 2>&1 | grep -v amdgpu | grep -A1 one_frame
timeout 100 python3 scripts/layout_bench.py records 64 2>&1 | grep -v amdgpu | grep -A1 one_frame
timeout 100 python3 scripts/layout_bench.py mix 300 2>&1 | grep -v amdgpu | grep -A1 one_frame
timeout 600 python3 -m pytest tests/test_compress_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -m gpu -q -x 2>&1 | tail -3
