cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 120 python3 scripts/n1_prof.py 1024 2>&1 | grep -v amdgpu || { echo "N1 1024 FAILED/HUNG"; exit 1; }
timeout 100 python3 scripts/layout_bench.py mix 1024 2>&1 | grep one_frame
timeout 400 python3 -m pytest tests/test_bigframe_gpu.py tests/test_decode_gpu.py -m gpu -q -x 2>&1 | tail -3
