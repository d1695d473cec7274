#!/bin/bash
# Every fuzzer of scripts/ once, with the rounds and seeds DESIGN.md section 5 quotes for a "short" pass (about ten GPU-minutes);
# scale=10 makes the long pass.  Run on a GPU box: gpurun -- scripts/fuzz_all.sh [seed=1] [scale=1]
# Each fuzzer prints its own verdict and exits non-zero on a disagreement, a hang (its own timeouts) or a fault.
seed=${1:-1}; scale=${2:-1}
cd "${GRAFT_REPO_ROOT:-.}"
fail=0
run() { echo "=== $*"; timeout 1800 python "$@" 2>&1 | grep -v amdgpu.ids | tail -4; [ ${PIPESTATUS[0]} -eq 0 ] || { echo "FAILED: $*"; fail=1; }; }
run scripts/fuzz_decode.py $((40 * scale)) $seed
run scripts/fuzz_decode.py $((4 * scale)) $seed wide
run scripts/fuzz_bigframe.py $((20 * scale)) $seed
run scripts/fuzz_bigframe.py $((10 * scale)) $seed dict
run scripts/fuzz_dict.py $((20 * scale)) $seed
run scripts/fuzz_encode.py $((200 * scale)) $seed
run scripts/fuzz_stream.py $((60 * scale)) $seed
run scripts/fuzz_stream_decode.py $((100 * scale)) $seed
run scripts/fuzz_multi.py $((100 * scale)) $seed
run scripts/fuzz_big.py $((6 * scale)) $seed
run scripts/fuzz_device.py $((200 * scale)) $seed
run scripts/fuzz_seek.py $((300 * scale)) $seed
run scripts/fuzz_runs.py $((20 * scale)) $seed
run scripts/fuzz_ctx.py $((300 * scale)) $seed
run scripts/fuzz_threads.py $((20 * scale)) 6 $seed
[ $fail -eq 0 ] && echo "ALL FUZZERS AGREE" || echo "SOME FUZZER FAILED"
exit $fail
