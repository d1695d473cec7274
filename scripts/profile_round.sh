#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, rocprofv3 kernel traces + stats of the bench command (headline layout), of
# the DEFAULT one-frame layout's round trip (scripts/one_frame_prof.py) and of the large-frame decode of a reference frame
# (scripts/n1_prof.py), and PMC passes (each counter set in its own run, never combined with other trace domains): FETCH_SIZE /
# WRITE_SIZE for both layouts, L2 hit / miss and fabric read requests for the headline.  Outputs land in gpurun_out/;
# scripts/summarize_profiles.py <round> turns them into the summaries kept under profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_bench gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_l2 gpurun_out/prof_one_frame gpurun_out/pmc_of_fetch gpurun_out/pmc_of_write
timeout 900 python3 bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full_err.log
tail -c 600 gpurun_out/bench_full_err.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-c2 --no-extra > gpurun_out/prof_bench_line.json 2> gpurun_out/prof_bench_err.log
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-c2 --no-extra > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-c2 --no-extra > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d gpurun_out/pmc_l2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-c2 --no-extra > /dev/null 2>&1
# the default (one-frame) layout
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one_frame -- python3 scripts/one_frame_prof.py 1024 3 > gpurun_out/one_frame_run.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmc_of_fetch -- python3 scripts/one_frame_prof.py 1024 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmc_of_write -- python3 scripts/one_frame_prof.py 1024 2 > /dev/null 2>&1
# the large-frame pipeline under the same profiler: only the stats table travels back
rm -rf /tmp/prof_n1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n1 -- python3 scripts/n1_prof.py 1024 > gpurun_out/n1_run.log 2>&1
cp "$(ls /tmp/prof_n1/*/*kernel_stats.csv | head -1)" gpurun_out/n1_kernel_stats.csv 2>/dev/null
# the deep match finder of levels >= 6 (zl_deep.hip): level 9 on the bench mix, level 12 on text
rm -rf /tmp/prof_deep
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_deep -- python3 scripts/dev/deep_prof.py 9 mix > gpurun_out/deep9_run.log 2>&1
cp "$(ls /tmp/prof_deep/*/*kernel_stats.csv | head -1)" gpurun_out/deep9_kernel_stats.csv 2>/dev/null
rm -rf /tmp/prof_deep
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_deep -- python3 scripts/dev/deep_prof.py 12 text > gpurun_out/deep12_run.log 2>&1
cp "$(ls /tmp/prof_deep/*/*kernel_stats.csv | head -1)" gpurun_out/deep12_text_kernel_stats.csv 2>/dev/null
# the price-based parse of levels 16 - 22 (zl_opt.hip): level 16 on the bench mix
rm -rf /tmp/prof_deep
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_deep -- python3 scripts/dev/deep_prof.py 16 mix > gpurun_out/opt16_run.log 2>&1
cp "$(ls /tmp/prof_deep/*/*kernel_stats.csv | head -1)" gpurun_out/opt16_kernel_stats.csv 2>/dev/null
# config C4: 100 000 small objects with a dictionary through the batch entries (the small-object kernels of round 6)
rm -rf /tmp/prof_c4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python3 scripts/dev/c4_quick.py > gpurun_out/c4_run.log 2>&1
cp "$(ls /tmp/prof_c4/*/*kernel_stats.csv | head -1)" gpurun_out/c4_kernel_stats.csv 2>/dev/null
# keep what travels back small: csv summaries only
find gpurun_out -name "*.db" -delete 2>/dev/null
du -sh gpurun_out
