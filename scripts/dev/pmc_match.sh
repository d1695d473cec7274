#!/bin/bash
# PMC passes over the match kernel, in-tree library vs a variant (ZL_HIP_LIB is read by zstdlite_amd/_lib.py)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_avail.txt 2>&1
for v in base old; do
  if [ "$v" = base ]; then unset ZL_HIP_LIB; else export ZL_HIP_LIB=$PWD/zstdlite_amd/build/variants/libzl_hip.$v.so; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1)); rm -rf /tmp/pmc_$v_$i
    timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pmc_${v}_$i -- python3 scripts/dev/pmc_match.py text > /tmp/pmc_log_${v}_$i.txt 2>&1 || tail -3 /tmp/pmc_log_${v}_$i.txt
    echo "== $v set $i"; python3 scripts/pmc_util.py /tmp/pmc_${v}_$i 2>/dev/null | grep -A12 "zl_match"
  done
done
