#!/bin/bash
# Per-kernel registers / spills / LDS / occupancy of one .hip source, as the compiler reports them (library kernels that a source
# instantiates -- rocPRIM's -- are left out).
# usage: scripts/kernel_resources.sh zstdlite_amd/csrc/zl_encode.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /TotalSGPRs:|  VGPRs:|Spill|ScratchSize|Occupancy|LDS Size/ {v=$0; sub(/.*remark: +/,"",v); sub(/ \[-Rpass.*/,"",v); row[name]=row[name] " | " v}
       END {for (k in row) if (k !~ /rocprim|hipcub/) print k row[k]}' | sort
