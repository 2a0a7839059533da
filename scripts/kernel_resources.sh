#!/bin/sh
# registers, spills and occupancy of the kernels whose mangled name matches $1 (default: all), from a device-only compile
#   sh scripts/kernel_resources.sh [pattern] [-DFLAG ...]
R=$(cd "$(dirname "$0")/.." && pwd)
PAT=${1:-.}; [ $# -gt 0 ] && shift
cd $R/gipuma_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -c \
    -Rpass-analysis=kernel-resource-usage --offload-device-only "$@" -o /dev/null gipuma_hip.hip 2>&1 |
  awk -v pat="$PAT" '/Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name); show = name ~ pat }
       show && /VGPRs:|VGPRs Spill|SGPRs Spill|Occupancy/ { v=$0; sub(/.*remark: +/,"",v); sub(/ \[-R.*/,"",v); line = line "  " v }
       show && /LDS Size/ { print name; print "   " line; line="" }'
