#!/bin/bash
# VGPRs / SGPRs / LDS / occupancy of the kernels as hipcc reports them (no GPU needed).
# usage: scripts/kernel_resources.sh [kernel-name-substring] [-DDEFINE ...]
cd "$(dirname "$0")/.."
pat=${1:-k_map_fused}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -c "$@" \
  -Rpass-analysis=kernel-resource-usage ct_mapreduce_amd/csrc/ctmr_engine.hip -o /dev/null 2>&1 |
  grep -A12 "Function Name: .*${pat}" | grep -E "Function Name|VGPRs:|SGPRs:|Spill|Occupancy|LDS Size|ScratchSize"
