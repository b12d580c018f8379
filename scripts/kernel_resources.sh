#!/bin/bash
# VGPR / SGPR / spill / LDS figures of the kernels whose name matches $1 (default: the three-role rollout kernel),
# compiled with the flags of torchdriveenv_amd/build.py plus any extra -D flags given after the pattern
PAT=${1:-env_rollout_trio_kernelILi16ELb0}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
  -fno-slp-vectorize -fno-vectorize -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null \
  "$(dirname "$0")/../torchdriveenv_amd/csrc/tde_kernels.hip" 2>&1 |
  grep -A 12 "Function Name: .*$PAT" | grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //'
