#!/bin/bash
# VGPR / SGPR / spill / scratch / LDS figures of ONE kernel instantiation in seconds (the whole translation unit takes 80 s):
#   scripts/probe_kernel.sh 'tde::env_step_trio_kernel<16, false, false, true>(tde_config, tde_world, tde_state, uint32_t)' [-D...]
SIG=$1; shift
D=$(mktemp -d)
echo "template __global__ void $SIG;" > "$D/tde_probe.inc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
  -fno-slp-vectorize -fno-vectorize -Rpass-analysis=kernel-resource-usage --offload-device-only "$@" \
  -DTDE_KERNEL_PROBE "-I$D" -c -o /dev/null "$(dirname "$0")/../torchdriveenv_amd/csrc/tde_kernels.hip" 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|Spill|ScratchSize|Occupancy|LDS Size|error" | sed 's/.*remark: [^ ]* //; s/ \[-Rpass.*//'
rm -rf "$D"
