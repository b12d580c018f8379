// tde_kernels.hip — libtde_hip.so as ONE translation unit: every unit of the library included here.  build.py compiles the units
// side by side instead (about a third of the wall time on 8 cores); this form serves the A/B and resource-probe scripts
// (scripts/build_variant.sh, scripts/kernel_resources.sh, scripts/probe_kernel.sh), which pass one source file to hipcc.
#ifndef TDE_KERNEL_PROBE
#define TDE_TU_API 1
#endif
#include "tde_kernels.h"

// Register / spill figures of a few kernels in seconds instead of the minutes of the whole library:
// -DTDE_KERNEL_PROBE -I<dir of a tde_probe.inc holding explicit instantiations> --offload-device-only (scripts/probe_kernel.sh)
#ifdef TDE_KERNEL_PROBE
#include "tde_probe.inc"
#else
#include "tde_api.hip"
#include "tde_step_trio.hip"
#include "tde_step_wide.hip"
#include "tde_step_wide8.hip"
#include "tde_step_solo.hip"
#include "tde_step_solo_mag.hip"
#include "tde_rollout_trio.hip"
#include "tde_rollout_duo.hip"
#include "tde_rollout_solo.hip"
#endif
