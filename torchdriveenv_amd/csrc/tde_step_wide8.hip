// tde_step_wide8.hip — the closed-loop step's two-role kernel for 128 agent slots per env (env_step_wide_kernel: tde_kernels.h) in its
// eight-wavefront form (drive / drive / judge / judge + two sweep helpers and two offroad helpers per env).  The two forms are two translation units: sixteen
// instantiations of this kernel in one unit were the library's longest compile (49 s of a 60-s build).
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_wide8(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, void *stream)
{
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
#define TDE_LAUNCH_WIDE3(L, O, M) tde::env_step_wide_kernel<L, O, M, 8><<<(unsigned)st->B, 8 * tde::kWave, 0, (hipStream_t)stream>>>(args, st->action)
#define TDE_LAUNCH_WIDE(L, O) do { if (st->magnitudes) TDE_LAUNCH_WIDE3(L, O, true); else TDE_LAUNCH_WIDE3(L, O, false); } while (0)
    if (st->obs) { if (lights) TDE_LAUNCH_WIDE(true, true); else TDE_LAUNCH_WIDE(false, true); }
    else { if (lights) TDE_LAUNCH_WIDE(true, false); else TDE_LAUNCH_WIDE(false, false); }
#undef TDE_LAUNCH_WIDE
#undef TDE_LAUNCH_WIDE3
    return launch_status("tde_env_step");
}

}  // namespace tde_host
