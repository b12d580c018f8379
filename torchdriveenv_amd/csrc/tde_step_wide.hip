// tde_step_wide.hip — the closed-loop step's two-role kernel for 128 agent slots per env (env_step_wide_kernel: tde_kernels.h) and its
// launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, int waves, void *stream)
{
    if (st->A != 128) return bad("tde_env_step: the two-role wide kernel serves 128 agent slots per env");
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    // waves = 4: drive / drive / judge / judge; 8: + two sweep helpers and two offroad helpers per env (tde_kernels.h)
#define TDE_LAUNCH_WIDE3(L, O, M, NW) tde::env_step_wide_kernel<L, O, M, NW><<<(unsigned)st->B, NW * tde::kWave, 0, (hipStream_t)stream>>>(args, st->action)
#define TDE_LAUNCH_WIDE(L, O)                                                                                   \
    do {                                                                                                        \
        if (st->magnitudes) { if (waves == 8) TDE_LAUNCH_WIDE3(L, O, true, 8); else TDE_LAUNCH_WIDE3(L, O, true, 4); }   \
        else { if (waves == 8) TDE_LAUNCH_WIDE3(L, O, false, 8); else TDE_LAUNCH_WIDE3(L, O, false, 4); }                \
    } while (0)
    if (st->obs) { if (lights) TDE_LAUNCH_WIDE(true, true); else TDE_LAUNCH_WIDE(false, true); }
    else { if (lights) TDE_LAUNCH_WIDE(true, false); else TDE_LAUNCH_WIDE(false, false); }
#undef TDE_LAUNCH_WIDE
#undef TDE_LAUNCH_WIDE3
    return launch_status("tde_env_step");
}

}  // namespace tde_host
