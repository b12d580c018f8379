// tde_step_wide.hip — the closed-loop step's two-role kernel for 128 agent slots per env (env_step_wide_kernel: tde_kernels.h) and its
// launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, void *stream)
{
    if (st->A != 128) return bad("tde_env_step: the two-role wide kernel serves 128 agent slots per env");
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
#define TDE_LAUNCH_WIDE(L, O)                                                                                                              \
    do {                                                                                                                                   \
        if (st->magnitudes) tde::env_step_wide_kernel<L, O, true><<<(unsigned)st->B, 4 * tde::kWave, 0, (hipStream_t)stream>>>(args, st->action);  \
        else tde::env_step_wide_kernel<L, O, false><<<(unsigned)st->B, 4 * tde::kWave, 0, (hipStream_t)stream>>>(args, st->action);               \
    } while (0)
    if (st->obs) { if (lights) TDE_LAUNCH_WIDE(true, true); else TDE_LAUNCH_WIDE(false, true); }
    else { if (lights) TDE_LAUNCH_WIDE(true, false); else TDE_LAUNCH_WIDE(false, false); }
#undef TDE_LAUNCH_WIDE
    return launch_status("tde_env_step");
}

}  // namespace tde_host
