// tde_step_trio.hip — the closed-loop step's three-role kernel (env_step_trio_kernel: tde_kernels.h) and its launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_trio(const tde_config *cfg, const tde_world *world, const tde_state *st, uint32_t act_hash, void *stream)
{
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const unsigned ng = (unsigned)(((int64_t)st->B * st->A + tde::kWave - 1) / tde::kWave);
#define TDE_LAUNCH_STEP3(AA, L, O)                                                                                                   \
    do {                                                                                                                           \
        if (st->magnitudes) tde::env_step_trio_kernel<AA, L, O, true><<<ng, 3 * tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, act_hash); \
        else tde::env_step_trio_kernel<AA, L, O, false><<<ng, 3 * tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, act_hash);               \
    } while (0)
#define TDE_LAUNCH_STEP3_A(AA)                                                                       \
    if (st->obs) { if (lights) TDE_LAUNCH_STEP3(AA, true, true); else TDE_LAUNCH_STEP3(AA, false, true); } \
    else { if (lights) TDE_LAUNCH_STEP3(AA, true, false); else TDE_LAUNCH_STEP3(AA, false, false); }
    if (st->A == 8) { TDE_LAUNCH_STEP3_A(8) } else if (st->A == 16) { TDE_LAUNCH_STEP3_A(16) } else if (st->A == 32) { TDE_LAUNCH_STEP3_A(32) }
    else return bad("tde_env_step: the three-role kernel serves 8, 16 or 32 agent slots per env");
#undef TDE_LAUNCH_STEP3_A
#undef TDE_LAUNCH_STEP3
    return launch_status("tde_env_step");
}

}  // namespace tde_host
