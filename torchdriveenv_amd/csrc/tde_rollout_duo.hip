// tde_rollout_duo.hip — the persistent rollout's two-role kernel (env_rollout_duo_kernel: up to 64 agent slots per env) and its launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_rollout_duo(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream)
{
    const unsigned nb = (unsigned)(((int64_t)st->B * st->A + tde::kWave - 1) / tde::kWave);
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const bool big = (world->hints & TDE_WORLD_LARGE_GRID) != 0;      // corner classes from the 2-bit class map (tde_abi.h)
    const uint32_t act_hash = act_cfg_hash(*cfg, *world);             // keys the world's first-step gap cache (tde_first_gap)
#define TDE_LAUNCH_DUO(L, G) TDE_DISPATCH_A(st->A, tde::env_rollout_duo_kernel<kA, L, G><<<nb, 2 * tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro, act_hash))
    if (lights) { if (big) { TDE_LAUNCH_DUO(true, true); } else { TDE_LAUNCH_DUO(true, false); } }
    else { if (big) { TDE_LAUNCH_DUO(false, true); } else { TDE_LAUNCH_DUO(false, false); } }
#undef TDE_LAUNCH_DUO
    return launch_status("tde_env_rollout");
}

}  // namespace tde_host
