// tde_rollout_solo.hip — the persistent rollout's one-role kernel (env_rollout_kernel: any slot count; at 128 slots the workgroup
// is the env's two wavefronts) and its launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_rollout_solo(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream)
{
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    if (st->A == 128) {
        if (lights) tde::env_rollout_kernel<128, true><<<(unsigned)st->B, 128, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro);
        else tde::env_rollout_kernel<128, false><<<(unsigned)st->B, 128, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro);
        return launch_status("tde_env_rollout");
    }
    const unsigned nb = (unsigned)(((int64_t)st->B * st->A + tde::kWave - 1) / tde::kWave);
    if (lights) {
        TDE_DISPATCH_A(st->A, tde::env_rollout_kernel<kA, true><<<nb, tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro));
    } else {
        TDE_DISPATCH_A(st->A, tde::env_rollout_kernel<kA, false><<<nb, tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro));
    }
    return launch_status("tde_env_rollout");
}

}  // namespace tde_host
