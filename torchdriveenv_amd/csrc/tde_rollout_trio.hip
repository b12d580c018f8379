// tde_rollout_trio.hip — the persistent rollout's three-role kernel (env_rollout_trio_kernel: 8 / 16 / 32 agent slots per env)
// and the two-role kernel for 128 slots (env_rollout_wide_kernel), with their launchers.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_rollout_trio(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream)
{
    const unsigned nb = (unsigned)(((int64_t)st->B * st->A + tde::kWave - 1) / tde::kWave);
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const bool big = (world->hints & TDE_WORLD_LARGE_GRID) != 0;      // corner classes from the 2-bit class map (tde_abi.h)
    const uint32_t act_hash = act_cfg_hash(*cfg, *world);             // keys the world's first-step gap cache (tde_first_gap)
    (void)act_hash;
#if TDE_ROLLOUT_CONST_ARGS
    {
        tde::RolloutArgs ra{*cfg, *world, *st, *ro, act_cfg_hash(*cfg, *world)};
        hipError_t ec = hipMemcpyToSymbolAsync(HIP_SYMBOL(tde::g_rollout_args), &ra, sizeof(ra), 0, hipMemcpyHostToDevice, (hipStream_t)stream);
        if (ec != hipSuccess) return fail("tde_env_rollout (argument block)", ec);
    }
#define TDE_LAUNCH_TRIO2(AA, L, G) tde::env_rollout_trio_kernel<AA, L, G><<<nb, 3 * tde::kWave, 0, (hipStream_t)stream>>>(0)
#else
#define TDE_LAUNCH_TRIO2(AA, L, G) tde::env_rollout_trio_kernel<AA, L, G><<<nb, 3 * tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro, act_hash)
#endif
#define TDE_LAUNCH_TRIO(AA)                                                                          \
    if (lights) { if (big) TDE_LAUNCH_TRIO2(AA, true, true); else TDE_LAUNCH_TRIO2(AA, true, false); } \
    else { if (big) TDE_LAUNCH_TRIO2(AA, false, true); else TDE_LAUNCH_TRIO2(AA, false, false); }
    if (st->A == 8) { TDE_LAUNCH_TRIO(8) } else if (st->A == 16) { TDE_LAUNCH_TRIO(16) } else if (st->A == 32) { TDE_LAUNCH_TRIO(32) }
    else return bad("tde_env_rollout: the three-role kernel serves 8, 16 or 32 agent slots per env");
#undef TDE_LAUNCH_TRIO
#undef TDE_LAUNCH_TRIO2
    return launch_status("tde_env_rollout");
}

int launch_rollout_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, int waves, void *stream)
{
    // two roles, four wavefronts per env of 128 slots (waves = 8: + two sweep helpers and two offroad helpers): one env per workgroup
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
#if TDE_WIDE_ROLLOUT_BLOCK
    (void)world;
#define TDE_LAUNCH_RW(L, NW) tde::env_rollout_wide_kernel<L, NW><<<(unsigned)st->B, NW * tde::kWave, 0, (hipStream_t)stream>>>(args, *ro)
#else
    (void)args;
#define TDE_LAUNCH_RW(L, NW) tde::env_rollout_wide_kernel<L, NW><<<(unsigned)st->B, NW * tde::kWave, 0, (hipStream_t)stream>>>(*cfg, *world, *st, *ro)
#endif
    if (waves == 8) { if (lights) TDE_LAUNCH_RW(true, 8); else TDE_LAUNCH_RW(false, 8); }
    else { if (lights) TDE_LAUNCH_RW(true, 4); else TDE_LAUNCH_RW(false, 4); }
#undef TDE_LAUNCH_RW
    return launch_status("tde_env_rollout");
}

}  // namespace tde_host
