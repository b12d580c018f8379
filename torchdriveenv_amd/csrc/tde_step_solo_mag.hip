// tde_step_solo_mag.hip — the one-role step kernel (env_step_kernel: tde_kernels.h) with the infraction magnitudes (MAG = true: tde_state.magnitudes), and its launcher.
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_solo_mag(const tde_config *cfg, const tde_world *world, const tde_state *st, void *stream)
{
    constexpr bool kMag = true;
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const unsigned nb = (unsigned)(((int64_t)st->B * st->A + tde::kBlock - 1) / tde::kBlock);
#define TDE_LAUNCH_FORM(AA, L, O, G, W) \
    tde::env_step_kernel<AA, L, O, G, W, kMag><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(*cfg, *world, *st, st->action, (float *)nullptr, st->done_bits)
#define TDE_LAUNCH_LO(AA, G, W)                                                                                            \
    if (st->obs) { if (lights) TDE_LAUNCH_FORM(AA, true, true, G, W); else TDE_LAUNCH_FORM(AA, false, true, G, W); }       \
    else { if (lights) TDE_LAUNCH_FORM(AA, true, false, G, W); else TDE_LAUNCH_FORM(AA, false, false, G, W); }
    if ((world->hints & TDE_WORLD_LARGE_GRID) && (st->A == 32 || st->A == 64)) {      // (up to 16 slots per env the class map is read anyway)
        if (st->A == 32) { TDE_LAUNCH_LO(32, true, TDE_WIDE_WAVES) } else { TDE_LAUNCH_LO(64, true, TDE_WIDE_WAVES) }
    } else if (st->A == 128 && st->B > 6 * cu_count()) {   // (128 slots, more than a residency round of the 3-per-SIMD form: 4 per SIMD)
        TDE_LAUNCH_LO(128, false, 4)
    } else {
        TDE_DISPATCH_A128(st->A, TDE_LAUNCH_LO(kA, false, TDE_WIDE_WAVES));
    }
#undef TDE_LAUNCH_LO
#undef TDE_LAUNCH_FORM
    return launch_status("tde_env_step");
}

}  // namespace tde_host
