// tde_step_wide.hip — the closed-loop step's two-role kernel for 128 agent slots per env (env_step_wide_kernel: tde_kernels.h) in its
// four-wavefront form (drive / drive / judge / judge), and the launcher of both forms.  The two forms are two translation units: sixteen
// instantiations of this kernel in one unit were the library's longest compile (49 s of a 60-s build).
#include "tde_kernels.h"
#include "tde_host.h"

namespace tde_host {

int launch_step_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, int waves, void *stream)
{
    if (st->A != 128) return bad("tde_env_step: the two-role wide kernel serves 128 agent slots per env");
    if (waves == 8) return launch_step_wide8(args, cfg, st, stream);
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
#define TDE_LAUNCH_WIDE3(L, O, M) tde::env_step_wide_kernel<L, O, M, 4><<<(unsigned)st->B, 4 * tde::kWave, 0, (hipStream_t)stream>>>(args, st->action)
#define TDE_LAUNCH_WIDE(L, O) do { if (st->magnitudes) TDE_LAUNCH_WIDE3(L, O, true); else TDE_LAUNCH_WIDE3(L, O, false); } while (0)
    if (st->obs) { if (lights) TDE_LAUNCH_WIDE(true, true); else TDE_LAUNCH_WIDE(false, true); }
    else { if (lights) TDE_LAUNCH_WIDE(true, false); else TDE_LAUNCH_WIDE(false, false); }
#undef TDE_LAUNCH_WIDE
#undef TDE_LAUNCH_WIDE3
    return launch_status("tde_env_step");
}

}  // namespace tde_host
