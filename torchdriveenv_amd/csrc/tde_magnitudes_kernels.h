// tde_magnitudes_kernels.h — the stand-alone magnitude kernels (tde_ego_infractions, tde_env_post_step: include/tde_hip.h) over the
// wavefront-wide functions of tde_magnitudes.h.  Included by tde_kernels.hip after the step path's own definitions (Agent,
// reset_lane, kBlock).
#pragma once
#include "tde_magnitudes.h"

namespace tde {

TDE_DEV EgoBox ego_box(const tde_state &st, int64_t g0)
{
    EgoBox b;
    b.x = st.x[g0]; b.y = st.y[g0];
    sincos_f32(st.psi[g0], b.s, b.c);
    b.hl = 0.5f * st.len[g0]; b.hw = 0.5f * st.wid[g0];
    return b;
}

// the boxes of an env's slots from the state arrays (tde_ego_infractions / tde_env_post_step run on a state in global memory)
struct StateRows {
    const tde_state &st;
    int64_t g0;
    TDE_DEV bool operator()(int j, float &x, float &y, float &c, float &s, float &hl, float &hw) const
    {
        if (!st.present[g0 + j]) return false;
        x = st.x[g0 + j]; y = st.y[g0 + j];
        sincos_f32(st.psi[g0 + j], s, c);
        hl = 0.5f * st.len[g0 + j]; hw = 0.5f * st.wid[g0 + j];
        return true;
    }
};

// out[e] = (offroad magnitude, collision magnitude = sum of IoUs, number of overlapping agents, 0) of env e's ego on the CURRENT
// state, whatever its flags say; one wavefront per env (every env has work: the operator form, tde_ego_infractions)
#ifdef TDE_TU_API      // (a non-template kernel: compiled by the one unit that launches it)
__global__ __launch_bounds__(kBlock) void ego_infractions_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    __shared__ float poly[kBlock / kWave][32];                     // per wavefront: box_iou_wave's vertex lists
    const int lane = (int)(threadIdx.x & 63u);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    if (e >= st.B) return;                                           // (wave-uniform)
    const int64_t g0 = (int64_t)e * st.A;
    float omag = 0.0f;
    float2 cm = make_float2(0.0f, 0.0f);
    if (st.present[g0]) {
        const EgoBox eb = ego_box(st, g0);
        cm = ego_collision_mag_of(st.A, lane, eb, StateRows{st, g0}, poly[threadIdx.x >> 6]);
        if (cfg.flags & TDE_F_OFFROAD) {
            const tde_map m = w.maps[reinterpret_cast<const int4 *>(w.scn)[st.scn[e]].x];
            omag = ego_offroad_mag_wave(cfg, w, m, eb, lane);
        }
    }
    if (lane == 0) reinterpret_cast<float4 *>(out)[e] = make_float4(omag, cm.x, cm.y, 0.0f);
}
#endif

// tde_env_post_step: what follows a step that was launched WITHOUT TDE_F_AUTORESET, in one launch -
// (a) out[e] = the magnitudes of the ego's infractions on the state that step left, GATED by the flags it stored: a magnitude is
//     non-zero only under its flag (collision: the same predicate; offroad: a corner beyond the threshold has d^2 > thr^2, and
//     sqrt(d^2) <= thr for d^2 <= RN(thr * thr) since RN(sqrt(RN(x * x))) = x), so the 98 % of the envs without an infraction
//     are not looked at;
// (b) with TDE_F_AUTORESET the re-spawn of the envs it finished (env_reset_kernel's stores for mask = terminated | truncated)
//     and, when the state carries the compact observation, that of the new episode (state_obs_kernel's expression).
// One wavefront per env (a workgroup of four wavefronts per env - a corner each - was tried: 32 768 wavefronts to launch for the
// ~150 that have work, 36 vs 31 us per step of the env; profiles/r04_z_magnitudes_cost.txt).
template <int A>
__global__ __launch_bounds__(kBlock) void env_post_step_kernel(tde_config cfg, tde_world w, tde_state st, float *__restrict__ out)
{
    __shared__ float poly[kBlock / kWave][32];
    const int lane = (int)(threadIdx.x & 63u), wv = (int)(threadIdx.x >> 6);
    const int e = (int)(blockIdx.x * (kBlock / kWave) + wv);
    if (e >= st.B) return;                                           // (wave-uniform, like every branch below but the slot guards)
    const int64_t g0 = (int64_t)e * A;
    if (out) {
        float omag = 0.0f;
        float2 cm = make_float2(0.0f, 0.0f);
        const bool do_coll = st.present[g0] && st.collided[g0] != 0;
        const bool do_off = st.present[g0] && (cfg.flags & TDE_F_OFFROAD) && st.offroad[g0] != 0;
        if (do_coll || do_off) {
            const EgoBox eb = ego_box(st, g0);
            if (do_coll) cm = ego_collision_mag_of(A, lane, eb, StateRows{st, g0}, poly[wv]);
            if (do_off) {
                const tde_map m = w.maps[reinterpret_cast<const int4 *>(w.scn)[st.scn[e]].x];
                omag = ego_offroad_mag_wave(cfg, w, m, eb, lane);
            }
        }
        if (lane == 0) reinterpret_cast<float4 *>(out)[e] = make_float4(omag, cm.x, cm.y, 0.0f);
    }
    if (!(cfg.flags & TDE_F_AUTORESET) || !(st.terminated[e] | st.truncated[e])) return;
    Cold cold;
    fill_cold(cold, cfg, w);
    const int episode = st.episode[e];
    for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + lane;
        if (a >= A) continue;
        Agent ag;
        EnvRegs er{0, 0, 0, 0, episode};
        reset_lane<A, false>(cfg, cold, e, a, ag, er);
        const int64_t g = g0 + a;
        store_agent_dynamic(st, g, ag);
        store_agent_static(st, g, ag);
        st.collided[g] = 0;
        st.offroad[g] = 0;
        if (a == 0) {
            st.scn[e] = er.scn; st.steps[e] = 0; st.target_idx[e] = 1; st.reached[e] = 0; st.episode[e] = er.episode;
            if (st.ep_return) st.ep_return[e] = 0.0;
            if (st.obs) {
                const bool has = 1 < reinterpret_cast<const int4 *>(w.scn)[er.scn].y;
                float fwd = 0.0f, lat = 0.0f;
                if (has) {
                    const double2 t = reinterpret_cast<const double2 *>(w.wp_xy)[(int64_t)er.scn * w.NW + 1];
                    float s, c;
                    sincos_f32(ag.psi, s, c);
                    const float dx = (float)t.x - ag.x, dy = (float)t.y - ag.y;
                    fwd = dx * c + dy * s;
                    lat = dy * c - dx * s;
                }
                float4 *ob = reinterpret_cast<float4 *>(st.obs) + 2 * (int64_t)e;
                ob[0] = make_float4(ag.x, ag.y, ag.psi, ag.v);
                ob[1] = make_float4(fwd, lat, has ? 1.0f : 0.0f, 0.0f);
            }
        }
    }
}

}  // namespace tde
