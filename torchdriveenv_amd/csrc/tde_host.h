// tde_host.h — what the translation units of libtde_hip.so share on the HOST side: the error slot of tde_last_error, the
// dispatch-by-slot-count macros and the launchers each kernel family's unit exports to the C-ABI entry points in tde_api.hip.
// Everything here has hidden visibility (the library is built with -fvisibility=hidden; only TDE_API symbols are exported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tde_hip.h"

namespace tde { struct StepArgs; }

namespace tde_host {

// the calling thread's error slot (256 bytes), defined in tde_api.hip
char *err_buf();
constexpr int kErrLen = 256;

inline int fail(const char *what, hipError_t e)
{
    snprintf(err_buf(), kErrLen, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

inline int bad(const char *msg)
{
    snprintf(err_buf(), kErrLen, "%s", msg);
    return (int)hipErrorInvalidValue;
}

inline int launch_status(const char *what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail(what, e);
}

// number of CUs of the current device (cached per device and thread)
inline int cu_count()
{
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        cached = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cached_dev = dev;
    }
    return cached;
}

// hash of what the NPC controller depends on besides the state (tde_act_cache; act_key_steps)
inline uint32_t act_cfg_hash(const tde_config &cfg, const tde_world &w)
{
#ifdef TDE_ACT_KEY_PLAIN          // (A/B builds: the round-3 key, the step counter alone)
    return 0u;
#endif
    uint32_t h = cfg.flags & (TDE_F_NPC | TDE_F_REPLAY | TDE_F_TRAFFIC_LIGHTS);
    // the identity of the tables the controller reads (routes, spawn records, stop lines, light phases, maps, scenarios): a caller
    // that swaps or rebuilds the world under an unchanged state gets the actions recomputed, not replayed
    const void *tabs[7] = {w.route_xy, w.spawn, w.stoplines, w.phases, w.maps, w.scn, w.replay_states};
    for (int i = 0; i < 7; ++i) {
        const uint64_t a = (uint64_t)(uintptr_t)tabs[i];
        h = (h ^ (uint32_t)a) * 0x9E3779B1u;
        h = (h ^ (uint32_t)(a >> 32)) * 0x9E3779B1u;
    }
    const int32_t dims[6] = {w.n_routes, w.RW, w.n_replay, w.RT, w.n_scn, w.n_maps};
    for (int i = 0; i < 6; ++i) h = (h ^ (uint32_t)dims[i]) * 0x9E3779B1u;
    const float c[10] = {cfg.npc_k_steer, cfg.npc_k_speed, cfg.npc_gap_s0, cfg.npc_cone_k, cfg.npc_lane_half, cfg.npc_reach,
                         cfg.npc_max_accel, cfg.npc_max_steer, cfg.npc_cone_range, cfg.dt};
    for (int i = 0; i < 10; ++i) {
        uint32_t b;
        memcpy(&b, &c[i], 4);
        h = (h ^ b) * 0x9E3779B1u;
    }
    return h;
}

// ---- one launcher per kernel family; each is defined by the unit that instantiates the family (build.py compiles them side by
// side).  All return 0 or a hipError_t value with the message in the error slot. -------------------------------------------------
// tde_step_trio.hip: env_step_trio_kernel<A in {8, 16, 32}, LIGHTS, OBS, MAG>
int launch_step_trio(const tde_config *cfg, const tde_world *world, const tde_state *st, uint32_t act_hash, void *stream);
// tde_step_wide.hip: env_step_wide_kernel<LIGHTS, OBS, MAG> (128 agent slots per env, two roles)
// (`args` = the launch's argument block in device memory, tde_api.hip: step_args; `st` = the caller's struct - shape, optional outputs, action)
int launch_step_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, int waves, void *stream);
// tde_step_wide8.hip: the same kernel's eight-wavefront instantiations (launch_step_wide hands waves == 8 over)
int launch_step_wide8(const tde::StepArgs *args, const tde_config *cfg, const tde_state *st, void *stream);
// tde_step_solo.hip / tde_step_solo_mag.hip: env_step_kernel<A, LIGHTS, OBS, BIG, WAVES, MAG = false / true>
int launch_step_solo(const tde_config *cfg, const tde_world *world, const tde_state *st, void *stream);
int launch_step_solo_mag(const tde_config *cfg, const tde_world *world, const tde_state *st, void *stream);
// tde_rollout_trio.hip: env_rollout_trio_kernel<A in {8, 16, 32}, LIGHTS, BIG>; env_rollout_wide_kernel<LIGHTS> (128 slots)
int launch_rollout_trio(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream);
int launch_rollout_wide(const tde::StepArgs *args, const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, int waves, void *stream);
// tde_rollout_duo.hip: env_rollout_duo_kernel<A <= 64, LIGHTS, BIG>
int launch_rollout_duo(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream);
// tde_rollout_solo.hip: env_rollout_kernel<A <= 128, LIGHTS>
int launch_rollout_solo(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, void *stream);

}  // namespace tde_host

#define TDE_DISPATCH_A(A, ...)                                  \
    switch (A) {                                                \
        case 1: { constexpr int kA = 1; __VA_ARGS__; } break;   \
        case 2: { constexpr int kA = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int kA = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int kA = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int kA = 16; __VA_ARGS__; } break; \
        case 32: { constexpr int kA = 32; __VA_ARGS__; } break; \
        case 64: { constexpr int kA = 64; __VA_ARGS__; } break; \
    }
#define TDE_DISPATCH_A128(A, ...)                               \
    if ((A) == 128) { constexpr int kA = 128; __VA_ARGS__; } else TDE_DISPATCH_A(A, __VA_ARGS__)
