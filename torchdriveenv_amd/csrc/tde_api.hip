// tde_api.hip — the C-ABI of libtde_hip.so (include/tde_hip.h): argument checks, the choice of kernel form, the operator-level and
// reset / raster / magnitude kernels' launches.  The step and rollout kernel families are instantiated by their own translation
// units (tde_step_*.hip, tde_rollout_*.hip: tde_host.h lists their launchers).  No torch types anywhere: plain device pointers,
// sizes and a hipStream_t.
#define TDE_TU_API 1
#include <mutex>
#include <vector>
#include "tde_kernels.h"
#include "tde_host.h"

// ------------------------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------------------------
static thread_local char g_err[tde_host::kErrLen] = "";
char *tde_host::err_buf() { return g_err; }
using tde_host::bad;
using tde_host::cu_count;
using tde_host::fail;

static bool pow2_le64(int A) { return A >= 1 && A <= TDE_MAX_AGENTS && (A & (A - 1)) == 0; }
// (up to 64 slots an env lives inside one wavefront and every kernel form applies; 128 = TDE_MAX_AGENTS: an env spans two
//  wavefronts - tde_env_step runs the one-role kernel's generic form (TDE_DISPATCH_A128), tde_env_rollout the persistent
//  env_rollout_wide_kernel (two roles, four wavefronts per env) at every batch size, the one-role persistent kernel only under
//  tde_kernel_override(1, 0); forced forms 2 and 3 do not exist at 128 slots and leave the choice as it is)

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + tde::kBlock - 1) / tde::kBlock); }

// which kernel form tde_env_rollout / tde_env_step launch: 0 = by group shape and batch size (the measured choice), else
// forced (tde_kernel_override: parity tests of every form, A/B runs)
static std::atomic<int> g_force_rollout{0}, g_force_step{0};

extern "C" {

int tde_abi_version(void) { return TDE_ABI_VERSION; }

int tde_kernel_override(int rollout_team, int step_team)
{
    if (rollout_team < 0 || rollout_team > 3 || step_team < 0 || step_team > 3)
        return bad("tde_kernel_override: rollout_team in {0, 1, 2, 3}, step_team in {0, 1, 2, 3}");
    g_force_rollout = rollout_team;
    g_force_step = step_team;
    return 0;
}

const char *tde_last_error(void) { return g_err; }

int tde_kinematics_step(int64_t n, float *x, float *y, float *psi, float *v, const float *lr, const uint8_t *present,
                        const float *action, float dt, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(tde::kinematics_kernel, dim3(blocks_for(n)), dim3(tde::kBlock), 0, (hipStream_t)stream, n, x, y,
                       psi, v, lr, present, action, dt);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_kinematics_step", e);
}

int tde_compute_collision(int32_t B, int32_t A, const float *x, const float *y, const float *psi, const float *len,
                          const float *wid, const uint8_t *present, uint8_t *out, void *stream)
{
    if (!pow2_le64(A)) return bad("tde_compute_collision: A must be a power of two in [1,128]");
    if (B <= 0) return 0;
    const unsigned nb = blocks_for((int64_t)B * A);
    TDE_DISPATCH_A128(A, tde::collide_kernel<kA, false><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(
                          B, const_cast<float *>(x), const_cast<float *>(y), const_cast<float *>(psi), (float *)nullptr,
                          (const float *)nullptr, len, wid, present, (const float *)nullptr, 0.0f, out));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_compute_collision", e);
}

int tde_kin_collide_step(int32_t B, int32_t A, float *x, float *y, float *psi, float *v, const float *lr,
                         const float *len, const float *wid, const uint8_t *present, const float *action, float dt,
                         uint8_t *collided, void *stream)
{
    if (!pow2_le64(A)) return bad("tde_kin_collide_step: A must be a power of two in [1,128]");
    if (B <= 0) return 0;
    const unsigned nb = blocks_for((int64_t)B * A);
    TDE_DISPATCH_A128(A, tde::collide_kernel<kA, true><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(
                          B, x, y, psi, v, lr, len, wid, present, action, dt, collided));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_kin_collide_step", e);
}

int tde_compute_offroad(int32_t B, int32_t A, const float *x, const float *y, const float *psi, const float *len,
                        const float *wid, const uint8_t *present, const tde_world *world, const int32_t *map_of_env,
                        float threshold, uint8_t *out, void *stream)
{
    if (!world) return bad("tde_compute_offroad: world is NULL");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(tde::offroad_kernel, dim3(blocks_for((int64_t)B * A)), dim3(tde::kBlock), 0, (hipStream_t)stream,
                       B, A, x, y, psi, len, wid, present, *world, map_of_env, threshold, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_compute_offroad", e);
}

int tde_waypoint_reward(const tde_config *cfg, int32_t n, const float *pre_x, const float *pre_y, const float *pre_psi,
                        const float *pre_v, const float *x, const float *y, const float *psi, const float *v,
                        const uint8_t *offroad, const uint8_t *collided, const uint8_t *tl_violation,
                        const double *wp_xy, const int32_t *wp_n, int32_t NW, const int32_t *scn, int32_t *steps,
                        int32_t *target_idx, int32_t *reached, float *reward, uint8_t *terminated, uint8_t *truncated,
                        double *info, int32_t *info_reached, void *stream)
{
    if (!cfg) return bad("tde_waypoint_reward: cfg is NULL");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(tde::reward_kernel, dim3(blocks_for(n)), dim3(tde::kBlock), 0, (hipStream_t)stream, *cfg, n,
                       pre_x, pre_y, pre_psi, pre_v, x, y, psi, v, offroad, collided, tl_violation, wp_xy, wp_n, NW, scn,
                       steps, target_idx, reached, reward, terminated, truncated, info, info_reached);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_waypoint_reward", e);
}

static int check_env_args(const char *fn, const tde_config *cfg, const tde_world *w, const tde_state *st)
{
    if (!cfg || !w || !st) { snprintf(g_err, sizeof(g_err), "%s: NULL argument", fn); return (int)hipErrorInvalidValue; }
    if (!pow2_le64(st->A)) { snprintf(g_err, sizeof(g_err), "%s: A must be a power of two in [1,128]", fn); return (int)hipErrorInvalidValue; }
    if (w->A != st->A) { snprintf(g_err, sizeof(g_err), "%s: world.A (%d) != state.A (%d)", fn, w->A, st->A); return (int)hipErrorInvalidValue; }
    // sqrt_cr_f32 (the controller's braking-distance speed) is exact for arguments that are zero or in the normal fp32
    // range: amax times a length difference of metres is, for any sensible amax
    if ((cfg->flags & TDE_F_NPC) && !(cfg->npc_max_steer >= 0.0f)) {          // clampf(v, -smax, smax) needs lo <= hi
        snprintf(g_err, sizeof(g_err), "%s: config.npc_max_steer must be >= 0 (got %g)", fn, (double)cfg->npc_max_steer);
        return (int)hipErrorInvalidValue;
    }
    if ((cfg->flags & TDE_F_NPC) && !(cfg->npc_max_accel >= 1e-3f && cfg->npc_max_accel <= 1e3f)) {
        snprintf(g_err, sizeof(g_err), "%s: config.npc_max_accel must be in [1e-3, 1e3] m/s^2 (got %g)", fn, (double)cfg->npc_max_accel);
        return (int)hipErrorInvalidValue;
    }
    return 0;
}

int tde_env_reset(const tde_config *cfg, const tde_world *world, const tde_state *st, const uint8_t *mask,
                  void *stream)
{
    int rc = check_env_args("tde_env_reset", cfg, world, st);
    if (rc) return rc;
    if (st->B <= 0) return 0;
    const unsigned nb = blocks_for((int64_t)st->B * st->A);
    TDE_DISPATCH_A128(st->A, tde::env_reset_kernel<kA><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(*cfg, *world, *st, mask));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_env_reset", e);
}

// `load_slots`: the agent slots stepping on the device at the same time - the batch's own, or the whole batch's when this is one
// of the sub-batches tde_env_step_render runs side by side (the choice of kernel form is a matter of load)
// The (world, configuration) pairs whose first-step gap cache this process has filled: tde_env_step / tde_env_rollout fill it on
// first use.  A stale entry (the table's memory re-used by another world at the same address with the same tables) only costs
// speed: the kernels check every entry's key and fall back to the whole controller.
static std::atomic<uint64_t> g_fg_memo[16];
static std::atomic<unsigned> g_fg_next{0};
static uint64_t fg_memo_word(const tde_world &w, uint32_t hash)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint64_t x = (uint64_t)(uintptr_t)w.first_gap * 0x9E3779B97F4A7C15ull;
    x ^= ((uint64_t)hash << 8) ^ (uint64_t)(dev & 0xff);
    return x | 1ull;
}

static int first_gaps_launch(const tde_config *cfg, const tde_world *world, void *stream, bool use_memo)
{
    if (!world->first_gap || world->A > 2 * tde::kWave || world->n_scn <= 0) return 0;
    if (!(cfg->flags & TDE_F_NPC) || !(cfg->flags & TDE_F_NPC_FIRST_STEP)) return 0;
    const uint32_t hash = tde_host::act_cfg_hash(*cfg, *world);
    const uint64_t word = fg_memo_word(*world, hash);
    if (use_memo)
        for (auto &m : g_fg_memo)
            if (m.load(std::memory_order_relaxed) == word) return 0;
    const unsigned nb = blocks_for((int64_t)world->n_scn * world->A);
    TDE_DISPATCH_A128(world->A, tde::first_gap_kernel<kA><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(*cfg, *world, hash | 1u));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("tde_first_gaps", e);
    g_fg_memo[g_fg_next.fetch_add(1, std::memory_order_relaxed) % 16].store(word, std::memory_order_relaxed);
    return 0;
}

// ---- argument blocks of the 128-slot two-role one-step kernel (tde::StepArgs, tde_kernels.h) -----------------------------------------
// One IMMUTABLE block per distinct (configuration, world, state without its action pointer, controller hash) and device: a pinned
// host copy and its device twin, slots of two pools allocated on first use and never freed or rewritten - a HIP graph that captured a
// launch (or the block's upload) keeps reading valid memory for the life of the process.  A block is uploaded once per stream
// that uses it, ahead of the first launch on that stream (identical bytes: concurrent uploads are benign).  No block - the pools are
// full (kArgBlocks distinct argument sets), or the first use of an argument set (or of a stream) falls inside a stream capture, where
// neither an allocation nor an un-replayed upload can be relied on - means nullptr: the caller launches the one-role kernel, which
// takes its arguments by value.
namespace {
constexpr int kArgBlocks = 2048;
struct ArgEntry { int device; int slot; uint64_t hash; std::vector<void *> streams; };
struct ArgPool { tde::StepArgs *host = nullptr, *dev = nullptr; int used = 0; };
std::mutex g_arg_mu;
std::vector<ArgEntry> g_arg_entries;
ArgPool g_arg_pool[16];
}  // namespace

static const tde::StepArgs *step_args(const tde_config *cfg, const tde_world *world, const tde_state *st, uint32_t act_hash, void *stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    tde::StepArgs now;
    memset(&now, 0, sizeof(now));
    now.cfg = *cfg; now.w = *world; now.st = *st; now.st.action = nullptr; now.act_hash = act_hash;
    tde::fill_cold(now.cold, *cfg, *world);                              // (host-filled: the kernel reads it from the block)
    std::lock_guard<std::mutex> lock(g_arg_mu);
    ArgPool &pool = g_arg_pool[dev];
    ArgEntry *ent = nullptr;
    // a closed loop passes the same arguments call after call: the entry of the last call first, then the table behind a hash
    static size_t last = 0;
    if (last < g_arg_entries.size() && g_arg_entries[last].device == dev && memcmp(&pool.host[g_arg_entries[last].slot], &now, sizeof(now)) == 0)
        ent = &g_arg_entries[last];
    uint64_t hash = 0xcbf29ce484222325ull;
    if (!ent) {
        uint64_t words[sizeof(now) / 8];
        static_assert(sizeof(now) % 8 == 0, "StepArgs is a whole number of 8-byte words");
        memcpy(words, &now, sizeof(now));
        for (uint64_t x : words) hash = (hash ^ x) * 0x100000001b3ull;
        for (size_t i = 0; i < g_arg_entries.size(); ++i) {
            ArgEntry &e = g_arg_entries[i];
            if (e.device == dev && e.hash == hash && memcmp(&pool.host[e.slot], &now, sizeof(now)) == 0) { ent = &e; last = i; break; }
        }
    }
    if (ent)
        for (void *s : ent->streams)
            if (s == stream) return &pool.dev[ent->slot];
    // a new argument set, or a known one on a new stream: neither an allocation nor an upload that only a replay would execute can
    // be relied on inside a stream capture
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    if (!ent) {
        if (!pool.host) {
            if (hipHostMalloc((void **)&pool.host, sizeof(tde::StepArgs) * kArgBlocks, hipHostMallocDefault) != hipSuccess) { pool.host = nullptr; (void)hipGetLastError(); return nullptr; }
            if (hipMalloc((void **)&pool.dev, sizeof(tde::StepArgs) * kArgBlocks) != hipSuccess) { (void)hipHostFree(pool.host); pool.host = nullptr; (void)hipGetLastError(); return nullptr; }
        }
        if (pool.used >= kArgBlocks) return nullptr;
        memcpy(&pool.host[pool.used], &now, sizeof(now));
        g_arg_entries.push_back(ArgEntry{dev, pool.used++, hash, {}});
        last = g_arg_entries.size() - 1;
        ent = &g_arg_entries.back();
    }
    if (hipMemcpyAsync(&pool.dev[ent->slot], &pool.host[ent->slot], sizeof(tde::StepArgs), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    ent->streams.push_back(stream);
    return &pool.dev[ent->slot];
}

static int env_step_launch(const tde_config *cfg, const tde_world *world, const tde_state *st, void *stream, int64_t load_slots)
{
    int rc = check_env_args("tde_env_step", cfg, world, st);
    if (rc) return rc;
    if (st->B <= 0) return 0;
    if (!st->action) return bad("tde_env_step: state.action is NULL");
    if ((cfg->flags & TDE_F_OFFROAD) && TDE_STEP_CLS2 && !world->cell_cls2) return bad("tde_env_step: world.cell_cls2 is NULL (ABI 7 class map)");
    const bool lights = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    // With the lookup caches present (and a group shape the three-role kernels are built for) the step runs as three
    // wavefronts per 64 agent slots; tde_kernel_override forces one form (A/B runs).
    // Which one wins is a matter of load (bench.py --mode step --step-kernel solo|trio, us per step at 16 agents per env,
    // profiles/r03_d_step_matrix.txt):
    //   envs     2048   4096   8192   16384   32768
    //   3 roles  7.26   7.91   9.45   16.58   27.34
    //   1 role   9.19   9.37  10.76   14.56   24.25
    // (round 2: 3 roles 12.3 at 8192 envs - its re-spawn path, the tail every launch waits for, recomputed the next step's
    //  controller and walked the record -> route table chain: +3.4 us with TDE_F_AUTORESET; now +1).  Three roles up to
    // 131 072 agent slots, one role above (configs[4]: 8192 x 32).  tde_kernel_override(0, 1 | 3) forces one.
    const int force = g_force_step;
    // (slot entries pack route / replay ids into 20 bits and their lengths into 12: tde_abi.h)
    const int32_t id_max = (1 << TDE_CACHE_ID_BITS) - 1, len_max = 1 << (32 - TDE_CACHE_ID_BITS);
    const bool packable = world->n_routes < id_max && world->n_replay < id_max && world->RW < len_max && world->RT < len_max;
    const bool trio_ok = st->slot_cache && st->env_cache && packable && (st->A == 8 || st->A == 16 || st->A == 32);
    const bool want_trio = force == 3 || (force == 0 && load_slots <= 131072);
    // 128 agent slots per env: two roles (four wavefronts per env) when the action cache is there and the batch is at most HALF a residency
    // round of that form (2 workgroups per CU: 512 envs) - the next step's controller then runs beside the judges' sweeps.  us per step, ~122 agents
    // per env, one role / two roles (profiles/r06_p_wide_step_forms.txt): 1 env 13.4 / 8.9 (the reference's own operating point),
    // 64 envs 15.5 / 13.1, 256 16.1 / 14.4, 512 16.4 / 15.5, 1024 19.1 / 18.8 through ctypes but 17.8 / 18.5 through the extension (the sweeps' VALU
    // work fills the chip either way), 2048 29.1 / 31.9.
    // tde_kernel_override(0, 1) forces the one-role kernel, (0, 2) the two-role one at any batch size.
    if (st->A == 128 && st->act_cache && (force == 2 || (force != 1 && st->B <= 4 * cu_count()))) {
        tde_state local = *st;
        if (!packable) local.slot_cache = nullptr;                           // (the kernel then walks the table chain)
        const uint32_t hash = tde_host::act_cfg_hash(*cfg, *world);
        if (const tde::StepArgs *args = step_args(cfg, world, &local, hash, stream)) {
            rc = first_gaps_launch(cfg, world, stream, true);                // (first use of this world with this configuration)
            if (rc) return rc;
            // eight wavefronts per env while the batch leaves the CUs issue slots to spare (half a residency round), four above;
            // tde_kernel_override(0, 2) = the four-wavefront form at any batch size
            const int waves = (force != 2 && TDE_WIDE_STEP_WAVES8 && st->B <= 2 * cu_count()) ? 8 : 4;
            return tde_host::launch_step_wide(args, cfg, st, waves, stream);
        }
    }
    if (trio_ok && want_trio) {
        rc = first_gaps_launch(cfg, world, stream, true);                    // (first use of this world with this configuration)
        if (rc) return rc;
        return tde_host::launch_step_trio(cfg, world, st, tde_host::act_cfg_hash(*cfg, *world), stream);
    }
    // the one-role kernel (its forms - the class map on large grids, four wavefronts per SIMD for big 128-slot batches - are chosen
    // by the launcher); with / without tde_state.magnitudes are two translation units
    return st->magnitudes ? tde_host::launch_step_solo_mag(cfg, world, st, stream) : tde_host::launch_step_solo(cfg, world, st, stream);
}

int tde_env_step(const tde_config *cfg, const tde_world *world, const tde_state *st, void *stream)
{
    return env_step_launch(cfg, world, st, stream, st ? (int64_t)st->B * st->A : 0);
}

// env arrays advanced by e0 envs, agent arrays by e0 * A slots: the shard [e0, e0 + n) of a batch as a tde_state of its own
static tde_state state_slice(const tde_state &s, int64_t e0, int32_t n)
{
    tde_state t = s;
    const int64_t g0 = e0 * s.A;
#define TDE_ADV(p, k) if (t.p) t.p += (k)
    TDE_ADV(x, g0); TDE_ADV(y, g0); TDE_ADV(psi, g0); TDE_ADV(v, g0); TDE_ADV(len, g0); TDE_ADV(wid, g0); TDE_ADV(lr, g0);
    TDE_ADV(vdes, g0); TDE_ADV(route_wp, g0); TDE_ADV(present, g0); TDE_ADV(collided, g0); TDE_ADV(offroad, g0);
    TDE_ADV(scn, e0); TDE_ADV(steps, e0); TDE_ADV(target_idx, e0); TDE_ADV(reached, e0); TDE_ADV(episode, e0);
    TDE_ADV(action, 2 * e0); TDE_ADV(reward, e0); TDE_ADV(terminated, e0); TDE_ADV(truncated, e0); TDE_ADV(tl_violation, e0);
    TDE_ADV(info, 4 * e0); TDE_ADV(info_reached, e0); TDE_ADV(done_bits, e0); TDE_ADV(obs, 8 * e0); TDE_ADV(ep_return, e0);
    TDE_ADV(ep_final, e0); TDE_ADV(ep_final_len, e0); TDE_ADV(slot_cache, g0); TDE_ADV(env_cache, e0); TDE_ADV(act_cache, g0 + e0); TDE_ADV(magnitudes, 4 * e0);
#undef TDE_ADV
    t.B = n;
    return t;
}

static int rollout_launch(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro, int team,
                          void *stream)
{
    if (team != 1) {                                                         // (the role-split kernels read the first-step gap cache)
        const int rc = first_gaps_launch(cfg, world, stream, true);
        if (rc) return rc;
    }
    if (team == 3) return tde_host::launch_rollout_trio(cfg, world, st, ro, stream);
    if (team == 1) return tde_host::launch_rollout_solo(cfg, world, st, ro, stream);
    return tde_host::launch_rollout_duo(cfg, world, st, ro, stream);
}

int tde_env_rollout(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_rollout *ro,
                    void *stream)
{
    int rc = check_env_args("tde_env_rollout", cfg, world, st);
    if (rc) return rc;
    if (!ro) return bad("tde_env_rollout: rollout is NULL");
    if (st->B <= 0 || ro->K <= 0) return 0;
    if (!ro->actions) return bad("tde_env_rollout: rollout.actions is NULL");
    if (ro->ldb != 0 && ro->ldb < st->B) return bad("tde_env_rollout: rollout.ldb must be 0 (= B) or >= B");
    if (st->A > 64) {
        // 128 slots per env: the one-role persistent kernel with the env's two wavefronts as its workgroup (the role-split kernels
        // keep an env inside one wavefront)
        tde_rollout r128 = *ro;
        if (r128.ldb == 0) r128.ldb = st->B;
        // Two roles at every batch size - us per step at ~122 agents per env, 256 / 1024 / 2048 / 4096 envs: two roles (four wavefronts
        // per env) 6.2 / 8.9 / 18.0 / 34.7, one role 10.6 / 11.4 / 23.3 / 38.9 (profiles/r04_z_wide2_waves.txt; round 6: 7.3 at 1024 envs).
        // tde_kernel_override(1, 0) forces the one-role kernel.
        if (g_force_rollout == 1) return tde_host::launch_rollout_solo(cfg, world, st, &r128, stream);
        const tde::StepArgs *args = nullptr;
#if TDE_WIDE_ROLLOUT_BLOCK       // (A/B: the arguments from a block in device memory, as the one-step kernel's)
        args = step_args(cfg, world, st, 0u, stream);
        if (!args) return tde_host::launch_rollout_solo(cfg, world, st, &r128, stream);
#endif
        // eight wavefronts per env while the batch leaves the CUs issue slots to spare (half a residency round), four above - us per
        // step, four / eight: 1 env 5.05 / 3.49, 64 envs 5.22 / 3.69, 256 envs 5.35 / 3.81, 512 envs 5.42 / 4.01, 256 envs with lights
        // 6.61 / 5.01 (profiles/r06_z_wide_128.txt); tde_kernel_override(2, 0) = the four-wavefront form at any batch size
        const int waves = (g_force_rollout != 2 && TDE_WIDE_ROLLOUT_WAVES8 && st->B <= 2 * cu_count()) ? 8 : 4;
        return tde_host::launch_rollout_wide(args, cfg, world, st, &r128, waves, stream);
    }
    // Which persistent kernel: one, two or three wavefronts per group of 64 agent slots (tde_kernel_override(1 | 2 | 3, 0)
    // forces one; a forced trio still needs 8, 16 or 32 agents per env).  Interleaved same-process A/B, 40 launches each, median
    // us per step (scripts/ab_rollout.py duo:... trio:..., profiles/r02_e_rollout_matrix.txt): three roles win at 8 and 16
    // agents per env, without traffic lights (3.17 vs 3.50, 3.06 vs 3.54) and with them (5.32 vs 5.55, 4.55 vs 5.01), and
    // at 32 without lights (3.70 vs 4.11); at 32 WITH lights the 32-row sweeps plus the stop-line loops spill under the
    // 80-VGPR cap and the two-role kernel (128 VGPRs, four wavefronts per SIMD) is faster (5.18 vs 7.66); at 64 the two
    // are equal (5.03) and two roles run.
    const int forced = g_force_rollout;
    const bool lights0 = (cfg->flags & TDE_F_TRAFFIC_LIGHTS) != 0;
    const bool trio_shape = st->A == 8 || st->A == 16 || st->A == 32;
    int team = forced ? forced : (st->A == 8 || st->A == 16 || (st->A == 32 && !lights0)) ? 3 : 2;
    if (team == 3 && !trio_shape) team = 2;
    tde_rollout r = *ro;
    if (r.ldb == 0) r.ldb = st->B;
    // The two- and three-role kernels are tuned for ONE residency round of the chip: 8 workgroups (groups of 64 agent slots)
    // per CU - 8192 envs x 16 agents on 256 CUs.  A larger batch as one grid runs its later rounds badly (16 384 envs:
    // 1784 us per 250-step launch against 2 x 737; profiles/r03_c_scale_envs.txt), so it is cut into consecutive launches
    // of one round each on the same stream: envs are independent, every launch runs at the tuned shape, and the last,
    // partial one is simply a smaller batch.  (A batch of less than 1.5 rounds stays one launch: its few extra workgroups
    // slip in as the first ones finish.)
    const int64_t groups = ((int64_t)st->B * st->A + tde::kWave - 1) / tde::kWave;
    const int64_t round = 8 * (int64_t)cu_count();
    if (team == 1 || 2 * groups < 3 * round) return rollout_launch(cfg, world, st, &r, team, stream);
    const int64_t envs_per_round = round * (tde::kWave / st->A);
    for (int64_t e0 = 0; e0 < st->B; e0 += envs_per_round) {
        const int32_t n = (int32_t)((st->B - e0 < envs_per_round) ? st->B - e0 : envs_per_round);
        tde_config c = *cfg;
        c.env_base = cfg->env_base + (uint32_t)e0;                           // the reset RNG is keyed by the global env index
        const tde_state s = state_slice(*st, e0, n);
        tde_rollout rr = r;
        rr.actions = r.actions + 2 * e0;
        if (rr.reward) rr.reward += e0;
        if (rr.done) rr.done += e0;
        rc = rollout_launch(&c, world, &s, &rr, team, stream);
        if (rc) return rc;
    }
    return 0;
}

// the argument checks of a render request, without launching anything (tde_render_ego; tde_env_step_render runs them before
// its first launch, so that a bad request leaves the state of every sub-batch untouched)
static int check_render_args(const char *who, const tde_world *world, const tde_render *rd)
{
    char msg[200];
    auto say = [&](const char *what) { snprintf(msg, sizeof(msg), "%s: %s", who, what); return bad(msg); };
    if (!rd || !rd->out) return say("render/out is NULL");
    // (the layer plane in LDS holds the image rounded up to multiples of 8 in both directions)
    if (rd->H <= 0 || rd->W <= 0 || (rd->W % 4) != 0 || (rd->H % 4) != 0 || (rd->H * rd->W) % 16 != 0 ||
        ((rd->H + 7) & ~7) * ((rd->W + 7) & ~7) > tde::kRasterMaxPix || rd->H > 256 || rd->W > 256)
        return say("H and W must be positive multiples of 4 (at most 256) whose product, each rounded up to a multiple of 8, is <= 4096");
    if (rd->phase < 0) return say("phase must be >= 0 (keep it reduced modulo n_stack)");
    if (!(rd->fov > 0.0f)) return say("fov must be positive");
    if (!world->cell_cls2 || !world->cell_sub || !world->cell_word || !world->cell_tri || !world->cell_coarse) return say("the world has no grid index tables");
    return 0;
}

int tde_render_ego(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_render *rd,
                   void *stream)
{
    int rc = check_env_args("tde_render_ego", cfg, world, st);
    if (rc) return rc;
    rc = check_render_args("tde_render_ego", world, rd);
    if (rc) return rc;
    if (st->B <= 0) return 0;
    if (rd->n_stack > 1 && !rd->layers && !rd->only)       // (a masked call re-renders the newest frame in place)
        tde::frame_shift_kernel<<<st->B, tde::kBlock, 0, (hipStream_t)stream>>>(rd->out, rd->H * rd->W, rd->n_stack);
    tde::RenderArgs ra;
    ra.maps = world->maps; ra.cell_word = world->cell_word; ra.cell_tri = world->cell_tri; ra.scn_tab = world->scn;
    ra.wp_xy = world->wp_xy; ra.stoplines = world->stoplines; ra.phases = world->phases;
    ra.x = st->x; ra.y = st->y; ra.psi = st->psi; ra.len = st->len; ra.wid = st->wid; ra.present = st->present;
    ra.scn = st->scn; ra.steps = st->steps; ra.target_idx = st->target_idx;
    ra.rd = *rd;
    ra.thr2 = cfg->offroad_threshold_squared ? cfg->offroad_threshold : cfg->offroad_threshold * cfg->offroad_threshold;
    ra.flags = cfg->flags; ra.NW = world->NW; ra.A = st->A;
    const float res = rd->fov / (float)rd->W;
    ra.K8 = tde::raster_block_clearance(8, res); ra.K4 = tde::raster_block_clearance(4, res);
    ra.cell_cls2 = world->cell_cls2; ra.cell_sub = world->cell_sub; ra.cell_coarse = world->cell_coarse;
    const int vpg = tde::kViewsPerGroup * TDE_RENDER_VPWAVE;
    const unsigned ng = (unsigned)((st->B + vpg - 1) / vpg);
    if (rd->H == 64 && rd->W == 64) tde::render_views_kernel<64><<<ng, tde::kWave * tde::kViewsPerGroup, 0, (hipStream_t)stream>>>(ra, st->B);
    else tde::render_views_kernel<0><<<ng, tde::kWave * tde::kViewsPerGroup, 0, (hipStream_t)stream>>>(ra, st->B);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_render_ego", e);
}

int tde_env_reset_render(const tde_config *cfg, const tde_world *world, const tde_state *st, const uint8_t *mask,
                         const tde_render *rd, void *stream)
{
    int rc = check_env_args("tde_env_reset_render", cfg, world, st);
    if (rc) return rc;
    if (!mask) return bad("tde_env_reset_render: mask is NULL (a full reset renders with tde_render_ego)");
    rc = check_render_args("tde_env_reset_render", world, rd);   // (before the reset: a bad request re-spawns nothing)
    if (rc) return rc;
    rc = tde_env_reset(cfg, world, st, mask, stream);
    if (rc) return rc;
    tde_render r = *rd;
    r.fresh = mask;                                              // the re-spawned views' older frames restart blank,
    r.only = mask;                                               // only their newest frame is rendered again, in place
    return tde_render_ego(cfg, world, st, &r, stream);
}

int tde_env_step_render(const tde_config *cfg, const tde_world *world, const tde_state *st, const tde_render *rd,
                        void *const *streams, int32_t n_streams)
{
    int rc = check_env_args("tde_env_step_render", cfg, world, st);
    if (rc) return rc;
    if (!streams || n_streams < 1 || n_streams > 16) return bad("tde_env_step_render: streams is NULL or n_streams not in [1, 16]");
    if (rd) {                                              // before the first launch: a failing call advances no sub-batch
        rc = check_render_args("tde_env_step_render", world, rd);
        if (rc) return rc;
    }
    if (st->B <= 0) return 0;
    // equal shares rounded up to whole groups of 64 envs (any A: slices start on wavefront and workgroup boundaries)
    const int64_t share = ((((int64_t)st->B + n_streams - 1) / n_streams) + 63) & ~(int64_t)63;
    int i = 0;
    for (int64_t e0 = 0; e0 < st->B; e0 += share, ++i) {
        const int32_t n = (int32_t)((st->B - e0 < share) ? st->B - e0 : share);
        tde_config c = *cfg;
        c.env_base = cfg->env_base + (uint32_t)e0;                           // the reset RNG is keyed by the global env index
        const tde_state s = state_slice(*st, e0, n);
        // (the sub-batches run side by side: configs[4] in two halves, us per timestep: one-role steps 46.0, three-role 49.1,
        //  profiles/r03_f_config5_streams_matrix.txt - the kernel form follows the whole batch's load)
        rc = env_step_launch(&c, world, &s, streams[i], (int64_t)st->B * st->A);
        if (rc) return rc;
        if (rd) {
            tde_render r = *rd;
            const int64_t ns = rd->n_stack > 1 ? rd->n_stack : 1, plane = (int64_t)rd->H * rd->W;
            if (r.out) r.out += e0 * 3 * ns * plane;
            if (r.layers) r.layers += e0 * ns * plane;
            if (r.fresh) r.fresh += e0;
            if (r.only) r.only += e0;
            rc = tde_render_ego(&c, world, &s, &r, streams[i]);
            if (rc) return rc;
        }
    }
    return 0;
}

int tde_ego_infractions(const tde_config *cfg, const tde_world *world, const tde_state *st, float *out, void *stream)
{
    int rc = check_env_args("tde_ego_infractions", cfg, world, st);
    if (rc) return rc;
    if (!out) return bad("tde_ego_infractions: out is NULL");
    if (st->B <= 0) return 0;
    const unsigned nb = (unsigned)((st->B + (tde::kBlock / tde::kWave) - 1) / (tde::kBlock / tde::kWave));
    tde::ego_infractions_kernel<<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(*cfg, *world, *st, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_ego_infractions", e);
}

int tde_env_post_step(const tde_config *cfg, const tde_world *world, const tde_state *st, float *magnitudes, void *stream)
{
    int rc = check_env_args("tde_env_post_step", cfg, world, st);
    if (rc) return rc;
    if (st->B <= 0) return 0;
    if (!st->terminated || !st->truncated || !st->collided || !st->offroad) return bad("tde_env_post_step: the state lacks the step's flag arrays");
    const unsigned nb = (unsigned)((st->B + (tde::kBlock / tde::kWave) - 1) / (tde::kBlock / tde::kWave));
    TDE_DISPATCH_A128(st->A, tde::env_post_step_kernel<kA><<<nb, tde::kBlock, 0, (hipStream_t)stream>>>(*cfg, *world, *st, magnitudes));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_env_post_step", e);
}

int tde_first_gaps(const tde_config *cfg, const tde_world *world, void *stream)
{
    if (!cfg || !world) return bad("tde_first_gaps: NULL argument");
    if (!pow2_le64(world->A)) return bad("tde_first_gaps: world.A must be a power of two in [1,128]");
    return first_gaps_launch(cfg, world, stream, false);
}

int tde_state_obs(const tde_world *world, const tde_state *st, float *out, void *stream)
{
    if (!world || !st || !out) return bad("tde_state_obs: world/state/out is NULL");
    if (st->B <= 0) return 0;
    if (!st->x || !st->y || !st->psi || !st->v || !st->scn || !st->target_idx || !st->steps || !world->scn || !world->wp_xy)
        return bad("tde_state_obs: a required state / world pointer is NULL");
    tde::state_obs_kernel<<<blocks_for(st->B), tde::kBlock, 0, (hipStream_t)stream>>>(*world, *st, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("tde_state_obs", e);
}

}  // extern "C"

// host-side table build (no kernels): tde_grid_build / tde_grid_free
#include "tde_gridbuild.h"
