/*
 * tde_hip.h — C-ABI of libtde_hip.so: the MI355X (gfx950) implementation of the per-timestep env step path of
 * inverted-ai/torchdriveenv.  Plain pointers (DEVICE memory unless stated), sizes, and a HIP stream passed as
 * void* (hipStream_t; NULL = the default stream).  No torch types.  All entry points are asynchronous on the given
 * stream, allocate nothing, and return 0 on success or a hipError_t value (message: tde_last_error()).
 *
 * Each entry point names the reference interface it replaces (file:line into /root/reference/torchdriveenv/).
 * The reference has no FFI of its own (it is pure Python over torchdrivesim); INTEGRATION.md shows the ctypes stub a
 * maintainer would add and how GymEnv/WaypointSuiteEnv would call it.
 *
 * Struct arguments (tde_config, tde_world, tde_state, tde_rollout: include/tde_abi.h) are HOST structs whose pointer
 * members are device pointers; they are copied by value into the kernel arguments at launch.
 */
#ifndef TDE_HIP_H
#define TDE_HIP_H

#include "tde_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TDE_API __attribute__((visibility("default")))

/* TDE_ABI_VERSION the library was built with. */
TDE_API int tde_abi_version(void);

/* Message of the last failing call on this thread ("" if none). */
TDE_API const char *tde_last_error(void);

/* Test / tuning hook (no reference counterpart): tde_env_rollout and tde_env_step each have several kernel forms (one, two
 * or three wavefronts per 64 agent slots) and pick one by group shape and batch size; this forces a form for the calling
 * process - 0 = automatic (default), rollout_team 1 | 2 | 3, step_team 1 | 3 (2 at 128 slots) - so that every form can be held against the
 * oracle and A/B-timed.  A forced three-wavefront form still needs 8, 16 or 32 agents per env; at 128 agent slots per env
 * rollout_team / step_team 1 = the one-role kernel, 2 = the two-role kernel in its four-wavefront form at any batch size (the
 * library itself takes the eight-wavefront form up to half a residency round), 3 leaves the choice as it is.  The two values are atomics: a call from one thread while another launches is a race on the choice, not on memory. */
TDE_API int tde_kernel_override(int rollout_team, int step_team);

/* ---- operator level: the SimulatorInterface methods GymEnv calls ------------------------------------ */

/* simulator.step(action) restricted to the kinematic model — KinematicBicycle.step for n agents.
 * Replaces: gym_env.py:117 (simulator.step), model built at gym_env.py:245-247 (lr = rear_axis_offset).
 * x,y,psi,v: in/out [n]; lr [n]; present [n] or NULL; action [n][2] = (acceleration, steering). */
TDE_API int tde_kinematics_step(int64_t n, float *x, float *y, float *psi, float *v, const float *lr,
                                const uint8_t *present, const float *action, float dt, void *stream);

/* simulator.compute_collision() > 0 per agent: strict OBB overlap with any other present agent of the same env.
 * Replaces: gym_env.py:143, 415, 428 (CollisionMetric.nograd, gym_env.py:48).  All arrays [B*A]; out u8 [B*A]. */
TDE_API int tde_compute_collision(int32_t B, int32_t A, const float *x, const float *y, const float *psi,
                                  const float *len, const float *wid, const uint8_t *present, uint8_t *out,
                                  void *stream);

/* simulator.compute_offroad() > 0 per agent: a box corner farther than `threshold` from the drivable mesh of the
 * env's map.  Replaces: gym_env.py:142, 415, 427 (mesh: gym_env.py:184,260).  map_of_env [B]; out u8 [B*A]. */
TDE_API int tde_compute_offroad(int32_t B, int32_t A, const float *x, const float *y, const float *psi,
                                const float *len, const float *wid, const uint8_t *present, const tde_world *world,
                                const int32_t *map_of_env, float threshold, uint8_t *out, void *stream);

/* Fused kinematics + all-pairs collision for every agent (BASELINE.json configs[1]: "bicycle kinematics + OBB
 * collision only").  action [B*A][2].  Replaces: gym_env.py:117 followed by :143. */
TDE_API int tde_kin_collide_step(int32_t B, int32_t A, float *x, float *y, float *psi, float *v, const float *lr,
                                 const float *len, const float *wid, const uint8_t *present, const float *action,
                                 float dt, uint8_t *collided, void *stream);

/* The part of the step the reference owns, batched over n envs: environment_steps += 1 (gym_env.py:116),
 * WaypointSuiteEnv.get_reward (:396-411), check_reach_target (:391-394), is_terminated (:413-417), is_truncated
 * (:134-135), get_info terms (:419-437), waypoint advance (:378-383).
 * pre_* = ego state before simulator.step (last_x.. :371-375), x.. = after; flags u8 [n]; tl_violation may be NULL.
 * wp_xy f64 [S][NW][2], wp_n [S], scn [n]; steps/target_idx/reached in/out [n];
 * info f64 [n][4] = psi_smoothness, speed_smoothness, psi_reward, dist_reward (or NULL); info_reached [n] or NULL. */
TDE_API int tde_waypoint_reward(const tde_config *cfg, int32_t n, const float *pre_x, const float *pre_y,
                                const float *pre_psi, const float *pre_v, const float *x, const float *y,
                                const float *psi, const float *v, const uint8_t *offroad, const uint8_t *collided,
                                const uint8_t *tl_violation, const double *wp_xy, const int32_t *wp_n, int32_t NW,
                                const int32_t *scn, int32_t *steps, int32_t *target_idx, int32_t *reached,
                                float *reward, uint8_t *terminated, uint8_t *truncated, double *info,
                                int32_t *info_reached, void *stream);

/* ---- env level: the fused hot path ---------------------------------------------------------------------------- */

/* WaypointSuiteEnv.reset for the envs selected by mask (u8 [B], NULL = all): scenario draw, start pose/speed/heading
 * noise, spawn of NPC slots, counters.  Replaces: gym_env.py:319-367 + the initial tensors of build_simulator
 * (:192-198, :241-247).  Network parts (IAI initialize, :236-238) are out of scope. */
TDE_API int tde_env_reset(const tde_config *cfg, const tde_world *world, const tde_state *state, const uint8_t *mask,
                          void *stream);

/* One timestep of every env in ONE kernel: SingleAgentWrapper.step -> WaypointSuiteEnv.step -> GymEnv.step
 * (gym_env.py:453-461, 369-389, 115-120): bicycle kinematics, heuristic NPC controller (in place of the IAI call,
 * :285-294), replay override (:275-283), all-pairs OBB collision, drivable-mesh offroad, WaypointSuite reward,
 * termination/truncation, info, waypoint advance, and (TDE_F_AUTORESET) in-place re-spawn of finished envs.
 * Reads state->action [B][2]; writes state in place.
 * (At 128 agent slots the kernel reads (cfg, world, state) from an immutable argument block the library keeps in device memory, one
 *  per distinct argument set, uploaded on `stream` at first use; every other form takes them by value.  INTEGRATION.md, section 1.) */
TDE_API int tde_env_step(const tde_config *cfg, const tde_world *world, const tde_state *state, void *stream);

/* K consecutive timesteps with the ego actions taken from a resident [K][B][2] buffer (open-loop / action-repeat
 * rollouts; how bench.py drives the path without a host round trip per step).  Per-step reward [K][B] and done bits
 * [K][B] are written to `rollout`. */
TDE_API int tde_env_rollout(const tde_config *cfg, const tde_world *world, const tde_state *state,
                            const tde_rollout *rollout, void *stream);

/* GymEnv.get_obs / render: simulator.render_egocentric() for the ego of every env -> uint8 [B][3*n_stack][H][W]
 * (channels first, obs space gym_env.py:95; frame stack as VecFrameStack(n_stack=3, channels_order="first"),
 * examples/rl_training.py:160).  Replaces: gym_env.py:122-124, 152-155.  Layer/palette definition: tde_abi.h. */
TDE_API int tde_render_ego(const tde_config *cfg, const tde_world *world, const tde_state *state,
                           const tde_render *render, void *stream);

/* The re-spawn of the envs an SB3-style auto-reset has just seen finish, and their first observation, in ONE call: tde_env_reset
 * for the envs with mask[e] != 0 (uint8 [B], required) followed by tde_render_ego of exactly those views - their newest frame
 * rendered again in place, their older stack frames blanked (render->fresh and render->only are set to `mask` by the call, so its entries must be 0 or 1: the rasteriser reads bits 0-1 of a fresh byte; phase
 * = the phase of the LAST full render).  Replaces: the reset() + get_obs() a VecEnv issues for finished envs
 * (gym_env.py:319-349, 122-124; examples/rl_training.py:159-160). */
TDE_API int tde_env_reset_render(const tde_config *cfg, const tde_world *world, const tde_state *state, const uint8_t *mask,
                                 const tde_render *render, void *stream);

/* One timestep + (render != NULL) the birdview of every env, as n_streams contiguous sub-batches of the batch, sub-batch i
 * launched on streams[i] (hipStream_t; the step, then the rasteriser): GymEnv.step followed by get_obs (gym_env.py:115-124)
 * for every env, the step of one sub-batch overlapping the rasteriser of another - both kernels are latency-bound alone
 * (DESIGN.md section 5, round 3).  Results are those of tde_env_step + tde_render_ego on the whole batch (envs are independent
 * and the reset RNG is keyed by the global env index: each sub-batch runs with env_base advanced to its first env).
 * Sub-batches are cut at multiples of 64 envs.  The caller orders the streams against the producer of state->action and the
 * consumer of the outputs (events); consecutive calls on the same streams are ordered per sub-batch, which is all the
 * path needs.  n_streams in [1, 16]. */
TDE_API int tde_env_step_render(const tde_config *cfg, const tde_world *world, const tde_state *state,
                                const tde_render *render, void *const *streams, int32_t n_streams);

/* Compact kinematic observation of every env's ego, float32 [B][8] (no reference counterpart; the reference only has
 * the birdview of gym_env.py:122-124): x, y, psi, v, offset of the current target waypoint in the ego frame (forward,
 * left; 0 when the route is finished, gym_env.py:378-383), 1 while a target exists, environment_steps.  One launch
 * instead of a dozen framework ops per step in closed-loop use. */
TDE_API int tde_state_obs(const tde_world *world, const tde_state *state, float *out, void *stream);

/* Infraction MAGNITUDES of every env's ego, float32 [B][4] = (offroad, collision, number of overlapping agents, 0): what the
 * reference's info dict holds under "offroad" / "collision" - simulator.compute_offroad() and compute_collision() for the exposed
 * agent, gym_env.py:427-428 (Monitor logs them, examples/rl_training.py:128) - where the step path only needs `> 0`.  offroad = sum
 * over the four box corners of clamp(dist - offroad_threshold, min = 0), dist = distance of the corner to the drivable mesh (the
 * SQUARED distance under tde_config.offroad_threshold_squared); collision = sum over the other present agents whose box overlaps
 * the ego's of the IoU of the two boxes (CollisionMetric.nograd's published form, gym_env.py:48).  Evaluated on the state as it
 * is: call it after a step WITHOUT TDE_F_AUTORESET and before tde_env_reset re-spawns the finished envs.  Cost: a corner within a
 * few metres of the mesh is a handful of table look-ups; the nearest triangle of a corner r metres away is found by scanning the
 * grid cells of a square of side ~ 2r around it - microseconds up to ~ 10 m - and, once that square would hold more cells than
 * (eight times) the map has triangles, by the definition itself: the minimum over all the map's triangles (tde_world.tri), 64 per
 * trip - tens of microseconds for an ego that keeps driving hundreds of metres off a junction map under
 * terminated_at_infraction = 0 (a millisecond by the scan alone).  The same holds for tde_state.magnitudes of tde_env_step. */
TDE_API int tde_ego_infractions(const tde_config *cfg, const tde_world *world, const tde_state *state, float *out, void *stream);

/* What follows a step that was launched WITHOUT TDE_F_AUTORESET, in one launch (one wavefront per env):
 *  (a) magnitudes != NULL: magnitudes[e] = tde_ego_infractions' four values for the state that step left - computed only for the
 *      envs whose ego the step flagged (state.collided / state.offroad of slot 0; a magnitude is zero without its flag), so the
 *      flags in `state` must be those of that step;
 *  (b) with TDE_F_AUTORESET in config->flags: tde_env_reset for the envs the step finished (terminated | truncated; the two
 *      arrays keep the step's values) and, when state.obs is set, the compact observation of their new episode.
 * step (no auto-reset) + this = the one-launch step's results, plus the reference's info["offroad" | "collision"] values
 * (gym_env.py:427-428).  Replaces: simulator.compute_offroad() / compute_collision() for the exposed agent followed by the
 * reset() a VecEnv issues for finished envs (gym_env.py:319-349; examples/rl_training.py:159-160). */
TDE_API int tde_env_post_step(const tde_config *config, const tde_world *world, const tde_state *state, float *magnitudes, void *stream);

/* (ABI 11) Fills the world's first-step gap cache (tde_world.first_gap, see tde_first_gap in tde_abi.h) for `config`: per (scenario, NPC
 * slot) what the NPC controller finds on the FIRST step of an episode apart from the ego - min(leader gap over the scenario's other
 * NPCs at their spawn poses, gap to a stop line that is red at step one) - keyed by the controller hash of (config, world).  With it
 * the role-split kernels give a re-spawned env its first NPC actions (TDE_F_NPC_FIRST_STEP; the reference's NPCs are driven from the
 * first simulator.step: IAIWrapper, gym_env.py:285-294) from one exact test against the ego's row instead of the controller's sweep;
 * without it (or with entries of another configuration) they run the whole controller: same results either way.
 * tde_env_step / tde_env_rollout call this themselves, on their stream, the first time they meet a (world, configuration) pair (a
 * small per-process memo of the pairs already served): a caller only needs it to choose WHEN the launch happens (e.g. outside a
 * stream capture).  One launch of n_scn * A lanes; up to 64 agent slots per env (the 128-slot kernels do not use the cache: no-op). */
TDE_API int tde_first_gaps(const tde_config *config, const tde_world *world, void *stream);

/* ---- host side: static tables ------------------------------------------------------------------------------------ */

/* Offroad grid index of ONE drivable mesh - what the simulator prepares once per map from the road mesh it is constructed
 * with (Simulator(road_mesh=map_cfg.road_mesh), gym_env.py:184, 260) so that compute_offroad (:142) is a cell look-up plus a
 * few candidate triangles instead of a pass over the mesh.  HOST pointers, no GPU involved, synchronous, n_threads host
 * threads (0 = all).  tri = [n_tri][6] fp32 vertices ax,ay,bx,by,cx,cy (the fp32 values the kernels and the oracle see);
 * threshold = the effective offroad distance in metres (sqrt of the threshold under offroad_threshold_squared); cell = cell
 * edge; margin = classification margin (0.05: absorbs fp32 evaluation at coordinates of kilometres); near_range (ABI 10) = how far
 * beyond the threshold, in metres, the coarse tiles carry a NEAR LIST (tde_world.tile_near: the triangles among which the nearest
 * one of any point of the tile is found - what the MAGNITUDE of the offroad infraction needs, gym_env.py:427; 0 = none, the
 * kernels then scan the grid around a corner; a few metres cover an ego that left the road within the last step or two).  Classes and lists are
 * conservative (csrc/tde_gridbuild.h), so the kernels' masks equal a brute-force pass over every triangle.  The result is
 * owned by the library until tde_grid_free. */
TDE_API int tde_grid_build(const float *tri, int32_t n_tri, float threshold, float cell, float margin, float near_range,
                           int32_t n_threads, tde_grid **out);
TDE_API void tde_grid_free(tde_grid *grid);

#ifdef __cplusplus
}
#endif
#endif /* TDE_HIP_H */
