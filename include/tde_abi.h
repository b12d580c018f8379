/*
 * tde_abi.h — plain-C data contract of the batched driving-env step path.
 *
 * Shared by the HIP library (include/tde_hip.h -> libtde_hip.so, device pointers) and by the CPU
 * oracle (oracle/tde_oracle.c -> libtde_oracle.so, host pointers).  It is an interface description
 * only: no algorithm lives here.  Every struct is laid out with natural alignment and explicit
 * padding so that a ctypes.Structure (torchdriveenv_amd/_abi.py) mirrors it field for field.
 *
 * Vocabulary follows the reference (inverted-ai/torchdriveenv):
 *   env      one simulator instance           (reference: one WaypointSuiteEnv, gym_env.py:303)
 *   agent    one vehicle slot of an env; slot 0 is the ego, the rest are NPCs
 *            (reference: npc_mask = all True except index 0, gym_env.py:269-271)
 *   state    (x, y, psi, v) per agent         (reference: simulator.get_state() -> (B,A,4), gym_env.py:371-375)
 *   attrs    (length, width, rear_axis_offset) (reference: agent_attributes (B,A,3), gym_env.py:241-247,261)
 *   scenario one entry of a WaypointSuite      (reference: gym_env.py:63-68, data/validation_cases.yml)
 *   replay   pre-recorded NPC trajectory       (reference: replay_states/replay_mask, gym_env.py:275-283)
 *
 * Layout: struct-of-arrays, env-major.  Agent arrays have B*A elements, element (e,a) at e*A + a.
 * A must be a power of two, 1..128.  Up to 64 an env never straddles a 64-lane wavefront and every kernel form applies (the
 * persistent rollout kernels, the three-role forms); A = 128 (the reference assembles up to ~100 agents per env,
 * gym_env.py:216-237; pad with absent slots) spans two wavefronts of a workgroup: the one-role kernels, whose sweeps take the
 * env's rows as two halves of 64 with a 64-bit candidate mask each (same results), and a persistent rollout kernel of their own (two roles
 * in four wavefronts per env).
 */
#ifndef TDE_ABI_H
#define TDE_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDE_ABI_VERSION 11
#define TDE_MAX_AGENTS 128

/* feature bits of tde_config.flags */
#define TDE_F_NPC        (1u << 0)  /* heuristic NPC controller drives slots 1..A-1 (else they coast: zero action,
                                       which is what NPCWrapper feeds the kinematic model before teleporting) */
#define TDE_F_REPLAY     (1u << 1)  /* replay override (gym_env.py:275-283) */
#define TDE_F_OFFROAD    (1u << 2)  /* drivable-mesh offroad test (gym_env.py:142,415,427) */
#define TDE_F_REWARD     (1u << 3)  /* WaypointSuite reward/advance/terminate/truncate (gym_env.py:369-437) */
#define TDE_F_AUTORESET  (1u << 4)  /* envs that finished this step are re-spawned in place (VecEnv semantics) */
#define TDE_F_EGO_ONLY_ATTRS (1u << 5) /* cfg.ego_only: random ego attrs at reset (gym_env.py:192-198) */
#define TDE_F_TRAFFIC_LIGHTS (1u << 6) /* red-light stop-line violation, third term of is_terminated (gym_env.py:415) */
#define TDE_F_NPC_FIRST_STEP (1u << 7) /* the NPC controller acts from the FIRST step of an episode, as the reference's NPCs do
                                          (gym_env.py:285-294: IAIWrapper predicts from step one) - part of TDE_F_ALL and what the
                                          host mirror sets by default (SimulatorConfig.npc_first_step).  Every kernel form honours
                                          it: the role-split rollout kernels run the controller a second time on the spawn rows of
                                          an env they re-spawned, the three-role step kernel computes a re-spawned env's first
                                          actions in the next launch's prologue.  Bit clear = the round-4 / 5 rule, kept as the
                                          opt-out: the NPCs coast through step one with the zero action (nothing to compute after
                                          a re-spawn: ~0.1 us per rollout step, ~0.5 us per closed-loop launch cheaper) */
#define TDE_F_ALL (TDE_F_NPC | TDE_F_REPLAY | TDE_F_OFFROAD | TDE_F_REWARD | TDE_F_AUTORESET | TDE_F_NPC_FIRST_STEP)

/* cell classes of the offroad grid index (HIP side only; the oracle is brute force over triangles) */
#define TDE_CELL_MAX_TRIS 255u
#define TDE_CLEARANCE_UNIT 0.125f
#define TDE_CELL_EMPTY 0u  /* every point of the cell is farther than threshold from every triangle  */
#define TDE_CELL_MIXED 1u  /* test the cell's candidate triangles                                    */
#define TDE_CELL_FULL  2u  /* every point of the cell is within threshold of some triangle            */
#define TDE_CELL_SUB   4   /* a MIXED cell carries TDE_CELL_SUB x TDE_CELL_SUB sub-cell classes (tde_world.cell_tri) */
#define TDE_COARSE_CELLS 4 /* a coarse tile of tde_world.cell_coarse is TDE_COARSE_CELLS x TDE_COARSE_CELLS cells */
#define TDE_COARSE_UNIT 0.25f /* metres per count of a coarse tile's clearance */

typedef struct tde_config {
    /* reward constants are Python floats (float64) in the reference: EnvConfig, gym_env.py:34-54 */
    double waypoint_bonus;      /* 100.  gym_env.py:39 */
    double heading_penalty;     /* 25.   gym_env.py:40 */
    double distance_bonus;      /* 1.    gym_env.py:41 */
    double distance_cutoff;     /* 0.5   gym_env.py:42 (shipped training configs use 0.25) */
    double reach_radius;        /* 3.    literal at gym_env.py:394 */
    uint64_t seed;              /* EnvConfig.seed, gym_env.py:45; keys the counter-based reset RNG */
    float dt;                   /* 0.1   KinematicBicycle default; render_fps 10, gym_env.py:75 */
    float offroad_threshold;    /* 0.5   TorchDriveConfig default (not overridden at gym_env.py:46-49) */
    /* heuristic NPC controller (replaces the IAI call at gym_env.py:285-294) */
    float npc_k_steer;          /* steering gain on sin(heading error) */
    float npc_k_speed;          /* accel gain on (v_des - v) */
    float npc_gap_s0;           /* standstill gap [m] */
    float npc_cone_k;           /* yield cone: corridor half width grows by this per metre ahead */
    float npc_lane_half;        /* half width of the look-ahead corridor [m] */
    float npc_reach;            /* NPC route waypoint switch radius [m] */
    float npc_max_accel;        /* NPC accel/brake limit [m/s^2] (NPCs are not bound by the ego's action range) */
    float npc_max_steer;        /* 0.3  action range gym_env.py:84 */
    int32_t max_steps;          /* 200   gym_env.py:37 */
    int32_t terminated_at_infraction; /* 1  gym_env.py:44 */
    uint32_t flags;             /* TDE_F_* */
    float npc_cone_range;       /* yield cone length [m] */
    uint32_t env_base;          /* global index of this shard's env 0: the reset RNG is keyed by env_base + e, so
                                   a batch sharded over GPUs replays exactly the episodes of the unsharded batch */
    int32_t offroad_threshold_squared; /* 0: a corner is off the road when its DISTANCE to the mesh exceeds
                                   offroad_threshold (d^2 > thr^2); 1: when its SQUARED distance does (d^2 > thr), the reading
                                   under which torchdrivesim's pytorch3d-style point-mesh distance is already squared.
                                   Upstream is unpinned here (source absent): both readings are built and tested.  The grid
                                   index of a World is built for the effective distance (thr or sqrt(thr)). */
} tde_config;

/* One drivable-surface map: triangle soup + uniform grid index + its traffic lights.  A descriptor is 80 bytes that POINT into the
 * world's tables, so several descriptors may share one mesh and grid (the same cell_base / tri_base / rec_base / cls2_base /
 * coarse_base) and differ only in stop_base / n_stop / phase_base / n_phase / cycle_steps: the lights of ONE neighbourhood of a
 * town each ("light groups": a light is a bit of a 32-bit mask and the kernels walk every stop line of a scenario's descriptor) -
 * tde_scenario.map selects the descriptor. */
typedef struct tde_map {
    float ox, oy;               /* grid origin (lower-left corner of cell (0,0)) */
    float cell;                 /* cell edge [m] */
    float inv_cell;             /* 1/cell */
    int32_t nx, ny;             /* grid size in cells */
    int32_t cell_base;          /* first cell word of this map in tde_world.cell_word; rows are stored with a pitch of
                                   2^row_shift >= nx words: word of cell (ix,iy) at cell_base + (iy << row_shift) + ix */
    int32_t tri_base;           /* first triangle of this map in tri */
    int32_t n_tri;
    int32_t stop_base, n_stop;  /* stop lines of this map in tde_world.stoplines */
    int32_t phase_base, n_phase;/* traffic-light cycle of this map in tde_world.phases */
    int32_t cycle_steps;        /* length of the cycle in env steps (0: no lights) */
    int32_t row_shift;          /* log2 of the row pitch of the map's cell words (>= 7) */
    int32_t cls2_base;          /* first 128-byte tile of this map in tde_world.cell_cls2 (in tiles) */
    int32_t rec_base;           /* (ABI 9) first candidate record of this map in tde_world.cell_tri: the 22-bit record offset of a
                                   cell word counts from here, so every map has 2^22 records of its own (a town mesh of 6e4
                                   triangles needs ~2e5 after identical lists are shared; one global offset ran out at the
                                   second town) */
    int32_t coarse_base;        /* (ABI 9) first 128-byte line of this map in tde_world.cell_coarse (in lines) */
    int32_t near_base;          /* (ABI 10) first word of this map in tde_world.tile_near: the word of coarse tile (cx, cy) is
                                   near_base + cy * (nx / TDE_COARSE_CELLS) + cx */
    int32_t _pad0;              /* 80 bytes: the kernels read the struct with 16-byte loads */
} tde_map;

/* A stop line: an oriented box across an inbound lane, governed by traffic light `light` of its map
 * (reference: map_cfg.stoplines with agent_type 'traffic_light', gym_env.py:183). */
typedef struct tde_stopline {
    float x, y, c, s;           /* centre, cos/sin of its heading */
    float hl, hw;               /* half extents */
    int32_t light;              /* light index (bit in tde_light_phase.red_mask) */
    int32_t _pad0;
} tde_stopline;

/* One phase of a map's light cycle: it lasts until step `end_step` of the cycle; lights in red_mask are red. */
typedef struct tde_light_phase {
    int32_t end_step;
    uint32_t red_mask;
} tde_light_phase;

/* One (scenario, slot) spawn record: everything a slot gets at reset, 64 B so that the kernels read it with four
 * 16-B loads.  Slot 0 (ego) only uses len/wid/lr: its pose is sampled (gym_env.py:351-367). */
typedef struct tde_spawn {
    float x, y, psi, v;         /* initial state (scenario agent_states, gym_env.py:222-224) */
    float len, wid, lr, vdes;   /* attrs (gym_env.py:225-226) + NPC desired speed */
    int32_t route;              /* NPC route id or -1 */
    int32_t route_wp;           /* first route waypoint the NPC heads for */
    int32_t route_n;            /* number of waypoints of that route (0 if none) */
    int32_t replay;             /* replay row id or -1 (car_sequence_suite, gym_env.py:275-283) */
    int32_t replay_len;         /* T of that replay row (0 if none) */
    int32_t present;            /* slot used by this scenario */
    float tgx0, tgy0;           /* (ABI 7) route_xy[route][route_wp]: the first route target (0 without a route), so that a
                                   re-spawned slot has its controller target with the record itself - one dependent table
                                   look-up less on the re-spawn path, which is the tail of every one-step launch */
} tde_spawn;

/* One WaypointSuite entry (gym_env.py:63-68). */
typedef struct tde_scenario {
    int32_t map;                /* map id */
    int32_t wp_n;               /* number of ego waypoints */
    float start_heading;        /* the start heading when the world carries no heading table (tde_world.NH == 0): the direction of
                                   the first waypoint segment, a stand-in for find_lanelet_directions (gym_env.py:359) */
    int32_t _pad0;
} tde_scenario;

/* (ABI 11) One entry of the first-step gap cache, tde_world.first_gap: what the NPC controller of slot a finds on the FIRST step of
 * an episode of scenario s apart from the ego - min(leader gap over the scenario's other NPCs, gap to a red stop line) at the spawn
 * state, which depends on the scenario's tables and the controller's constants only, not on the episode.  `key` = the launch's
 * controller hash | 1 (what tde_act_cache's key carries: config.flags & (NPC | REPLAY | TRAFFIC_LIGHTS), the npc_* constants, the
 * identity of the world's tables); an entry with another key (zero-initialised memory) is recomputed and rewritten by the lane
 * that meets it, as ONE 8-byte store.  With a valid entry the first step of a re-spawned env costs the role-split kernels one
 * exact test against the ego's row instead of the controller's sweep over the env's slots; same minimum, same bits. */
typedef struct tde_first_gap {
    float gap;
    uint32_t key;
} tde_first_gap;

/* Static world tables, replicated per GPU.  All pointers live in the address space of the
 * library they are handed to (device for libtde_hip, host for libtde_oracle). */
typedef struct tde_world {
    const tde_map *maps;        /* [n_maps] */
    const float *tri;           /* [n_tri_total][6]  ax,ay,bx,by,cx,cy (what the oracle's brute force reads) */
    const uint32_t *cell_word;  /* [n_cells_total] grid index (kernels): bits 0-1 TDE_CELL_*; MIXED cells: bits 2-9
                                   number of candidate triangles, bits 10-31 first record in cell_tri counted from the map's
                                   rec_base (cells with the same candidates share their records); FULL / EMPTY
                                   cells: bits 2-9 clearance in units of TDE_CLEARANCE_UNIT (every point that close to the cell
                                   lies in a cell of the same class) */
    const float *cell_tri;      /* [n_records + 16][12] per-cell candidate triangles, packed for 16-B loads:
                                   ax,ay,bx,by | cx,cy,1/|ab|^2,1/|bc|^2 | 1/|ca|^2,len,0,0 - `len` (ABI 10, the bits of an int32):
                                   at the first record of a NEAR LIST (tile_near) the length of that list, else 0.  The table ends
                                   with 16 zero records: the magnitude kernels fetch the first 16 records of a near list before
                                   they know its length */
    const uint32_t *cell_cls2;  /* (ABI 7, rasteriser) the cell classes alone, 2 bits per cell, in 128-byte tiles of 32 x 16 cells
                                   (8 m x 4 m at 0.25 m cells: a 35 m view touches ~50 cache lines of it, against one line per
                                   look-up in cell_word).  Tile (tx, ty) = cells [32 tx, 32 tx + 32) x [16 ty, 16 ty + 16) of a
                                   map is tile cls2_base + (ty << (row_shift - 5)) + tx; inside a tile word 2 * (iy & 15) +
                                   ((ix >> 4) & 1) holds the 16 cells ix & ~15 .. of row iy, cell ix at bits 2 * (ix & 15) */
    const uint32_t *cell_sub;   /* (ABI 7, rasteriser) one word per cell, meaningful for MIXED cells: the 2-bit TDE_CELL_* class
                                   of each of the cell's 4 x 4 sub-cells, sub-cell (sx, sy) at bits 2 * (4 * sy + sx),
                                   conservative like the cell classes (2 mm margin).  Same number of words per map as
                                   cell_word, in 128-byte tiles of 8 x 4 cells: cell (ix, iy) at
                                   cell_base + ((((iy >> 2) << (row_shift - 3)) + (ix >> 3)) << 5) + ((iy & 3) << 3) + (ix & 7) */
    const uint8_t *cell_coarse; /* (ABI 9, rasteriser) one byte per coarse tile of 4 x 4 cells: bits 0-1 TDE_CELL_FULL / TDE_CELL_EMPTY when
                                   all 16 cells have that class, else TDE_CELL_MIXED; bits 2-7 the tile's clearance in
                                   TDE_COARSE_UNITs, rounded down (every point that close to ANY point of the tile lies in a cell of
                                   the tile's class; 0 for MIXED).  In 128-byte lines of 16 x 8 tiles (16 m x 8 m at 0.25 m
                                   cells): tile (cx, cy) = cells [4 cx, 4 cx + 4) x [4 cy, 4 cy + 4) of a map is byte
                                   ((coarse_base + ((cy >> 3) << (row_shift - 6)) + (cx >> 4)) << 7) + ((cy & 7) << 4) + (cx & 15).
                                   The block pyramid of a 35 m view reads ~20 lines of it (1 MB per km^2) where the clearance
                                   field of cell_word cost one line per look-up (66 MB per km^2) */
    const uint32_t *tile_near;  /* (ABI 10, infraction magnitudes) one word per coarse tile of 4 x 4 cells, row-major per map
                                   (tde_map.near_base): 0 = the tile has no near list (farther than threshold + near_range from the
                                   mesh: the kernels scan the grid); 0xFFFFFFFF = all 16 cells FULL (every point of the tile within
                                   the threshold: the offroad magnitude's term is 0); else 1 + the first record, counted from the
                                   map's rec_base, of the tile's NEAR LIST in cell_tri: a superset of the triangles one of which is
                                   the nearest triangle of any point of the tile, so the minimum of the point-triangle distances
                                   over the list IS the distance to the mesh (the CPU checker's brute-force minimum, same bits) */
    const tde_scenario *scn;    /* [S] */
    const double *wp_xy;        /* [S][NW][2] ego waypoints, float64 like the YAML lists (gym_env.py:314,394) */
    const tde_spawn *spawn;     /* [S][A] */
    const float *route_xy;      /* [R][RW][2] NPC route polylines */
    const float *replay_states; /* [P][RT][4] */
    const tde_stopline *stoplines;   /* [n_stop_total] */
    const tde_light_phase *phases;   /* [n_phase_total] */
    const float *start_psi;     /* (ABI 11) [S][NH] the ego's start heading along the first waypoint segment of every scenario: entry j
                                   = the lane direction at the point p0 + (j + 0.5) / NH * (p1 - p0), what the reference takes from
                                   find_lanelet_directions(lanelet_map, x, y) at the sampled start point (gym_env.py:359-361); an
                                   episode that starts at fraction f of the segment reads entry floor(f * NH).  Unused (may be a
                                   one-element dummy) when NH == 0: tde_scenario.start_heading then */
    tde_first_gap *first_gap;   /* (ABI 11) [S][A] the first-step gap cache (see tde_first_gap): the ONE table of the world the
                                   kernels WRITE - scratch that belongs to the world's device copy, zero-initialised by its owner;
                                   NULL: no cache (every first step runs the controller's sweep).  The CPU checker ignores it */
    int32_t n_maps, n_scn, NW, A;
    int32_t n_routes, RW, n_replay, RT;
    int32_t hints;              /* (ABI 9) TDE_WORLD_*: properties of the tables that select a kernel form (never results) */
    int32_t NH;                 /* (ABI 11) entries per scenario of start_psi; 0: none */
} tde_world;
/* tde_world.hints: some map's grid is large (more than 2^21 cells, ~360 m x 360 m at 0.25 m cells): the persistent rollout kernels
 * and the 32- / 64-slot one-step kernel then take the corner classes from the 2-bit class map (cell_cls2, 1/16 of cell_word's
 * footprint) and fetch a cell word only for a corner in a MIXED cell.  Same masks either way; on a 1 km^2 town 3.98 -> 3.02 us
 * per rollout step, on 280 m junction maps 2.92 -> 2.99 (profiles/r04_b_*). */
#define TDE_WORLD_LARGE_GRID (1 << 0)

/* What tde_grid_build (include/tde_hip.h) returns for ONE map: the offroad grid index of a drivable mesh, row-major
 * [ny * nx] HOST arrays owned by the library (tde_grid_free).  The caller packs them into the tde_world tables: a cell's word
 * = cell_class | cell_count << 2 | cell_first << 10, rows at a pitch of 2^row_shift; cell_cls2 / cell_sub in their tiles;
 * cell_tri = the packed records of triangles rec_tri[] (torchdriveenv_amd/world.py: assemble_world). */
typedef struct tde_grid {
    float ox, oy, cell;         /* grid origin (integers) and cell edge */
    int32_t nx, ny;             /* multiples of 8; >= 2 EMPTY cells on every side of the mesh */
    int64_t n_lists;            /* distinct candidate lists */
    int64_t n_records;          /* their total length (< 2^22) */
    uint8_t *cell_class;        /* TDE_CELL_* */
    uint8_t *cell_count;        /* MIXED: number of candidates (<= TDE_CELL_MAX_TRIS); FULL / EMPTY: clearance in TDE_CLEARANCE_UNITs */
    uint32_t *cell_first;       /* MIXED: first record of the cell's list in rec_tri (map-relative) */
    uint32_t *cell_sub;         /* MIXED: the 2-bit classes of the cell's TDE_CELL_SUB x TDE_CELL_SUB sub-cells */
    int32_t *rec_tri;           /* [n_records] triangle index (into the mesh handed in) of every record */
    uint32_t *tile_near;        /* (ABI 10) [(ny / 4) * (nx / 4)] row-major: the tde_world.tile_near word of every coarse tile */
    int32_t *rec_len;           /* (ABI 10) [n_records] length of the near list that starts at this record, else 0 */
    int64_t n_near_lists;       /* (ABI 10) tiles with a near list */
} tde_grid;

/* Optional lookup caches of the closed-loop step (tde_env_step): what the step needs from the scenario tables for a
 * slot / an env, kept next to the state so that a one-step launch starts with independent loads instead of the chain
 * scenario -> spawn record -> route table (three dependent L2 / HBM round trips before the first useful instruction).
 * Each entry carries the key it was formed for; the kernel uses it only when the key equals the current state and
 * rebuilds it otherwise, so any writer of the state (reset, rollout, host edits) leaves the caches correct by
 * construction.  An entry is valid when its TDE_CACHE_VALID bit is set (zero-initialised memory is invalid). */
typedef struct tde_slot_cache {  /* 32 bytes (ABI 8; 48 before) */
    int32_t scn;                /* key: the env's scenario */
    int32_t key;                /* key: this slot's route waypoint index (bits 0-15) | (config.flags & (TDE_F_NPC | TDE_F_REPLAY)) << 16
                                   (the ids below are what the launch that formed the entry found under those flags) | TDE_CACHE_VALID */
    float tgx, tgy;             /* NPC: current route waypoint */
    uint32_t route;             /* NPC: route id + 1 (bits 0-19; 0: none) | route length << 20 */
    uint32_t replay;            /* replay row id + 1 (bits 0-19; 0: none) | replay length << 20 */
    float tgx2, tgy2;           /* NPC: the route waypoint after the current one (a waypoint switch then needs no look-up
                                   on the step's dependent chain); undefined when route_wp + 1 >= route length */
} tde_slot_cache;
/* (ids below 2^20 - 1 and lengths below 4096: tde_env_step uses its one-role kernel, which needs no caches, for a world beyond) */
#define TDE_CACHE_ID_BITS 20

/* The NPC controller's actions for the NEXT step, computed at the end of a step behind the judges (they only need the state
 * after the step).  [B * (A + 1)] entries of 8 bytes: env e's slots at e * (A + 1) + a, then ONE key entry for the env at
 * e * (A + 1) + A holding, as two int32, the (episode, environment_steps) pair of the state the actions were computed from
 * (episode < 0: invalid); bits 20-31 of the second word carry a hash of what else the controller depends on - config.flags &
 * (NPC | REPLAY | TRAFFIC_LIGHTS) and the npc_* constants (ABI 9) - so a caller that changes those between two launches gets
 * the actions recomputed, not replayed.  (ABI 8: 16 bytes per slot with a key each before.) */
typedef struct tde_act_cache {
    float acc, beta;
} tde_act_cache;

typedef struct tde_env_cache {
    int32_t scn, target_idx;    /* key */
    int32_t n_wp, map;          /* waypoints of the scenario; its map id; bit 30 of n_wp = entry valid */
    double wtx, wty;            /* current ego target (undefined when target_idx >= n_wp) */
} tde_env_cache;
#define TDE_CACHE_VALID (1 << 30)

/* Mutable per-env / per-agent state and per-step outputs.  Caller-allocated, written in place. */
typedef struct tde_state {
    /* agent arrays [B*A] */
    float *x, *y, *psi, *v;     /* kinematic state (R4) */
    float *len, *wid, *lr;      /* attrs */
    float *vdes;                /* NPC desired speed */
    int32_t *route_wp;          /* NPC current route waypoint index (route / replay ids come from the spawn record) */
    uint8_t *present;           /* present mask (gym_env.py:262-263: all ones in the reference) */
    uint8_t *collided;          /* out: compute_collision() > 0 per agent (R9) */
    uint8_t *offroad;           /* out: compute_offroad() > 0 per agent (R10) */
    /* env arrays [B] */
    int32_t *scn;               /* current scenario (current_waypoint_suite_idx, gym_env.py:320) */
    int32_t *steps;             /* environment_steps (gym_env.py:116) */
    int32_t *target_idx;        /* current_target_idx (gym_env.py:325,379) */
    int32_t *reached;           /* reached_waypoint_num (gym_env.py:338,406) */
    int32_t *episode;           /* number of resets so far (RNG counter) */
    const float *action;        /* in: [B][2] ego (accel, steer), raw units (gym_env.py:83-94) */
    float *reward;              /* out [B] (R6) */
    uint8_t *terminated;        /* out [B] (R8) */
    uint8_t *truncated;         /* out [B] (R11) */
    uint8_t *tl_violation;      /* out [B]: compute_traffic_lights_violations() > 0 for the ego (gym_env.py:144,415,429) */
    double *info;               /* out [B][4] psi_smoothness, speed_smoothness, psi_reward, dist_reward (R12; Python floats
                                   in the reference, hence float64); may be NULL */
    int32_t *info_reached;      /* out [B] reached_waypoint_num as reported by get_info (gym_env.py:425,431); may be NULL */
    uint8_t *done_bits;         /* out [B], tde_env_step only, may be NULL: the ego's flags of THIS step before any in-place
                                   re-spawn clears them (what get_info reports at a terminal step, gym_env.py:426-429):
                                   bit0 terminated, bit1 truncated, bit2 offroad, bit3 collided, bit4 red-light violation
                                   (the layout of tde_rollout.done) */
    float *obs;                 /* out [B][8], tde_env_step only, may be NULL: the compact ego observation of tde_state_obs
                                   (x, y, psi, v, target offset forward / left, target flag, steps) of the state AFTER
                                   the step and any re-spawn - the same values as a tde_state_obs call would give, without
                                   the second launch */
    double *ep_return;          /* in/out [B], tde_env_step / tde_env_reset only, may be NULL: sum of the rewards of the running
                                   episode (float64, as Monitor sums Python floats: examples/rl_training.py:123-128); zeroed by a
                                   reset and by an in-place re-spawn */
    double *ep_final;           /* out [B], tde_env_step only, may be NULL: written for the envs that finish at THIS step with the
                                   return of the episode that just ended (its length is environment_steps at that step: the k
                                   of done_bits' step, also reported as ep_final_len); other entries keep their value */
    int32_t *ep_final_len;      /* out [B], with ep_final */
    tde_slot_cache *slot_cache; /* in/out [B*A], tde_env_step only, may be NULL (with env_cache): see above.  With both caches
                                   present (and 8, 16 or 32 agents per env) tde_env_step runs its three-role kernel */
    tde_env_cache *env_cache;   /* in/out [B] */
    tde_act_cache *act_cache;   /* in/out [B*(A+1)], may be NULL: with it the three-role step applies the stored NPC actions at once
                                   and computes the next step's behind the judges.  The key cannot see a state that was edited
                                   from outside with its counters unchanged: zero / invalidate the cache after such an edit
                                   (EnvState.load does) */
    float *magnitudes;          /* out [B][4], tde_env_step only, may be NULL (ABI 10): the MAGNITUDES of the ego's infractions at
                                   THIS step - what the reference's info dict holds under "offroad" / "collision"
                                   (gym_env.py:427-428: simulator.compute_offroad() / compute_collision() for the exposed agent) -
                                   as tde_ego_infractions defines them: (sum over the box corners of clamp(distance to the mesh -
                                   offroad_threshold, 0), sum of the IoUs with the agents the ego overlaps, their number, 0), of
                                   the state the step left BEFORE any in-place re-spawn; zeros without the matching flag */
    int32_t B, A;
} tde_state;

/* K-step open-loop rollout buffers (actions resident in HBM; per-step env outputs). */
typedef struct tde_rollout {
    const float *actions;       /* [K][B][2] */
    float *reward;              /* [K][B] */
    uint8_t *done;              /* [K][B] bit0 terminated, bit1 truncated, bit2 ego offroad, bit3 ego collided,
                                   bit4 ego red-light violation */
    int32_t K;
    int32_t ldb;                /* row pitch of the three [K][..] buffers in envs; 0 = B (how a shard [e0, e0 + n) of a larger
                                   batch is stepped on buffers of the whole batch: pointers advanced by e0, ldb = the batch) */
} tde_rollout;

/* ego-centred, ego-aligned birdview raster (R13; BASELINE config 5).  Observation space (3,64,64) uint8, channels
 * first (gym_env.py:95).  Pixel (row r, col c) is sampled at its centre; the ego sits at the image centre with its
 * heading pointing up: forward = (H/2 - (r+.5))*res, left = (W/2 - (c+.5))*res, res = fov / W.
 * Layers, painted in this order: background, drivable surface (= within offroad_threshold of the mesh, the same
 * predicate the offroad infraction uses), stop lines of the map coloured by their light's state at the env's current
 * step (only with TDE_F_TRAFFIC_LIGHTS: the traffic controls the reference hands to the renderer, gym_env.py:187-189,
 * 259-266), remaining ego waypoints (discs), NPC boxes, ego box.
 * With n_stack > 1 the output holds the last n_stack frames, oldest first (SB3 VecFrameStack(channels_order="first"),
 * examples/rl_training.py:160): older frames are shifted down and the new frame is written last. */
#define TDE_RGB_BACKGROUND 255, 255, 255
#define TDE_RGB_ROAD       128, 128, 128
#define TDE_RGB_WAYPOINT    44, 160,  44
#define TDE_RGB_NPC         31, 119, 180
#define TDE_RGB_EGO        214,  39,  40
#define TDE_RGB_STOP_RED   255,   0,   0   /* stop line whose light is red */
#define TDE_RGB_STOP_GO      0, 255,   0   /* stop line whose light is not red */
#define TDE_WAYPOINT_RADIUS 1.0f
/* layer codes of the one-byte-per-pixel planes (tde_render.layers); the palette above in this order */
#define TDE_LAYER_BACKGROUND 0
#define TDE_LAYER_ROAD       1
#define TDE_LAYER_WAYPOINT   2
#define TDE_LAYER_NPC        3
#define TDE_LAYER_EGO        4
#define TDE_LAYER_STOP_RED   6
#define TDE_LAYER_STOP_GO    7
/* tde_render.flags */
#define TDE_RENDER_LEFT_HANDED   (1 << 0)  /* RendererConfig.left_handed_coordinates (True in the reference, gym_env.py:46):
                                              the lateral image axis is mirrored (world +y is to the RIGHT of +x) */
#define TDE_RENDER_PLAIN_EGO     (1 << 1)  /* highlight_ego_vehicle = False (gym_env.py:47 sets True): ego painted as an NPC */
typedef struct tde_render {
    uint8_t *out;               /* [B][3*max(n_stack,1)][H][W] uint8 */
    int32_t H, W;               /* 64, 64 (multiples of 4, H*W <= 4096: the view is staged in LDS as one layer byte per pixel) */
    float fov;                  /* metres covered by the image width (35 in torchdrivesim's default RendererConfig) */
    int32_t n_stack;            /* 0/1: single frame; n: frame stack of n */
    /* Frame stack without moving pixels (optional; NULL: the older frames of `out` are shifted in place by a launch of
     * their own).  `layers` is a caller-owned ring of the stack's frames as one LAYER byte per pixel,
     * uint8 [B][n_stack][H*W], initialised to TDE_LAYER_BLANK and kept between calls; `phase` counts the calls made on
     * this stack (0, 1, 2, ...).  Each call stores the new frame's layer plane in slot phase % n_stack and writes all
     * n_stack frames of `out` (oldest first) by expanding the ring through the palette: 4 KiB read + 4 KiB written per
     * older frame and view instead of 12 + 12, no ordering hazard, one launch.  A caller that clears a view's stack
     * (VecFrameStack on reset) fills its ring slots with TDE_LAYER_BLANK. */
    uint8_t *layers;
    int32_t phase;              /* >= 0; callers keep it reduced modulo n_stack */
    int32_t flags;              /* TDE_RENDER_* */
    const uint8_t *fresh;       /* optional u8 [B] (NULL: none): views with (fresh[e] & 3) != 0 just (re)started their episode -
                                   their older ring slots are blanked inside this call, so the first stacked observation of an
                                   episode holds (blank, ..., blank, frame 0) as VecFrameStack gives after a reset.  Bits 0-1
                                   are the done bits of tde_state.done_bits: after a step with TDE_F_AUTORESET pass done_bits
                                   itself (the finished envs were re-spawned in place); a plain 0/1 reset mask works too. */
    const uint8_t *only;        /* optional u8 [B] (NULL: all): render only these views; the others keep their ring, their
                                   `out` pixels and do not consume a ring slot (phase is per call, so a masked call must
                                   use the phase of the LAST full call: it re-renders the newest slot of the masked views) */
} tde_render;
#define TDE_LAYER_BLANK 5       /* palette entry (0, 0, 0): a frame that has not been rendered yet */

#ifdef __cplusplus
}
#endif
#endif /* TDE_ABI_H */
