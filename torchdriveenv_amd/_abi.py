"""ctypes mirror of include/tde_abi.h (field for field) plus helpers to fill the structs from numpy arrays
(host pointers, for the CPU oracle used by the tests) or torch tensors (device pointers, for libtde_hip.so).

PyTorch is plumbing here: it owns device memory and the HIP stream; nothing in the C-ABI mentions torch.
"""
import ctypes as C

import numpy as np

TDE_ABI_VERSION = 11
TDE_MAX_AGENTS = 128

F_NPC = 1 << 0
F_REPLAY = 1 << 1
F_OFFROAD = 1 << 2
F_REWARD = 1 << 3
F_AUTORESET = 1 << 4
F_EGO_ONLY_ATTRS = 1 << 5
F_TRAFFIC_LIGHTS = 1 << 6
F_NPC_FIRST_STEP = 1 << 7
F_ALL = F_NPC | F_REPLAY | F_OFFROAD | F_REWARD | F_AUTORESET | F_NPC_FIRST_STEP    # (TDE_F_ALL: the reference's NPC timing)

CELL_EMPTY, CELL_MIXED, CELL_FULL = 0, 1, 2

_p = C.c_void_p


class TdeConfig(C.Structure):
    _fields_ = [
        ("waypoint_bonus", C.c_double),
        ("heading_penalty", C.c_double),
        ("distance_bonus", C.c_double),
        ("distance_cutoff", C.c_double),
        ("reach_radius", C.c_double),
        ("seed", C.c_uint64),
        ("dt", C.c_float),
        ("offroad_threshold", C.c_float),
        ("npc_k_steer", C.c_float),
        ("npc_k_speed", C.c_float),
        ("npc_gap_s0", C.c_float),
        ("npc_cone_k", C.c_float),
        ("npc_lane_half", C.c_float),
        ("npc_reach", C.c_float),
        ("npc_max_accel", C.c_float),
        ("npc_max_steer", C.c_float),
        ("max_steps", C.c_int32),
        ("terminated_at_infraction", C.c_int32),
        ("flags", C.c_uint32),
        ("npc_cone_range", C.c_float),
        ("env_base", C.c_uint32),
        ("offroad_threshold_squared", C.c_int32),
    ]


class TdeMap(C.Structure):
    _fields_ = [
        ("ox", C.c_float), ("oy", C.c_float), ("cell", C.c_float), ("inv_cell", C.c_float),
        ("nx", C.c_int32), ("ny", C.c_int32), ("cell_base", C.c_int32), ("tri_base", C.c_int32),
        ("n_tri", C.c_int32), ("stop_base", C.c_int32), ("n_stop", C.c_int32), ("phase_base", C.c_int32),
        ("n_phase", C.c_int32), ("cycle_steps", C.c_int32), ("row_shift", C.c_int32), ("cls2_base", C.c_int32),
        ("rec_base", C.c_int32), ("coarse_base", C.c_int32), ("near_base", C.c_int32), ("_pad0", C.c_int32),
    ]


class TdeGrid(C.Structure):
    """tde_grid: what tde_grid_build returns for one map (host arrays owned by the library)"""
    _fields_ = [
        ("ox", C.c_float), ("oy", C.c_float), ("cell", C.c_float), ("nx", C.c_int32), ("ny", C.c_int32),
        ("n_lists", C.c_int64), ("n_records", C.c_int64),
        ("cell_class", C.POINTER(C.c_uint8)), ("cell_count", C.POINTER(C.c_uint8)),
        ("cell_first", C.POINTER(C.c_uint32)), ("cell_sub", C.POINTER(C.c_uint32)), ("rec_tri", C.POINTER(C.c_int32)),
        ("tile_near", C.POINTER(C.c_uint32)), ("rec_len", C.POINTER(C.c_int32)), ("n_near_lists", C.c_int64),
    ]


MAP_DTYPE = np.dtype([("ox", "f4"), ("oy", "f4"), ("cell", "f4"), ("inv_cell", "f4"), ("nx", "i4"), ("ny", "i4"),
                      ("cell_base", "i4"), ("tri_base", "i4"), ("n_tri", "i4"), ("stop_base", "i4"), ("n_stop", "i4"),
                      ("phase_base", "i4"), ("n_phase", "i4"), ("cycle_steps", "i4"), ("row_shift", "i4"), ("cls2_base", "i4"),
                      ("rec_base", "i4"), ("coarse_base", "i4"), ("near_base", "i4"), ("_pad0", "i4")])
STOPLINE_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("c", "f4"), ("s", "f4"), ("hl", "f4"), ("hw", "f4"),
                           ("light", "i4"), ("_pad0", "i4")])
PHASE_DTYPE = np.dtype([("end_step", "i4"), ("red_mask", "u4")])
assert MAP_DTYPE.itemsize == C.sizeof(TdeMap) == 80 and STOPLINE_DTYPE.itemsize == 32 and PHASE_DTYPE.itemsize == 8

SPAWN_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("psi", "f4"), ("v", "f4"), ("len", "f4"), ("wid", "f4"),
                        ("lr", "f4"), ("vdes", "f4"), ("route", "i4"), ("route_wp", "i4"), ("route_n", "i4"),
                        ("replay", "i4"), ("replay_len", "i4"), ("present", "i4"), ("tgx0", "f4"), ("tgy0", "f4")])
SCN_DTYPE = np.dtype([("map", "i4"), ("wp_n", "i4"), ("start_heading", "f4"), ("_pad0", "i4")])
assert SPAWN_DTYPE.itemsize == 64 and SCN_DTYPE.itemsize == 16

WORLD_PTRS = ["maps", "tri", "cell_word", "cell_tri", "cell_cls2", "cell_sub", "cell_coarse", "tile_near", "scn", "wp_xy", "spawn", "route_xy", "replay_states",
              "stoplines", "phases", "start_psi", "first_gap"]
WORLD_INTS = ["n_maps", "n_scn", "NW", "A", "n_routes", "RW", "n_replay", "RT", "hints", "NH"]
WORLD_LARGE_GRID = 1 << 0


class TdeWorld(C.Structure):
    _fields_ = [(n, _p) for n in WORLD_PTRS] + [(n, C.c_int32) for n in WORLD_INTS]


STATE_AGENT_F32 = ["x", "y", "psi", "v", "len", "wid", "lr", "vdes"]
STATE_AGENT_I32 = ["route_wp"]
STATE_AGENT_U8 = ["present", "collided", "offroad"]
STATE_ENV_I32 = ["scn", "steps", "target_idx", "reached", "episode"]
STATE_PTRS = (STATE_AGENT_F32 + STATE_AGENT_I32 + STATE_AGENT_U8 + STATE_ENV_I32 +
              ["action", "reward", "terminated", "truncated", "tl_violation", "info", "info_reached", "done_bits", "obs",
               "ep_return", "ep_final", "ep_final_len", "slot_cache", "env_cache", "act_cache", "magnitudes"])


class TdeState(C.Structure):
    _fields_ = [(n, _p) for n in STATE_PTRS] + [("B", C.c_int32), ("A", C.c_int32)]


class TdeRollout(C.Structure):
    _fields_ = [("actions", _p), ("reward", _p), ("done", _p), ("K", C.c_int32), ("ldb", C.c_int32)]


class TdeRender(C.Structure):
    _fields_ = [("out", _p), ("H", C.c_int32), ("W", C.c_int32), ("fov", C.c_float), ("n_stack", C.c_int32),
                ("layers", _p), ("phase", C.c_int32), ("flags", C.c_int32), ("fresh", _p), ("only", _p)]


LAYER_BLANK = 5
LAYER_STOP_RED, LAYER_STOP_GO = 6, 7
RENDER_LEFT_HANDED, RENDER_PLAIN_EGO = 1 << 0, 1 << 1
# palette of the layer codes 0..7 (tde_abi.h TDE_RGB_*): background, road, waypoint, NPC, ego, blank, stop red, stop go
PALETTE = ((255, 255, 255), (128, 128, 128), (44, 160, 44), (31, 119, 180), (214, 39, 40), (0, 0, 0), (255, 0, 0),
           (0, 255, 0))


def ptr_of(a):
    """Raw address of a contiguous numpy array or torch tensor (None -> NULL)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "arrays handed to the C-ABI must be contiguous"
        return a.ctypes.data
    # torch tensor
    assert a.is_contiguous(), "tensors handed to the C-ABI must be contiguous"
    return a.data_ptr()


def default_config(**over):
    """tde_config with the reference's defaults (EnvConfig, gym_env.py:34-54) and the NPC controller's."""
    cfg = TdeConfig()
    cfg.waypoint_bonus = 100.0
    cfg.heading_penalty = 25.0
    cfg.distance_bonus = 1.0
    cfg.distance_cutoff = 0.5
    cfg.reach_radius = 3.0
    cfg.seed = 0
    cfg.dt = 0.1
    cfg.offroad_threshold = 0.5
    cfg.npc_k_steer = 1.2
    cfg.npc_k_speed = 3.0
    cfg.npc_gap_s0 = 3.0
    cfg.npc_cone_k = 0.5
    cfg.npc_cone_range = 25.0
    cfg.npc_lane_half = 1.75
    cfg.npc_reach = 3.0
    cfg.npc_max_accel = 3.0
    cfg.npc_max_steer = 0.3
    cfg.max_steps = 200
    cfg.terminated_at_infraction = 1
    cfg.flags = F_ALL
    cfg.env_base = 0
    cfg.offroad_threshold_squared = 0
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(f"tde_config has no field {k!r}")
        setattr(cfg, k, v)
    return cfg


WORLD_DTYPES = {
    "maps": MAP_DTYPE, "tri": np.float32, "cell_word": np.uint32, "cell_tri": np.float32, "cell_cls2": np.uint32, "cell_sub": np.uint32, "cell_coarse": np.uint8, "tile_near": np.uint32, "scn": SCN_DTYPE,
    "wp_xy": np.float64, "spawn": SPAWN_DTYPE, "route_xy": np.float32, "replay_states": np.float32,
    "stoplines": STOPLINE_DTYPE, "phases": PHASE_DTYPE, "start_psi": np.float32,
    "first_gap": np.uint32,        # tde_first_gap [S][A] as pairs of words (gap bits, key): scratch the kernels fill, zeros at upload
}

STATE_DTYPES = {**{n: np.float32 for n in STATE_AGENT_F32}, **{n: np.int32 for n in STATE_AGENT_I32},
                **{n: np.uint8 for n in STATE_AGENT_U8}, **{n: np.int32 for n in STATE_ENV_I32},
                "action": np.float32, "reward": np.float32, "terminated": np.uint8, "truncated": np.uint8,
                "tl_violation": np.uint8,
                "info": np.float64, "info_reached": np.int32, "done_bits": np.uint8, "obs": np.float32,
                "ep_return": np.float64, "ep_final": np.float64, "ep_final_len": np.int32,
                "slot_cache": np.int32, "env_cache": np.int32, "act_cache": np.int32, "magnitudes": np.float32}          # opaque 32-byte records (tde_slot_cache / tde_env_cache)


def state_shapes(B, A):
    sh = {n: (B * A,) for n in STATE_AGENT_F32 + STATE_AGENT_I32 + STATE_AGENT_U8}
    sh.update({n: (B,) for n in STATE_ENV_I32})
    sh.update({"action": (B, 2), "reward": (B,), "terminated": (B,), "truncated": (B,), "tl_violation": (B,),
               "info": (B, 4),
               "info_reached": (B,), "done_bits": (B,), "obs": (B, 8), "ep_return": (B,), "ep_final": (B,),
               "ep_final_len": (B,), "slot_cache": (B * A, 8), "env_cache": (B, 8), "act_cache": (B * (A + 1), 2),
               "magnitudes": (B, 4)})
    return sh


def fill_world_struct(arrays, ints):
    w = TdeWorld()
    for n in WORLD_PTRS:
        setattr(w, n, ptr_of(arrays[n]))
    for n in WORLD_INTS:
        setattr(w, n, int(ints[n]))
    return w


def fill_state_struct(arrays, B, A):
    s = TdeState()
    for n in STATE_PTRS:
        setattr(s, n, ptr_of(arrays.get(n)))
    s.B, s.A = int(B), int(A)
    return s
