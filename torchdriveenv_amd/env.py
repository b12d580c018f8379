"""Host-side mirror of the reference's env surface over the HIP step path.

  * `BatchedWaypointEnv`  — B envs per GPU in one fused kernel per step; device-resident torch outputs (fast path) and
    an SB3-`VecEnv`-shaped numpy API (`step_async/step_wait`, auto-reset with `terminal_observation`), so
    `SubprocVecEnv` (ref examples/rl_training.py:159) is unnecessary.
  * `WaypointSuiteEnv` + `SingleAgentWrapper` — the B=1, one-exposed-agent interface the reference registers as
    'torchdriveenv-v0' (ref __init__.py:10, gym_env.py:303-487): same signatures, shapes, dtypes and info keys.

What stays in Python is only argument marshalling; every number comes from libtde_hip.so.  There is no CPU fallback.
"""
import math
import warnings

import numpy as np
import torch

from . import _abi, ops
from .config import EnvConfig, WaypointSuite, render_flags, to_tde_config, validate
from .state import EnvState
from .world import World, assemble_world, check_threshold, corridor_mesh, effective_offroad_distance

try:  # optional: neither gymnasium nor SB3 ships in this image
    import gymnasium as gym
    _GymEnvBase, _GymWrapperBase = gym.Env, gym.Wrapper
except Exception:  # pragma: no cover
    gym = None
    _GymEnvBase = object

    class _GymWrapperBase:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)


class Box:
    """stand-in for gym.spaces.Box when gymnasium is absent (same attributes)"""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype) if shape is None else np.full(shape, low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype) if shape is None else np.full(shape, high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)


def _box(low, high, shape=None, dtype=np.float32):
    if gym is not None:
        return gym.spaces.Box(low=low, high=high, shape=shape, dtype=dtype) if shape is not None else \
            gym.spaces.Box(low=np.asarray(low, dtype=dtype), high=np.asarray(high, dtype=dtype), dtype=dtype)
    return Box(low, high, shape, dtype)


# action space of the reference: acceleration in [-1, 1], steering in [-0.3, 0.3], raw units (ref gym_env.py:83-94)
ACTION_LOW, ACTION_HIGH = (-1.0, -0.3), (1.0, 0.3)


def _straight_route(x, y, psi, v):
    ahead = max(30.0, 25.0 * max(v, 1.0))
    return [(x + math.cos(psi) * d, y + math.sin(psi) * d) for d in np.arange(8.0, ahead, 8.0)]


def _mesh_of_location(road_meshes, loc):
    """the caller's drivable mesh for a location as a [n, 3, 2] float array, or None (-> synthetic corridor)"""
    if road_meshes is None:
        return None
    tri = road_meshes(loc) if callable(road_meshes) else road_meshes.get(loc)
    if tri is None:
        return None
    if isinstance(tri, (str, bytes)) or hasattr(tri, "__fspath__"):
        tri = np.load(tri)                                   # a .npy file of the triangles
    tri = np.asarray(tri, dtype=np.float64)
    if tri.ndim == 2 and tri.shape[1] == 6:
        tri = tri.reshape(-1, 3, 2)
    if tri.ndim != 3 or tri.shape[1:] != (3, 2) or len(tri) == 0 or not np.isfinite(tri).all():
        raise ValueError(f"road mesh of location {loc!r} must be a finite [n, 3, 2] (or [n, 6]) array of triangle vertices, "
                         f"got shape {tri.shape}")
    return tri


def mesh_from_verts_faces(verts, faces):
    """(V, 2+) vertices and (F, 3) vertex indices - the layout of torchdrivesim's `road_mesh` (ref gym_env.py:184: verts / faces of
    `map_cfg.road_mesh`) - as the [F, 3, 2] triangle soup `road_meshes` takes"""
    v = np.asarray(verts, dtype=np.float64).reshape(-1, np.asarray(verts).shape[-1])[:, :2]
    f = np.asarray(faces, dtype=np.int64).reshape(-1, 3)
    return v[f]


def traffic_lights_from_controller(stoplines, light_states, durations, dt=0.1):
    """The per-location traffic-light description `traffic_lights=` takes, from the pieces the reference's map config holds
    (ref gym_env.py:181-189: `map_cfg.stoplines` with agent_type 'traffic_light' and `map_cfg.traffic_light_controller`):

      stoplines     iterable of stop lines: objects / dicts with actor_id, x, y, orientation (or psi), length, width
                    (an `agent_type` other than 'traffic_light' - stop / yield signs - is skipped, as at gym_env.py:183)
      light_states  the controller's cycle: one mapping actor_id -> 'red' | 'yellow' | 'green' (anything but 'red' lets traffic
                    pass: compute_traffic_lights_violations only counts red lights) per phase
      durations     seconds each phase lasts (rounded to whole steps of `dt`, at least one)

    -> dict(stoplines=[(x, y, psi, length, width, light)], phases=[(n_steps, [red lights])], actor_ids=[...]) with `light` the
    index of the line's actor in actor_ids."""
    def get(o, *names, default=None):
        for n in names:
            v = o.get(n) if isinstance(o, dict) else getattr(o, n, None)
            if v is not None:
                return v
        return default

    ids, lines = [], []
    for sl in stoplines:
        if get(sl, "agent_type", default="traffic_light") != "traffic_light":
            continue
        aid = get(sl, "actor_id")
        if aid not in ids:
            ids.append(aid)
        lines.append((float(get(sl, "x")), float(get(sl, "y")), float(get(sl, "orientation", "psi", default=0.0)),
                      float(get(sl, "length", default=1.0)), float(get(sl, "width", default=3.5)), ids.index(aid)))
    assert len(light_states) == len(durations) and len(durations) >= 1, "one duration per phase of the light cycle"
    phases = []
    for states, sec in zip(light_states, durations):
        red = sorted(ids.index(a) for a, st in states.items() if a in ids and str(st).lower() == "red")
        phases.append((max(1, int(round(float(sec) / dt))), red))
    return dict(stoplines=lines, phases=phases, actor_ids=ids)


def _lights_of_location(traffic_lights, loc):
    """the caller's traffic lights of a location as dict(stoplines=[(x, y, psi, length, width, light)], phases=[(n_steps, [red])])
    with validated shapes, or None"""
    if traffic_lights is None or loc is None:
        return None
    spec = traffic_lights(loc) if callable(traffic_lights) else traffic_lights.get(loc)
    if spec is None or not spec.get("stoplines"):
        return None
    lines = [tuple(float(t) for t in sl[:5]) + (int(sl[5]),) for sl in spec["stoplines"]]
    phases = [(int(n), sorted(int(i) for i in red)) for n, red in spec["phases"]]
    if not phases or min(n for n, _ in phases) < 1 or min(sl[5] for sl in lines) < 0 or not np.isfinite(np.asarray([sl[:5] for sl in lines])).all():
        raise ValueError(f"traffic lights of location {loc!r}: stop lines are (x, y, psi, length, width, light >= 0), phases (n_steps >= 1, "
                         f"[red lights]) and there is at least one phase")
    return dict(stoplines=lines, phases=phases)


MAX_GROUP_LIGHTS = 32        # a light is a bit of a 32-bit mask (tde_light_phase.red_mask)


def _light_group(spec, polylines, radius):
    """The stop lines of `spec` a scenario sees: all of them when the location is small (at most 32 lights), else those within
    `radius` metres of the scenario's polylines - at most 32 distinct lights, the nearest ones (a light is a bit of a 32-bit mask
    and the kernels walk every stop line of a scenario's map descriptor) - with the lights re-numbered 0 .. n-1 inside the group.
    -> (sorted tuple of the kept stop-line indices, dict(stoplines, phases)) or (None, None) when nothing is in reach."""
    lines = spec["stoplines"]
    lights = sorted({sl[5] for sl in lines})
    keep = list(range(len(lines)))
    if len(lights) > MAX_GROUP_LIGHTS:
        pts = np.concatenate([np.asarray(pl, np.float64).reshape(-1, 2) for pl in polylines], 0)
        ctr = np.asarray([sl[:2] for sl in lines], np.float64)
        d = np.sqrt(((ctr[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).min(1)
        near = [i for i in range(len(lines)) if d[i] <= radius]
        by_light = {}
        for i in near:
            by_light[lines[i][5]] = min(by_light.get(lines[i][5], np.inf), d[i])
        if len(by_light) > MAX_GROUP_LIGHTS:
            order = sorted(by_light, key=lambda g: (by_light[g], g))
            warnings.warn(f"{len(by_light)} traffic lights within {radius:g} m of a scenario: the nearest {MAX_GROUP_LIGHTS} are kept "
                          f"(lower light_radius to choose differently)", stacklevel=3)
            by_light = {g: by_light[g] for g in order[:MAX_GROUP_LIGHTS]}
        keep = [i for i in near if lines[i][5] in by_light]
        lights = sorted(by_light)
    if not keep:
        return None, None
    local = {g: n for n, g in enumerate(lights)}
    grp = dict(stoplines=[lines[i][:5] + (local[lines[i][5]],) for i in keep],
               phases=[(n, [local[g] for g in red if g in local]) for n, red in spec["phases"]])
    return tuple(keep), grp


def _heading_table(start_headings, loc, p0, p1, n):
    """the lane direction at n points along the first waypoint segment p0 -> p1 (entry j at fraction (j + 0.5) / n), from the
    caller's heading field: what the reference reads with find_lanelet_directions(lanelet_map, x, y)[0] at the sampled start point
    (ref gym_env.py:359-361).  None when the caller gives no field for the location."""
    if start_headings is None or loc is None:
        return None
    f = start_headings
    if isinstance(f, dict):
        f = f.get(loc)
        if f is None:
            return None
        field = f if callable(f) else (lambda x, y, c=float(f): c)
    else:
        field = lambda x, y: start_headings(loc, x, y)       # noqa: E731
    out = []
    for j in range(n):
        t = (j + 0.5) / n
        psi = field(p0[0] + t * (p1[0] - p0[0]), p0[1] + t * (p1[1] - p0[1]))
        if psi is None:
            return None
        out.append(float(np.asarray(psi, dtype=np.float64).reshape(-1)[0]))    # (find_lanelet_directions returns a list: its first entry)
    if not np.isfinite(out).all():
        raise ValueError(f"start_headings of location {loc!r} returned a non-finite heading")
    return out


def world_from_waypoint_suite(data: WaypointSuite, agents_per_env=8, road_width=12.0, threshold=0.5,
                              background=None, background_radius=250.0, ego_only=False, road_meshes=None, near_range=None,
                              traffic_lights=None, light_radius=150.0, start_headings=None, heading_samples=16):
    """WaypointSuite -> World.  Agent ordering follows the reference: slot 0 ego, then the scenario's agents
    (ref gym_env.py:219-228); `car_sequence_suite[i][k]` replays slot k (ref gym_env.py:275-283).
    The CARLA town meshes the reference takes from torchdrivesim's package data are not available, so each scenario
    gets a synthetic drivable corridor around its waypoints, scenario agents and replay paths; non-replayed scenario
    agents get a straight route along their initial heading for the heuristic NPC controller.

    `background`: directory of background-traffic JSON files (ref gym_env.py:200-235), or a callable
    location -> dict from `loaders.load_background_traffic`.  As in the reference the ego takes the attributes of the
    file's first agent (:223) and the file's agents farther than 100 m from the ego start are kept (:227-231); the
    near field, which the reference fills through a remote INITIALIZE call (:232-235), stays empty.  Of the kept
    agents the nearest ones (within `background_radius`, so the synthetic corridor mesh stays local) fill the free
    slots after the scenario's own agents.  `ego_only`: the ego alone, no scenario / replay / background agents
    (ref gym_env.py:192-198).

    `road_meshes`: the drivable surface per location, as the reference takes it from `find_map_config(location).road_mesh`
    (ref gym_env.py:312, 184, 260): a dict location -> triangles ([n, 3, 2] or [n, 6] array, or the path of a .npy file holding
    one; `mesh_from_verts_faces` converts a verts / faces pair), or a callable location -> the same or None.  Scenarios of a
    location that has a mesh all run on ONE map built from it (one grid index per location); a location without one falls back to
    the synthetic corridor of its scenario.

    `traffic_lights`: the stop lines and light cycle per location, as the reference takes them from the map config
    (`map_cfg.stoplines` / `map_cfg.traffic_light_controller`, ref gym_env.py:181-189; fed to the NPCs :290-291, to is_terminated
    :415 and get_info :429): a dict location -> dict(stoplines=[(x, y, psi, length, width, light)], phases=[(n_steps, [red
    lights])]) or a callable location -> the same or None (`traffic_lights_from_controller` builds one from stop-line objects and
    a controller's state sequence).  The cycle restarts with every episode.  A location with more than 32 lights is cut into
    per-scenario neighbourhoods (the lines within `light_radius` metres of the scenario's waypoints, agents and replay paths; a
    light is a bit of a 32-bit mask): each distinct set becomes a light group - a map descriptor of its own that shares the
    location's mesh (assemble_world).  A world with lights steps with TDE_F_TRAFFIC_LIGHTS (BatchedWaypointEnv sets it): the ego's
    stop-line violation terminates the episode and shows in info["traffic_light_violation"], the NPCs stop at red lines, the
    birdview paints the lines.

    `start_headings`: the lane direction at the ego's start - what the reference reads with find_lanelet_directions(lanelet_map,
    x, y)[0] at the point it drew on the first waypoint segment (ref gym_env.py:357-361) - as a callable (location, x, y) -> psi
    or a dict location -> callable (x, y) -> psi (or a constant).  Sampled at `heading_samples` points along every scenario's
    first segment into the world's heading table (tde_world.start_psi); the episode's start heading is the entry of its drawn
    fraction + normal(0, 0.1).  Without it: the direction of the first segment.

    `near_range` (metres; default world.NEAR_RANGE = 2): how far beyond the offroad threshold the grid index carries NEAR LISTS, from
    which the MAGNITUDE of the offroad infraction (info["offroad"], ref gym_env.py:427) is two table look-ups; a corner farther out
    is still exact but found by scanning the grid, and beyond the reach where that square would hold more cells than the map has
    triangles by a walk over all the map's triangles (tens of microseconds per such ego and step).  Raise it (it costs table memory:
    a few records per square metre of the band) for `terminated_at_infraction=False` runs whose egos keep driving off the road."""
    from . import loaders
    meshes, scenarios = [], []
    map_of_location = {}
    light_groups, group_of = [], {}                  # (map id, kept stop lines) -> index into light_groups
    n = len(data.waypoint_suite)
    for i in range(n):
        wps = [tuple(p) for p in data.waypoint_suite[i]]
        scen = data.scenarios[i] if data.scenarios is not None and not ego_only else None
        seqs = (data.car_sequence_suite[i] if data.car_sequence_suite is not None and not ego_only else None) or {}
        seqs = {int(k): v for k, v in seqs.items()}
        polylines = [wps]
        agents = []
        if scen is not None and scen.agent_states:
            for k, (s, a) in enumerate(zip(scen.agent_states, scen.agent_attributes)):
                slot = k + 1
                x, y, psi, v = [float(t) for t in s[:4]]
                replay = seqs.get(slot)
                route = None
                if replay is None:
                    route = _straight_route(x, y, psi, v)
                    polylines.append([(x, y)] + route)
                else:
                    polylines.append([(r[0], r[1]) for r in replay[::10]] + [(replay[-1][0], replay[-1][1])])
                agents.append(dict(state=(x, y, psi, v), attr=tuple(float(t) for t in a[:3]), vdes=v, route=route,
                                   replay=[tuple(float(t) for t in r[:4]) for r in replay] if replay else None))
        for slot, replay in seqs.items():          # replay cars that are not scenario agents
            if slot - 1 >= len(agents) and replay:
                r0 = replay[0]
                agents.append(dict(state=tuple(float(t) for t in r0[:4]), attr=(5.0, 2.0, 1.9), vdes=0.0, route=None,
                                   replay=[tuple(float(t) for t in r[:4]) for r in replay]))
                polylines.append([(r[0], r[1]) for r in replay[::10]] + [(replay[-1][0], replay[-1][1])])
        ego_attr = None
        if background is not None and not ego_only:
            loc = data.locations[i] if data.locations else ""
            bt = background(loc) if callable(background) else loaders.pick_background_traffic(loc, background)
            if bt is not None and bt["agent_states"]:
                ego_attr = tuple(float(t) for t in bt["agent_attributes"][0][:3])
                far = [(math.dist(wps[0], s[:2]), k) for k, s in enumerate(bt["agent_states"])]
                far = sorted((d, k) for d, k in far if 100.0 < d <= background_radius)
                for _, k in far[:max(0, agents_per_env - 1 - len(agents))]:
                    x, y, psi, v = [float(t) for t in bt["agent_states"][k][:4]]
                    route = _straight_route(x, y, psi, v)
                    polylines.append([(x, y)] + route)
                    agents.append(dict(state=(x, y, psi, v), attr=tuple(float(t) for t in bt["agent_attributes"][k][:3]),
                                       vdes=v, route=route, replay=None))
        heading = math.atan2(wps[1][1] - wps[0][1], wps[1][0] - wps[0][0])
        loc = data.locations[i] if data.locations else None
        if loc is not None and loc in map_of_location:
            map_id = map_of_location[loc]
        else:
            tri = _mesh_of_location(road_meshes, loc) if loc is not None else None
            map_id = len(meshes)
            if tri is not None:
                map_of_location[loc] = map_id
                meshes.append(tri)
            else:
                meshes.append(corridor_mesh(polylines, width=road_width))
        if len(agents) > agents_per_env - 1:
            # the reference assembles up to ~100 agents per env (gym_env.py:216-237); an env here has agents_per_env slots
            # (a power of two <= TDE_MAX_AGENTS = 128): the scenario's own agents come first, what does not fit is dropped - loudly
            warnings.warn(f"scenario {i}: {len(agents)} non-ego agents but only {agents_per_env - 1} NPC slots "
                          f"(agents_per_env={agents_per_env}): the last {len(agents) - (agents_per_env - 1)} are dropped; "
                          f"raise agents_per_env (a power of two, at most {_abi.TDE_MAX_AGENTS})", stacklevel=2)
        scn = dict(map=map_id, waypoints=wps, start_heading=heading, agents=agents[:agents_per_env - 1])
        if ego_attr is not None:
            scn["ego_attr"] = ego_attr
        spec = _lights_of_location(traffic_lights, loc)
        if spec is not None:
            key, grp = _light_group(spec, polylines, float(light_radius))
            if key is not None:
                if (map_id, loc, key) not in group_of:
                    group_of[(map_id, loc, key)] = len(light_groups)
                    light_groups.append(dict(map=map_id, **grp))
                scn["lights"] = group_of[(map_id, loc, key)]
        table = _heading_table(start_headings, loc, wps[0], wps[1], int(heading_samples))
        if table is not None:
            scn["start_headings"] = table
        scenarios.append(scn)
    if any("start_headings" in sc for sc in scenarios):      # one table size per world: the others repeat their segment's direction
        for sc in scenarios:
            sc.setdefault("start_headings", [sc["start_heading"]] * int(heading_samples))
    from .world import NEAR_RANGE
    return assemble_world(meshes, scenarios, agents_per_env, threshold=threshold, light_groups=light_groups,
                          near_range=NEAR_RANGE if near_range is None else float(near_range))


class _LazyInfo(dict):
    """dict of per-env info tensors (ref gym_env.py:419-437) whose values are built when first read.  The entries are VIEWS of the
    env's own output buffers (like reward / terminated / truncated): valid until the next step() overwrites them - clone what
    must be kept."""

    KEYS = ("offroad", "collision", "traffic_light_violation", "is_success")
    EXTRA = ("reached_waypoint_num", "psi_smoothness", "speed_smoothness", "psi_reward", "dist_reward")

    ALL = KEYS + EXTRA

    def __init__(self, st, B, A, magnitudes=None):
        self._st, self._BA = st, (B, A)
        self._mag = magnitudes                # float32 [B, 4] (offroad, collision, count, -) of tde_ego_infractions, or None: 0 / 1 indicators

    @property
    def _keys(self):
        return self.ALL if self._st["info"] is not None else self.KEYS

    @property
    def _ego(self):
        return slice(0, self._BA[0] * self._BA[1], self._BA[1])

    def _make(self, k):
        st = self._st
        bits = st["done_bits"]                # the ego's flags of this step, kept when the env re-spawned in place
        if k == "offroad":
            if self._mag is not None:
                return self._mag[:, 0]
            return ((bits >> 2) & 1).float() if bits is not None else st["offroad"][self._ego].float()
        if k == "collision":
            if self._mag is not None:
                return self._mag[:, 1]
            return ((bits >> 3) & 1).float() if bits is not None else st["collided"][self._ego].float()
        if k == "traffic_light_violation":
            return st["tl_violation"].float()
        if k == "is_success":
            return st["truncated"].view(torch.bool)
        if k == "reached_waypoint_num":
            return st["info_reached"]
        return st["info"][:, ("psi_smoothness", "speed_smoothness", "psi_reward", "dist_reward").index(k)]

    def __missing__(self, k):
        if k not in self._keys:
            raise KeyError(k)
        v = self._make(k)
        dict.__setitem__(self, k, v)
        return v

    def __contains__(self, k):
        return k in self._keys

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def keys(self):
        return list(self._keys)

    def items(self):
        return [(k, self[k]) for k in self._keys]

    def values(self):
        return [self[k] for k in self._keys]

    def get(self, k, default=None):
        return self[k] if k in self._keys else default


class BatchedWaypointEnv:
    """`num_envs` independent WaypointSuite envs stepped by one fused HIP kernel per timestep.

    step(actions [B,2]) -> (obs, reward f32[B], terminated bool[B], truncated bool[B], info dict of [B] tensors), all
    device-resident torch tensors.  obs is the ego-centred birdview uint8 [B, 3*frame_stack, 64, 64] (obs_mode
    "birdview", the reference's observation, ref gym_env.py:95,122-124) or a compact float32 [B, 8] kinematic vector
    (obs_mode "state").  Finished envs are re-spawned inside the same kernel (auto_reset=True); with a frame stack their
    older frames restart blank in the same observation (VecFrameStack semantics, ref examples/rl_training.py:160).
    reward / terminated / truncated are the env's own buffers, overwritten by the next step (clone what must be
    kept).  (Replaying step + observation from a captured HIP graph was measured and is slower than the two direct
    launches: 27.9 vs 18.5 us per step at 8192 envs.)

    The SB3 `VecEnv` interface (numpy in / out, `terminal_observation`, `TimeLimit.truncated`, Monitor's
    `info["episode"]`) is `as_vec_env()` -> WaypointVecEnv."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 10}   # ref gym_env.py:73-76

    def __init__(self, cfg: EnvConfig, data, num_envs, agents_per_env=16, device=None, obs_mode="birdview",
                 frame_stack=1, auto_reset=True, with_info=True, background=None, env_base=0, binding="ext",
                 info_magnitudes=True, road_meshes=None, near_range=None, traffic_lights=None, start_headings=None, light_radius=150.0,
                 heading_samples=16):
        """binding: "ext" = launches go through the PyTorch-ROCm C++ extension (csrc/tde_torch_ext.cpp), "ctypes" = through
        the ctypes binding of the same C-ABI (ops.py); both call the very same entry points of libtde_hip.so.
        info_magnitudes (default): info["offroad"] / info["collision"] hold the MAGNITUDES the reference reports there (ref
        gym_env.py:427-428: sum over the ego's corners of clamp(distance - threshold, 0); sum of the IoUs with the agents the ego
        overlaps), written by the step kernel itself for the egos it flagged, before a finished env is re-spawned
        (tde_state.magnitudes: still ONE launch per step).  False: 0 / 1 indicators (the kernel then skips the magnitudes).
        traffic_lights / start_headings (light_radius, heading_samples): the stop lines + light cycle and the lane-direction field per
        location, as the reference takes them from the map config (ref gym_env.py:181-189, 359-361): world_from_waypoint_suite.
        road_meshes / near_range: the drivable mesh per location and the reach of the grid index's near lists when `data` is a
        WaypointSuite (world_from_waypoint_suite).  With `terminated_at_infraction=False` an ego may drive far off the road, where
        the exact offroad magnitude is found by a scan of the grid or a walk over the map's triangles (tens of microseconds per such
        ego and step; `near_range` moves that point outwards, info_magnitudes=False skips the work)."""
        validate(cfg)
        if binding not in ("ext", "ctypes"):
            raise ValueError("binding must be 'ext' or 'ctypes'")                                              # ref gym_env.py:79-80 and the fields this path rejects
        if obs_mode not in ("birdview", "state"):
            raise ValueError("obs_mode must be 'birdview' or 'state'")
        self.config = cfg
        dev = device or cfg.device or ("cuda" if torch.cuda.is_available() else None)
        if dev is None or not torch.cuda.is_available():
            raise RuntimeError("torchdriveenv_amd needs a HIP device: there is no CPU path")
        self.torch_device = torch.device(dev if str(dev) != "cuda" else "cuda:0")
        if background is None and cfg.use_background_traffic:        # ref gym_env.py:200-203: the packaged directory
            from .loaders import pick_background_traffic
            background = pick_background_traffic                     # (searched under TORCHDRIVEENV_DATA; None if absent)
        sim = cfg.simulator
        self.world = data if isinstance(data, World) else world_from_waypoint_suite(
            data, agents_per_env,
            threshold=effective_offroad_distance(sim.offroad_threshold, sim.offroad_threshold_squared),
            background=background if cfg.use_background_traffic else None, ego_only=cfg.ego_only, road_meshes=road_meshes,
            near_range=near_range, traffic_lights=traffic_lights, light_radius=light_radius, start_headings=start_headings,
            heading_samples=heading_samples)
        check_threshold(self.world, sim.offroad_threshold, sim.offroad_threshold_squared,
                        "EnvConfig.simulator.offroad_threshold")   # a prebuilt World bakes its threshold into the grid
        self.A = self.world.A
        self.num_envs = int(num_envs)
        seed = cfg.seed if cfg.seed is not None else int(np.random.randint(0, 2**31 - 1))  # ref helpers.py:39-41
        self.seed_value = seed
        flags = _abi.F_NPC | _abi.F_REPLAY | _abi.F_OFFROAD | _abi.F_REWARD
        if auto_reset:
            flags |= _abi.F_AUTORESET
        if cfg.ego_only:
            flags |= _abi.F_EGO_ONLY_ATTRS
        if self.world.has_lights:
            flags |= _abi.F_TRAFFIC_LIGHTS
        if cfg.simulator.npc_first_step:                             # ref gym_env.py:285-294: the NPCs act from step one
            flags |= _abi.F_NPC_FIRST_STEP
        self.tde_cfg = to_tde_config(cfg, seed, flags)
        self.tde_cfg.env_base = int(env_base)                        # shard of a larger batch: sharding.ShardedBatchedEnv
        self.dworld = self.world.to_device(self.torch_device)
        # obs_mode "state": tde_env_step writes the compact observation itself (no second launch per step)
        self.info_magnitudes = bool(info_magnitudes)
        self.state = EnvState(self.num_envs, self.A, device=self.torch_device, with_info=with_info,
                              with_obs=(obs_mode == "state"), with_magnitudes=self.info_magnitudes)
        self.obs_mode, self.frame_stack = obs_mode, max(1, int(frame_stack))
        r = cfg.simulator.renderer
        self._res, self._fov = int(r.res), float(r.fov)
        self._rflags = render_flags(cfg)
        self._obs = self._stack = None
        self.action_space = _box(ACTION_LOW, ACTION_HIGH)
        self.observation_space = (_box(0, 255, (3 * self.frame_stack, self._res, self._res), np.uint8)
                                  if obs_mode == "birdview" else _box(-np.inf, np.inf, (8,), np.float32))
        self.reward_range = (-float("inf"), float("inf"))           # ref gym_env.py:97
        self._vec = None
        self._h = None
        self._mag = self.state["magnitudes"]                         # float32 [B, 4] written by the step (None: indicators)
        if binding == "ext":
            from . import _ext
            self._h = _ext.env_handle(self.tde_cfg, self.dworld, self.state)
        # the world's first-step gap cache for this configuration, now - not inside the first step() (which may be under a stream capture)
        ops.first_gaps(self.tde_cfg, self.dworld)

    @property
    def auto_reset(self):
        return bool(self.tde_cfg.flags & _abi.F_AUTORESET)

    # ---- device-resident API ------------------------------------------------------------------------------
    def reset(self, seed=None, options=None, mask=None):
        """re-spawn all envs (or those in `mask`, uint8/bool [B]).  `seed` is ignored like in the reference
        (ref gym_env.py:107-109); seeding is EnvConfig.seed.  With a mask only the re-spawned envs' observations
        change: their newest frame is re-rendered in place and their older frames are blanked; the other envs keep
        their frame stack exactly as the last step left it."""
        m = None
        if mask is not None:
            # 0 / 1 bytes: the rasteriser reads bits 0-1 of a `fresh` byte (tde_render.fresh), so a mask like done_bits with only
            # infraction bits set must not re-spawn an env and leave its older stack frames un-blanked
            m = (torch.as_tensor(mask, device=self.torch_device) != 0).to(torch.uint8).contiguous()
        if m is not None and self.obs_mode == "birdview" and self._obs is not None:
            # the SB3-style auto-reset: the re-spawn and the re-spawned views' first observation in ONE C-ABI call
            if self._stack is not None:
                self._obs = self._stack.reset_rerender(self.tde_cfg, self.dworld, self.state, m, self._fov)
            elif self._h is not None:
                self._h.reset_render(m, int(self.tde_cfg.flags), self._obs, self._res, self._res, self._fov, 1, None, 0, self._rflags)
            else:
                ops.env_reset_render(self.tde_cfg, self.dworld, self.state, m, self._obs, self._res, self._res, self._fov, 1,
                                     flags=self._rflags)
            return self._obs
        if self._h is not None:
            self._h.reset(m, int(self.tde_cfg.flags))
        else:
            ops.env_reset(self.tde_cfg, self.dworld, self.state, m)
        if self.obs_mode == "state" or m is None or self._obs is None:
            if self._stack is not None:
                self._stack.clear()                                  # VecFrameStack clears the stack on reset
            return self.get_obs()
        if self._stack is not None:
            self._obs = self._stack.rerender(self.tde_cfg, self.dworld, self.state, m, self._fov)
        else:
            self._render1(self._obs, only=m)
        return self._obs

    def _flag_views(self):
        """the state's terminated / truncated bytes seen as bool (no copy), formed once: the buffers never move"""
        v = self.__dict__.get("_tt_views")
        st = self.state
        if v is None or v[0] is not st["terminated"]:
            v = self._tt_views = (st["terminated"], st["terminated"].view(torch.bool), st["truncated"].view(torch.bool))
        return v[1], v[2]

    def step(self, actions):
        # (the host side of a step is what a closed loop of small kernels waits for: no tensor op that is not needed)
        a = actions
        if not (torch.is_tensor(a) and a.dtype is torch.float32 and a.device == self.torch_device and a.dim() == 2
                and a.shape[0] == self.num_envs and a.shape[1] == 2 and a.is_contiguous()):
            a = torch.as_tensor(actions, dtype=torch.float32, device=self.torch_device).reshape(self.num_envs, 2).contiguous()
        if self._h is not None:
            self._h.step(a, int(self.tde_cfg.flags))
        else:
            ops.env_step(self.tde_cfg, self.dworld, self.state, action=a)
        st = self.state
        if self.obs_mode == "state":
            obs = st["obs"]
        else:
            fresh = None
            if self.auto_reset and self.frame_stack > 1:
                # envs that finished were re-spawned inside the kernel: their frame stack restarts blank.  The render
                # kernel reads bits 0-1 of the mask, which is exactly the done part of done_bits
                fresh = st["done_bits"] if st["done_bits"] is not None else (st["terminated"] | st["truncated"])
            obs = self.get_obs(fresh)
        # uint8 0/1 flags seen as bool without a copy; info entries are only computed when they are read
        term, trunc = self._flag_views()
        return obs, st["reward"], term, trunc, _LazyInfo(st, self.num_envs, self.A, magnitudes=self._mag)

    def _step_then_post_step(self, a):
        """the round-4 form of a step with magnitudes, kept for A/B runs and as a second witness of the fused path
        (tests/test_gpu_magnitudes.py): step without in-kernel re-spawn -> tde_env_post_step (the magnitudes of the infractions the
        step flagged + the re-spawn of the envs it finished, one launch) -> the observation.  Same results as step(), one launch
        more."""
        st = self.state
        if self._mag is None:
            raise RuntimeError("needs info_magnitudes=True")
        full = int(self.tde_cfg.flags)
        flags = full & ~_abi.F_AUTORESET
        post = flags | (full & _abi.F_AUTORESET)            # (WaypointVecEnv.step_wait clears the flag: it re-spawns the envs itself)
        self.tde_cfg.flags = flags
        try:
            if self._h is not None:
                self._h.step(a, flags)
                self._h.post_step(self._mag, post)
            else:
                ops.env_step(self.tde_cfg, self.dworld, st, action=a)
                self.tde_cfg.flags = post
                ops.env_post_step(self.tde_cfg, self.dworld, st, self._mag)
        finally:
            self.tde_cfg.flags = full
        term, trunc = self._flag_views()
        info = _LazyInfo(st, self.num_envs, self.A, magnitudes=self._mag)
        if self.obs_mode == "state":
            obs = st["obs"]                                           # written by the step, refreshed by the re-spawn
        else:
            # the finished envs were re-spawned: their frame stacks restart blank (bits 0-1 of done_bits = the step's done flags)
            fresh = None
            if (full & _abi.F_AUTORESET) and self.frame_stack > 1:
                fresh = st["done_bits"] if st["done_bits"] is not None else (st["terminated"] | st["truncated"])
            obs = self.get_obs(fresh)
        return obs, st["reward"], term, trunc, info

    def rollout(self, actions):
        """K open-loop steps from a resident [K,B,2] action tensor -> (reward [K,B], done bits [K,B]).  (The Monitor-style
        episode statistics of the closed-loop API are not maintained across a rollout: they follow from the returned
        arrays.)"""
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.torch_device).contiguous()
        if self._h is not None:
            reward = torch.empty(a.shape[:2], dtype=torch.float32, device=self.torch_device)
            done = torch.empty(a.shape[:2], dtype=torch.uint8, device=self.torch_device)
            self._h.rollout(a, reward, done, int(self.tde_cfg.flags))
            return reward, done
        return ops.env_rollout(self.tde_cfg, self.dworld, self.state, a)

    def get_obs(self, fresh=None):
        if self.obs_mode == "state":
            # x, y, psi, v, target offset (forward, left) in the ego frame, target-exists flag, environment_steps
            if self._h is not None:
                self._h.state_obs(self.state["obs"])
                return self.state["obs"]
            return ops.state_obs(self.dworld, self.state, self.state["obs"])
        if self.frame_stack > 1:
            if self._stack is None:
                self._stack = ops.FrameStack(self.num_envs, self.frame_stack, self._res, self._res, self.torch_device,
                                             flags=self._rflags, handle=self._h)
            self._obs = self._stack.render(self.tde_cfg, self.dworld, self.state, self._fov, fresh=fresh)
            return self._obs
        if self._obs is None:
            self._obs = torch.zeros((self.num_envs, 3, self._res, self._res), dtype=torch.uint8, device=self.torch_device)
        self._render1(self._obs)
        return self._obs

    def _render1(self, out, only=None):
        """single-frame raster of every (or the masked) view into `out`"""
        if self._h is not None:
            self._h.render(out, self._res, self._res, self._fov, 1, None, 0, self._rflags, None, only)
        else:
            ops.render_ego(self.tde_cfg, self.dworld, self.state, self._res, self._res, self._fov, 1, out,
                           flags=self._rflags, only=only)

    def get_info(self):
        """info schema of the reference (ref gym_env.py:419-437), one entry per env; a mapping whose tensors are formed
        on first access (a training loop that never reads `psi_smoothness` does not pay for it)"""
        return _LazyInfo(self.state, self.num_envs, self.A, magnitudes=self._mag)

    def render(self):
        """(B, H, W, 3) uint8 of the current ego views (ref gym_env.py:152-155)"""
        img = ops.render_ego(self.tde_cfg, self.dworld, self.state, self._res, self._res, self._fov, 1, flags=self._rflags)
        return img.permute(0, 2, 3, 1).cpu().numpy()

    def close(self):
        pass

    def state_dict(self):
        """snapshot of every mutable buffer (checkpoint / parity replays), the frame-stack ring included"""
        sd = {k: v.clone() for k, v in self.state.arrays.items() if v is not None}
        if self._stack is not None:
            sd["_frame_stack"] = self._stack.state_dict()
        return sd

    def load_state_dict(self, sd):
        for k, v in sd.items():
            if k == "_frame_stack":
                if self._stack is None:
                    self._stack = ops.FrameStack(self.num_envs, self.frame_stack, self._res, self._res, self.torch_device,
                                                 flags=self._rflags, handle=self._h)
                self._stack.load_state_dict(v)
                self._obs = self._stack.obs
            else:
                self.state.arrays[k].copy_(v)

    # ---- SB3 VecEnv-shaped numpy API --------------------------------------------------------------------
    def as_vec_env(self, **kw):
        """the SB3 `VecEnv` over this batch (one shared adapter per env object)"""
        if self._vec is None:
            self._vec = WaypointVecEnv(self, **kw)
        return self._vec

    def step_async(self, actions):
        self.as_vec_env().step_async(actions)

    def step_wait(self):
        return self.as_vec_env().step_wait()

    def vec_step(self, actions):
        return self.as_vec_env().step(actions)

    def vec_reset(self):
        return self.as_vec_env().reset()


try:  # optional: stable_baselines3 does not ship in this image
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:  # pragma: no cover
    _SB3VecEnv = None


class LazyInfos:
    """The `infos` list of a VecEnv step (one mapping per env) without building num_envs dicts per step: entry i is
    assembled from the per-env arrays when it is read.  Behaves like a list of dicts for the consumers SB3 has
    (`infos[i]`, iteration, `len`, `.get("episode")`, `.get("terminal_observation")`, `.get("TimeLimit.truncated")`)."""

    INFO_COLS = ("psi_smoothness", "speed_smoothness", "psi_reward", "dist_reward")

    def __init__(self, n, cols, terminal):
        self._n, self._cols, self._terminal = n, cols, terminal   # cols: name -> [n] array; terminal: env -> extra entries
        self._made = {}                                           # env -> the dict handed out (writes to it stick)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        d = self._made.get(i)
        if d is None:
            # the SAME dict on every access: SB3 wrappers rewrite infos[i]["terminal_observation"] in step_wait
            # (VecNormalize, VecTransposeImage, VecFrameStack) and read it back through infos[i] later
            d = {k: v[i].item() for k, v in self._cols.items()}
            extra = self._terminal.get(i)
            if extra:
                d.update(extra)
            self._made[i] = d
        return d

    def __setitem__(self, i, d):
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        self._made[i] = d

    @property
    def columns(self):
        """name -> per-env array of every info column (what ShardedBatchedEnv concatenates)"""
        return self._cols

    @property
    def terminal(self):
        """env -> the extra entries of the envs that finished at this step (terminal_observation, episode)"""
        return self._terminal

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def column(self, key):
        """the whole per-env column of `key` as one array (what a vectorised consumer should read)"""
        return self._cols[key]


class LazyTerminal:
    """env -> the extra info entries of the envs that finished at a step (`terminal_observation`, Monitor's `episode`), as a
    read-only mapping that builds an entry when it is asked for: at 8192 envs some env finishes at every step, and a dict per
    finished env per step was a Python loop on the step's critical path.  `idx`: the finished envs, ascending; `obs_of(n)`: the
    terminal observation of the n-th of them; ep_r / ep_l: per-ENV arrays (or None)."""

    def __init__(self, idx, obs_of, ep_r, ep_l, t):
        self._idx, self._obs_of, self._r, self._l, self._t = idx, obs_of, ep_r, ep_l, t

    def _n(self, i):
        n = int(np.searchsorted(self._idx, i))
        return n if n < len(self._idx) and self._idx[n] == i else None

    def get(self, i, default=None):
        n = self._n(i)
        if n is None:
            return default
        ex = {"terminal_observation": self._obs_of(n)}
        if self._r is not None:                                  # Monitor: round(sum(rewards), 6), len(rewards), elapsed
            ex["episode"] = {"r": round(float(self._r[i]), 6), "l": int(self._l[i]), "t": self._t}
        return ex

    def __contains__(self, i):
        return self._n(i) is not None

    def __len__(self):
        return len(self._idx)

    def __iter__(self):
        return (int(i) for i in self._idx)

    def __getitem__(self, i):
        ex = self.get(i)
        if ex is None:
            raise KeyError(i)
        return ex

    def keys(self):
        return [int(i) for i in self._idx]

    def items(self):
        return ((int(i), self.get(int(i))) for i in self._idx)


class WaypointVecEnv(_SB3VecEnv if _SB3VecEnv is not None else object):
    """stable_baselines3 `VecEnv` over a BatchedWaypointEnv, replacing `SubprocVecEnv` + `Monitor` + `VecFrameStack` of the
    reference's trainer (ref examples/rl_training.py:121-129, 159-160): numpy actions in, (obs, rewards, dones, infos)
    out, finished envs re-spawned at once with `terminal_observation`, `TimeLimit.truncated` and Monitor's
    `episode = {"r", "l", "t"}` in their info.  Subclasses SB3's VecEnv when stable_baselines3 is importable; otherwise
    the same methods on a plain class (tests/test_features_cpu.py walks the abstract interface).

    Per step: one fused step launch, one observation launch, ONE packed device-to-host copy of all per-env outputs and
    one of the observation, one stream synchronisation; `infos` is a LazyInfos.  copy_obs=False hands out views of
    rings of three pinned buffers (observations, rewards, info columns and terminal observations stay valid for the two following
    steps) instead of fresh arrays.  With the compact observation (obs_mode="state") a step is ONE stream synchronisation, the
    re-spawn of the finished envs included; the birdview needs a second one for the re-spawned envs' first frames."""

    def __init__(self, env, copy_obs=True, info_keywords=("offroad", "collision", "traffic_light_violation", "is_success",
                                                           "reached_waypoint_num", "psi_smoothness", "speed_smoothness",
                                                           "psi_reward", "dist_reward"), obs_buffers=None):
        """obs_buffers: optional list of host tensors (pinned, or page-locked by the caller with hipHostRegister) of the
        observation's shape to use as the ring the device-to-host copies land in - how a shard of ShardedBatchedEnv lets
        its observations arrive directly in its slice of the gathered buffer"""
        import time

        self.env = env
        if _SB3VecEnv is not None:
            _SB3VecEnv.__init__(self, env.num_envs, env.observation_space, env.action_space)
        else:
            self.num_envs, self.observation_space, self.action_space = env.num_envs, env.observation_space, env.action_space
            self.reset_infos = [{} for _ in range(env.num_envs)]
            self._seeds = [None] * env.num_envs
            self._options = [{} for _ in range(env.num_envs)]
        self.render_mode = "rgb_array"
        self.copy_obs = bool(copy_obs)
        self.info_keywords = tuple(k for k in info_keywords
                                   if env.state["info"] is not None or k not in LazyInfos.INFO_COLS + ("reached_waypoint_num",))
        self._t_start = time.time()
        self._pending = None
        shape = (env.num_envs,) + tuple(env.observation_space.shape)
        odt = torch.uint8 if env.obs_mode == "birdview" else torch.float32
        if obs_buffers is not None:
            for b in obs_buffers:
                if tuple(b.shape) != shape or b.dtype != odt or b.is_cuda or not b.is_contiguous():
                    raise ValueError(f"obs_buffers must be contiguous host tensors of shape {shape} and dtype {odt}")
            self._obs_ring = list(obs_buffers)
        else:
            self._obs_ring = [torch.empty(shape, dtype=odt, pin_memory=True) for _ in range(1 if copy_obs else 3)]
        self._turn = 0
        self._act_pin = self._act_np = self._act_dev = None
        self._obs2_ring = None

    # ---- helpers
    def _obs_to_host(self, obs, rows=None):
        buf = self._obs_ring[self._turn % len(self._obs_ring)]
        buf.copy_(obs, non_blocking=True)
        return buf

    def reset(self):
        obs = self.env.reset()
        buf = self._obs_to_host(obs)
        torch.cuda.current_stream(self.env.torch_device).synchronize()
        self._turn += 1
        self.reset_infos = [{} for _ in range(self.num_envs)]
        out = buf.numpy()
        return out.copy() if self.copy_obs else out

    def step_async(self, actions):
        self._pending = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        import time

        if self._pending is None:
            raise RuntimeError("step_wait() without step_async()")
        env = self.env
        st = env.state
        acts, self._pending = self._pending, None
        auto = env.auto_reset
        # the actions cross the bus from a pinned staging buffer, asynchronously (a pageable source makes the copy synchronous)
        if self._act_pin is None:
            self._act_pin = torch.empty((self.num_envs, 2), dtype=torch.float32, pin_memory=True)
            self._act_np = self._act_pin.numpy()
            self._act_dev = torch.empty((self.num_envs, 2), dtype=torch.float32, device=env.torch_device)
        self._act_np[...] = acts.reshape(self.num_envs, 2)
        self._act_dev.copy_(self._act_pin, non_blocking=True)
        # terminal_observation needs the pre-reset frame: run the step without in-kernel reset, then re-spawn the
        # finished envs with the masked reset kernel
        env.tde_cfg.flags &= ~_abi.F_AUTORESET
        try:
            obs, _, _, _, _ = env.step(self._act_dev)
        finally:
            if auto:
                env.tde_cfg.flags |= _abi.F_AUTORESET
        one_sync = auto and env.obs_mode == "state" and env._h is not None
        pre = None
        if one_sync:
            # compact observation: the observation the step left goes to an internal ring (the finished envs' terminal
            # observations), then the re-spawn of the finished envs (tde_env_post_step: one launch, it rewrites exactly their
            # observation rows) and the copy of the 32-byte rows into the caller-visible ring are queued behind it - ONE
            # synchronisation per step, no mask upload, no gather of the finished rows (round 4: two synchronisations per step)
            if self._obs2_ring is None:
                self._obs2_ring = [torch.empty(self._obs_ring[0].shape, dtype=self._obs_ring[0].dtype, pin_memory=True) for _ in range(3)]
            pre = self._obs2_ring[self._turn % 3]
            pre.copy_(obs, non_blocking=True)
            st.copy_outputs_async(ring=3)                            # ONE copy of every per-env output, queued (no synchronisation yet)
            env._h.post_step(None, int(env.tde_cfg.flags))
            buf = self._obs_to_host(st["obs"])
        else:
            buf = self._obs_to_host(obs)
            st.copy_outputs_async(ring=3)
        torch.cuda.current_stream(env.torch_device).synchronize()
        out = st.outputs_views()
        self._turn += 1
        obs_np = buf.numpy()
        pre_np = pre.numpy() if one_sync else obs_np                 # the observation the step left (terminal for the finished envs)
        term, trunc = out["terminated"].view(np.bool_), out["truncated"].view(np.bool_)      # (0 / 1 bytes seen as bool: no copy)
        done = term | trunc
        bits = out.get("done_bits")
        cols = {"TimeLimit.truncated": trunc & ~term}
        mag = out.get("magnitudes")                                  # (the reference's magnitudes, part of the same packed copy)
        for k in self.info_keywords:
            if k == "offroad":
                cols[k] = mag[:, 0] if mag is not None else ((bits >> 2) & 1).astype(np.float32) if bits is not None else None
            elif k == "collision":
                cols[k] = mag[:, 1] if mag is not None else ((bits >> 3) & 1).astype(np.float32) if bits is not None else None
            elif k == "traffic_light_violation":
                cols[k] = out["tl_violation"].astype(np.float32)
            elif k == "is_success":
                cols[k] = trunc
            elif k == "reached_waypoint_num":
                cols[k] = out["info_reached"]
            else:
                cols[k] = out["info"][:, LazyInfos.INFO_COLS.index(k)]
        cols = {k: v for k, v in cols.items() if v is not None}
        rew = out["reward"]
        if self.copy_obs:                                            # fresh arrays: nothing handed out aliases a staging buffer
            cols = {k: v.copy() for k, v in cols.items()}
            rew = rew.copy()
        terminal = {}
        idx = np.flatnonzero(done)
        if len(idx):
            ep_r, ep_l = out.get("ep_final"), out.get("ep_final_len")
            t = round(time.time() - self._t_start, 6)
            if one_sync and not self.copy_obs:
                obs_of = lambda n: pre_np[idx[n]]                    # noqa: E731  (a view of the internal ring: valid for the next two steps)
            else:
                term_rows = pre_np[idx]                              # (fancy index: a copy, taken before the rows are overwritten below)
                obs_of = lambda n: term_rows[n]                      # noqa: E731
            if auto and not one_sync:
                # Only the re-spawned envs are reset and re-rendered; only their first observations cross the bus.  The reset
                # mask is formed ON the device from the step's own outputs (no upload), the gathered rows land in a pinned
                # staging buffer with an asynchronous copy, and the stream is synchronised a second time
                new = env.reset(mask=st["terminated"] | st["truncated"])
                sel = new.index_select(0, torch.from_numpy(idx).to(env.torch_device, non_blocking=True))
                pin = self._sel_staging(len(idx), sel)
                pin.copy_(sel, non_blocking=True)
                torch.cuda.current_stream(env.torch_device).synchronize()
                obs_np[idx] = pin.numpy()
            if self.copy_obs and ep_r is not None:
                ep_r, ep_l = ep_r.copy(), ep_l.copy()
            terminal = LazyTerminal(idx, obs_of, ep_r, ep_l, t)
        if self.copy_obs:
            obs_np = obs_np.copy()
        return obs_np, rew, done, LazyInfos(self.num_envs, cols, terminal)

    def _sel_staging(self, n, like):
        """pinned host rows for the first observations of the envs that re-spawned at this step (grown on demand)"""
        cap = getattr(self, "_sel_pin", None)
        if cap is None or cap.shape[0] < n:
            rows = max(256, 1 << (int(n) - 1).bit_length())
            self._sel_pin = cap = torch.empty((rows,) + tuple(like.shape[1:]), dtype=like.dtype, pin_memory=True)
        return cap[:n]

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.env, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self.env, method_name)(*method_args, **method_kwargs)] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def seed(self, seed=None):                                      # ref gym_env.py:149-150 (a no-op there too)
        return [None] * self.num_envs

    def get_images(self):
        return list(self.env.render())

    def render(self, mode=None):
        return self.env.render()

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else list(indices)


class WaypointSuiteEnv(_GymEnvBase):
    """B = 1 env with the reference's (B, A_exposed) = (1, 1) tensor interface (ref gym_env.py:303-437): `step` takes
    a (1,1,2) float32 tensor and returns obs (1,1,3,64,64) uint8 ndarray, reward float, terminated bool, truncated
    bool, info with (1,1) tensors — what `SingleAgentWrapper` then squeezes."""

    metadata = {"render_modes": ["video", "rgb_array"], "render_fps": 10}

    def __init__(self, cfg: EnvConfig, data, agents_per_env=8, road_meshes=None, traffic_lights=None, start_headings=None):
        self.config = cfg
        # info["offroad"] / info["collision"] carry the MAGNITUDES of compute_offroad() / compute_collision(), as the reference's
        # get_info reports them (ref gym_env.py:427-428; Monitor logs them, examples/rl_training.py:128)
        self._env = BatchedWaypointEnv(cfg, data, num_envs=1, agents_per_env=agents_per_env, obs_mode="birdview",
                                       frame_stack=1, auto_reset=False, info_magnitudes=True, road_meshes=road_meshes,
                                       traffic_lights=traffic_lights, start_headings=start_headings)
        self.torch_device = self._env.torch_device
        self.render_mode = cfg.render_mode
        self.max_environment_steps = cfg.max_environment_steps
        self.action_space = _box(ACTION_LOW, ACTION_HIGH)
        self.observation_space = _box(0, 255, (3, 64, 64), np.uint8)
        self.reward_range = (-float("inf"), float("inf"))
        self.collision_threshold = 0.0
        self.offroad_threshold = 0.0
        self._obs_pin = None

    @property
    def environment_steps(self):
        return int(self._env.state["steps"][0])

    @property
    def current_target_idx(self):
        return int(self._env.state["target_idx"][0])

    @property
    def reached_waypoint_num(self):
        return int(self._env.state["reached"][0])

    def reset(self, seed=None, options=None):                       # ref gym_env.py:319-349
        obs = self._env.reset()
        return obs.cpu().numpy().reshape(1, 1, 3, 64, 64).astype(np.uint8), {}

    def step(self, action):                                         # ref gym_env.py:369-389
        """one timestep.  Everything the reference's step returns crosses the bus in TWO asynchronous copies - the observation
        and the packed per-env outputs - behind ONE stream synchronisation (a `.item()` / `.cpu()` per returned value was a dozen
        synchronisations per step); the (1, 1) info tensors are therefore host tensors (the reference's live on `torch_device`;
        SingleAgentWrapper moves them to the CPU either way, gym_env.py:463-472)."""
        env = self._env
        st = env.state
        obs, _, _, _, _ = env.step(torch.as_tensor(action, dtype=torch.float32).reshape(1, 2))
        if self._obs_pin is None:
            self._obs_pin = torch.empty((1, 3, 64, 64), dtype=torch.uint8, pin_memory=True)
        self._obs_pin.copy_(obs, non_blocking=True)
        out = st.fetch_outputs()                                     # (the one synchronisation: the observation copy is ahead of it)
        mag = out["magnitudes"][0]
        t11 = lambda v: torch.tensor([[v]], dtype=torch.float32)    # noqa: E731
        trunc = bool(out["truncated"][0])
        info = {"offroad": t11(mag[0]), "collision": t11(mag[1]), "traffic_light_violation": t11(float(out["tl_violation"][0])),
                "is_success": trunc}
        if "info" in out:
            inf = out["info"][0]
            info.update(reached_waypoint_num=int(out["info_reached"][0]), psi_smoothness=float(inf[0]), speed_smoothness=float(inf[1]),
                        psi_reward=float(inf[2]), dist_reward=float(inf[3]))
        return (self._obs_pin.numpy().reshape(1, 1, 3, 64, 64).copy(), float(out["reward"][0]), bool(out["terminated"][0]), trunc, info)

    def render(self):                                               # ref gym_env.py:152-157
        if self.render_mode == "rgb_array":
            return self._env.render()[0]
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        pass


class SingleAgentWrapper(_GymWrapperBase):
    """Removes batch and agent dimensions (ref gym_env.py:440-487): numpy (2,) action in, obs (3,64,64) uint8,
    reward float, terminated bool, truncated bool, info with 0-d CPU tensors out."""

    def __init__(self, env):
        super().__init__(env)

    def reset(self, **kwargs):
        obs, info = self.env.reset(**kwargs)
        return self.transform_out(obs), info

    def step(self, action):
        action = torch.Tensor(action).unsqueeze(0).unsqueeze(0).to(self.env.torch_device)   # ref :454
        obs, reward, terminated, truncated, info = self.env.step(action)
        return (self.transform_out(obs), self.transform_out(reward), self.transform_out(terminated), truncated,
                self.transform_out(info))

    def transform_out(self, x):                                     # ref gym_env.py:463-472
        if torch.is_tensor(x):
            return x.squeeze(0).squeeze(0).cpu()
        if isinstance(x, dict):
            return {k: self.transform_out(v) for k, v in x.items()}
        if isinstance(x, np.ndarray):
            return self.transform_out(torch.tensor(x)).cpu().numpy()
        return x

    def transform_in(self, x):                                      # ref gym_env.py:474-481 (unused upstream too: kept for the surface)
        if torch.is_tensor(x):
            return x.unsqueeze(0).unsqueeze(0)
        if isinstance(x, dict):
            return {k: self.transform_in(v) for k, v in x.items()}
        return x

    def render(self, *args, **kwargs):
        return self.env.render(*args, **kwargs)

    def close(self):
        self.env.close()


def make(cfg: EnvConfig, data, agents_per_env=8, road_meshes=None, traffic_lights=None, start_headings=None):
    """what gym.make('torchdriveenv-v0', args={'cfg': cfg, 'data': data}) returns in the reference (ref __init__.py:10).
    `road_meshes`, `traffic_lights`, `start_headings`: what the reference takes from torchdrivesim's map config of the location
    (`find_map_config`: road mesh ref gym_env.py:184, stop lines + light controller :181-189, lanelet directions :359) - see
    world_from_waypoint_suite"""
    return SingleAgentWrapper(WaypointSuiteEnv(cfg=cfg, data=data, agents_per_env=agents_per_env, road_meshes=road_meshes,
                                               traffic_lights=traffic_lights, start_headings=start_headings))


if gym is not None:  # pragma: no cover
    try:
        gym.register('torchdriveenv-v0', entry_point=lambda args: make(args['cfg'], args['data'], road_meshes=args.get('road_meshes'),
                                                                       traffic_lights=args.get('traffic_lights'),
                                                                       start_headings=args.get('start_headings')))
    except Exception:
        pass
