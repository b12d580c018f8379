"""Deterministic synthetic worlds for tests and bench.py (SURVEY §8d recipe).

  * maps: three-way / four-way junctions of 7 m-wide two-lane roads, ~200 triangles each;
  * agents: attribute statistics of the 4,355 vehicles in the reference's
    torchdriveenv/resources/background_traffic/*.json (length N(4.80,0.31) in [3.83,6.89], width N(2.07,0.11) in
    [1.67,3.03], rear_axis_offset N(1.83,0.12) in [1.46,2.62], speed |N(6.3,4.3)| in [0,24.4]);
  * ego waypoints: 5-20 per scenario, spacing 13-15 m (statistics of torchdriveenv/data/training_cases.yml:102-2980);
  * NPCs on lane centre lines, >= 10 m apart at spawn, each with a lane-following route; a few parked (replay) cars.
"""
import math

import numpy as np

from .world import assemble_world, disc_mesh, strip_mesh

LANE = 3.5
ROAD_W = 2 * LANE


def _right(t):
    return np.array([t[1], -t[0]])


def _resample(poly, spacing, first_offset=0.0):
    """points every `spacing` metres along a polyline, starting `first_offset` metres in"""
    poly = np.asarray(poly, np.float64)
    seg = np.diff(poly, axis=0)
    L = np.hypot(seg[:, 0], seg[:, 1])
    cum = np.concatenate([[0.0], np.cumsum(L)])
    out = []
    s = first_offset
    while s <= cum[-1] + 1e-9:
        i = min(len(L) - 1, int(np.searchsorted(cum, s, side="right") - 1))
        f = (s - cum[i]) / L[i] if L[i] > 0 else 0.0
        out.append(poly[i] + f * seg[i])
        s += spacing
    return np.asarray(out)


class Junction:
    """arms radiate from the origin; arm k has unit direction d_k and length len_k"""

    def __init__(self, angles_deg, lengths):
        self.d = [np.array([math.cos(math.radians(a)), math.sin(math.radians(a))]) for a in angles_deg]
        self.len = list(lengths)

    def mesh(self):
        parts = [strip_mesh([(0.0, 0.0), tuple(d * L)], ROAD_W, 5.0) for d, L in zip(self.d, self.len)]
        parts.append(disc_mesh((0.0, 0.0), 0.5 * ROAD_W + 5.5, 12))  # widened junction box
        return np.concatenate(parts, 0)

    def pose(self, arm, outbound, s, off):
        """point `s` metres out on an arm, `off` metres to the right of the travel direction; -> (position, heading)"""
        d = self.d[arm]
        t = d if outbound else -d
        return d * s + _right(t) * off, math.atan2(t[1], t[0])

    def lights(self, s_stop=11.0):
        """one light per arm with a stop line across its inbound lane; the first two arms (the main road) share a
        green, the side arm(s) get the other one, with all-red clearance phases (10 Hz steps)"""
        stop = []
        for arm, d in enumerate(self.d):
            t = -d                                         # inbound travel direction
            pos = d * s_stop + _right(t) * (0.5 * LANE)
            stop.append((float(pos[0]), float(pos[1]), math.atan2(t[1], t[0]), 1.0, LANE, arm))
        main, side = [0, 1], list(range(2, len(self.d)))
        phases = [(80, side), (15, main + side), (50, main), (15, main + side)]
        return dict(stoplines=stop, phases=phases)

    def lane(self, arm, outbound, s0, s1, step=2.0):
        """lane-centre polyline on an arm between distances s0 -> s1 from the junction centre"""
        d = self.d[arm]
        t = d if outbound else -d
        off = _right(t) * (0.5 * LANE)
        n = max(2, int(abs(s1 - s0) / step) + 1)
        return np.asarray([d * s + off for s in np.linspace(s0, s1, n)])

    def path(self, arm_in, arm_out, s_start, s_end=None):
        """inbound on arm_in from s_start to the junction, then outbound on arm_out"""
        if s_end is None:
            s_end = self.len[arm_out] - 30.0
        a = self.lane(arm_in, False, s_start, 7.0)
        b = self.lane(arm_out, True, 7.0, s_end)
        return np.concatenate([a, b], 0)


def _attrs(rng):
    L = float(np.clip(rng.normal(4.80, 0.31), 3.83, 6.89))
    W = float(np.clip(rng.normal(2.07, 0.11), 1.67, 3.03))
    lr = float(np.clip(rng.normal(1.83, 0.12), 1.46, 2.62))
    return (L, W, lr)


def _junction_scenario(J, m, rng, A, n_parked, min_gap=10.0):
    """one WaypointSuite-like scenario around junction `J` (a Junction or a TownJunction) of map `m`: the ego's waypoints
    through the junction, A - 1 NPCs on the lanes of its arms (>= 10 m apart at spawn, lane-following routes), the first
    `n_parked` of them parked at the kerb as constant replay rows.  `min_gap`: spacing of the spawns (3.4 m still separates the
    two lanes of a road: the crowded scenes of tests/test_gpu_wide.py)"""
    n_arm = len(J.len)
    a_in = int(rng.integers(n_arm))
    a_out = int((a_in + 1 + rng.integers(n_arm - 1)) % n_arm)
    n_wp = int(rng.integers(5, 21))
    spacing = float(rng.uniform(13.0, 15.0))
    s_start = float(min(J.len[a_in] - 5.0, rng.uniform(60.0, 100.0)))
    ego_path = J.path(a_in, a_out, s_start)
    wps = _resample(ego_path, spacing)[:n_wp]
    if len(wps) < 2:
        wps = _resample(ego_path, spacing)[:2]
    heading = math.atan2(wps[1, 1] - wps[0, 1], wps[1, 0] - wps[0, 0])
    # spawn slots: every 12 m on every lane
    slots = []
    for arm in range(n_arm):
        for outbound in (False, True):
            s = 14.0
            while s < J.len[arm] - 45.0:
                slots.append((arm, outbound, s))
                s += 12.0
    order = rng.permutation(len(slots))
    agents = []
    used = []
    ego_seg = np.stack([wps[0], wps[1]])
    for idx in order:
        if len(agents) >= A - 1:
            break
        arm, outbound, s = slots[idx]
        pos, psi = J.pose(arm, outbound, s, 0.5 * LANE)
        # keep clear of the ego's start segment and of other spawns
        ap = pos - ego_seg[0]
        ab = ego_seg[1] - ego_seg[0]
        tt = np.clip((ap @ ab) / (ab @ ab), 0, 1)
        if np.hypot(*(ap - tt * ab)) < 12.0:
            continue
        if any(np.hypot(*(pos - u)) < min_gap for u in used):
            continue
        used.append(pos)
        parked = len(agents) < n_parked
        if parked:
            # a parked car hugging the road edge, recorded as a constant replay row (env_utils.py:86-91)
            pos, psi = J.pose(arm, outbound, s, LANE - 1.1)
            st = (float(pos[0]), float(pos[1]), psi, 0.0)
            agents.append(dict(state=st, attr=(4.6, 1.9, 1.8), vdes=0.0, route=None, replay=[st] * 220))
            continue
        speed = float(np.clip(abs(rng.normal(6.3, 4.3)), 0.0, 24.4))
        speed = min(speed, 9.0)
        if outbound:
            route = J.lane(arm, True, s + 8.0, J.len[arm] - 30.0)
        else:
            out_arm = int((arm + 1 + rng.integers(n_arm - 1)) % n_arm)
            route = J.path(arm, out_arm, max(8.0, s - 8.0))
        route = _resample(route, 6.0)[:32]
        agents.append(dict(state=(float(pos[0]), float(pos[1]), psi, speed), attr=_attrs(rng), vdes=max(speed, 3.0),
                           route=route, replay=None))
    return dict(map=m, waypoints=wps, start_heading=heading, agents=agents, ego_attr=_attrs(rng))


def synthetic_world(n_scn=64, A=16, seed=0, n_maps=4, n_parked=1, cell=0.25, threshold=0.5, lights=True, near_range=None):
    """World with `n_scn` scenarios of A-1 NPCs each on `n_maps` junction maps.  Deterministic in `seed`.
    near_range: how far beyond the threshold the grid index carries near lists (world.NEAR_RANGE by default; 0: none)."""
    rng = np.random.default_rng(seed)
    juncs = []
    for m in range(n_maps):
        if m % 2 == 0:  # three-way
            side = rng.uniform(70.0, 110.0)
            ang = [0.0, 180.0, side]
        else:  # four-way
            ang = [0.0, 180.0, rng.uniform(75.0, 105.0), rng.uniform(255.0, 285.0)]
        juncs.append(Junction(ang, [rng.uniform(110.0, 140.0) for _ in ang]))
    meshes = [j.mesh() for j in juncs]
    scenarios = [_junction_scenario(juncs[si % n_maps], si % n_maps, rng, A, n_parked) for si in range(n_scn)]
    from .world import NEAR_RANGE
    return assemble_world(meshes, scenarios, A, threshold=threshold, cell=cell,
                          lights=[j.lights() for j in juncs] if lights else None,
                          near_range=NEAR_RANGE if near_range is None else near_range)


# ------------------------------------------------------------------------------------------------
# a town: the operating size of the reference's maps (a CARLA town's drivable mesh, 1e4 - 1e5 triangles: gym_env.py:184, 260)
# ------------------------------------------------------------------------------------------------
class Town:
    """n x n streets `spacing` metres apart (n^2 four-way junctions) under a smooth warp, so that no street is straight or
    axis-aligned, plus one diagonal avenue; every street is a ribbon of two lane strips cut every `ds` metres."""

    def __init__(self, n=10, spacing=100.0, ext=45.0, amp=6.0, wavelength=430.0, ds=1.5):
        self.n, self.spacing, self.ext, self.amp, self.k, self.ds = n, spacing, ext, amp, 2.0 * math.pi / wavelength, ds
        self.span = (n - 1) * spacing

    def F(self, u, v):
        """street coordinates (metres along the two street families) -> world"""
        u, v = np.asarray(u, np.float64), np.asarray(v, np.float64)
        return np.stack([u + self.amp * np.sin(self.k * v), v + self.amp * np.sin(self.k * u + 1.0)], -1)

    def ribbon(self, pts, width=ROAD_W, strips=2):
        """triangulated ribbon around a polyline: vertices offset along the (averaged) normals, so bends leave no gaps"""
        pts = np.asarray(pts, np.float64)
        t = np.gradient(pts, axis=0)
        t /= np.hypot(t[:, 0], t[:, 1])[:, None]
        nrm = np.stack([-t[:, 1], t[:, 0]], -1)
        lat = np.linspace(-0.5 * width, 0.5 * width, strips + 1)
        G = pts[:, None, :] + lat[None, :, None] * nrm[:, None, :]            # [N][strips + 1][2]
        a, b, c, d = G[:-1, :-1], G[1:, :-1], G[1:, 1:], G[:-1, 1:]
        return np.concatenate([np.stack([a, b, c], -2).reshape(-1, 3, 2), np.stack([a, c, d], -2).reshape(-1, 3, 2)], 0)

    def mesh(self):
        s = np.arange(-self.ext, self.span + self.ext + 1e-9, self.ds)
        parts = []
        for i in range(self.n):
            c = np.full_like(s, i * self.spacing)
            parts.append(self.ribbon(self.F(s, c)))          # a street along u
            parts.append(self.ribbon(self.F(c, s)))          # a street along v
        parts.append(self.ribbon(self.F(s, s)))              # the diagonal avenue
        for i in range(self.n):
            for j in range(self.n):
                parts.append(disc_mesh(self.F(i * self.spacing, j * self.spacing), 0.5 * ROAD_W + 5.5, 16))
        return np.concatenate(parts, 0)


class TownJunction:
    """the four arms of junction (i, j) of a Town, with the interface of Junction (len, pose, lane, path)"""
    DIRS = ((1.0, 0.0), (-1.0, 0.0), (0.0, 1.0), (0.0, -1.0))

    def __init__(self, town, i, j, max_len=240.0):
        self.town, self.u0, self.v0 = town, i * town.spacing, j * town.spacing
        lo, hi = -town.ext, town.span + town.ext
        self.len = [min(max_len, (hi - self.u0) if du > 0 else (self.u0 - lo) if du < 0 else
                        (hi - self.v0) if dv > 0 else (self.v0 - lo)) for du, dv in self.DIRS]

    def _frame(self, arm, s):
        """centre line point(s) `s` metres out on an arm and the unit tangent pointing outwards"""
        du, dv = self.DIRS[arm]
        s = np.asarray(s, np.float64)
        p = self.town.F(self.u0 + s * du, self.v0 + s * dv)
        e = 0.05
        t = self.town.F(self.u0 + (s + e) * du, self.v0 + (s + e) * dv) - self.town.F(self.u0 + (s - e) * du, self.v0 + (s - e) * dv)
        return p, t / np.hypot(t[..., 0], t[..., 1])[..., None]

    def pose(self, arm, outbound, s, off):
        p, d = self._frame(arm, s)
        t = d if outbound else -d
        return p + _right(t) * off, math.atan2(t[1], t[0])

    def lane(self, arm, outbound, s0, s1, step=2.0):
        n = max(2, int(abs(s1 - s0) / step) + 1)
        p, d = self._frame(arm, np.linspace(s0, s1, n))
        t = d if outbound else -d
        return p + np.stack([t[:, 1], -t[:, 0]], -1) * (0.5 * LANE)

    def path(self, arm_in, arm_out, s_start, s_end=None):
        if s_end is None:
            s_end = self.len[arm_out] - 30.0
        return np.concatenate([self.lane(arm_in, False, s_start, 7.0), self.lane(arm_out, True, 7.0, s_end)], 0)

    def stoplines(self, light_main, light_side, s_stop=11.0):
        """a stop line across the inbound lane of every arm; arms 0 / 1 (the street along u) obey `light_main`, 2 / 3 `light_side`"""
        out = []
        for arm in range(4):
            pos, psi = self.pose(arm, False, s_stop, 0.5 * LANE)
            out.append((float(pos[0]), float(pos[1]), psi, 1.0, LANE, light_main if arm < 2 else light_side))
        return out


def synthetic_town(n_scn=256, A=16, seed=0, n_streets=10, spacing=100.0, ext=45.0, ds=1.5, n_parked=1, cell=0.25,
                   threshold=0.5, min_gap=10.0, n_signals=0, signal_reach=1):
    """ONE map of town size - n_streets^2 junctions on ~((n_streets - 1) * spacing + 2 * ext)^2 metres, >= 5e4 triangles at the
    defaults (1 km x 1 km, 100 junctions) - with `n_scn` scenarios spread over its interior junctions, each built like a
    synthetic_world scenario (A - 1 NPCs on the arms around its junction).  `n_signals` of the scenario junctions (the first ones in
    scenario order; up to all of them) are signalised: a stop line per arm, main street / side street lights, the same cycle
    everywhere.  The scenarios at a junction see the lights of that junction and of its four neighbours - a LIGHT GROUP
    (assemble_world: a map descriptor that shares the town's grid and carries <= 5 x 4 stop lines, <= 10 lights), since the
    kernels walk every stop line of a scenario's descriptor and a light is a bit of a 32-bit mask (`signal_reach=0`: the junction's own
    lights only - 4 stop lines, the cost of the junction maps).  Deterministic in `seed`."""
    rng = np.random.default_rng(seed)
    town = Town(n_streets, spacing, ext, ds=ds)
    inner = [(i, j) for i in range(1, n_streets - 1) for j in range(1, n_streets - 1)]
    order = rng.permutation(len(inner))
    assert 0 <= n_signals <= len(inner), f"the town has {len(inner)} interior junctions"
    signalised = {inner[order[q]] for q in range(n_signals)}
    phases = lambda n: [(80, [2 * k + 1 for k in range(n)]), (15, list(range(2 * n))), (50, [2 * k for k in range(n)]), (15, list(range(2 * n)))]
    groups, group_of = [], {}
    for q in range(min(n_scn, len(inner)) if n_signals else 0):
        i, j = inner[order[q]]
        around = ((i, j), (i + 1, j), (i - 1, j), (i, j + 1), (i, j - 1))[:1 + 4 * signal_reach]
        members = [ij for ij in around if ij in signalised]                                   # own junction first
        if members:
            stop = []
            for k, (mi, mj) in enumerate(members):
                stop += TownJunction(town, mi, mj).stoplines(2 * k, 2 * k + 1)
            group_of[q] = len(groups)
            groups.append(dict(map=0, stoplines=stop, phases=phases(len(members))))
    scenarios = []
    for si in range(n_scn):
        q = si % len(inner)
        i, j = inner[order[q]]
        sc = _junction_scenario(TownJunction(town, i, j), 0, rng, A, n_parked, min_gap)
        if q in group_of:
            sc["lights"] = group_of[q]
        scenarios.append(sc)
    return assemble_world([town.mesh()], scenarios, A, threshold=threshold, cell=cell, light_groups=groups)
