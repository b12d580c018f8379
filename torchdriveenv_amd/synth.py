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


def synthetic_world(n_scn=64, A=16, seed=0, n_maps=4, n_parked=1, cell=0.25, threshold=0.5, lights=True):
    """World with `n_scn` scenarios of A-1 NPCs each on `n_maps` junction maps.  Deterministic in `seed`."""
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
    scenarios = []
    for si in range(n_scn):
        m = si % n_maps
        J = juncs[m]
        n_arm = len(J.d)
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
            d = J.d[arm]
            t = d if outbound else -d
            pos = d * s + _right(t) * (0.5 * LANE)
            # keep clear of the ego's start segment and of other spawns
            ap = pos - ego_seg[0]
            ab = ego_seg[1] - ego_seg[0]
            tt = np.clip((ap @ ab) / (ab @ ab), 0, 1)
            if np.hypot(*(ap - tt * ab)) < 12.0:
                continue
            if any(np.hypot(*(pos - u)) < 10.0 for u in used):
                continue
            used.append(pos)
            psi = math.atan2(t[1], t[0])
            parked = len(agents) < n_parked
            if parked:
                # a parked car hugging the road edge, recorded as a constant replay row (env_utils.py:86-91)
                pos = d * s + _right(t) * (LANE - 1.1)
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
        scenarios.append(dict(map=m, waypoints=wps, start_heading=heading, agents=agents, ego_attr=_attrs(rng)))
    return assemble_world(meshes, scenarios, A, threshold=threshold, cell=cell,
                          lights=[j.lights() for j in juncs] if lights else None)
