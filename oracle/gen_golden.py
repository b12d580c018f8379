#!/usr/bin/env python3
"""Generate golden vectors for the reference-OWNED part of the step path by running the reference's own
`torchdriveenv/gym_env.py` (WaypointSuiteEnv.step / get_reward / check_reach_target / is_terminated /
is_truncated / get_info, and SingleAgentWrapper's squeeze) unmodified, with

  * the third-party modules it imports (gymnasium, invertedai, torchdrivesim, cv2 ...) stubbed in sys.modules —
    they are not installed in this image and are not on the path under test, and
  * a scripted fake simulator that replays a fixed fp32 state trajectory and fixed infraction values.

Runs ONLY in the build container (it reads /root/reference).  It writes inputs + expected outputs as data to
tests/golden/reward_golden.json; neither this harness's stubs nor any reference source travel to the GPU box.

    python oracle/gen_golden.py            # regenerate tests/golden/reward_golden.json
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("TDE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "reward_golden.json")


# ------------------------------------------------------------------------------------------------
# stubs for the third-party imports at gym_env.py:12-28
# ------------------------------------------------------------------------------------------------
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Env:
        pass

    class Wrapper:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            if name == "env":
                raise AttributeError(name)
            return getattr(self.env, name)

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kw):
            return self.env.reset(**kw)

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces = mod("gymnasium.spaces", Box=Box)
    mod("gymnasium", Env=Env, Wrapper=Wrapper, spaces=spaces, register=lambda *a, **k: None)

    class _Any:
        def __init__(self, *a, **k):
            pass

    mod("invertedai")
    mod("invertedai.common", AgentState=_Any, Point=_Any, AgentAttributes=_Any, RecurrentState=_Any,
        TrafficLightState=_Any)
    mod("invertedai.api")
    mod("invertedai.error", InvalidRequestError=Exception)
    sys.modules["invertedai"].api = sys.modules["invertedai.api"]
    sys.modules["invertedai"].error = sys.modules["invertedai.error"]
    mod("torchdrivesim")
    mod("torchdrivesim.behavior")
    mod("torchdrivesim.behavior.iai", IAIWrapper=_Any, iai_conditional_initialize=None,
        unpack_attributes=None, unpack_states=None)
    mod("torchdrivesim.goals", WaypointGoal=_Any)
    mod("torchdrivesim.kinematic", KinematicBicycle=_Any)
    mod("torchdrivesim.rendering", renderer_from_config=None)
    mod("torchdrivesim.rendering.base", RendererConfig=_Any)
    mod("torchdrivesim.utils", Resolution=_Any)
    mod("torchdrivesim.lanelet2", find_lanelet_directions=None)
    mod("torchdrivesim.map", find_map_config=None, traffic_controls_from_map_config=None)
    mod("torchdrivesim.traffic_lights", current_light_state_tensor_from_controller=None)

    class CollisionMetric:
        nograd = "nograd"

    mod("torchdrivesim.simulator", TorchDriveConfig=_Any, SimulatorInterface=_Any, BirdviewRecordingWrapper=_Any,
        Simulator=_Any, HomogeneousWrapper=_Any, CollisionMetric=CollisionMetric)
    mod("omegaconf", OmegaConf=_Any)
    mod("cv2")


class ScriptedSimulator:
    """Implements exactly the SimulatorInterface methods the env calls (gym_env.py:117,123,127,142-144)."""

    def __init__(self, states, offroad, collision, tl):
        self.states = states  # [T+1][4] fp32
        self.offroad, self.collision, self.tl = offroad, collision, tl  # [T+1]
        self.t = 0

    def get_state(self):
        return torch.tensor(self.states[self.t], dtype=torch.float32).reshape(1, 1, 4)

    def step(self, action):
        assert tuple(action.shape) == (1, 1, 2) and action.dtype == torch.float32  # gym_env.py:454
        self.t += 1

    def render_egocentric(self):
        return torch.zeros(1, 1, 3, 64, 64)

    def compute_offroad(self):
        return torch.tensor([[self.offroad[self.t]]], dtype=torch.float32)

    def compute_collision(self):
        return torch.tensor([[self.collision[self.t]]], dtype=torch.float32)

    def compute_traffic_lights_violations(self):
        return torch.tensor([[self.tl[self.t]]], dtype=torch.float32)


def make_env(gym_env, cfg_over, waypoints, sim):
    cfg = gym_env.EnvConfig(**cfg_over)
    env = object.__new__(gym_env.WaypointSuiteEnv)  # __init__ needs torchdrivesim map assets (gym_env.py:312)
    gym_env.GymEnv.__init__(env, cfg=cfg, simulator=sim)
    env.config = cfg
    env.torch_device = torch.device("cpu")
    # what WaypointSuiteEnv.reset sets (gym_env.py:325-339, 352)
    env.waypoints = waypoints
    env.current_target_idx = 1
    env.current_target = waypoints[1]
    env.last_x = env.last_y = env.last_psi = None
    env.last_obs = env.last_reward = env.last_info = None
    env.reached_waypoint_num = 0
    env.environment_steps = 0
    return gym_env.SingleAgentWrapper(env), env


def f32(v):
    return float(np.float32(v))


def run_case(gym_env, name, cfg_over, waypoints, states, offroad=None, collision=None, tl=None):
    T = len(states) - 1
    offroad = offroad or [0.0] * (T + 1)
    collision = collision or [0.0] * (T + 1)
    tl = tl or [0.0] * (T + 1)
    states = [[f32(v) for v in s] for s in states]
    sim = ScriptedSimulator(states, offroad, collision, tl)
    wrapped, env = make_env(gym_env, cfg_over, waypoints, sim)
    steps = []
    for t in range(T):
        pre_target = env.current_target_idx
        obs, reward, terminated, truncated, info = wrapped.step(np.array([0.5, 0.1], dtype=np.float32))
        assert obs.shape == (3, 64, 64) and obs.dtype == np.uint8  # R1 squeeze
        assert isinstance(reward, float) and isinstance(truncated, bool)
        steps.append(dict(
            reward=reward, terminated=bool(terminated), truncated=bool(truncated),
            target_idx_before=pre_target, target_idx_after=env.current_target_idx,
            offroad=float(info["offroad"]), collision=float(info["collision"]),
            traffic_light_violation=float(info["traffic_light_violation"]),
            is_success=bool(info["is_success"]), reached_waypoint_num=int(info["reached_waypoint_num"]),
            psi_smoothness=float(info["psi_smoothness"]), psi_reward=float(info["psi_reward"]),
            dist_reward=float(info["dist_reward"]), speed_smoothness=float(info["speed_smoothness"]),
            info_tensor_shapes=[list(info[k].shape) for k in ("offroad", "collision", "traffic_light_violation")],
        ))
    cfg = gym_env.EnvConfig(**cfg_over)
    return dict(name=name,
                config=dict(waypoint_bonus=cfg.waypoint_bonus, heading_penalty=cfg.heading_penalty,
                            distance_bonus=cfg.distance_bonus, distance_cutoff=cfg.distance_cutoff,
                            max_environment_steps=cfg.max_environment_steps,
                            terminated_at_infraction=cfg.terminated_at_infraction),
                waypoints=[[float(a), float(b)] for a, b in waypoints], states=states, offroad=offroad,
                collision=collision, traffic_light_violation=tl, steps=steps)


def bicycle_traj(rng, start, n, lr=1.9):
    """generic fp32-ish trajectory (any smooth motion will do: the reward only sees consecutive states)"""
    x, y, psi, v = start
    out = [[x, y, psi, v]]
    for _ in range(n):
        a, b = rng.uniform(-1, 1), rng.uniform(-0.3, 0.3)
        v = v + a * 0.1
        x = x + v * math.cos(psi + b) * 0.1
        y = y + v * math.sin(psi + b) * 0.1
        psi = psi + v / lr * math.sin(b) * 0.1
        psi = (math.pi + psi) % (2 * math.pi) - math.pi
        out.append([x, y, psi, v])
    return out


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    gym_env = importlib.import_module("torchdriveenv.gym_env")
    shipped = dict(waypoint_bonus=100., heading_penalty=25., distance_bonus=1., distance_cutoff=0.25)  # examples/env_configs
    cases = []

    # Three-Way validation scenario waypoints (data/validation_cases.yml:8-19), read as data
    import yaml
    with open(os.path.join(REF, "torchdriveenv", "data", "validation_cases.yml")) as f:
        val = yaml.safe_load(f)
    wp0 = val["waypoint_suite"][0]

    # 1. drive straight through three waypoints 2 m apart: consecutive waypoints both within 3 m, last one -> None
    wps = [[0.0, 0.0], [10.0, 0.0], [12.0, 0.0], [14.0, 0.0]]
    st = [[5.0 + 0.6 * k, 0.0, 0.0, 6.0] for k in range(20)]
    cases.append(run_case(gym_env, "straight_consecutive_waypoints", shipped, wps, st))

    # 2. displacement exactly at / just above / just below the cutoff (strict >) for cutoff 0.25 and default 0.5
    xs = [0.0, 0.25, 0.5, 0.75 + 2 ** -20, 1.0, 1.5, 2.0 + 2 ** -18, 2.5 - 2 ** -18, 2.5]
    st = [[x, 0.0, 0.0, 1.0] for x in xs]
    cases.append(run_case(gym_env, "cutoff_edges_0p25", shipped, [[0.0, 0.0], [100.0, 0.0]], st))
    cases.append(run_case(gym_env, "cutoff_edges_default", dict(), [[0.0, 0.0], [100.0, 0.0]], st))

    # 3. reach radius edges: distance to the waypoint exactly 3, just under, just over (strict <)
    wps = [[0.0, 0.0], [10.0, 0.0], [50.0, 0.0]]
    st = [[6.0, 0.0, 0.0, 1.0], [7.0, 0.0, 0.0, 1.0], [7.0 + 2 ** -20, 0.0, 0.0, 1.0], [10.0, 3.0, 0.0, 1.0],
          [10.0, 2.9999, 0.0, 1.0], [13.0, 0.0, 0.0, 1.0], [47.0, 0.0, 0.0, 1.0], [47.5, 0.0, 0.0, 1.0],
          [48.0, 0.0, 0.0, 1.0], [49.0, 0.0, 0.0, 1.0]]
    cases.append(run_case(gym_env, "reach_radius_edges", shipped, wps, st))

    # 4. heading wrap across +-pi and large heading changes
    st = [[0.0, 0.0, 3.13, 5.0], [0.5, 0.0, 3.1415, 5.0], [1.0, 0.0, -3.1415, 5.0], [1.5, 0.0, -3.0, 5.0],
          [2.0, 0.0, 3.0, 5.2], [2.5, 0.0, 1.5, 4.0], [3.0, 0.0, 1.49, 4.0], [3.5, 0.0, -1.6, 4.1]]
    cases.append(run_case(gym_env, "heading_wrap", shipped, [[0.0, 0.0], [100.0, 0.0]], st))

    # 5. truncation: max_environment_steps = 5 and the default 200
    st = [[0.3 * k, 0.0, 0.0, 3.0] for k in range(8)]
    cases.append(run_case(gym_env, "truncate_5", dict(shipped, max_environment_steps=5), [[0.0, 0.0], [100.0, 0.0]], st))
    st = [[0.3 * k, 0.01 * k, 0.001 * k, 3.0] for k in range(203)]
    cases.append(run_case(gym_env, "truncate_200", shipped, [[0.0, 0.0], [30.0, 1.0], [100.0, 0.0]], st))

    # 6. infractions, one flag at a time, then with terminated_at_infraction=False
    st = [[0.3 * k, 0.0, 0.0, 3.0] for k in range(8)]
    off = [0, 0, 0.7, 0, 0, 0, 0, 0]
    col = [0, 0, 0, 0, 0.2, 0, 0, 0]
    tl = [0, 0, 0, 0, 0, 0, 1.0, 0]
    cases.append(run_case(gym_env, "infractions", shipped, [[0.0, 0.0], [100.0, 0.0]], st, off, col, tl))
    cases.append(run_case(gym_env, "infractions_not_terminating", dict(shipped, terminated_at_infraction=False),
                          [[0.0, 0.0], [100.0, 0.0]], st, off, col, tl))

    # 7. generic trajectories along the Three-Way waypoints, shipped reward constants and defaults
    rng = np.random.default_rng(20240229)
    for k in range(6):
        p0, p1 = wp0[0], wp0[1]
        psi0 = math.atan2(p1[1] - p0[1], p1[0] - p0[0]) + rng.normal(0, 0.1)
        traj = bicycle_traj(rng, (p0[0], p0[1], psi0, rng.uniform(0, 10)), 120)
        cases.append(run_case(gym_env, f"threeway_generic_{k}", shipped if k % 2 == 0 else dict(), wp0, traj))
    # 8. a trajectory that follows the Three-Way polyline so that every waypoint is reached
    pts = np.asarray(wp0)
    traj, pos, i = [], pts[0].copy(), 1
    while i < len(pts) and len(traj) < 260:
        d = pts[i] - pos
        L = float(np.hypot(*d))
        if L < 0.4:
            i += 1
            continue
        psi = math.atan2(d[1], d[0])
        traj.append([pos[0], pos[1], psi, 4.0])
        pos = pos + d / L * 0.4
    cases.append(run_case(gym_env, "threeway_follow_all_waypoints", dict(shipped, max_environment_steps=400), wp0, traj))

    # 9. BASELINE configs[0] input data: the Three-Way validation scenario (validation case 0), as data
    three = dict(source="torchdriveenv/data/validation_cases.yml case 0 (Three-Way, README.md:23)",
                 location=val["locations"][0], waypoints=wp0,
                 car_sequences={str(k): v for k, v in (val["car_sequence_suite"][0] or {}).items()},
                 scenario=dict(agent_states=val["scenarios"][0]["agent_states"],
                               agent_attributes=val["scenarios"][0]["agent_attributes"]),
                 # the only shipped replay car (case 1: 300 identical states, validation_cases.yml:86-1289)
                 parked_replay_example=dict(state=val["car_sequence_suite"][1][1][0],
                                            length=len(val["car_sequence_suite"][1][1])))
    with open(os.path.join(os.path.dirname(OUT), "threeway_scenario.json"), "w") as f:
        json.dump(three, f)

    meta = dict(generator="oracle/gen_golden.py",
                reference="inverted-ai/torchdriveenv torchdriveenv/gym_env.py (WaypointSuiteEnv + SingleAgentWrapper, "
                          "run unmodified over a scripted simulator; third-party imports stubbed)",
                python=sys.version.split()[0], torch=torch.__version__, numpy=np.__version__,
                n_cases=len(cases), n_steps=sum(len(c["steps"]) for c in cases))
    with open(OUT, "w") as f:
        json.dump(dict(meta=meta, cases=cases), f)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
