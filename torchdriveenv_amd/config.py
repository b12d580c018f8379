"""Config / data carriers with the reference's field names and defaults (ref gym_env.py:34-68).

`EnvConfig.simulator` held a torchdrivesim `TorchDriveConfig` in the reference (gym_env.py:46-49); here it is this
package's own `SimulatorConfig` with the fields the step path uses."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional


@dataclass
class RendererConfig:
    left_handed_coordinates: bool = True      # ref gym_env.py:46
    highlight_ego_vehicle: bool = True        # ref gym_env.py:47
    res: int = 64                             # observation space (3,64,64), ref gym_env.py:95
    fov: float = 35.0                         # metres across the image (torchdrivesim default)


@dataclass
class SimulatorConfig:
    renderer: RendererConfig = field(default_factory=RendererConfig)
    collision_metric: str = "nograd"          # CollisionMetric.nograd, ref gym_env.py:48
    left_handed_coordinates: bool = True      # ref gym_env.py:49
    offroad_threshold: float = 0.5            # TorchDriveConfig default (not overridden by the env)
    # how the threshold meets the point-to-mesh distance (torchdrivesim internals are not in the reference repository):
    # False: distance > threshold; True: SQUARED distance > threshold (pytorch3d-style point_mesh distance)
    offroad_threshold_squared: bool = False
    # heuristic NPC controller (stands where the IAI call was, ref gym_env.py:285-294)
    npc_k_steer: float = 1.2
    npc_k_speed: float = 3.0
    npc_gap_s0: float = 3.0
    npc_cone_k: float = 0.5
    npc_cone_range: float = 25.0
    npc_lane_half: float = 1.75
    npc_reach: float = 3.0
    npc_max_accel: float = 3.0
    npc_max_steer: float = 0.3
    # the NPC controller acts from the FIRST step of an episode, as the reference's NPCs do (ref gym_env.py:285-294: IAIWrapper
    # predicts from step one) -> TDE_F_NPC_FIRST_STEP.  False: the NPCs coast through step one (the rule of rounds 4 / 5, slightly
    # cheaper: a re-spawn then leaves nothing to recompute)
    npc_first_step: bool = True


@dataclass
class EnvConfig:
    ego_only: bool = False
    max_environment_steps: int = 200
    frame_stack: int = 3
    waypoint_bonus: float = 100.
    heading_penalty: float = 25.
    distance_bonus: float = 1.
    distance_cutoff: float = 0.5
    use_background_traffic: bool = True
    terminated_at_infraction: bool = True
    seed: Optional[int] = None
    simulator: SimulatorConfig = field(default_factory=SimulatorConfig)
    render_mode: Optional[str] = "rgb_array"
    video_filename: Optional[str] = "rendered_video.mp4"
    video_res: Optional[int] = 1024
    video_fov: Optional[float] = 500
    device: Optional[str] = None


@dataclass
class Scenario:
    agent_states: List[List[float]] = None
    agent_attributes: List[List[float]] = None
    recurrent_states: List[List[float]] = None


@dataclass
class WaypointSuite:
    locations: List[str] = None
    waypoint_suite: List[List[List[float]]] = None
    car_sequence_suite: List[Optional[Dict[int, List[List[float]]]]] = None
    scenarios: List[Optional[Scenario]] = None


def validate(cfg: EnvConfig):
    """Reject what this path does not implement instead of silently ignoring it (the reference accepts these through
    TorchDriveConfig, ref gym_env.py:46-49, 75-80, 295-297)."""
    sim = cfg.simulator
    if str(getattr(sim.collision_metric, "name", sim.collision_metric)) != "nograd":
        raise NotImplementedError(f"collision_metric={sim.collision_metric!r}: only CollisionMetric.nograd (the reference's "
                                  "setting, gym_env.py:48) is implemented: strict OBB overlap")
    if cfg.render_mode == "video":
        raise NotImplementedError("render_mode='video' (BirdviewRecordingWrapper, gym_env.py:295-297) is not part of the "
                                  "step path; use render_mode='rgb_array' and record the frames render() returns")
    if cfg.render_mode not in (None, "rgb_array"):
        raise NotImplementedError                                   # ref gym_env.py:79-80
    if sim.left_handed_coordinates != sim.renderer.left_handed_coordinates:
        raise NotImplementedError("simulator.left_handed_coordinates and renderer.left_handed_coordinates differ: the "
                                  "reference sets both (gym_env.py:46-49); here the flag mirrors the birdview's lateral axis "
                                  "(the kinematic model is built with its default handedness, gym_env.py:245)")
    if sim.npc_cone_k < 0:
        raise ValueError("simulator.npc_cone_k must be >= 0")
    if sim.offroad_threshold <= 0:
        raise ValueError("simulator.offroad_threshold must be > 0")
    if not sim.npc_max_steer >= 0:
        raise ValueError("simulator.npc_max_steer must be >= 0")
    if not 1e-3 <= sim.npc_max_accel <= 1e3:
        raise ValueError("simulator.npc_max_accel must be in [1e-3, 1e3] m/s^2 (tde_env_* reject anything else)")


def render_flags(cfg: EnvConfig):
    """tde_render.flags of this config (RendererConfig, ref gym_env.py:46-47)"""
    from . import _abi

    r = cfg.simulator.renderer
    return ((_abi.RENDER_LEFT_HANDED if r.left_handed_coordinates else 0) |
            (0 if r.highlight_ego_vehicle else _abi.RENDER_PLAIN_EGO))


def to_tde_config(cfg: EnvConfig, seed: int, flags: int):
    """EnvConfig -> the C-ABI's tde_config"""
    from . import _abi

    validate(cfg)
    sim = cfg.simulator
    return _abi.default_config(
        waypoint_bonus=float(cfg.waypoint_bonus), heading_penalty=float(cfg.heading_penalty),
        distance_bonus=float(cfg.distance_bonus), distance_cutoff=float(cfg.distance_cutoff), seed=int(seed),
        max_steps=int(cfg.max_environment_steps), terminated_at_infraction=int(bool(cfg.terminated_at_infraction)),
        offroad_threshold=float(sim.offroad_threshold), npc_k_steer=sim.npc_k_steer, npc_k_speed=sim.npc_k_speed,
        npc_gap_s0=sim.npc_gap_s0, npc_cone_k=sim.npc_cone_k, npc_cone_range=sim.npc_cone_range,
        npc_lane_half=sim.npc_lane_half, npc_reach=sim.npc_reach, npc_max_accel=sim.npc_max_accel,
        npc_max_steer=sim.npc_max_steer, flags=flags,
        offroad_threshold_squared=int(bool(sim.offroad_threshold_squared)))
