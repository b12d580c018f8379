"""The host-side mirror of the reference surface on a real GPU: B=1 Three-Way scenario through the reference-shaped API
(BASELINE configs[0]) checked against the oracle stepping the same world, and the batched / VecEnv-shaped API."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from torchdriveenv_amd import _abi  # noqa: E402
from torchdriveenv_amd.config import EnvConfig, Scenario, WaypointSuite  # noqa: E402
from torchdriveenv_amd.env import BatchedWaypointEnv, SingleAgentWrapper, WaypointSuiteEnv, make  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def threeway_suite():
    t = json.load(open(os.path.join(ROOT, "tests", "golden", "threeway_scenario.json")))
    wp, sc, parked = t["waypoints"], t["scenario"], t["parked_replay_example"]
    extra = [[wp[2][0] + 4.0, wp[2][1] + 3.0, 0.0, 0.0], [wp[4][0] - 3.0, wp[4][1] + 4.0, 0.0, 0.0]]
    # all four NPCs replayed: the two scenario agents drive straight at their scenario speed, two parked cars
    seqs = {}
    for slot, s in enumerate(sc["agent_states"], start=1):
        x, y, psi, v = s
        seqs[slot] = [[x + np.cos(psi) * v * 0.1 * k, y + np.sin(psi) * v * 0.1 * k, psi, v] for k in range(120)]
    seqs[3] = [extra[0]] * parked["length"]
    seqs[4] = [extra[1]] * parked["length"]
    return WaypointSuite(locations=[t["location"]], waypoint_suite=[wp], car_sequence_suite=[seqs],
                         scenarios=[Scenario(agent_states=sc["agent_states"] + extra,
                                             agent_attributes=sc["agent_attributes"] + [[5.0, 2.0, 2.0]] * 2)])


def test_config1_threeway_b1_reference_api_vs_oracle():
    cfg = EnvConfig(seed=3, distance_cutoff=0.25)
    env = make(cfg, threeway_suite(), agents_per_env=8)
    assert isinstance(env, SingleAgentWrapper) and isinstance(env.env, WaypointSuiteEnv)
    obs, info = env.reset()
    assert obs.shape == (3, 64, 64) and obs.dtype == np.uint8 and info == {}
    inner = env.env._env
    # the oracle steps the same world from the same reset
    hs = EnvState(1, 8)
    ocfg = inner.tde_cfg
    oracle.env_reset(ocfg, inner.world, hs)
    assert np.array_equal(hs["x"].view(np.uint32), inner.state["x"].cpu().numpy().view(np.uint32))
    total, n_term = 0.0, 0
    for t in range(200):
        action = np.array([0.6, 0.0 if t < 25 else -0.02], dtype=np.float32)
        obs, reward, terminated, truncated, info = env.step(action)
        hs["action"][...] = action
        oracle.env_step(ocfg, inner.world, hs)
        # reference shapes / types (gym_env.py:453-472)
        assert obs.shape == (3, 64, 64) and obs.dtype == np.uint8
        assert isinstance(reward, float) and isinstance(terminated, bool) and isinstance(truncated, bool)
        for k in ("offroad", "collision", "traffic_light_violation"):
            assert torch.is_tensor(info[k]) and info[k].dim() == 0 and info[k].device.type == "cpu"
        for k in ("is_success", "reached_waypoint_num", "psi_smoothness", "psi_reward", "dist_reward",
                  "speed_smoothness"):
            assert k in info
        assert np.float32(reward) == hs["reward"][0]
        assert terminated == bool(hs["terminated"][0]) and truncated == bool(hs["truncated"][0])
        # the reference's info semantics (gym_env.py:427-428): the MAGNITUDES of compute_offroad() / compute_collision()
        mag = oracle.ego_infractions(ocfg, inner.world, hs)[0]
        assert np.float32(info["offroad"]).view(np.uint32) == mag[0].view(np.uint32)
        assert np.float32(info["collision"]).view(np.uint32) == mag[1].view(np.uint32)
        assert (float(info["collision"]) > 0) == bool(hs["collided"][0]) and (float(info["offroad"]) > 0) == bool(hs["offroad"][0])
        assert info["reached_waypoint_num"] == hs["info_reached"][0]
        assert np.array_equal(inner.state["x"].cpu().numpy().view(np.uint32), hs["x"].view(np.uint32))
        assert np.array_equal(obs, oracle.render_ego(ocfg, inner.world, hs, flags=inner._rflags)[0])   # (left-handed raster: the reference default)
        total += reward
        if terminated or truncated:
            n_term += 1
            break
    assert env.env.environment_steps == int(hs["steps"][0]) and n_term == 1
    assert env.env.reached_waypoint_num >= 1          # it drove through at least the first waypoint
    img = env.render()
    assert img.shape == (64, 64, 3) and img.dtype == np.uint8


@pytest.mark.parametrize("terminate", [True, False])
def test_make_step_info_carries_the_reference_magnitudes(terminate):
    """make(...).step(): info["offroad"] / info["collision"] are the magnitudes the reference reports (gym_env.py:419-437: the
    values of simulator.compute_offroad() / compute_collision() for the exposed agent, which Monitor logs,
    examples/rl_training.py:128) - here the oracle's, bit for bit, over episodes whose ego leaves the road and runs into the
    parked cars; with terminated_at_infraction the episode ends at the first non-zero one"""
    cfg = EnvConfig(seed=5, distance_cutoff=0.25, terminated_at_infraction=terminate, max_environment_steps=120)
    env = make(cfg, threeway_suite(), agents_per_env=8)
    inner = env.env._env
    assert inner.info_magnitudes
    ocfg = inner.tde_cfg
    n_off = n_col = n_eps = 0
    for ep in range(6):
        env.reset()
        hs = EnvState(1, 8)
        hs.load({k: v for k, v in inner.state.host().items() if k in hs.arrays and hs.arrays[k] is not None})
        steer = (-0.3, 0.3, 0.12, -0.12, 0.05, 0.0)[ep]
        for t in range(120):
            action = np.array([0.8, steer if t > 5 else 0.0], dtype=np.float32)
            obs, reward, terminated, truncated, info = env.step(action)
            hs["action"][...] = action
            oracle.env_step(ocfg, inner.world, hs)
            mag = oracle.ego_infractions(ocfg, inner.world, hs)[0]
            assert info["offroad"].dtype == torch.float32 and info["offroad"].dim() == 0
            assert np.float32(info["offroad"]).view(np.uint32) == mag[0].view(np.uint32), (ep, t)
            assert np.float32(info["collision"]).view(np.uint32) == mag[1].view(np.uint32), (ep, t)
            assert np.float32(reward) == hs["reward"][0] and terminated == bool(hs["terminated"][0])
            n_off += mag[0] > 0
            n_col += mag[1] > 0
            if terminate and (mag[0] > 0 or mag[1] > 0):
                assert terminated
            if terminated or truncated:
                break
        n_eps += 1
    assert n_off > 0 and n_eps == 6
    if not terminate:
        assert n_off > 20                      # the ego kept driving off the road: distances of metres, not 0 / 1
    env.close()


def test_road_mesh_hook_runs_the_suite_on_the_callers_mesh(tmp_path):
    """world_from_waypoint_suite(road_meshes=...): a caller who has the location's road mesh (the reference:
    find_map_config(location).road_mesh, gym_env.py:312, 184, 260) gets THAT map instead of a synthetic corridor - here the synthetic
    town's mesh saved to .npy, five scenarios on it sharing one map, HIP == oracle over an episode each"""
    from torchdriveenv_amd.env import mesh_from_verts_faces, world_from_waypoint_suite
    from torchdriveenv_amd.synth import Town

    town = Town(n=3, spacing=100.0, ext=30.0)
    tri = town.mesh()
    path = tmp_path / "Town_synth.npy"
    np.save(path, tri.astype(np.float32))
    # five waypoint routes along the town's first street, shifted along it
    suites, locs = [], []
    for k in range(5):
        x0 = 20.0 + 25.0 * k
        pts = [town.F(x0 + 14.0 * i, 0.0) for i in range(8)]
        suites.append([[float(p[0]), float(p[1])] for p in pts])
        locs.append("Town_synth")
    data = WaypointSuite(locations=locs, waypoint_suite=suites, car_sequence_suite=[None] * 5, scenarios=[None] * 5)
    # verts / faces form of the same mesh through the converter
    verts = tri.reshape(-1, 2)
    faces = np.arange(len(verts)).reshape(-1, 3)
    assert np.array_equal(mesh_from_verts_faces(verts, faces), tri)
    for meshes in ({"Town_synth": str(path)}, lambda loc: tri if loc == "Town_synth" else None):
        world = world_from_waypoint_suite(data, agents_per_env=8, road_meshes=meshes)
        assert world.ints["n_maps"] == 1 and (world.arrays["scn"]["map"] == 0).all()        # one map for the location
        assert world.arrays["maps"]["n_tri"][0] == len(tri)
    cfg = EnvConfig(seed=2, distance_cutoff=0.25)
    env = BatchedWaypointEnv(cfg, data, num_envs=40, agents_per_env=8, obs_mode="state", road_meshes={"Town_synth": str(path)},
                             info_magnitudes=True)
    assert env.world.ints["n_maps"] == 1
    hs = EnvState(40, 8)
    ocfg = _abi.TdeConfig.from_buffer_copy(env.tde_cfg)
    oracle.env_reset(ocfg, env.world, hs)
    env.reset()
    rng = np.random.default_rng(0)
    n_off = 0
    for t in range(150):
        act = np.stack([rng.uniform(-0.2, 1, 40), rng.uniform(-0.3, 0.3, 40)], -1).astype(np.float32)
        _, rew, term, trunc, info = env.step(torch.from_numpy(act).cuda())
        hs["action"][...] = act
        oracle.env_step(ocfg, env.world, hs)
        assert np.array_equal(rew.cpu().numpy().view(np.uint32), hs["reward"].view(np.uint32)), t
        n_off += int((info["offroad"] > 0).sum())
    for k in ("x", "y", "psi", "v", "steps", "episode", "scn", "offroad", "collided"):
        assert np.array_equal(env.state[k].cpu().numpy(), hs[k]), k
    assert n_off > 5                           # egos did leave the town's streets: the mesh is the caller's, not a corridor


def test_batched_env_device_api_and_vecenv_adapter(small_world):
    cfg = EnvConfig(seed=11, distance_cutoff=0.25, frame_stack=3)
    B = 64
    env = BatchedWaypointEnv(cfg, small_world, num_envs=B, device="cuda:0", frame_stack=3)
    obs = env.reset()
    assert obs.shape == (B, 9, 64, 64) and obs.dtype == torch.uint8 and obs.is_cuda
    g = torch.Generator().manual_seed(0)
    ndone = 0
    for t in range(30):
        a = torch.stack([torch.rand(B, generator=g) * 2 - 1, torch.rand(B, generator=g) * 0.6 - 0.3], -1)
        obs, rew, term, trunc, info = env.step(a)
        assert rew.shape == (B,) and term.dtype == torch.bool and obs.is_cuda
        assert set(info) >= {"offroad", "collision", "traffic_light_violation", "is_success", "reached_waypoint_num",
                             "psi_smoothness", "speed_smoothness", "psi_reward", "dist_reward"}
        ndone += int((term | trunc).sum())
    # newest frame is last in the stack; the previous call's newest frame moved one slot down
    prev_last = obs[:, 6:9].clone()
    obs2, *_ = env.step(torch.zeros(B, 2))
    keep = ~(env.state["steps"] == 0)          # envs that did not just reset
    assert torch.equal(obs2[keep][:, 3:6], prev_last[keep])
    # SB3-shaped numpy path with terminal_observation
    env2 = BatchedWaypointEnv(cfg, small_world, num_envs=B, device="cuda:0", frame_stack=1)
    o = env2.vec_reset()
    assert o.shape == (B, 3, 64, 64) and o.dtype == np.uint8
    seen_terminal = False
    rng = np.random.default_rng(1)
    for t in range(60):
        o, r, d, infos = env2.vec_step(np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1))
        assert o.shape == (B, 3, 64, 64) and r.shape == (B,) and d.dtype == bool and len(infos) == B
        for i in np.nonzero(d)[0]:
            assert infos[i]["terminal_observation"].shape == (3, 64, 64)
            assert int(env2.state["steps"][i]) == 0        # re-spawned
            seen_terminal = True
    assert seen_terminal
    sd = env2.state_dict()
    env2.vec_step(np.zeros((B, 2), np.float32))
    env2.load_state_dict(sd)
    assert torch.equal(env2.state["x"], sd["x"])


def test_rollout_api_and_state_obs(small_world):
    cfg = EnvConfig(seed=2)
    env = BatchedWaypointEnv(cfg, small_world, num_envs=128, device="cuda:0", obs_mode="state", with_info=False)
    o = env.reset()
    assert o.shape == (128, 8) and o.dtype == torch.float32
    acts = torch.zeros(50, 128, 2, device="cuda:0")
    acts[..., 0] = 0.3
    r, d = env.rollout(acts)
    assert r.shape == (50, 128) and d.shape == (50, 128) and torch.isfinite(r).all()


def test_state_obs_operator_matches_numpy(small_world):
    """tde_state_obs: x, y, psi, v, target offset in the ego frame, target flag, steps — against plain numpy"""
    cfg = EnvConfig(seed=4)
    env = BatchedWaypointEnv(cfg, small_world, num_envs=96, device="cuda:0", obs_mode="state", with_info=True)
    env.reset()
    acts = torch.zeros(96, 2, device="cuda:0")
    acts[:, 0] = 0.5
    for _ in range(30):
        o, r, term, trunc, info = env.step(acts)
    st = {k: v.cpu().numpy() for k, v in env.state.arrays.items() if v is not None}
    A = env.A
    x, y, psi, v = (st[k][::A] for k in ("x", "y", "psi", "v"))
    w = small_world
    n_wp = w.arrays["scn"]["wp_n"][st["scn"]]
    wp = w.arrays["wp_xy"].reshape(-1, w.ints["NW"], 2)
    has = st["target_idx"] < n_wp
    ti = np.minimum(st["target_idx"], n_wp - 1)
    tgt = wp[st["scn"], ti].astype(np.float32)
    dx, dy = tgt[:, 0] - x, tgt[:, 1] - y
    c, s = np.cos(psi), np.sin(psi)
    exp = np.stack([x, y, psi, v, np.where(has, dx * c + dy * s, 0), np.where(has, dy * c - dx * s, 0),
                    has.astype(np.float32), st["steps"].astype(np.float32)], -1)
    got = o.cpu().numpy()
    assert got.shape == (96, 8)
    np.testing.assert_array_equal(got[:, [0, 1, 2, 3, 6, 7]], exp[:, [0, 1, 2, 3, 6, 7]])
    np.testing.assert_allclose(got[:, 4:6], exp[:, 4:6], rtol=0, atol=2e-4)
    # flags come back as bool views of the uint8 state, info entries are formed on access
    assert term.dtype == torch.bool and trunc.dtype == torch.bool
    assert set(info.keys()) >= {"offroad", "collision", "traffic_light_violation", "is_success", "psi_smoothness"}
    assert info["offroad"].shape == (96,) and info["offroad"].dtype == torch.float32
    assert "psi_reward" in info and info.get("nope") is None


def test_terminal_info_survives_in_place_respawn(small_world):
    """get_info at a terminal step reports the infraction that ended the episode (ref gym_env.py:426-429) even though
    the env was re-spawned inside the same kernel (tde_state.done_bits)"""
    cfg = EnvConfig(seed=5)
    B = 256
    env = BatchedWaypointEnv(cfg, small_world, num_envs=B, device="cuda:0", obs_mode="state", with_info=True)
    env.reset()
    g = torch.Generator().manual_seed(1)
    seen = 0
    for t in range(120):
        a = torch.stack([torch.rand(B, generator=g) * 2 - 1, torch.rand(B, generator=g) * 0.6 - 0.3], -1)
        obs, rew, term, trunc, info = env.step(a)
        if term.any():
            # (info["collision"] is the SUM OF IoUs, as the reference reports it: an overlap thinner than fp32 resolves clips to an
            #  area of exactly 0 although the mask's strict-SAT predicate - which decides `terminated` - holds; the number of
            #  overlapping agents, magnitudes[:, 2], is the mask's own predicate)
            overlaps = env.state["magnitudes"][:, 2]
            infr = (info["offroad"] > 0) | (overlaps > 0) | (info["traffic_light_violation"] > 0)
            assert torch.equal(infr, term), t                       # terminated <=> an infraction is reported
            assert ((info["collision"] > 0) <= (overlaps > 0)).all()
            assert (env.state["steps"][term] == 0).all()            # ... although those envs already re-spawned
            seen += int(term.sum())
    assert seen > 10


@pytest.mark.parametrize("flags", [_abi.F_ALL, _abi.F_ALL & ~_abi.F_AUTORESET, _abi.F_NPC | _abi.F_OFFROAD])
def test_step_writes_the_same_observation_as_state_obs(small_world, flags):
    """tde_state.obs: the compact observation written by tde_env_step itself equals a tde_state_obs call on the state
    after the step, bit for bit - through re-spawns, finished episodes without re-spawn, and with the reward path off"""
    from torchdriveenv_amd import ops
    from torchdriveenv_amd.state import EnvState

    cfg = _abi.default_config(seed=9, flags=flags, max_steps=40)
    dw = small_world.to_device("cuda:0")
    B = 192
    st = EnvState(B, small_world.A, device="cuda:0", with_obs=True)
    ops.env_reset(cfg, dw, st)
    g = torch.Generator().manual_seed(2)
    for t in range(90):
        a = torch.stack([torch.rand(B, generator=g) * 2 - 1, torch.rand(B, generator=g) * 0.6 - 0.3], -1).cuda()
        ops.env_step(cfg, dw, st, action=a)
        want = ops.state_obs(dw, st)
        assert torch.equal(st["obs"], want), (t, int((st["obs"] != want).any(1).sum()))


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_loader_built_validation_worlds_hip_vs_oracle(case, tmp_path):
    """the reference's validation cases go load_waypoint_suite_data -> (pick_background_traffic) ->
    world_from_waypoint_suite -> the HIP step, 200 steps of 64 envs with in-place re-spawns, against the oracle on the same
    world: every state array and output bit for bit, and the birdview of the last step pixel for pixel.  Case 2 (Town03)
    also takes background agents from a background-traffic file in the reference's schema (ref gym_env.py:200-235)."""
    from tests.golden_util import BACKGROUND_DIR, write_validation_suite_yaml
    from torchdriveenv_amd import ops
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.loaders import load_waypoint_suite_data, pick_background_traffic

    val = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "validation_cases.yml")))
    one = WaypointSuite(locations=val.locations[case:case + 1], waypoint_suite=val.waypoint_suite[case:case + 1],
                        scenarios=val.scenarios[case:case + 1], car_sequence_suite=val.car_sequence_suite[case:case + 1])
    bg = None
    if case == 2:
        bt = pick_background_traffic(one.locations[0], BACKGROUND_DIR)
        assert bt is not None and len(bt["agent_states"]) == 24
        bg = lambda loc: bt                                                   # noqa: E731
    world = world_from_waypoint_suite(one, agents_per_env=8, background=bg, background_radius=250.0)
    n_present = int(world.arrays["spawn"]["present"].sum())
    if case == 2:
        assert n_present == 8                                                 # ego + 2 scenario agents + 5 background agents
    if case == 1:
        assert world.arrays["spawn"][0, 1]["replay_len"] == 300              # the parked replay car
    B, A, DEV = 64, 8, "cuda:0"
    cfg = _abi.default_config(seed=5 + case, distance_cutoff=0.25, max_steps=60)
    dw = world.to_device(DEV)
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(case)
    for t in range(200):
        # mostly forward with some steering noise: episodes end by truncation, by leaving the corridor and by collisions
        act = np.stack([rng.uniform(0.0, 1.0, B), rng.normal(0.0, 0.05, B).clip(-0.3, 0.3)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(torch.from_numpy(act).to(DEV))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
        if t % 40 == 39 or t == 199:
            h, d = hs.host(), ds.host()
            for k in h:
                if k != "action":
                    assert np.array_equal(h[k].view(np.uint8), d[k].view(np.uint8)), (case, t, k)
    assert hs["episode"].max() >= 3                                           # re-spawns happened
    want = oracle.render_ego(cfg, world, hs)
    got = ops.render_ego(cfg, dw, ds).cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("obs_mode,frame_stack", [("state", 1), ("birdview", 1), ("birdview", 3)])
def test_step_is_capturable_in_a_hip_graph(small_world, obs_mode, frame_stack):
    """BatchedWaypointEnv.step makes no host synchronisation and no allocation whose address a later step depends on: ONE
    timestep (policy stand-in -> step -> observation) captured as a HIP graph and replayed T times == T eager steps of a second env
    on the same actions (state, rewards, flags, observations: same bits).  The frame-stack ring advances on the host
    (its phase is a kernel argument), so n_stack > 1 is captured as one graph PER PHASE."""
    cfg = EnvConfig(seed=23, distance_cutoff=0.25, max_environment_steps=40)
    B, T = 96, 90
    kw = dict(num_envs=B, device="cuda:0", obs_mode=obs_mode, frame_stack=frame_stack)
    eager, graphed = BatchedWaypointEnv(cfg, small_world, **kw), BatchedWaypointEnv(cfg, small_world, **kw)
    assert torch.equal(eager.reset(), graphed.reset())
    g = torch.Generator().manual_seed(5)
    acts = torch.stack([torch.rand(T, B, generator=g) * 1.4 - 0.4, torch.rand(T, B, generator=g) * 0.4 - 0.2], -1).to("cuda:0")
    act = torch.zeros(B, 2, device="cuda:0")                 # the graph's action buffer
    side = torch.cuda.Stream(device="cuda:0")
    graphs, outs = [], []
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for ph in range(frame_stack):                        # (warm-up outside the capture, then one graph per ring phase)
            act.copy_(acts[ph])
            gr = torch.cuda.CUDAGraph()
            if ph == 0:
                snap = graphed.state_dict()
            with torch.cuda.graph(gr, stream=side):
                out = graphed.step(act)
            graphs.append(gr)
            outs.append(out)
        graphed.load_state_dict(snap)                        # (capture does not execute: rewind the host-side ring phase too)
        torch.cuda.synchronize()
        for t in range(T):
            act.copy_(acts[t])
            graphs[t % frame_stack].replay()
            o_g, r_g, te_g, tr_g, _ = outs[t % frame_stack]
            o_e, r_e, te_e, tr_e, _ = eager.step(acts[t])
            assert torch.equal(o_g, o_e) and torch.equal(r_g, r_e) and torch.equal(te_g, te_e) and torch.equal(tr_g, tr_e), t
    for k in ("x", "y", "psi", "v", "steps", "episode", "scn", "target_idx"):
        assert torch.equal(eager.state[k], graphed.state[k]), k
    assert int(eager.state["episode"].max()) >= 2            # re-spawns happened inside the replays


def _validation_lights_and_headings(val):
    """per location of the validation suite: stop lines across every scenario's route at its second and fourth waypoint (two lights,
    alternating red), and a lane-direction field that bends with the position"""
    import math

    lights = {}
    for loc, wps in zip(val.locations, val.waypoint_suite):
        d = lights.setdefault(loc, dict(stoplines=[], phases=[(25, [0]), (10, []), (25, [1]), (10, [])]))
        for n, light in ((1, 0), (3, 1)):
            if n < len(wps):
                p, q = wps[n - 1], wps[n]
                d["stoplines"].append((q[0], q[1], math.atan2(q[1] - p[1], q[0] - p[0]), 2.0, 9.0, light))

    def field(loc, x, y):
        i = val.locations.index(loc)
        p, q = val.waypoint_suite[i][0], val.waypoint_suite[i][1]
        return math.atan2(q[1] - p[1], q[0] - p[0]) + 0.004 * (x - p[0]) - 0.003 * (y - p[1])
    return lights, field


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_validation_worlds_with_traffic_lights_and_start_headings_hip_vs_oracle(case, tmp_path):
    """The reference's five validation cases with what its map config adds to every env (ref gym_env.py:181-189, 290-291, 359-361, 415,
    429): stop lines + a light cycle (`traffic_lights=`) and a lane-direction field for the start heading (`start_headings=`), through
    world_from_waypoint_suite -> the HIP step with TDE_F_TRAFFIC_LIGHTS against the oracle on the same world: every state array and
    output bit for bit over 200 steps of 64 envs with re-spawns (start headings read from the table every time), red-line violations
    terminating episodes, the NPCs stopping at red lines, the birdview with the painted lines pixel for pixel."""
    from tests.golden_util import write_validation_suite_yaml
    from torchdriveenv_amd import ops
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.loaders import load_waypoint_suite_data

    val = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "validation_cases.yml")))
    lights, field = _validation_lights_and_headings(val)
    one = WaypointSuite(locations=val.locations[case:case + 1], waypoint_suite=val.waypoint_suite[case:case + 1],
                        scenarios=val.scenarios[case:case + 1], car_sequence_suite=val.car_sequence_suite[case:case + 1])
    world = world_from_waypoint_suite(one, agents_per_env=8, traffic_lights=lights, start_headings=field)
    assert world.has_lights and world.ints["NH"] == 16 and world.arrays["maps"]["n_stop"].max() >= 1
    B, A, DEV = 64, 8, "cuda:0"
    cfg = _abi.default_config(seed=15 + case, distance_cutoff=0.25, max_steps=60, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS)
    dw = world.to_device(DEV)
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    assert np.array_equal(hs["psi"].view(np.uint32), ds["psi"].cpu().numpy().view(np.uint32))
    assert len(np.unique(hs["psi"][::A])) > B // 2                                  # table entry + noise
    rng = np.random.default_rng(case)
    n_tl = 0
    for t in range(200):
        act = np.stack([rng.uniform(0.0, 1.0, B), rng.normal(0.0, 0.04, B).clip(-0.3, 0.3)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(torch.from_numpy(act).to(DEV))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
        n_tl += int(hs["tl_violation"].sum())
        if t % 40 == 39 or t == 199:
            h, d = hs.host(), ds.host()
            for k in h:
                if k != "action":
                    assert np.array_equal(h[k].view(np.uint8), d[k].view(np.uint8)), (case, t, k)
    assert hs["episode"].max() >= 3 and n_tl > 0                                    # re-spawns and red-line violations happened
    want = oracle.render_ego(cfg, world, hs)
    got = ops.render_ego(cfg, dw, ds).cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())


def test_make_with_traffic_lights_terminates_at_a_red_line_and_reports_the_violation(tmp_path):
    """The reference-shaped surface: make(cfg, data, road_meshes=, traffic_lights=, start_headings=) (what gym.make('torchdriveenv-v0')
    returns, ref __init__.py:10) on validation case 0 with its map's stop lines: an ego driven straight ahead crosses the first line
    while its light is red -> terminated, info['traffic_light_violation'] > 0 (ref gym_env.py:415, 429), at the very step the oracle
    says, with the oracle's rewards; with the light green over that stretch the same drive passes the line."""
    from tests.golden_util import write_validation_suite_yaml
    from torchdriveenv_amd.env import corridor_mesh, make, world_from_waypoint_suite
    from torchdriveenv_amd.loaders import load_waypoint_suite_data

    val = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "validation_cases.yml")))
    lights, field = _validation_lights_and_headings(val)
    one = WaypointSuite(locations=val.locations[:1], waypoint_suite=val.waypoint_suite[:1], scenarios=val.scenarios[:1],
                        car_sequence_suite=val.car_sequence_suite[:1])
    mesh = {val.locations[0]: corridor_mesh([one.waypoint_suite[0]], width=14.0)}
    ecfg = EnvConfig(seed=11, distance_cutoff=0.25, use_background_traffic=False, device="cuda:0")
    kw = dict(agents_per_env=8, road_meshes=mesh, start_headings=field)
    results = {}
    for name, spec in (("red", lights), ("green", {k: dict(v, phases=[(200, [])]) for k, v in lights.items()})):
        env = make(ecfg, one, traffic_lights=spec, **kw)
        inner = env.env._env
        assert inner.world.has_lights and (inner.tde_cfg.flags & _abi.F_TRAFFIC_LIGHTS) and (inner.tde_cfg.flags & _abi.F_NPC_FIRST_STEP)
        world = world_from_waypoint_suite(one, agents_per_env=8, traffic_lights=spec, road_meshes=mesh, start_headings=field)
        hs = EnvState(1, 8, with_magnitudes=True)
        oracle.env_reset(inner.tde_cfg, world, hs)
        obs, _ = env.reset()
        assert obs.shape == (3, 64, 64) and np.float32(inner.state["psi"][0].item()) == hs["psi"][0]
        saw = []
        for t in range(80):
            a = np.asarray([1.0, 0.0], np.float32)
            hs["action"][...] = a[None]
            oracle.env_step(inner.tde_cfg, world, hs)
            obs, r, term, trunc, info = env.step(a)
            assert np.float32(r) == hs["reward"][0] and term == bool(hs["terminated"][0]) and float(info["traffic_light_violation"]) == float(hs["tl_violation"][0])
            saw.append(float(info["traffic_light_violation"]))
            if term or trunc:
                break
        results[name] = (t, term, saw[-1])
    t_red, term_red, tl_red = results["red"]
    assert term_red and tl_red > 0 and t_red < 79                                   # stopped by the red line
    assert results["green"][2] == 0.0 and results["green"][0] > t_red              # the same drive passes it on green


def test_batched_and_vec_env_report_red_line_violations(tmp_path):
    """BatchedWaypointEnv / WaypointVecEnv on a WaypointSuite WITH `traffic_lights=`: the world carries light groups, the env steps with
    TDE_F_TRAFFIC_LIGHTS (nobody has to ask), the device-side info and the SB3-shaped per-env infos both report the ego's red-line
    violations, and they are the oracle's"""
    from tests.golden_util import write_validation_suite_yaml
    from torchdriveenv_amd.env import world_from_waypoint_suite
    from torchdriveenv_amd.loaders import load_waypoint_suite_data

    val = load_waypoint_suite_data(write_validation_suite_yaml(str(tmp_path / "validation_cases.yml")))
    lights, field = _validation_lights_and_headings(val)
    B = 96
    ecfg = EnvConfig(seed=31, distance_cutoff=0.25, use_background_traffic=False, max_environment_steps=60, device="cuda:0")
    env = BatchedWaypointEnv(ecfg, val, num_envs=B, agents_per_env=8, obs_mode="state", traffic_lights=lights, start_headings=field)
    assert env.world.has_lights and (env.tde_cfg.flags & _abi.F_TRAFFIC_LIGHTS) and env.world.ints["NH"] == 16
    world = world_from_waypoint_suite(val, agents_per_env=8, traffic_lights=lights, start_headings=field)
    hs = EnvState(B, 8)
    oracle.env_reset(env.tde_cfg, world, hs)
    env.reset()
    rng = np.random.default_rng(5)
    seen = 0
    for t in range(120):
        act = np.stack([rng.uniform(0.2, 1.0, B), rng.normal(0.0, 0.03, B).clip(-0.3, 0.3)], -1).astype(np.float32)
        hs["action"][...] = act
        oracle.env_step(env.tde_cfg, world, hs)
        _, rew, term, trunc, info = env.step(torch.from_numpy(act).to("cuda:0"))
        tl = info["traffic_light_violation"].cpu().numpy()
        assert np.array_equal(tl, hs["tl_violation"].astype(np.float32)) and np.array_equal(rew.cpu().numpy().view(np.uint32), hs["reward"].view(np.uint32))
        seen += int(tl.sum())
    assert seen > 0 and int(hs["episode"].max()) >= 2
    # the SB3-shaped path: the column is there and carries violations too
    venv = BatchedWaypointEnv(ecfg, val, num_envs=B, agents_per_env=8, obs_mode="state", traffic_lights=lights, start_headings=field).as_vec_env()
    venv.reset()
    n = 0
    for t in range(120):
        act = np.stack([rng.uniform(0.2, 1.0, B), np.zeros(B)], -1).astype(np.float32)
        _, _, dones, infos = venv.step(act)
        n += int(np.asarray(infos.column("traffic_light_violation")).sum())
        if dones.any():
            i = int(np.flatnonzero(dones)[0])
            assert "terminal_observation" in infos[i] and "traffic_light_violation" in infos[i]
    assert n > 0
