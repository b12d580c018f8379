"""Infraction MAGNITUDES of the ego (tde_ego_infractions): what the reference's info dict holds under "offroad" / "collision"
(ref gym_env.py:427-428: simulator.compute_offroad() / compute_collision() for the exposed agent; Monitor logs them,
examples/rl_training.py:128).  Upstream's values are unpinned (torchdrivesim absent): the oracle defines them - sum over the ego's
corners of clamp(distance to the mesh - threshold, 0), brute force over every triangle; number of agents the ego overlaps - and
the kernel, which finds a corner's nearest triangle through the grid index (candidate lists, then a growing scan), must return
the same bits, however far off the road the ego is.  collision = sum of the IoUs (Sutherland-Hodgman clip, fp32) with the agents
the ego overlaps - CollisionMetric.nograd's published form - plus their number."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from tests.test_gpu_parity import assert_state_equal, dev  # noqa: E402
from torchdriveenv_amd import _abi, ops  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402

DEV = "cuda:0"


def _wander(world, cfg, B, A, steps, seed):
    """B envs stepped `steps` times by the oracle with infractions that do not end the episode: egos end up anywhere"""
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    rng = np.random.default_rng(seed)
    for _ in range(steps):
        hs["action"][...] = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        oracle.env_step(cfg, world, hs)
    return hs


@pytest.mark.parametrize("squared", [0, 1])
def test_ego_infraction_magnitudes_bit_exact(small_world, squared):
    from torchdriveenv_amd.synth import synthetic_world

    world = small_world if not squared else synthetic_world(n_scn=8, A=16, seed=0, n_maps=2, threshold=float(np.sqrt(0.5)))
    cfg = _abi.default_config(seed=3, flags=_abi.F_ALL & ~_abi.F_AUTORESET, terminated_at_infraction=0, max_steps=10_000,
                              offroad_threshold_squared=squared)
    B, A = 300, 16
    dw = world.to_device(DEV)
    ds = EnvState(B, A, device=DEV)
    for steps in (15, 45):
        hs = _wander(world, cfg, B, A, steps, seed=steps)
        ds.load(hs.host())
        want = oracle.ego_infractions(cfg, world, hs)
        got = ops.ego_infractions(cfg, dw, ds).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
        off = hs["offroad"].reshape(B, A)[:, 0] > 0
        assert np.array_equal(want[:, 0] > 0, off)                   # the magnitude is positive exactly where the mask is set
        assert np.array_equal(want[:, 2] > 0, hs["collided"].reshape(B, A)[:, 0] > 0)      # overlap count <=> the collision mask
        assert ((want[:, 1] > 0) <= (want[:, 2] > 0)).all() and (want[:, 1] <= want[:, 2]).all() and (want[:, 3] == 0).all()
    assert want[:, 0].max() > 20.0 and (want[:, 0] > 0).sum() > 50 and (want[:, 0] == 0).sum() > 20


def test_ego_infraction_magnitudes_town_and_128_slots(town):
    from torchdriveenv_amd.synth import synthetic_town

    cfg = _abi.default_config(seed=5, flags=_abi.F_ALL & ~_abi.F_AUTORESET, terminated_at_infraction=0, max_steps=10_000)
    for world, B, A in ((town, 96, 16), (synthetic_town(n_scn=4, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4), 16, 128)):
        hs = _wander(world, cfg, B, A, 40, seed=A)
        ds = EnvState(B, A, device=DEV)
        ds.load(hs.host())
        want = oracle.ego_infractions(cfg, world, hs)
        got = ops.ego_infractions(cfg, world.to_device(DEV), ds).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (A, np.abs(got - want).max())
        assert (want[:, 0] > 0).any()
    assert want[:, 2].max() >= 1.0 and 0.0 < want[:, 1].max() <= want[:, 2].max()   # crowded scenes: the ego overlaps somebody


@pytest.mark.parametrize("A,squared", [(16, 0), (16, 1), (128, 0)])
def test_post_step_equals_magnitudes_then_masked_reset(small_world, A, squared):
    """tde_env_post_step after a step without TDE_F_AUTORESET == tde_ego_infractions (every env, ungated) + tde_env_reset of the
    finished envs + tde_state_obs, and == the oracle: magnitudes, state, compact observation, bit for bit - the gate (magnitudes
    only for the envs the step flagged) never changes a value"""
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world

    world = small_world
    if squared:
        world = synthetic_world(n_scn=8, A=16, seed=0, n_maps=2, threshold=float(np.sqrt(0.5)))
    if A == 128:
        world = synthetic_town(n_scn=4, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
    B = 96 if A == 16 else 24
    flags = _abi.F_ALL & ~_abi.F_AUTORESET
    cfg = _abi.default_config(seed=17, flags=flags, max_steps=30, offroad_threshold_squared=squared, distance_cutoff=0.25)
    cfg_post = _abi.TdeConfig.from_buffer_copy(cfg)
    cfg_post.flags |= _abi.F_AUTORESET
    dw = world.to_device(DEV)
    hs = EnvState(B, A)
    d1, d2 = EnvState(B, A, device=DEV, with_obs=True), EnvState(B, A, device=DEV, with_obs=True)
    oracle.env_reset(cfg, world, hs)
    d1.load(hs.host()); d2.load(hs.host())
    rng = np.random.default_rng(4)
    mag = torch.zeros(B, 4, device=DEV)
    n_done = n_off = 0
    for t in range(70):
        act = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        hs["action"][...] = act
        oracle.env_step(cfg, world, hs)
        want = oracle.ego_infractions(cfg, world, hs)
        done = (hs["terminated"] | hs["truncated"]).astype(np.uint8)
        oracle.env_reset(cfg, world, hs, done)
        for d in (d1, d2):
            ops.env_step(cfg, dw, d, action=dev(act))
        ops.env_post_step(cfg_post, dw, d1, mag)                                  # one launch
        got2 = ops.ego_infractions(cfg, dw, d2)                                    # three
        ops.env_reset(cfg, dw, d2, dev(done))
        ops.state_obs(dw, d2, d2["obs"])
        assert np.array_equal(mag.cpu().numpy().view(np.uint32), want.view(np.uint32)), t
        assert torch.equal(mag, got2), t
        assert torch.equal(d1["obs"], d2["obs"]), t
        n_done += int(done.sum()); n_off += int((want[:, 0] > 0).sum())
    assert_state_equal(hs.host(), d1.host(), "post_step vs oracle")
    assert_state_equal(d2.host(), d1.host(), "post_step vs the three launches")
    assert n_done > 20 and n_off > 5


@pytest.mark.parametrize("obs_mode,frame_stack", [("birdview", 3), ("state", 1)])
def test_batched_env_info_magnitudes_equals_the_one_launch_step(small_world, obs_mode, frame_stack):
    """BatchedWaypointEnv(info_magnitudes=True) - step without in-kernel re-spawn, then tde_env_post_step - gives the
    observations, rewards, flags and episodes of the one-launch step, and info["offroad"] / info["collision"] are the oracle's
    magnitudes of the state the step left (the reference's info semantics) instead of 0 / 1"""
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv

    B = 160
    kw = dict(num_envs=B, agents_per_env=16, obs_mode=obs_mode, frame_stack=frame_stack)
    plain = BatchedWaypointEnv(EnvConfig(seed=8, distance_cutoff=0.25), small_world, **kw)
    mag = BatchedWaypointEnv(EnvConfig(seed=8, distance_cutoff=0.25), small_world, info_magnitudes=True, **kw)
    hs = EnvState(B, 16)
    cfg_na = _abi.TdeConfig.from_buffer_copy(plain.tde_cfg)
    cfg_na.flags &= ~_abi.F_AUTORESET
    oracle.env_reset(cfg_na, small_world, hs)
    assert torch.equal(plain.reset(), mag.reset())
    rng = np.random.default_rng(2)
    seen = 0
    for t in range(90):
        act = np.stack([rng.uniform(-0.5, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        a = dev(act)
        o1, r1, te1, tr1, i1 = plain.step(a)
        o2, r2, te2, tr2, i2 = mag.step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), t
        hs["action"][...] = act
        oracle.env_step(cfg_na, small_world, hs)
        want = oracle.ego_infractions(cfg_na, small_world, hs)
        assert np.array_equal(i2["offroad"].cpu().numpy().view(np.uint32), want[:, 0].view(np.uint32)), t
        assert np.array_equal(i2["collision"].cpu().numpy().view(np.uint32), want[:, 1].view(np.uint32)), t
        assert torch.equal(i1["offroad"] > 0, i2["offroad"] > 0) and torch.equal(i1["collision"] > 0, i2["collision"] > 0)
        seen += int((want[:, 0] > 0).sum())
        done = (hs["terminated"] | hs["truncated"]).astype(np.uint8)
        if done.any():
            oracle.env_reset(cfg_na, small_world, hs, done)
    for k in ("x", "y", "psi", "v", "steps", "episode", "scn"):
        assert torch.equal(plain.state[k], mag.state[k]), k
    assert seen > 10 and int(plain.state["episode"].max()) > 1


def test_vecenv_with_info_magnitudes_equals_the_plain_vecenv(small_world):
    """the SB3 path over BatchedWaypointEnv(info_magnitudes=True): WaypointVecEnv clears TDE_F_AUTORESET for its own masked reset,
    so tde_env_post_step computes magnitudes and re-spawns NOTHING - observations, rewards, dones, terminal observations and
    episode statistics equal the plain VecEnv's, and the info columns carry the magnitudes"""
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv, WaypointVecEnv

    B = 96
    cfg = EnvConfig(seed=12, distance_cutoff=0.25, max_environment_steps=40)
    kw = dict(num_envs=B, agents_per_env=16, obs_mode="state")
    v0 = WaypointVecEnv(BatchedWaypointEnv(cfg, small_world, **kw))
    v1 = WaypointVecEnv(BatchedWaypointEnv(cfg, small_world, info_magnitudes=True, **kw))
    assert np.array_equal(v0.reset(), v1.reset())
    rng = np.random.default_rng(3)
    n_done = n_mag = 0
    for t in range(90):
        acts = np.stack([rng.uniform(-0.5, 1, B), rng.uniform(-0.3, 0.3, B)], -1)
        o0, r0, d0, i0 = v0.step(acts)
        o1, r1, d1, i1 = v1.step(acts)
        assert np.array_equal(o0, o1) and np.array_equal(r0, r1) and np.array_equal(d0, d1), t
        for i in np.nonzero(d0)[0]:
            a, b = i0[int(i)], i1[int(i)]
            assert np.array_equal(a["terminal_observation"], b["terminal_observation"]) and a["episode"]["r"] == b["episode"]["r"] and a["episode"]["l"] == b["episode"]["l"]
            assert (a["offroad"] > 0) == (b["offroad"] > 0) and (a["collision"] > 0) == (b["collision"] > 0)
            if b["offroad"] > 0:
                n_mag += 1
                assert a["offroad"] == 1.0 and b["offroad"] != 1.0       # an indicator there, a distance here
            n_done += 1
    for k in ("x", "y", "psi", "episode", "scn", "steps"):
        assert torch.equal(v0.env.state[k], v1.env.state[k]), k
    assert n_done > 30 and n_mag > 3
