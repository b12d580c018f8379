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
from torchdriveenv_amd import _abi, _lib, ops  # noqa: E402
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
def test_batched_env_fused_magnitudes_equal_indicator_env_and_post_step_form(small_world, obs_mode, frame_stack):
    """BatchedWaypointEnv (default: info_magnitudes=True, tde_state.magnitudes written by the ONE-launch step) gives the
    observations, rewards, flags and episodes of the indicator env (info_magnitudes=False) and of the round-4 two-launch form
    (step without re-spawn, then tde_env_post_step), and info["offroad"] / info["collision"] are the oracle's magnitudes of the state
    the step left (the reference's info semantics, gym_env.py:427-428) instead of 0 / 1"""
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv

    B = 160
    kw = dict(num_envs=B, agents_per_env=16, obs_mode=obs_mode, frame_stack=frame_stack)
    plain = BatchedWaypointEnv(EnvConfig(seed=8, distance_cutoff=0.25), small_world, info_magnitudes=False, **kw)
    mag = BatchedWaypointEnv(EnvConfig(seed=8, distance_cutoff=0.25), small_world, **kw)
    two = BatchedWaypointEnv(EnvConfig(seed=8, distance_cutoff=0.25), small_world, **kw)
    assert mag.info_magnitudes and plain.state["magnitudes"] is None
    hs = EnvState(B, 16)
    cfg_na = _abi.TdeConfig.from_buffer_copy(plain.tde_cfg)
    cfg_na.flags &= ~_abi.F_AUTORESET
    oracle.env_reset(cfg_na, small_world, hs)
    o0 = plain.reset()
    assert torch.equal(o0, mag.reset()) and torch.equal(o0, two.reset())
    rng = np.random.default_rng(2)
    seen = 0
    for t in range(90):
        act = np.stack([rng.uniform(-0.5, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        a = dev(act)
        o1, r1, te1, tr1, i1 = plain.step(a)
        o2, r2, te2, tr2, i2 = mag.step(a)
        o3, r3, te3, tr3, i3 = two._step_then_post_step(a)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(te1, te2) and torch.equal(tr1, tr2), t
        assert torch.equal(o3, o2) and torch.equal(r3, r2) and torch.equal(te3, te2) and torch.equal(tr3, tr2), t
        hs["action"][...] = act
        oracle.env_step(cfg_na, small_world, hs)
        want = oracle.ego_infractions(cfg_na, small_world, hs)
        assert np.array_equal(hs["magnitudes"].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(i2["offroad"].cpu().numpy().view(np.uint32), want[:, 0].view(np.uint32)), t
        assert np.array_equal(i2["collision"].cpu().numpy().view(np.uint32), want[:, 1].view(np.uint32)), t
        assert torch.equal(mag.state["magnitudes"], two.state["magnitudes"]), t
        assert torch.equal(i1["offroad"] > 0, i2["offroad"] > 0) and torch.equal(i1["collision"] > 0, i2["collision"] > 0)
        seen += int((want[:, 0] > 0).sum())
        done = (hs["terminated"] | hs["truncated"]).astype(np.uint8)
        if done.any():
            oracle.env_reset(cfg_na, small_world, hs, done)
    for k in ("x", "y", "psi", "v", "steps", "episode", "scn"):
        assert torch.equal(plain.state[k], mag.state[k]) and torch.equal(two.state[k], mag.state[k]), k
    assert seen > 10 and int(plain.state["episode"].max()) > 1


def _step_forms(A):
    return (("trio", "solo") if A in (8, 16, 32) else ("solo",))


@pytest.mark.parametrize("which", ["junctions_8192x16", "town_2048x16", "town_lights_512x16", "wide_64x128", "a32_1024", "a64_256", "a4_512",
                                   "squared_threshold_512x16", "a8_lights_512"])
def test_one_launch_step_magnitudes_equal_step_plus_post_step_and_oracle(which, town):
    """tde_env_step with tde_state.magnitudes (one launch, every kernel form) == step without TDE_F_AUTORESET + tde_env_post_step
    (round 4's two launches) == the oracle's step: magnitudes, state, outputs, bit for bit - on BASELINE configs[2]'s batch, the
    town (large grid), a signalised town, 128 slots per env, and the other group shapes"""
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world

    lights, squared = False, 0
    if which == "squared_threshold_512x16":     # (the other reading of the threshold: distances are SQUARED distances, tde_abi.h)
        world, B, A, T, squared = synthetic_world(n_scn=8, A=16, seed=0, n_maps=2, threshold=float(np.sqrt(0.5))), 512, 16, 60, 1
    elif which == "a8_lights_512":
        world, B, A, T, lights = synthetic_world(n_scn=8, A=8, seed=1, n_maps=2), 512, 8, 60, True
    elif which == "junctions_8192x16":
        world, B, A, T = synthetic_world(n_scn=64, A=16, seed=0, n_maps=4), 8192, 16, 12
    elif which == "town_2048x16":
        world, B, A, T = town, 2048, 16, 25
    elif which == "town_lights_512x16":
        world, B, A, T, lights = synthetic_town(n_scn=64, A=16, seed=3, n_streets=5, spacing=100.0, ext=45.0, n_signals=9), 512, 16, 60, True
    elif which == "wide_64x128":
        world, B, A, T = synthetic_town(n_scn=4, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4), 64, 128, 30
    elif which == "a32_1024":
        world, B, A, T = synthetic_world(n_scn=16, A=32, seed=2, n_maps=2), 1024, 32, 40
    elif which == "a64_256":
        world, B, A, T = synthetic_world(n_scn=8, A=64, seed=3, n_maps=2), 256, 64, 40
    else:
        world, B, A, T = synthetic_world(n_scn=8, A=4, seed=4, n_maps=2), 512, 4, 60
    flags = _abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0)
    cfg = _abi.default_config(seed=21, flags=flags, max_steps=40, distance_cutoff=0.25, offroad_threshold_squared=squared)
    cfg_na = _abi.TdeConfig.from_buffer_copy(cfg)
    cfg_na.flags &= ~_abi.F_AUTORESET
    dw = world.to_device(DEV)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    forms = _step_forms(A)
    fused = [EnvState(B, A, device=DEV, with_obs=True) for _ in forms]
    two = EnvState(B, A, device=DEV, with_obs=True, with_magnitudes=False)
    for d in fused + [two]:
        d.load(hs.host())
    mag2 = torch.zeros(B, 4, device=DEV)
    rng = np.random.default_rng(6)
    n_off = n_col = n_done = 0
    try:
        for t in range(T):
            act = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, world, hs)
            a = dev(act)
            for form, d in zip(forms, fused):
                _lib.kernel_override(step=form)
                ops.env_step(cfg, dw, d, action=a)
            _lib.kernel_override()
            ops.env_step(cfg_na, dw, two, action=a)
            ops.env_post_step(cfg, dw, two, mag2)
            want = hs["magnitudes"]
            for form, d in zip(forms, fused):
                got = d["magnitudes"].cpu().numpy()
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (which, form, t, np.abs(got - want).max())
            assert np.array_equal(mag2.cpu().numpy().view(np.uint32), want.view(np.uint32)), (which, t)
            n_off += int((want[:, 0] > 0).sum()); n_col += int((want[:, 1] > 0).sum())
            n_done += int(((hs["done_bits"] & 3) != 0).sum())
        for form, d in zip(forms, fused):
            assert_state_equal(hs.host(), d.host(), f"{which}: fused step ({form}) vs oracle")
            for k in ("x", "y", "psi", "v", "steps", "episode", "scn", "obs", "reward"):
                assert torch.equal(d[k], two[k]), (which, form, k)
    finally:
        _lib.kernel_override()
    assert n_done > 0 and n_off + n_col > 0, (n_done, n_off, n_col)
    if which in ("junctions_8192x16", "town_2048x16"):
        assert n_off > 20 and n_col > 5


def test_vecenv_with_info_magnitudes_equals_the_plain_vecenv(small_world):
    """the SB3 path over BatchedWaypointEnv (default info_magnitudes=True): WaypointVecEnv clears TDE_F_AUTORESET for its own masked
    reset, the step kernel writes the magnitudes into the packed output arena (one device-to-host copy for everything) -
    observations, rewards, dones, terminal observations and episode statistics equal the indicator VecEnv's, and the info columns
    carry the magnitudes"""
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv, WaypointVecEnv

    B = 96
    cfg = EnvConfig(seed=12, distance_cutoff=0.25, max_environment_steps=40)
    kw = dict(num_envs=B, agents_per_env=16, obs_mode="state")
    v0 = WaypointVecEnv(BatchedWaypointEnv(cfg, small_world, info_magnitudes=False, **kw))
    v1 = WaypointVecEnv(BatchedWaypointEnv(cfg, small_world, **kw))
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


@pytest.mark.parametrize("near_range", [0.0, 0.25])
def test_magnitudes_through_the_grid_scans_when_no_near_list_reaches(near_range):
    """a world WITHOUT near lists (near_range = 0), and one whose lists end a quarter metre beyond the threshold: every / most
    flagged corners then take the fall-back scans of the grid - the low-register one inside the three-role step kernel, the
    two-records-per-trip one in the one-role kernel and in tde_ego_infractions - and the values are still the oracle's brute force,
    bit for bit, for egos that wander metres off the road (terminated_at_infraction = 0)"""
    from torchdriveenv_amd.synth import synthetic_world

    world = synthetic_world(n_scn=8, A=16, seed=0, n_maps=2, near_range=near_range)
    tn = world.arrays["tile_near"]
    listed = int(((tn != 0) & (tn != 0xFFFFFFFF)).sum())
    assert (listed == 0) if near_range == 0.0 else (listed > 0)
    cfg = _abi.default_config(seed=31, flags=_abi.F_ALL & ~_abi.F_AUTORESET, terminated_at_infraction=0, max_steps=10_000)
    B, A = 256, 16
    dw = world.to_device(DEV)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    forms = ("trio", "solo")
    ds = [EnvState(B, A, device=DEV) for _ in forms]
    for d in ds:
        d.load(hs.host())
    rng = np.random.default_rng(8)
    n_off = far = 0
    try:
        for t in range(45):
            act = np.stack([rng.uniform(-0.2, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, world, hs)
            want = hs["magnitudes"]
            for form, d in zip(forms, ds):
                _lib.kernel_override(step=form)
                ops.env_step(cfg, dw, d, action=dev(act))
                got = d["magnitudes"].cpu().numpy()
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (form, t, np.abs(got - want).max())
            _lib.kernel_override()
            assert np.array_equal(ops.ego_infractions(cfg, dw, ds[0]).cpu().numpy().view(np.uint32), want.view(np.uint32)), t
            n_off += int((want[:, 0] > 0).sum()); far += int((want[:, 0] > 8.0).sum())
    finally:
        _lib.kernel_override()
    assert n_off > 500 and far > 50            # corners metres beyond any list: the scans grew their squares


@pytest.mark.parametrize("A", [16, 32, 64])
def test_magnitudes_far_off_a_small_map_take_the_triangle_walk(A):
    """egos tens to hundreds of metres off a map of a few hundred triangles (terminated_at_infraction = 0 lets them drive on): the
    square a grid scan would have to walk holds more cells than the map has triangles, and the scans switch to the oracle's own
    definition - the minimum over ALL triangles from the raw vertices (tde_world.tri).  Same bits in every kernel form and in
    tde_ego_infractions, outside the grid too."""
    from torchdriveenv_amd.synth import synthetic_world

    world = synthetic_world(n_scn=8, A=A, seed=1, n_maps=2)
    cfg = _abi.default_config(seed=5, flags=_abi.F_ALL & ~_abi.F_AUTORESET, terminated_at_infraction=0, max_steps=10_000)
    B = 128
    dw = world.to_device(DEV)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    rng = np.random.default_rng(3)
    # the egos are put at 10 .. 600 m from their spawn points, in any direction (half of them beyond the grid's padding)
    r = np.exp(rng.uniform(np.log(10.0), np.log(600.0), B)).astype(np.float32)
    th = rng.uniform(0, 2 * np.pi, B)
    x, y = hs["x"].reshape(B, A), hs["y"].reshape(B, A)
    x[:, 0] += (r * np.cos(th)).astype(np.float32); y[:, 0] += (r * np.sin(th)).astype(np.float32)
    forms = _step_forms(A)
    ds = [EnvState(B, A, device=DEV) for _ in forms]
    for d in ds:
        d.load(hs.host())
    try:
        for t in range(6):
            act = np.stack([rng.uniform(0, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, world, hs)
            want = hs["magnitudes"]
            for form, d in zip(forms, ds):
                _lib.kernel_override(step=form)
                ops.env_step(cfg, dw, d, action=dev(act))
                got = d["magnitudes"].cpu().numpy()
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (form, t, np.abs(got - want).max())
            _lib.kernel_override()
            assert np.array_equal(ops.ego_infractions(cfg, dw, ds[0]).cpu().numpy().view(np.uint32), want.view(np.uint32)), t
    finally:
        _lib.kernel_override()
    assert (want[:, 0] > 4 * 8.0).sum() > B // 2 and want[:, 0].max() > 1000.0       # (four corners, each hundreds of metres out)
