"""More than 64 agent slots per env.  The reference assembles up to ~100 agents per simulator (ref gym_env.py:216-237:
`len(states) + density < 100`, `agent_count = max(95 - ..., 0)`); TDE_MAX_AGENTS is 128: an env then spans two wavefronts of
a workgroup and the one-role kernels run their generic forms (every row walked with the exact tests), a rollout is a
sequence of one-step launches, the rasteriser takes the slots 64 at a time.  Same bar: every array bit-identical to the
oracle, every pixel equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from tests.test_gpu_parity import assert_state_equal, dev, random_agents  # noqa: E402
from torchdriveenv_amd import _abi, ops  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def crowded_town():
    """a small town whose scenarios hold ~100 agents in 128 slots (the rest absent: the reference's ~96 agents, padded)"""
    from torchdriveenv_amd.synth import synthetic_town

    w = synthetic_town(n_scn=6, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
    n = w.arrays["spawn"]["present"].reshape(-1, 128).sum(1)
    assert n.min() >= 96 and n.max() <= 128
    return w


def test_collision_mask_128_slots():
    rng = np.random.default_rng(128)
    B, A = 24, 128
    ag = random_agents(rng, B, A, spread=60.0)
    want = oracle.compute_collision(B, A, ag["x"], ag["y"], ag["psi"], ag["length"], ag["width"], ag["present"])
    got = ops.compute_collision(B, A, dev(ag["x"]), dev(ag["y"]), dev(ag["psi"]), dev(ag["length"]), dev(ag["width"]),
                                dev(ag["present"])).cpu().numpy()
    assert np.array_equal(got, want) and 0 < want.sum() < want.size


@pytest.mark.parametrize("lights", [False, True, "crowded"])
def test_env_step_rollout_and_birdview_128_slots(crowded_town, lights):
    """lights = "crowded": ~100 agents per env AND every junction signalised (light groups of up to 20 stop lines: the NPCs queue
    at red lines in platoons - the busiest form of the 128-slot sweeps and of the four-lines-per-trip stop-line loops)"""
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world

    world = crowded_town
    if lights == "crowded":
        world = synthetic_town(n_scn=6, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4, n_signals=4)
        assert world.ints["n_maps"] == 5 and world.arrays["maps"]["n_stop"].max() >= 12
    elif lights:
        world = synthetic_world(n_scn=4, A=128, seed=9, n_maps=2)
    flags = _abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0)
    cfg = _abi.default_config(seed=41, distance_cutoff=0.25, flags=flags, max_steps=50)
    B, A = 20, 128
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    assert_state_equal(hs.host(), ds.host(), "reset, 128 slots")
    rng = np.random.default_rng(7)
    for t in range(70):
        act = np.stack([rng.uniform(-0.3, 1, B), rng.uniform(-0.2, 0.2, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
        if t % 10 == 0 or t > 66:
            assert_state_equal(hs.host(), ds.host(), f"step {t}, 128 slots, lights={lights}")
        if t in (0, 33, 69):
            want = oracle.render_ego(cfg, world, hs)
            got = ops.render_ego(cfg, dw, ds).cpu().numpy()
            assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ at step {t}"
    assert hs["episode"].max() > 1 and hs["collided"].sum() >= 0
    from torchdriveenv_amd import _lib

    try:
        for team in ("duo", "solo", None):                            # two roles (four wavefronts per env), one role, the dispatch rule
            _lib.kernel_override(rollout=team)
            K = 25 if team else 60                                    # (60 steps: re-spawns inside the launch, max_steps = 50)
            actions = np.stack([rng.uniform(-0.3, 1, (K, B)), rng.uniform(-0.2, 0.2, (K, B))], -1).astype(np.float32)
            hr, hd = oracle.env_rollout(cfg, world, hs, actions)
            dr, dd = ops.env_rollout(cfg, dw, ds, dev(actions))
            assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32)) and np.array_equal(dd.cpu().numpy(), hd), team
            assert_state_equal(hs.host(), ds.host(), f"rollout, 128 slots, {team}")
    finally:
        _lib.kernel_override()


@pytest.mark.parametrize("lights", [False, True])
@pytest.mark.parametrize("form", [None, "duo", "solo"])
def test_both_step_forms_at_128_slots_with_every_output(crowded_town, lights, form):
    """tde_env_step at 128 agent slots: the two-role kernel (round 6: drive / judge wavefronts for each half of the env's slots, the next
    step's controller beside the judges, actions through tde_act_cache) in its eight-wavefront form (None: what the library picks for a
    small batch - sweep and offroad helpers beside the drivers and judges) and its four-wavefront form ("duo"), and the one-role kernel
    (tde_kernel_override(0, 1); also what a
    state without the action cache gets), with info terms, done bits, episode statistics, the compact observation and the infraction
    magnitudes: 60 steps with re-spawns == the oracle, every array bit for bit; a state edit behind the cache's back (load) is survived"""
    from torchdriveenv_amd import _lib
    from torchdriveenv_amd.synth import synthetic_world

    world = synthetic_world(n_scn=4, A=128, seed=9, n_maps=2) if lights else crowded_town
    cfg = _abi.default_config(seed=43, distance_cutoff=0.25, flags=_abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0), max_steps=25)
    B, A = 24, 128
    hs = EnvState(B, A, with_magnitudes=True)                         # (the oracle does not form the compact observation: state_obs below)
    ds = EnvState(B, A, device=DEV, with_obs=True, with_magnitudes=True)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(11)
    _lib.kernel_override(step=form)
    try:
        for t in range(60):
            act = np.stack([rng.uniform(-0.3, 1, B), rng.uniform(-0.25, 0.25, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, world, hs)
            ops.env_step(cfg, dw, ds, action=dev(act))
            if t % 7 == 0 or t > 56:
                assert_state_equal(hs.host(), ds.host(), f"step {t}, 128 slots, form {form}, lights={lights}")
                assert torch.equal(ds["obs"], ops.state_obs(dw, ds))
            if t == 30:                                           # the caches do not survive a reload: recomputed, same results
                ds.load(ds.host())
        assert int(hs["episode"].max()) >= 3 and float(hs["magnitudes"][:, :2].max()) >= 0.0
    finally:
        _lib.kernel_override()


def test_batched_env_with_128_slots(crowded_town):
    """the host mirror end to end: BatchedWaypointEnv(agents_per_env = 128), birdview observation, 30 steps == the oracle"""
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv

    env = BatchedWaypointEnv(EnvConfig(seed=3, distance_cutoff=0.25), crowded_town, num_envs=12, obs_mode="birdview")
    hs = EnvState(12, 128)
    oracle.env_reset(env.tde_cfg, crowded_town, hs)
    obs = env.reset()
    assert obs.shape == (12, 3, 64, 64)
    rng = np.random.default_rng(1)
    for _ in range(30):
        act = np.stack([rng.uniform(0, 1, 12), rng.uniform(-0.1, 0.1, 12)], -1).astype(np.float32)
        hs["action"][...] = act
        oracle.env_step(env.tde_cfg, crowded_town, hs)
        obs, rew, term, trunc, info = env.step(torch.from_numpy(act).to(DEV))
        assert np.array_equal(rew.cpu().numpy().view(np.uint32), hs["reward"].view(np.uint32))
    want = oracle.render_ego(env.tde_cfg, crowded_town, hs, flags=env._rflags)
    assert np.array_equal(obs.cpu().numpy(), want)


def test_step_128_slots_above_one_residency_round():
    """tde_env_step at 128 slots and more than 1536 envs launches the form compiled for four wavefronts per SIMD (another register
    allocation of the same source): 1600 envs, a few steps with re-spawns == oracle"""
    from torchdriveenv_amd.synth import synthetic_world

    world = synthetic_world(n_scn=4, A=128, seed=9, n_maps=2)
    cfg = _abi.default_config(seed=5, distance_cutoff=0.25, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS, max_steps=3)
    B, A = 1600, 128
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV, with_obs=True)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(3)
    for t in range(7):
        act = np.stack([rng.uniform(-0.3, 1, B), rng.uniform(-0.2, 0.2, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
    assert_state_equal(hs.host(), ds.host(), "1600 envs x 128 slots, step form for large batches")
    assert torch.equal(ds["obs"], ops.state_obs(dw, ds)) and int(hs["episode"].max()) >= 3


def test_wide_step_argument_blocks_across_states_streams_and_graph_capture(crowded_town):
    """The two-role 128-slot step kernel takes its arguments from an immutable block in device memory (tde_api.hip: step_args), one per
    distinct argument set, uploaded once per stream.  Two batches stepped alternately on two streams, then a HIP graph that captured
    three launches (after a warm-up on the capturing stream: blocks in place) and is replayed, then a THIRD batch whose first step
    ever falls inside a capture (no block can be made there: the library launches the one-role kernel): all == the oracle."""
    world = crowded_town
    cfg = _abi.default_config(seed=47, distance_cutoff=0.25, max_steps=12)
    dw = world.to_device(DEV)
    A = 128
    rng = np.random.default_rng(5)
    sizes = (3, 5, 2)
    hosts = [EnvState(B, A) for B in sizes]
    devs = [EnvState(B, A, device=DEV) for B in sizes]
    for h, d in zip(hosts, devs):
        oracle.env_reset(cfg, world, h)
        ops.env_reset(cfg, dw, d)
    torch.cuda.synchronize()

    def host_step(i, act):
        hosts[i]["action"][...] = act
        oracle.env_step(cfg, world, hosts[i])

    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    for t in range(20):                                           # two states, two streams, alternating
        for i in (0, 1):
            act = np.stack([rng.uniform(-0.3, 1, sizes[i]), rng.uniform(-0.25, 0.25, sizes[i])], -1).astype(np.float32)
            host_step(i, act)
            with torch.cuda.stream(streams[(i + t) & 1]):
                devs[i]["action"].copy_(dev(act))
                ops.env_step(cfg, dw, devs[i])
            torch.cuda.synchronize()
    for i in (0, 1):
        assert_state_equal(hosts[i].host(), devs[i].host(), f"batch {i} on alternating streams")
    # a captured graph of three launches, replayed four times (the action buffer is the state's own: refilled between replays)
    side = torch.cuda.Stream(device=DEV)
    acts = [np.stack([rng.uniform(-0.3, 1, sizes[0]), rng.uniform(-0.25, 0.25, sizes[0])], -1).astype(np.float32) for _ in range(3)]
    abuf = [dev(a) for a in acts]
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(2):                                        # warm-up on the capturing stream
            host_step(0, acts[k]); ops.env_step(cfg, dw, devs[0], action=abuf[k])
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for k in range(3):
                ops.env_step(cfg, dw, devs[0], action=abuf[k])
    torch.cuda.synchronize()
    for r in range(4):
        graph.replay()
        for k in range(3):
            host_step(0, acts[k])
    torch.cuda.synchronize()
    assert_state_equal(hosts[0].host(), devs[0].host(), "captured two-role launches, replayed")
    # first use of an argument set inside a capture
    g2 = torch.cuda.CUDAGraph()
    a2 = np.stack([rng.uniform(-0.3, 1, sizes[2]), rng.uniform(-0.25, 0.25, sizes[2])], -1).astype(np.float32)
    b2 = dev(a2)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g2, stream=side):
            ops.env_step(cfg, dw, devs[2], action=b2)
    for r in range(15):
        g2.replay()
        host_step(2, a2)
    torch.cuda.synchronize()
    assert_state_equal(hosts[2].host(), devs[2].host(), "first step of a batch inside a capture")
    assert int(hosts[2]["episode"].max()) >= 2


def test_two_role_step_kernel_forced_at_a_full_residency_round(crowded_town):
    """tde_kernel_override(0, 2): the two-role step kernel at 1100 envs (above the 4 x CUs the library itself would give it) == oracle"""
    from torchdriveenv_amd import _lib

    cfg = _abi.default_config(seed=9, distance_cutoff=0.25, max_steps=4)
    B, A = 1100, 128
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = crowded_town.to_device(DEV)
    oracle.env_reset(cfg, crowded_town, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(3)
    _lib.kernel_override(step="duo")
    try:
        for t in range(9):
            act = np.stack([rng.uniform(-0.3, 1, B), rng.uniform(-0.2, 0.2, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, crowded_town, hs)
            ops.env_step(cfg, dw, ds, action=dev(act))
    finally:
        _lib.kernel_override()
    assert_state_equal(hs.host(), ds.host(), "1100 envs x 128 slots, two roles")
    assert int(hs["episode"].max()) >= 3


@pytest.mark.parametrize("seed", range(8))
def test_rollout_fuzz_128_slots(seed, crowded_town):
    """the two-role 128-slot rollout kernel - eight wavefronts per env (what the library picks for these batches: sweep and offroad
    helpers beside drivers and judges), four (tde_kernel_override(2, 0)) and the one-role kernel (1, 0), by seed - over random flags
    (lights, no auto-reset, no offroad, no reward), launch lengths and episode lengths, consecutive launches: rewards, done bits and the
    whole state equal the oracle's after every launch"""
    from torchdriveenv_amd import _lib
    from torchdriveenv_amd.synth import synthetic_world

    rng = np.random.default_rng(700 + seed)
    world = crowded_town if seed % 2 else synthetic_world(n_scn=4, A=128, seed=20 + seed, n_maps=2)
    flags = _abi.F_ALL
    if rng.random() < 0.5:
        flags |= _abi.F_TRAFFIC_LIGHTS
    if rng.random() < 0.3:
        flags &= ~_abi.F_AUTORESET
    if rng.random() < 0.25:
        flags &= ~_abi.F_OFFROAD
    if seed == 5:
        flags &= ~_abi.F_REWARD
    cfg = _abi.default_config(seed=seed, distance_cutoff=0.25, flags=flags, max_steps=int(rng.choice([1, 2, 7, 40])))
    B, A = int(rng.integers(1, 14)), 128
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    form = [None, "duo", None, "duo", None, "duo", "solo", None][seed]
    _lib.kernel_override(rollout=form)
    try:
        for launch in range(3):
            K = int(rng.choice([1, 2, 9, 33]))
            actions = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
            hr, hd = oracle.env_rollout(cfg, world, hs, actions)
            dr, dd = ops.env_rollout(cfg, dw, ds, dev(actions))
            tag = f"seed {seed} form {form} B={B} K={K} flags={flags:#x} max_steps={cfg.max_steps} launch {launch}"
            assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32)), tag
            assert np.array_equal(dd.cpu().numpy(), hd), tag
            assert_state_equal(hs.host(), ds.host(), tag)
    finally:
        _lib.kernel_override()
