"""GPU parity on a TOWN-scale drivable mesh: the reference's operating size for the offroad mesh (a CARLA town handed to
Simulator(road_mesh=...), ref gym_env.py:184, 260, 312; SURVEY R10: 1e4 - 1e5 triangles) instead of the ~200-triangle
junction maps of the other tests.  One 1 km x 1 km map, 5.7e4 triangles, 100 junctions, 256 scenarios spread over it
(synth.synthetic_town); the index comes from the library's host builder (tde_grid_build: shared candidate lists, per-map
record base).  The oracle still passes over EVERY triangle of the map per corner / pixel (tde_oracle_point_near_mesh)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from tests.test_gpu_parity import assert_state_equal, dev, random_agents  # noqa: E402
from torchdriveenv_amd import _abi, _lib, ops  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402

DEV = "cuda:0"


def test_town_is_town_sized(town):
    m = town.arrays["maps"][0]
    assert town.ints["n_maps"] == 1 and town.ints["n_scn"] == 256
    assert m["n_tri"] >= 50_000
    assert m["nx"] * m["cell"] >= 1000.0 and m["ny"] * m["cell"] >= 1000.0          # >= 1 km x 1 km
    cls = town.arrays["cell_word"] & 3
    assert (cls == _abi.CELL_MIXED).sum() > 300_000 and (cls == _abi.CELL_FULL).sum() > 2_000_000
    # scenarios are spread: their first waypoints cover most of the town
    w0 = town.arrays["wp_xy"][:, 0]
    assert np.ptp(w0[:, 0]) > 600 and np.ptp(w0[:, 1]) > 600


def test_town_offroad_operator_bit_exact(town):
    """compute_offroad on poses scattered around the town's road edges (both sides of the threshold) == brute force"""
    w = town
    dw = w.to_device(DEV)
    rng = np.random.default_rng(3)
    B, A = 256, 16
    n = B * A
    m = w.arrays["maps"][0]
    tri = w.arrays["tri"][m["tri_base"]:m["tri_base"] + m["n_tri"]].reshape(-1, 3, 2).astype(np.float64)
    k = rng.integers(len(tri), size=n)
    p = tri[k, rng.integers(3, size=n)] + rng.normal(0, 1.5, (n, 2))
    ag = random_agents(rng, B, A)
    x, y = p[:, 0].astype(np.float32), p[:, 1].astype(np.float32)
    psi = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    moe = np.zeros(B, np.int32)
    want = oracle.compute_offroad(B, A, x, y, psi, ag["length"], ag["width"], ag["present"], w, moe)
    got = ops.compute_offroad(B, A, dev(x), dev(y), dev(psi), dev(ag["length"]), dev(ag["width"]), dev(ag["present"]),
                              dw, dev(moe)).cpu().numpy()
    assert np.array_equal(got, want)
    assert 0.2 < want.mean() < 0.9


def test_town_closed_loop_200_steps_and_birdviews_vs_oracle(town):
    """town world, B >= 64, A = 16, 200 steps + birdviews == the oracle's brute force over 5.7e4
    triangles: every state array bit-identical along the way, every pixel of every view at five points of the episode"""
    cfg = _abi.default_config(seed=31, distance_cutoff=0.25)
    B, A = 96, 16
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = town.to_device(DEV)
    oracle.env_reset(cfg, town, hs)
    ops.env_reset(cfg, dw, ds)
    assert_state_equal(hs.host(), ds.host(), "town reset")
    assert len(np.unique(hs["scn"])) > 60                       # the batch is spread over the town's scenarios
    rng = np.random.default_rng(4)
    n_off = n_col = n_done = 0
    for t in range(200):
        act = np.stack([rng.uniform(-0.5, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        if t % 40 < 25:
            act[:, 1] *= 0.1                                    # stretches of near-straight driving: waypoints get reached
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg, town, hs)
        ops.env_step(cfg, dw, ds)
        n_off += int(hs["offroad"].sum()); n_col += int(hs["collided"].sum())
        n_done += int((hs["terminated"] | hs["truncated"]).sum())
        if t % 20 == 0 or t >= 197:
            assert_state_equal(hs.host(), ds.host(), f"town step {t}")
        if t in (0, 37, 90, 150, 199):
            want = oracle.render_ego(cfg, town, hs)
            got = ops.render_ego(cfg, dw, ds).cpu().numpy()
            assert np.array_equal(got, want), f"{(got != want).sum()} birdview pixels differ at step {t}"
            road = (want[:, 0] == 128).mean()
            assert 0.05 < road < 0.8
    assert n_off > 0 and n_col > 0 and n_done > B // 2 and hs["reached"].max() >= 1 and hs["episode"].max() > 1


@pytest.mark.parametrize("team", ["solo", "duo", "trio"])
def test_town_rollout_every_kernel_form(town, team):
    """the persistent rollout kernels on the town: rewards, done bits and the whole state equal the oracle's"""
    cfg = _abi.default_config(seed=32, distance_cutoff=0.25)
    B, A, K = 128, 16, 120
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = town.to_device(DEV)
    oracle.env_reset(cfg, town, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(6)
    actions = np.stack([rng.uniform(-0.3, 1, (K, B)), rng.uniform(-0.25, 0.25, (K, B))], -1).astype(np.float32)
    hr, hd = oracle.env_rollout(cfg, town, hs, actions)
    _lib.kernel_override(rollout=team)
    try:
        dr, dd = ops.env_rollout(cfg, dw, ds, dev(actions))
        torch.cuda.synchronize()
    finally:
        _lib.kernel_override()
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32))
    assert np.array_equal(dd.cpu().numpy(), hd)
    assert_state_equal(hs.host(), ds.host(), f"town rollout ({team})")
    assert (hd & 1).any() and (hd & 4).any()


@pytest.mark.parametrize("step_team", ["solo", "trio"])
def test_town_one_step_kernel_forms_with_caches(town, step_team):
    cfg = _abi.default_config(seed=33, distance_cutoff=0.25)
    B, A = 128, 16
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = town.to_device(DEV)
    oracle.env_reset(cfg, town, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(8)
    _lib.kernel_override(step=step_team)
    try:
        for t in range(80):
            act = np.stack([rng.uniform(-0.3, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            hs["action"][...] = act
            ds["action"].copy_(dev(act))
            oracle.env_step(cfg, town, hs)
            ops.env_step(cfg, dw, ds)
        torch.cuda.synchronize()
    finally:
        _lib.kernel_override()
    assert_state_equal(hs.host(), ds.host(), f"town one-step kernel ({step_team})")


def test_town_full_size_8192x16_subset_vs_oracle(town):
    """BASELINE configs[2] size on the town: envs [0, 256) of the 8192-env batch == a 256-env oracle run, bit for bit"""
    cfg = _abi.default_config(seed=77, distance_cutoff=0.25)
    B, A, K, SUB = 8192, 16, 210, 256
    dw = town.to_device(DEV)
    rng = np.random.default_rng(1)
    actions = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
    ds = EnvState(B, A, device=DEV)
    ops.env_reset(cfg, dw, ds)
    r, d = ops.env_rollout(cfg, dw, ds, dev(actions))
    r, d, full = r.cpu().numpy(), d.cpu().numpy(), ds.host()
    hs = EnvState(SUB, A)
    oracle.env_reset(cfg, town, hs)
    hr, hd = oracle.env_rollout(cfg, town, hs, np.ascontiguousarray(actions[:, :SUB]))
    assert np.array_equal(r[:, :SUB].view(np.uint32), hr.view(np.uint32))
    assert np.array_equal(d[:, :SUB], hd)
    for k, a in hs.host().items():
        if k == "action":
            continue
        assert np.array_equal(np.ascontiguousarray(full[k][:a.shape[0]]).view(np.uint8), a.view(np.uint8)), k
    assert (d & 2).sum() > 0 and (d & 1).sum() > 0 and np.isfinite(r).all()
    assert len(np.unique(full["scn"])) == 256                   # every scenario of the town is in play


def test_town_config5_shape_8192x32_views_vs_oracle():
    """BASELINE configs[4] shape on the town (8192 envs x 32 agents + 64 x 64 birdview, sub-batches on three streams): the
    first 48 envs' state and pixels after 12 timesteps == the oracle's"""
    from torchdriveenv_amd.synth import synthetic_town

    world = synthetic_town(n_scn=256, A=32, seed=1)
    cfg = _abi.default_config(seed=55, distance_cutoff=0.25)
    B, A, SUB, T = 8192, 32, 48, 12
    dw = world.to_device(DEV)
    ds = EnvState(B, A, device=DEV)
    ops.env_reset(cfg, dw, ds)
    hs = EnvState(SUB, A)
    oracle.env_reset(cfg, world, hs)
    img = torch.zeros(B, 3, 64, 64, dtype=torch.uint8, device=DEV)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    rng = np.random.default_rng(2)
    for t in range(T):
        act = np.stack([rng.uniform(-0.5, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        a = dev(act)
        ops.fork_streams(streams, torch.device(DEV))
        ops.env_step_render(cfg, dw, ds, streams, action=a, out=img)
        ops.join_streams(streams, torch.device(DEV))
        hs["action"][...] = act[:SUB]
        oracle.env_step(cfg, world, hs)
    torch.cuda.synchronize()
    full = ds.host()
    for k, a in hs.host().items():
        if k in ("action", "info"):
            continue
        assert np.array_equal(np.ascontiguousarray(full[k][:a.shape[0]]).view(np.uint8), a.view(np.uint8)), k
    want = oracle.render_ego(cfg, world, hs)
    got = img[:SUB].cpu().numpy()
    assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ"


@pytest.mark.parametrize("team", ["trio", "duo"])
def test_town_with_signalised_junctions_rollout_and_step(team):
    """a town with signalised junctions - every one a light group of its own, i.e. a map descriptor that shares the town's grid
    (assemble_world) - next to scenarios without lights: the LARGE_GRID forms of the kernels WITH the traffic-light code (red
    stop lines as standing leaders, the ego's violation term, the lights in the birdview) == oracle"""
    from torchdriveenv_amd.synth import synthetic_town

    world = synthetic_town(n_scn=6, A=16, seed=3, n_streets=10, n_signals=4)          # full size: TDE_WORLD_LARGE_GRID is set
    maps = world.arrays["maps"]
    assert world.ints["hints"] & _abi.WORLD_LARGE_GRID and world.has_lights and world.ints["n_maps"] == 6
    assert maps[0]["n_stop"] == 0 and all(maps[k]["n_stop"] >= 4 and maps[k]["cell_base"] == maps[0]["cell_base"] for k in range(1, 6))
    # (scenario 4: no signal in reach; scenario 5: its own junction is plain, two of its neighbours are signalised)
    assert list(world.arrays["scn"]["map"]) == [1, 2, 3, 4, 0, 5] and maps[5]["n_stop"] == 8
    flags = _abi.F_ALL | _abi.F_TRAFFIC_LIGHTS
    cfg = _abi.default_config(seed=61, distance_cutoff=0.25, flags=flags, max_steps=170)
    B, A, K = 96, 16, 180
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(12)
    actions = np.stack([rng.uniform(0.0, 0.8, (K, B)), rng.uniform(-0.03, 0.03, (K, B))], -1).astype(np.float32)
    hr, hd = oracle.env_rollout(cfg, world, hs, actions)
    _lib.kernel_override(rollout=team)
    try:
        dr, dd = ops.env_rollout(cfg, dw, ds, dev(actions))
        torch.cuda.synchronize()
    finally:
        _lib.kernel_override()
    assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32)) and np.array_equal(dd.cpu().numpy(), hd)
    assert_state_equal(hs.host(), ds.host(), f"town with lights, rollout ({team})")
    assert (hd & 16).any()                                            # some ego ran a red light on the way
    for t in range(25):                                               # closed loop on from there (three-role step kernel with lights)
        act = np.stack([rng.uniform(0.0, 0.8, B), rng.uniform(-0.03, 0.03, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
    assert_state_equal(hs.host(), ds.host(), "town with lights, closed loop")
    want = oracle.render_ego(cfg, world, hs)
    got = ops.render_ego(cfg, dw, ds).cpu().numpy()
    assert np.array_equal(got, want)
    assert ((want[:, 0] == 255) & (want[:, 1] == 0) & (want[:, 2] == 0)).any() or ((want[:, 0] == 0) & (want[:, 1] == 255)).any()
