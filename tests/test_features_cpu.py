"""CPU-side coverage of the round-2 additions: the oracle's stop-line layer / left-handed raster / masked and fresh
frame-stack semantics, both readings of the offroad threshold, Monitor-style episode statistics, config validation,
the World's recorded grid threshold, the lazy per-env infos, and the SB3 VecEnv contract of WaypointVecEnv."""
import inspect

import numpy as np
import pytest

from oracle import oracle
from torchdriveenv_amd import _abi
from torchdriveenv_amd.config import EnvConfig, SimulatorConfig, RendererConfig, render_flags, validate
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.world import World, check_threshold, effective_offroad_distance


def _stepped(world, cfg, B=12, A=16, steps=30, seed=0):
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    rng = np.random.default_rng(seed)
    for _ in range(steps):
        hs["action"][...] = np.stack([rng.uniform(0, 1, B), rng.uniform(-0.1, 0.1, B)], -1).astype(np.float32)
        oracle.env_step(cfg, world, hs)
    return hs


def test_oracle_raster_stop_lines_follow_the_light_state(small_world):
    """with TDE_F_TRAFFIC_LIGHTS the stop lines of the map are painted over the road, red or green by their light at
    the env's current step; without the flag the image has none of these colours"""
    assert small_world.has_lights
    cfg = _abi.default_config(seed=2, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS, terminated_at_infraction=0)
    hs = _stepped(small_world, cfg)
    # put every ego on the first stop line of its map so that the line is in view
    maps, stops = small_world.arrays["maps"], small_world.arrays["stoplines"]
    scn_map = small_world.arrays["scn"]["map"]
    for e in range(hs.B):
        m = maps[scn_map[hs["scn"][e]]]
        sl = stops[m["stop_base"]]
        g = e * hs.A
        hs["x"][g], hs["y"][g] = sl["x"] - 6.0 * sl["c"], sl["y"] - 6.0 * sl["s"]
        hs["psi"][g] = np.arctan2(sl["s"], sl["c"])
    RED, GO = np.array(_abi.PALETTE[_abi.LAYER_STOP_RED]), np.array(_abi.PALETTE[_abi.LAYER_STOP_GO])
    seen = set()
    for k in (1, 40, 80, 120, 160):
        hs["steps"][:] = k
        img = oracle.render_ego(cfg, small_world, hs).transpose(0, 2, 3, 1).reshape(hs.B, -1, 3)
        for e in range(hs.B):
            m = maps[scn_map[hs["scn"][e]]]
            phases = small_world.arrays["phases"][m["phase_base"]:m["phase_base"] + m["n_phase"]]
            t = k % m["cycle_steps"]
            red = next(int(p["red_mask"]) for p in phases if t < p["end_step"])
            light = int(stops[m["stop_base"]]["light"])
            want = RED if (red >> light) & 1 else GO
            n_want = (img[e] == want).all(1).sum()
            assert n_want > 0, (e, k)
            seen.add(bool((red >> light) & 1))
    assert seen == {True, False}                       # both states were exercised
    plain = _abi.default_config(seed=2, flags=_abi.F_ALL, terminated_at_infraction=0)
    img = oracle.render_ego(plain, small_world, hs).transpose(0, 2, 3, 1).reshape(-1, 3)
    assert not (img == RED).all(1).any() and not (img == GO).all(1).any()


def test_oracle_left_handed_raster_is_the_mirror_image(small_world):
    cfg = _abi.default_config(seed=3, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS)
    hs = _stepped(small_world, cfg, steps=12)
    right = oracle.render_ego(cfg, small_world, hs)
    left = oracle.render_ego(cfg, small_world, hs, flags=_abi.RENDER_LEFT_HANDED)
    assert np.array_equal(left, right[..., ::-1]) and not np.array_equal(left, right)
    plain = oracle.render_ego(cfg, small_world, hs, flags=_abi.RENDER_PLAIN_EGO)
    assert (right[:, :, 32, 32] == np.array(_abi.PALETTE[4])).all()            # ego colour at the centre
    assert (plain[:, :, 32, 32] == np.array(_abi.PALETTE[3])).all()                              # painted as an NPC


def test_oracle_frame_stack_fresh_and_masked_calls(small_world):
    cfg = _abi.default_config(seed=4)
    hs = _stepped(small_world, cfg, B=6, steps=5)
    out = None
    frames = []
    for _ in range(3):
        hs["action"][...] = 0.3
        oracle.env_step(cfg, small_world, hs)
        out = oracle.render_ego(cfg, small_world, hs, n_stack=3, out=out)
        frames.append(oracle.render_ego(cfg, small_world, hs))
    assert np.array_equal(out, np.concatenate(frames, 1))
    # masked call: views 1 and 4 were re-spawned -> newest frame re-rendered in place, older frames blank; others untouched
    before = out.copy()
    mask = np.zeros(hs.B, np.uint8)
    mask[[1, 4]] = 1
    oracle.env_reset(cfg, small_world, hs, mask)
    out = oracle.render_ego(cfg, small_world, hs, n_stack=3, out=out, fresh=mask, only=mask)
    single = oracle.render_ego(cfg, small_world, hs)
    for e in range(hs.B):
        if mask[e]:
            assert not out[e, :6].any() and np.array_equal(out[e, 6:], single[e])
        else:
            assert np.array_equal(out[e], before[e])
    # fresh given as done bits (bit 0 / 1 set) on a full call: stack shifted, then the older frames of those views blanked
    bits = np.zeros(hs.B, np.uint8)
    bits[2] = 2
    bits[3] = 4 | 8                                     # infraction bits without done: NOT fresh
    prev = out.copy()
    out = oracle.render_ego(cfg, small_world, hs, n_stack=3, out=out, fresh=bits)
    assert not out[2, :6].any() and np.array_equal(out[3, :6], prev[3, 3:])


def test_offroad_threshold_both_readings():
    """a corner 0.6 m from the mesh: off the road when 0.5 bounds the distance, on it when 0.5 bounds the squared distance
    (0.36); a corner 0.75 m away (0.5625) is off the road under both"""
    from torchdriveenv_amd.world import assemble_world

    tri = np.array([[[0, 0], [40, 0], [40, 10]], [[0, 0], [40, 10], [0, 10]]], np.float64)     # a 40 x 10 m slab
    scn = [dict(map=0, waypoints=[(2.0, 5.0), (30.0, 5.0)], start_heading=0.0, agents=[])]
    for squared, thr_eff in ((False, 0.5), (True, float(np.sqrt(0.5)))):
        w = assemble_world([tri], scn, 1, threshold=thr_eff)
        assert abs(w.threshold - thr_eff) < 1e-12
        x = np.array([10.0, 10.0], np.float32)
        # box 4 x 2 m heading +x: its left corners are at y + 1
        y = np.array([10.0 + 0.6 - 1.0, 10.0 + 0.75 - 1.0], np.float32)
        psi = np.zeros(2, np.float32)
        L, W = np.full(2, 4.0, np.float32), np.full(2, 2.0, np.float32)
        thr2 = 0.5 if squared else 0.25
        got = oracle.compute_offroad(2, 1, x, y, psi, L, W, np.ones(2, np.uint8), w, np.zeros(2, np.int32),
                                     threshold=float(np.sqrt(thr2)))
        assert list(got) == ([0, 1] if squared else [1, 1])
        cfg = _abi.default_config(flags=_abi.F_OFFROAD, offroad_threshold=0.5, offroad_threshold_squared=int(squared))
        hs = EnvState(2, 1)
        oracle.env_reset(cfg, w, hs)
        hs["x"][:], hs["y"][:], hs["psi"][:], hs["v"][:] = x, y, psi, 0.0
        hs["len"][:], hs["wid"][:] = L, W
        hs["action"][...] = 0.0
        oracle.env_step(cfg, w, hs)
        assert list(hs["offroad"]) == ([0, 1] if squared else [1, 1])
        check_threshold(w, 0.5, squared)
        with pytest.raises(ValueError):
            check_threshold(w, 0.5, not squared)
    assert effective_offroad_distance(0.5, True) == pytest.approx(0.70710678)


def test_world_save_load_keeps_the_grid_threshold(small_world, tmp_path):
    p = str(tmp_path / "w.npz")
    small_world.save(p)
    w2 = World.load(p)
    assert w2.threshold == small_world.threshold == 0.5
    with pytest.raises(ValueError):
        check_threshold(w2, 0.75)


def test_config_fields_are_honoured_or_rejected():
    validate(EnvConfig())                                                         # the reference's defaults pass
    assert render_flags(EnvConfig()) == _abi.RENDER_LEFT_HANDED                  # gym_env.py:46-47
    cfg = EnvConfig(simulator=SimulatorConfig(renderer=RendererConfig(left_handed_coordinates=False,
                                                                      highlight_ego_vehicle=False),
                                              left_handed_coordinates=False))
    assert render_flags(cfg) == _abi.RENDER_PLAIN_EGO
    with pytest.raises(NotImplementedError):
        validate(EnvConfig(simulator=SimulatorConfig(collision_metric="iou")))
    with pytest.raises(NotImplementedError):
        validate(EnvConfig(render_mode="video"))
    with pytest.raises(NotImplementedError):
        validate(EnvConfig(render_mode="human"))
    with pytest.raises(NotImplementedError):
        validate(EnvConfig(simulator=SimulatorConfig(left_handed_coordinates=False)))   # differs from the renderer's
    from torchdriveenv_amd.config import to_tde_config
    c = to_tde_config(EnvConfig(simulator=SimulatorConfig(offroad_threshold_squared=True)), 1, _abi.F_ALL)
    assert c.offroad_threshold_squared == 1 and c.offroad_threshold == pytest.approx(0.5)


def test_oracle_episode_statistics_match_monitor(small_world):
    """ep_return is the float64 sum of the episode's rewards (Monitor: sum of Python floats), reported with the episode
    length at the terminal step, zeroed by the re-spawn"""
    cfg = _abi.default_config(seed=6, distance_cutoff=0.25)
    B, A = 24, 16
    hs = EnvState(B, A)
    oracle.env_reset(cfg, small_world, hs)
    rng = np.random.default_rng(1)
    acc, length, n_done = np.zeros(B), np.zeros(B, np.int64), 0
    for _ in range(260):
        hs["action"][...] = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        oracle.env_step(cfg, small_world, hs)
        acc += hs["reward"].astype(np.float64)
        length += 1
        done = (hs["terminated"] | hs["truncated"]).astype(bool)
        for e in np.nonzero(done)[0]:
            assert hs["ep_final"][e] == acc[e] and hs["ep_final_len"][e] == length[e]
            acc[e], length[e] = 0.0, 0
            n_done += 1
        assert np.array_equal(hs["ep_return"], acc)
    assert n_done > B


def test_lazy_infos_behave_like_a_list_of_dicts():
    from torchdriveenv_amd.env import LazyInfos

    cols = {"offroad": np.array([0.0, 1.0, 0.0], np.float32), "is_success": np.array([False, False, True]),
            "TimeLimit.truncated": np.array([False, False, True])}
    term = {2: {"terminal_observation": np.zeros(3), "episode": {"r": 1.5, "l": 7, "t": 0.1}}}
    infos = LazyInfos(3, cols, term)
    assert len(infos) == 3 and infos[1]["offroad"] == 1.0 and infos[-1]["episode"]["l"] == 7
    assert [i.get("episode") is not None for i in infos] == [False, False, True]
    assert isinstance(infos[0], dict) and infos[0].get("terminal_observation") is None
    assert infos[2]["TimeLimit.truncated"] is True and len(infos[0:2]) == 2
    assert infos.column("offroad") is cols["offroad"]
    with pytest.raises(IndexError):
        infos[3]
    # what SB3's wrappers do in step_wait (VecNormalize / VecTransposeImage / VecFrameStack): rewrite an entry through
    # infos[i][key] and read it back later - the same dict must come back on every access
    infos[2]["terminal_observation"] = "rewritten"
    assert infos[2]["terminal_observation"] == "rewritten" and infos[2] is infos[2]
    infos[0] = {"replaced": 1}
    assert infos[0] == {"replaced": 1} and [i.get("replaced") for i in infos] == [1, None, None]
    assert infos.columns is cols and infos.terminal is term


# The abstract interface of stable_baselines3.common.vec_env.base_vec_env.VecEnv (SB3 2.x): name -> parameter names.
# stable_baselines3 does not ship in this image, so the contract is written down here; when it is importable the class
# really subclasses it (and the abstract-method check of ABCMeta applies on instantiation).
SB3_VECENV_ABSTRACT = {
    "reset": ["self"],
    "step_async": ["self", "actions"],
    "step_wait": ["self"],
    "close": ["self"],
    "get_attr": ["self", "attr_name", "indices"],
    "set_attr": ["self", "attr_name", "value", "indices"],
    "env_method": ["self", "method_name", "method_args", "indices", "method_kwargs"],
    "env_is_wrapped": ["self", "wrapper_class", "indices"],
}
SB3_VECENV_CONCRETE = {"step": ["self", "actions"], "seed": ["self", "seed"], "get_images": ["self"],
                       "render": ["self", "mode"]}


def test_vecenv_adapter_walks_the_sb3_interface():
    from torchdriveenv_amd import env as E

    cls = E.WaypointVecEnv
    for name, params in {**SB3_VECENV_ABSTRACT, **SB3_VECENV_CONCRETE}.items():
        fn = getattr(cls, name, None)
        assert callable(fn), f"WaypointVecEnv lacks {name}"
        got = list(inspect.signature(fn).parameters)
        assert got == params, f"{name}{got} != {params}"
    try:
        from stable_baselines3.common.vec_env import VecEnv
    except Exception:
        VecEnv = None
    if VecEnv is not None:
        assert issubclass(cls, VecEnv) and not getattr(cls, "__abstractmethods__", None)
    # BatchedWaypointEnv keeps the old entry points as delegations
    for name in ("step_async", "step_wait", "vec_step", "vec_reset", "as_vec_env"):
        assert callable(getattr(E.BatchedWaypointEnv, name))


def test_lazy_terminal_mapping_builds_entries_on_demand():
    """the per-step `terminal` mapping of the VecEnv adapter (env -> terminal_observation + Monitor's episode): entries are
    formed when read, the mapping interface is what LazyInfos and ShardedBatchedEnv use (get / items / in / len)"""
    import numpy as np

    from torchdriveenv_amd.env import LazyInfos, LazyTerminal

    idx = np.array([2, 5])
    obs = np.arange(6, dtype=np.float32).reshape(2, 3)
    ep_r, ep_l = np.zeros(8), np.zeros(8, np.int32)                  # per-ENV arrays (the step's ep_final / ep_final_len outputs)
    ep_r[[2, 5]], ep_l[[2, 5]] = [1.23456789, -2.0], [7, 9]
    t = LazyTerminal(idx, lambda n: obs[n], ep_r, ep_l, 0.5)
    assert len(t) == 2 and 5 in t and 3 not in t and t.get(3) is None and sorted(t.keys()) == [2, 5]
    e = t[5]
    assert np.array_equal(e["terminal_observation"], obs[1]) and e["episode"] == {"r": -2.0, "l": 9, "t": 0.5}
    assert t.get(2)["episode"]["r"] == 1.234568                       # Monitor rounds the return to 6 digits
    assert dict(t.items()).keys() == {2, 5}
    with __import__("pytest").raises(KeyError):
        t[4]
    infos = LazyInfos(8, {"offroad": np.zeros(8, np.float32)}, t)
    assert infos[2]["episode"]["l"] == 7 and "terminal_observation" not in infos[3] and infos[2] is infos[2]
    no_stats = LazyTerminal(idx, lambda n: obs[n], None, None, 0.0)
    assert "episode" not in no_stats[2]


def test_single_agent_wrapper_transforms_mirror_the_reference():
    """SingleAgentWrapper.transform_out / transform_in (ref gym_env.py:463-481): tensors lose / gain the two leading singleton
    dimensions (and go to the CPU on the way out), dicts are mapped entry by entry, numpy arrays go through torch, anything else
    passes through - on a stand-in env (no GPU involved)"""
    import numpy as np
    import torch

    from torchdriveenv_amd.env import SingleAgentWrapper

    class Env:
        torch_device = torch.device("cpu")

    w = SingleAgentWrapper(Env())
    t = torch.arange(6.0).reshape(1, 1, 2, 3)
    out = w.transform_out({"a": t, "b": np.zeros((1, 1, 4), np.uint8), "c": 3.5, "d": {"e": torch.ones(1, 1)}})
    assert out["a"].shape == (2, 3) and out["b"].shape == (4,) and out["b"].dtype == np.uint8 and out["c"] == 3.5
    assert out["d"]["e"].dim() == 0 and out["d"]["e"].device.type == "cpu"
    back = w.transform_in({"a": out["a"], "c": 3.5})
    assert back["a"].shape == (1, 1, 2, 3) and torch.equal(back["a"], t) and back["c"] == 3.5
    assert w.torch_device == torch.device("cpu")           # attribute access falls through to the wrapped env
