"""Randomised shapes for the persistent kernels (`-m gpu`): agents per env, launch lengths from 1 step to several reward
windows, feature flags, short episodes (re-spawns at consecutive steps), consecutive launches - rewards, done bits and the
whole state equal the oracle's after every launch."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle  # noqa: E402
from torchdriveenv_amd import _abi, ops  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402
from tests.test_gpu_parity import assert_state_equal, dev  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDE_FUZZ_CASES", "12"))))
def test_rollout_fuzz(seed):
    from torchdriveenv_amd.synth import synthetic_world

    rng = np.random.default_rng(1000 + seed)
    A = int(rng.choice([8, 16, 16, 16, 32]))
    world = synthetic_world(n_scn=5, A=A, seed=200 + seed, n_maps=2)
    flags = _abi.F_ALL
    if rng.random() < 0.3:
        flags |= _abi.F_TRAFFIC_LIGHTS
    if rng.random() < 0.25:
        flags &= ~_abi.F_AUTORESET
    if rng.random() < 0.2:
        flags &= ~_abi.F_OFFROAD
    cfg = _abi.default_config(seed=seed, distance_cutoff=0.25, flags=flags, max_steps=int(rng.choice([1, 2, 3, 7, 40, 200])))
    B = int(rng.integers(1, 60)) if A < 32 else int(rng.integers(1, 20))
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    for launch in range(3):
        K = int(rng.choice([1, 2, A - 1, A, A + 1, 2 * A + 3, 50]))
        actions = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
        hr, hd = oracle.env_rollout(cfg, world, hs, actions)
        dr, dd = ops.env_rollout(cfg, dw, ds, dev(actions))
        tag = f"seed {seed} A={A} B={B} K={K} flags={flags:#x} max_steps={cfg.max_steps} launch {launch}"
        assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32)), tag
        assert np.array_equal(dd.cpu().numpy(), hd), tag
        assert_state_equal(hs.host(), ds.host(), tag)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDE_FUZZ_CASES", "8"))))
def test_step_fuzz(seed):
    """the closed-loop kernels (one launch per step; with the lookup caches: three roles, without: one) over random shapes,
    flags and episode lengths, compared with the oracle after EVERY step"""
    from torchdriveenv_amd.synth import synthetic_world

    rng = np.random.default_rng(5000 + seed)
    A = int(rng.choice([4, 8, 16, 16, 32]))
    world = synthetic_world(n_scn=5, A=A, seed=300 + seed, n_maps=2)
    flags = _abi.F_ALL
    if rng.random() < 0.3:
        flags |= _abi.F_TRAFFIC_LIGHTS
    if rng.random() < 0.25:
        flags &= ~_abi.F_AUTORESET
    cfg = _abi.default_config(seed=seed, distance_cutoff=0.25, flags=flags, max_steps=int(rng.choice([1, 2, 5, 30, 200])))
    B = int(rng.integers(1, 50)) if A < 32 else int(rng.integers(1, 16))
    hs = EnvState(B, A)
    ds = EnvState(B, A, device=DEV, with_cache=bool(rng.random() < 0.7))
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    for t in range(45):
        act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
        if t % 5 == 4 or t < 3:
            assert_state_equal(hs.host(), ds.host(), f"seed {seed} A={A} B={B} flags={flags:#x} max_steps={cfg.max_steps} step {t}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("TDE_FUZZ_CASES", "8"))))
def test_render_and_streams_fuzz(seed):
    """the rasteriser and tde_env_step_render over random shapes: image sizes (multiples of 4, padded planes, non-square),
    field of view, render flags, traffic lights, agents per env, forced step kernel, stream counts - pixels equal the oracle's,
    state equal the oracle's, after steps taken through the stream form"""
    from torchdriveenv_amd import _lib
    from torchdriveenv_amd.synth import synthetic_world

    rng = np.random.default_rng(9000 + seed)
    A = int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128]))            # (128: the two-role step kernel's argument blocks, one per sub-batch and stream)
    from torchdriveenv_amd.world import effective_offroad_distance

    squared = bool(rng.random() < 0.3)              # (the grid index is built for the effective distance of the reading)
    world = synthetic_world(n_scn=5, A=A, seed=400 + seed, n_maps=2, threshold=effective_offroad_distance(0.5, squared))
    flags = _abi.F_ALL
    if rng.random() < 0.4:
        flags |= _abi.F_TRAFFIC_LIGHTS
    cfg = _abi.default_config(seed=seed, distance_cutoff=0.25, flags=flags, max_steps=int(rng.choice([3, 10, 60])),
                              offroad_threshold_squared=int(squared))
    B = int(rng.integers(1, 40)) if A <= 16 else int(rng.integers(1, 12))
    H, W = [(64, 64), (64, 64), (32, 32), (36, 36), (60, 64), (8, 256), (64, 32), (16, 128), (4, 4)][int(rng.integers(0, 9))]
    fov = float(rng.choice([35.0, 20.0, 70.0]))
    rflags = int(rng.choice([0, _abi.RENDER_LEFT_HANDED, _abi.RENDER_PLAIN_EGO, _abi.RENDER_LEFT_HANDED | _abi.RENDER_PLAIN_EGO]))
    n_streams = int(rng.choice([1, 2, 3, 5]))
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV, with_cache=bool(rng.random() < 0.7))
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(n_streams)]
    img = torch.zeros((B, 3, H, W), dtype=torch.uint8, device=DEV)
    tag = f"seed {seed} A={A} B={B} {H}x{W} fov={fov} rflags={rflags} flags={flags:#x} streams={n_streams}"
    try:
        _lib.kernel_override(step=[None, "solo", "trio", "duo"][int(rng.integers(0, 4))])
        for t in range(24):
            act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
            hs["action"][...] = act
            oracle.env_step(cfg, world, hs)
            ops.fork_streams(streams, DEV)
            ops.env_step_render(cfg, dw, ds, streams, action=dev(act), out=img, H=H, W=W, fov=fov, flags=rflags)
            ops.join_streams(streams, DEV)
            if t % 6 == 5 or t < 2:
                want = oracle.render_ego(cfg, world, hs, H=H, W=W, fov=fov, flags=rflags)
                got = img.cpu().numpy()
                assert np.array_equal(got, want), (tag, t, int((got != want).sum()))
                assert_state_equal(hs.host(), ds.host(), f"{tag} step {t}")
    finally:
        _lib.kernel_override()
