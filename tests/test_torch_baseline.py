"""The batched-tensor CPU baseline (oracle/torch_step.py) against the scalar C oracle, teacher-forced step by step: same
masks away from their decision boundaries, kinematic state within the north star's 1e-5 (relative to the coordinate
scale), rewards equal up to the libm-vs-polynomial sine / cosine."""
import numpy as np

from oracle import oracle
from oracle.torch_step import TorchWorld, torch_env_step
from torchdriveenv_amd import _abi
from torchdriveenv_amd.state import EnvState


def test_batched_torch_step_tracks_the_c_oracle(small_world):
    cfg = _abi.default_config(seed=4, distance_cutoff=0.25)
    B, A = 64, 16
    a, b = EnvState(B, A), EnvState(B, A)
    oracle.env_reset(cfg, small_world, a)
    tw = TorchWorld(small_world)
    rng = np.random.default_rng(0)
    agree = {"collided": [], "offroad": [], "done": []}
    for t in range(120):
        b.load(a.host())                                     # teacher forcing: both start every step from the same state
        act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        a["action"][...] = act
        b["action"][...] = act
        keep = a.host()
        oracle.env_step(cfg, small_world, a)
        torch_env_step(cfg, small_world, tw, b, oracle_reset=oracle.env_reset)
        done = (a["terminated"] | a["truncated"]).astype(bool)
        live = np.repeat(~done, A)                           # finished envs were re-spawned: compare the others' state
        for k in ("x", "y", "psi", "v"):
            d = np.abs(a[k] - b[k])[live]
            if k == "psi":
                d = np.minimum(d, 2 * np.pi - d)
            assert d.max() <= 1e-5 * max(1.0, np.abs(keep[k]).max()), (t, k, d.max())
        assert np.array_equal(a["route_wp"][live], b["route_wp"][live]) or (a["route_wp"] != b["route_wp"]).mean() < 1e-3
        assert np.allclose(a["reward"], b["reward"], atol=2e-3)
        agree["collided"].append((a["collided"] == b["collided"])[live].mean())
        agree["offroad"].append((a["offroad"] == b["offroad"])[live].mean())
        agree["done"].append((done == (b["terminated"] | b["truncated"]).astype(bool)).mean())
    assert min(np.mean(v) for v in agree.values()) > 0.999
    assert a["episode"].max() > 1
