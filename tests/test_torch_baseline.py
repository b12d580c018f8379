"""The batched-tensor CPU baseline (oracle/torch_step.py) against the scalar C oracle, teacher-forced step by step: same
masks away from their decision boundaries, kinematic state within the north star's 1e-5 (relative to the coordinate
scale), rewards equal up to the libm-vs-polynomial sine / cosine."""
import numpy as np

from oracle import oracle
from oracle.torch_step import TorchWorld, torch_env_step
from torchdriveenv_amd import _abi
from torchdriveenv_amd.state import EnvState


def test_batched_torch_step_tracks_the_c_oracle(small_world):
    """the CPU twin of tests/test_gpu_torch_witness.py (oracle bits = HIP bits), held to the same bounds: state per element within
    1e-5 * max(1, |ref|), rewards within 1e-4, collision / offroad masks equal on every slot farther than 1e-4 m from its decision
    threshold (tests/witness_util.py); the slots inside the band are counted"""
    import os
    import tempfile

    from tests.witness_util import BAND, Observed, collision_margin, offroad_margin

    cfg = _abi.default_config(seed=4, distance_cutoff=0.25)
    B, A = 64, 16
    a, b = EnvState(B, A), EnvState(B, A)
    oracle.env_reset(cfg, small_world, a)
    tw = TorchWorld(small_world)
    scn_map = small_world.arrays["scn"]["map"].astype(np.int64)
    rng = np.random.default_rng(0)
    obs = Observed()
    for t in range(120):
        b.load(a.host())                                     # teacher forcing: both start every step from the same state
        act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        a["action"][...] = act
        b["action"][...] = act
        oracle.env_step(cfg, small_world, a)
        torch_env_step(cfg, small_world, tw, b, oracle_reset=oracle.env_reset)
        done = (a["terminated"] | a["truncated"]).astype(bool)
        live = np.repeat(~done, A) & (b["present"] != 0)     # finished envs were re-spawned: compare the others' state
        for k in ("x", "y", "psi", "v"):
            got, ref = a[k][live].astype(np.float64), b[k][live].astype(np.float64)
            err = np.abs(got - ref)
            if k == "psi":
                got = np.where(err > np.pi, got - np.sign(got - ref) * 2 * np.pi, got)
                err = np.abs(got - ref)
            assert (err <= 1e-5 * np.maximum(1.0, np.abs(ref))).all(), (t, k, float(err.max()))
            obs.state(got, ref)
        assert np.array_equal(a["route_wp"][live], b["route_wp"][live]) or (a["route_wp"] != b["route_wp"]).mean() < 1e-3
        assert np.allclose(a["reward"], b["reward"], rtol=0, atol=1e-4)
        obs.reward(a["reward"], b["reward"])
        obs.rec["slot_steps"] += int(live.sum())
        cm, om = collision_margin(b, B, A), offroad_margin(b, B, A, tw, scn_map, cfg.offroad_threshold)
        assert obs.mask("collided", a["collided"], b["collided"], cm, live) == 0, t
        assert obs.mask("offroad", a["offroad"], b["offroad"], om, live) == 0, t
        ego_band = np.minimum(cm, om).reshape(B, A)[:, 0] <= BAND
        assert (done == (b["terminated"] | b["truncated"]).astype(bool))[~ego_band].all(), t
    assert a["episode"].max() > 1
    assert obs.rec["collided_in_band"] + obs.rec["offroad_in_band"] < 0.001 * obs.rec["slot_steps"]
    obs.write(os.environ.get("TDE_WITNESS_OUT") or os.path.join(tempfile.gettempdir(), "tde_witness_cpu.json"),
              what="oracle/tde_oracle.c vs oracle/torch_step.py, 64 envs x 16 agents x 120 teacher-forced steps")
