"""long parity soak (one-off): closed loop and rollouts against the oracle over thousands of steps, lights on, short and long episodes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world
from tests.test_gpu_parity import assert_state_equal, dev

DEV = "cuda:0"
cases = [("junctions A=16 lights", synthetic_world(n_scn=16, A=16, seed=11, n_maps=4), 16, 96, 200),
         ("junctions A=32 lights", synthetic_world(n_scn=8, A=32, seed=12, n_maps=2), 32, 40, 60),
         ("town A=16, 6 signalised junctions", synthetic_town(n_scn=12, A=16, seed=13, n_streets=6, n_signals=6), 16, 64, 120),
         ("crowded A=128 lights", synthetic_town(n_scn=6, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4, n_signals=4), 128, 12, 90)]
for name, world, A, B, max_steps in cases:
    cfg = _abi.default_config(seed=77, distance_cutoff=0.25, flags=_abi.F_ALL | _abi.F_TRAFFIC_LIGHTS, max_steps=max_steps)
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV, with_obs=True)
    dw = world.to_device(DEV)
    oracle.env_reset(cfg, world, hs); ops.env_reset(cfg, dw, ds)
    rng = np.random.default_rng(1)
    T = 1500 if A < 128 else 400
    for t in range(T):
        act = np.stack([rng.uniform(-0.6, 1, B), rng.uniform(-0.25, 0.25, B)], -1).astype(np.float32)
        hs["action"][...] = act; ds["action"].copy_(dev(act))
        oracle.env_step(cfg, world, hs); ops.env_step(cfg, dw, ds)
        if t % 100 == 99:
            assert_state_equal(hs.host(), ds.host(), f"{name}: closed loop step {t}")
    for r in range(6 if A < 128 else 2):
        K = 250
        acts = np.stack([rng.uniform(-0.6, 1, (K, B)), rng.uniform(-0.25, 0.25, (K, B))], -1).astype(np.float32)
        hr, hd = oracle.env_rollout(cfg, world, hs, acts)
        dr, dd = ops.env_rollout(cfg, dw, ds, dev(acts))
        assert np.array_equal(dr.cpu().numpy().view(np.uint32), hr.view(np.uint32)) and np.array_equal(dd.cpu().numpy(), hd), (name, r)
        assert_state_equal(hs.host(), ds.host(), f"{name}: rollout {r}")
    print(f"{name}: {T} closed-loop steps + {(6 if A < 128 else 2) * 250} rollout steps == oracle; episodes up to {int(hs['episode'].max())}, tl violations seen {bool((hd & 16).any())}", flush=True)
