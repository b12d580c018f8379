"""BatchedWaypointEnv.step with info_magnitudes (step + tde_ego_infractions + masked reset) against the one-launch step: us per call"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd.config import EnvConfig
from torchdriveenv_amd.env import BatchedWaypointEnv
from torchdriveenv_amd.synth import synthetic_world

B, A = 8192, 16
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, lights=False)
g = torch.Generator().manual_seed(0)
rows = list(torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to("cuda:0"))
for mode in ("state", "birdview"):
    for mag in (False, True):
        env = BatchedWaypointEnv(EnvConfig(seed=3, distance_cutoff=0.25), world, num_envs=B, obs_mode=mode, with_info=True, info_magnitudes=mag)
        env.reset()
        best = 1e9
        for rep in range(3):
            for i in range(300): env.step(rows[i % 250])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(1000): env.step(rows[i % 250])
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
        print(f"{mode:9s} info_magnitudes={mag}: {best:.2f} us per step", flush=True)
