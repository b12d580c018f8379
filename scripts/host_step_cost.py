"""host-side cost of one BatchedWaypointEnv.step (64 envs: the kernel is shorter than the host path, so wall time per call IS the host path)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd.config import EnvConfig
from torchdriveenv_amd.env import BatchedWaypointEnv
from torchdriveenv_amd.synth import synthetic_world

world = synthetic_world(n_scn=8, A=16, seed=0, n_maps=2)
for B in (64, 8192):
    env = BatchedWaypointEnv(EnvConfig(seed=3), world, num_envs=B, obs_mode="state", with_info=True)
    env.reset()
    act = torch.zeros(B, 2, device=env.torch_device)
    fl = int(env.tde_cfg.flags)

    def timed(fn, n=20000):
        for _ in range(100): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    print(f"B={B}: handle.step {timed(lambda: env._h.step(act, fl)):.2f} us, env.step {timed(lambda: env.step(act)):.2f} us, "
          f"empty lambda {timed(lambda: None):.2f} us", flush=True)
env = BatchedWaypointEnv(EnvConfig(seed=3), world, num_envs=64, obs_mode="state", with_info=True)
env.reset()
act = torch.zeros(64, 2, device=env.torch_device)
pr = cProfile.Profile()
pr.enable()
for _ in range(20000): env.step(act)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
