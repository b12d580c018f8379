"""Runs a few K-step rollouts of the 8192x16 workload (for rocprofv3 --pmc / --kernel-trace passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world
TOWN = os.environ.get("TDE_WORLD") == "town"            # TDE_WORLD=town: the 1 km^2 town map instead of the junction maps

flags = int(sys.argv[1]) if len(sys.argv) > 1 else _abi.F_ALL
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B, A, K = 8192, 16, 250
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=256, A=A, seed=0) if TOWN else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=flags)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
for _ in range(reps):
    ops.env_rollout(cfg, dw, st, actions, reward, done)
torch.cuda.synchronize()
print("done", float(reward.sum()))
