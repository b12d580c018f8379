"""a few 250-step rollouts of 1024 envs x 128 agent slots (~122 present) for rocprofv3 passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A, K = 128, 250
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=32, A=A, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
for _ in range(3):
    ops.env_rollout(cfg, dw, st, actions)
torch.cuda.synchronize()
print("done")
