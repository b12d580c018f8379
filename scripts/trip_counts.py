import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
B, A, K = 8192, 16, 250
dev = torch.device("cuda:0"); lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4); dw = world.to_device(dev)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False); ops.env_reset(cfg, dw, st)
reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
ops.env_rollout(cfg, dw, st, actions, reward, done); torch.cuda.synchronize()
out = (C.c_ulonglong * 24)(); lib.tde_debug_stamps(out, 1)
ops.env_rollout(cfg, dw, st, actions, reward, done); torch.cuda.synchronize()
lib.tde_debug_stamps(out, 0)
# usage: python scripts/make_stamped_build.py trips && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/trip_counts.py
print("npc: sweeps", out[1], "exact trips per sweep", out[0] / out[1], "cand per lane-sweep", out[3] / (64 * out[1]))
print("coll: sweeps", out[5], "exact trips per sweep", out[4] / out[5], "cand per lane-sweep", out[6] / (64 * out[5]))
