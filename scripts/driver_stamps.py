import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
B = int(sys.argv[1]); A, K = 16, 250
dev = torch.device("cuda:0"); lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4); dw = world.to_device(dev)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False); ops.env_reset(cfg, dw, st)
reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
for _ in range(3): ops.env_rollout(cfg, dw, st, actions, reward, done)
torch.cuda.synchronize()
out = (C.c_ulonglong * 24)(); lib.tde_debug_stamps(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): ops.env_rollout(cfg, dw, st, actions, reward, done)
e1.record(); torch.cuda.synchronize()
lib.tde_debug_stamps(out, 0)
n = (B * A // 64) * 4 * K
print(B, "envs", e0.elapsed_time(e1) * 1e3 / (4 * K), "us/step; driver: work", out[0] / n, "wait A", out[1] / n, "fixup+commit", out[2] / n, "wait B", out[3] / n)
