"""us per batch step of the persistent rollout kernel vs number of envs (waves per SIMD = B*16/64/1024)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
A, K = 16, 250
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
    g = torch.Generator().manual_seed(0)
    actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    for _ in range(2): ops.env_rollout(cfg, dw, st, actions)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4): ops.env_rollout(cfg, dw, st, actions)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (4 * K)
    print(f"B={B:6d} waves/SIMD={B*A/64/1024:5.2f}  {us:7.2f} us/step  {B*A/us*1e6:.3e} agent-steps/s")
