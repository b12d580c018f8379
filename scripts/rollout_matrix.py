"""Rollout kernel step time over (agents per env, traffic lights) - run under TDE_ROLLOUT=solo|duo|trio to compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _lib, ops
_lib.kernel_override(rollout=os.environ.get('TDE_ROLLOUT'))   # (the scripts' own switch; the library reads no environment)
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

dev = torch.device("cuda:0")
K = 250
for A, B in ((8, 16384), (16, 8192), (32, 4096), (64, 2048)):
    world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
    dw = world.to_device(dev)
    g = torch.Generator().manual_seed(0)
    actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
    for lights in (False, True):
        flags = _abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0)
        cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=flags)
        st = EnvState(B, A, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        for _ in range(2):
            ops.env_rollout(cfg, dw, st, actions, reward, done)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(4):
            ops.env_rollout(cfg, dw, st, actions, reward, done)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (4 * K)
        print(f"{os.environ.get('TDE_ROLLOUT', 'default'):8s} A={A:2d} B={B:5d} lights={int(lights)}  {us:7.2f} us/step  {B * A / us * 1e6:.3e} agent-steps/s")
