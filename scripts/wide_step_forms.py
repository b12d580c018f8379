"""closed-loop step at 128 agent slots per env, us per step by batch size: the two-role kernel (env_step_wide_kernel) vs the one-role one"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town

dev = torch.device("cuda:0")
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
world = synthetic_town(n_scn=32, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
dw = world.to_device(dev)
for B in (1, 16, 64, 256, 512, 768, 1024, 1536, 2048):
    g = torch.Generator().manual_seed(0)
    actions = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    rows = [actions[i] for i in range(250)]
    out = []
    for form in ("solo", "duo", None):
        _lib.kernel_override(step=form)
        st = EnvState(B, 128, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        for i in range(300): ops.env_step(cfg, dw, st, action=rows[i % 250])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(1000): ops.env_step(cfg, dw, st, action=rows[i % 250])
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    _lib.kernel_override()
    print(f"B={B:5d} x 128 slots: one role {out[0]:7.2f} us, two roles {out[1]:7.2f} us, the library's choice {out[2]:7.2f} us per step", flush=True)
