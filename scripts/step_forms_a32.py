"""closed-loop step, 32 agents per env: one-role vs three-role kernel by batch size (us per step, HIP events over 3000 launches)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

A = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
_lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
for B in (512, 1024, 2048, 2752, 4096, 8192):
    g = torch.Generator().manual_seed(0)
    actions = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    rows = [actions[i] for i in range(250)]
    for kern in ("solo", "trio"):
        _lib.kernel_override(step=kern)
        st = EnvState(B, A, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        h = _ext.env_handle(cfg, dw, st)
        fl = int(cfg.flags)
        for i in range(1000): h.step(rows[i % 250], fl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(3000): h.step(rows[i % 250], fl)
        e1.record(); torch.cuda.synchronize()
        print(f"A={A} B={B:5d} {kern}: {e0.elapsed_time(e1) * 1e3 / 3000:.2f} us per step", flush=True)
_lib.kernel_override()
