"""closed-loop step at 8192 x 16: what each optional output costs (info terms, done bits + episode statistics, compact observation)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A = 8192, 16
dev = torch.device("cuda:0")
_lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
rows = [actions[i] for i in range(250)]
for info, obs, epi in ((False, False, False), (True, False, False), (False, True, False), (False, False, True), (True, False, True), (True, True, True)):
    st = EnvState(B, A, device=dev, with_info=info, with_obs=obs, with_episode=epi)
    ops.env_reset(cfg, dw, st)
    h = _ext.env_handle(cfg, dw, st)
    fl = int(cfg.flags)
    for i in range(1000): h.step(rows[i % 250], fl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(4000): h.step(rows[i % 250], fl)
    e1.record(); torch.cuda.synchronize()
    print(f"info={int(info)} obs={int(obs)} episode_stats={int(epi)}: {e0.elapsed_time(e1) * 1e3 / 4000:.2f} us per step", flush=True)
