"""tde_env_post_step by parts, on the state one step without TDE_F_AUTORESET left (8192 envs x 16, ~2 % finished): us per launch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A = 8192, 16
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, lights=False)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=True, with_obs=True)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
for i in range(300): ops.env_step(cfg, dw, st, action=acts[i % 250])
na = _abi.TdeConfig.from_buffer_copy(cfg); na.flags &= ~_abi.F_AUTORESET
ops.env_step(na, dw, st, action=acts[50])
done = (st["terminated"] | st["truncated"])
print(f"finished {int(done.sum())}, ego offroad {int(st['offroad'].view(B, A)[:, 0].sum())}, ego collided {int(st['collided'].view(B, A)[:, 0].sum())}")
mag = torch.zeros(B, 4, device=dev)
def timed(fn, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
import copy
coll0, off0 = st["collided"].clone(), st["offroad"].clone()
print(f"launch only (no magnitudes, no re-spawn) {timed(lambda: ops.env_post_step(na, dw, st, None)):.2f} us")
print(f"magnitudes only                          {timed(lambda: ops.env_post_step(na, dw, st, mag)):.2f} us")
st["offroad"].zero_()
print(f"magnitudes, collision flags only         {timed(lambda: ops.env_post_step(na, dw, st, mag)):.2f} us")
st["offroad"].copy_(off0); st["collided"].zero_()
print(f"magnitudes, offroad flags only           {timed(lambda: ops.env_post_step(na, dw, st, mag)):.2f} us   (max {float(mag[:, 0].max()):.2f} m)")
st["collided"].copy_(coll0)
print(f"re-spawn only                            {timed(lambda: ops.env_post_step(cfg, dw, st, None)):.2f} us")
