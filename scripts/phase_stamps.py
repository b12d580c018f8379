"""Per-phase wavefront time of the two-role rollout kernel from s_memtime stamps.
usage: python scripts/make_stamped_build.py duo && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py [envs]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A, K = 16, 250
dev = torch.device("cuda:0")
lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1)
actions = actions.float().contiguous().to(dev)
reward = torch.empty((K, B), device=dev)
done = torch.empty((K, B), dtype=torch.uint8, device=dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
for _ in range(4):
    ops.env_rollout(cfg, dw, st, actions, reward, done)
torch.cuda.synchronize()
out = (C.c_ulonglong * 24)()
lib.tde_debug_stamps(out, 1)
reps = 4
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.env_rollout(cfg, dw, st, actions, reward, done)
e1.record()
torch.cuda.synchronize()
lib.tde_debug_stamps(out, 0)
waves = B * A // 64
drive = ["controller + bicycle + sincos (spec)", "reward arithmetic (ego lane)", "wait A (done of prev step)",
         "re-spawn fixup + commit rows", "wait B"]
judge = ["wait A", "wait B (rows of this step)", "read rows + collision", "offroad resolve", "stop lines + flags",
         "publish + reset"]
us = e0.elapsed_time(e1) * 1e3 / (reps * K)
n = waves * reps * K
print(f"{B} envs: {us:.2f} us/step (stamped build)")
for title, names, off in (("drive wavefront", drive, 0), ("judge wavefront", judge, 12)):
    tot = sum(out[off:off + 12])
    print(f" {title}: {tot / n:.0f} memtime ticks per wave-step")
    for i, nm in enumerate(names):
        v = out[off + i]
        print(f"   {nm:34s} {v / n:8.1f} ticks  {100.0 * v / tot:5.1f} %")
