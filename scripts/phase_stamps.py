"""Per-phase wavefront time of the two- / three-role rollout kernels from s_memtime stamps.
usage: python scripts/make_stamped_build.py duo|trio && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py [duo|trio] [envs]
(duo: run under TDE_ROLLOUT=duo)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

MODE = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "duo"
_num = [a for a in sys.argv[1:] if a.isdigit()]
B = int(_num[0]) if _num else 8192
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
print(f"{B} envs: {us:.2f} us/step (stamped build, {MODE})")
groups = (("drive wavefront", drive, 0, 12), ("judge wavefront", judge, 12, 12))
if MODE == "trio":
    groups = (("drive wavefront", ["loop top (action prefetch, replay read)", "controller: prefilter sweep",
                                   "controller: exact loop", "controller: steering + speed", "bicycle",
                                   "route switch + sincos", "wait A (masks of prev step)", "done test + commit rows",
                                   "wait B", "route reload + loop end"], 0, 12),
              ("judge C wavefront", ["wait A", "settle (done byte, re-spawn)", "wait B", "read rows + collision",
                                     "reward (ego lane) + outputs"], 12, 6),
              ("judge O wavefront", ["wait A", "done test + re-spawn", "wait B", "offroad + stop lines"], 18, 6))
for title, names, off, width in groups:
    tot = sum(out[off:off + width])
    print(f" {title}: {tot / n:.0f} memtime ticks per wave-step")
    for i, nm in enumerate(names):
        v = out[off + i]
        print(f"   {nm:34s} {v / n:8.1f} ticks  {100.0 * v / tot:5.1f} %")
