"""Milestones of the three-role one-step kernel (make_stamped_build.py step3): ticks since the wavefront entered the kernel.
usage: python scripts/make_stamped_build.py step3 && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/step_stamps.py [--lights]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A, N = 8192, 16, 400
dev = torch.device("cuda:0")
lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
LIGHTS = "--lights" in sys.argv
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=_abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if LIGHTS else 0))
if "--coast" in sys.argv:                                 # the opt-out of TDE_F_NPC_FIRST_STEP
    cfg.flags &= ~_abi.F_NPC_FIRST_STEP
_lib.kernel_override(step="trio")
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)                        # a steady-state mix of episode ages
for i in range(50):
    ops.env_step(cfg, dw, st, action=acts[i])
torch.cuda.synchronize()
out = (C.c_ulonglong * 24)()
lib.tde_debug_stamps(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N):
    ops.env_step(cfg, dw, st, action=acts[i % 250])
e1.record()
torch.cuda.synchronize()
lib.tde_debug_stamps(out, 0)
n = (B * A // 64) * N
print(f"{e0.elapsed_time(e1) * 1e3 / N:.2f} us per step (stamped build, lights={int(LIGHTS)})")
names = {0: "D: entry -> state + cache loaded", 1: "D: controller", 2: "D: bicycle, route switch, sincos, rows", 3: "D: wait B",
         6: "D: next-step controller (action cache)", 4: "D: wait A", 5: "D: done test, re-spawn, stores",
         8: "C: entry -> prologue done", 9: "C: wait B", 12: "C: collision", 10: "C: reward (dist, reach, smoothness)", 11: "C: wait A",
         16: "O: entry -> prologue done", 17: "O: wait B", 18: "O: offroad + stop lines", 20: "O: the ego's psi term (float64 cosine)",
         19: "O: wait A"}
for i, nm in names.items():
    print(f"  {nm:42s} {out[i] / n:8.0f} ticks")

# the last launch's workgroups on the 100 MHz counter (slots 22 / 23: entry / the drive wavefront's last store issued; 7: held a finished env)
import numpy as np
raw = np.zeros((4096, 24), dtype=np.uint64)
lib.tde_debug_wg(raw.ctypes.data_as(C.c_void_p))
nwg = min(B * A // 64, 4096)
t0, t1, dn = raw[:nwg, 22].astype(np.int64), raw[:nwg, 23].astype(np.int64), raw[:nwg, 7] != 0
z = t0.min()
q = lambda v: " / ".join(f"{x * 0.01:6.2f}" for x in np.percentile(v, [0, 25, 50, 75, 95, 100]))
print(f"  workgroup start, us after the first (min / 25 % / median / 75 % / 95 % / max): {q(t0 - z)}")
print(f"  drive wavefront's end                                                      : {q(t1 - z)}")
print(f"  lifetime, workgroups without a finished env ({int((~dn).sum())})                      : {q((t1 - t0)[~dn])}")
if dn.any():
    print(f"  lifetime, workgroups that re-spawned an env ({int(dn.sum())})                        : {q((t1 - t0)[dn])}")
