"""Which phase makes a workgroup of the three-role one-step kernel slow?  ONE launch of the stamped build (make_stamped_build.py step3), every
workgroup's phase times (shader cycles) and its start / end on the 100 MHz counter; the slowest decile against the median workgroup.
usage: python scripts/make_stamped_build.py step3 && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/step_wg_spread.py [--lights]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A = 8192, 16
dev = torch.device("cuda:0")
lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
LIGHTS = "--lights" in sys.argv
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=_abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if LIGHTS else 0))
_lib.kernel_override(step="trio")
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)
names = {0: "D prologue", 1: "D controller (prologue)", 2: "D bicycle..rows", 3: "D wait B", 6: "D next-step controller", 4: "D wait A", 5: "D tail",
         8: "C prologue", 9: "C wait B", 12: "C collision", 10: "C reward", 11: "C wait A", 16: "O prologue", 17: "O wait B", 18: "O offroad", 20: "O psi term", 19: "O wait A"}
nwg = B * A // 64
rows = []
for rep in range(12):
    for i in range(20):
        ops.env_step(cfg, dw, st, action=acts[(rep * 21 + i) % 250])
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 24)()
    lib.tde_debug_stamps(out, 1)
    ops.env_step(cfg, dw, st, action=acts[(rep * 21 + 20) % 250])
    torch.cuda.synchronize()
    raw = np.zeros((4096, 24), dtype=np.uint64)
    lib.tde_debug_wg(raw.ctypes.data_as(C.c_void_p))
    rows.append(raw[:nwg].astype(np.int64))
R = np.concatenate(rows)                                  # [launches * nwg][24]
life = (R[:, 23] - R[:, 22]) * 0.01                       # us
start = np.concatenate([(r[:, 22] - r[:, 22].min()) * 0.01 for r in rows])
end = np.concatenate([(r[:, 23] - r[:, 22].min()) * 0.01 for r in rows])
dn = R[:, 7] != 0
slow = life >= np.percentile(life, 90)
late = end >= np.percentile(end, 95)
print(f"{len(rows)} launches x {nwg} workgroups; lifetime median {np.median(life):.2f} us, 90 % {np.percentile(life, 90):.2f}, max {life.max():.2f}; "
      f"launch = last end: median over launches {np.median([((r[:, 23] - r[:, 22].min()) * 0.01).max() for r in rows]):.2f} us")
print(f"share with a finished env: all {dn.mean():.3f}, slowest decile {dn[slow].mean():.3f}, last 5 % to end {dn[late].mean():.3f}")
print(f"start of the last 5 % to end: median {np.median(start[late]):.2f} us (all: {np.median(start):.2f})")
print(f"{'phase':28s} {'median wg':>10s} {'slow decile':>12s} {'last-to-end 5 %':>16s}   (shader cycles)")
for i, nm in names.items():
    print(f"{nm:28s} {np.median(R[:, i]):10.0f} {np.median(R[slow, i]):12.0f} {np.median(R[late, i]):16.0f}")
