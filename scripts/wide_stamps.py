"""Milestones of the two-role 128-slot one-step kernel (make_stamped_build.py wide): shader cycles between marks on wavefronts 0 (drive) and 2 (judge), workgroup start / end on the 100 MHz counter.
usage: python scripts/make_stamped_build.py wide && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/wide_stamps.py [B ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town

N = 400
dev = torch.device("cuda:0")
lib = _lib.load()
world = synthetic_town(n_scn=32, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
names = {17: "D: entry -> loads issued", 0: "D: -> (cold barrier: none since the cold block travels in the argument block)", 1: "D: state + cache entries landed, ctx, stored?", 2: "D: wait E",
         3: "D: bicycle, route switch, sincos, rows", 4: "D: wait B", 5: "D: next step's controller", 6: "D: wait A", 7: "D: re-spawn, stores (drained)",
         8: "J: entry -> prologue done", 9: "J: wait E + B", 10: "J: collision sweep, exact tests, publish", 11: "J: offroad", 12: "J: join the other judge",
         13: "J: reward, done flag", 14: "J: wait A", 15: "J: stores (drained)"}
for B in [int(x) for x in sys.argv[1:] if x.isdigit()] or [1, 256, 1024]:
    _lib.kernel_override(step="duo")
    st = EnvState(B, 128, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    g = torch.Generator().manual_seed(0)
    acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    for i in range(300):
        ops.env_step(cfg, dw, st, action=acts[i % 250])
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
    print(f"B = {B}: {e0.elapsed_time(e1) * 1e3 / N:.2f} us per step (stamped build)")
    for i, nm in names.items():
        print(f"  {nm:48s} {out[i] / (B * N):8.0f} cycles")
    # the last launch's workgroups on the 100 MHz real-time counter: when did they start and end, relative to the first start
    import numpy as np
    raw = np.zeros((4096, 24), dtype=np.uint64)
    lib.tde_debug_wg(raw.ctypes.data_as(C.c_void_p))
    n = min(B, 4096)
    t0, tD, tJ = (raw[:n, k].astype(np.int64) for k in (20, 21, 22))
    z = t0.min()
    q = lambda v: " / ".join(f"{x * 0.01:6.2f}" for x in np.percentile(v - z, [0, 25, 50, 75, 100]))
    print(f"  workgroup start (us after the first; min / 25 % / median / 75 % / max): {q(t0)}")
    print(f"  drive wavefront's end                                                 : {q(tD)}")
    print(f"  judge wavefront's end                                                 : {q(tJ)}")
    print(f"  workgroup lifetime                                                    : {q(np.maximum(tD, tJ) - t0 + z)}")
