"""distribution of per-view durations of the rasteriser (probe build: scripts/build_variant.sh ab/libskip128.so -DTDE_RASTER_SKIP=128)
usage: TDE_HIP_LIB=$PWD/ab/libskip128.so python scripts/view_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
A, B = 32, 8192
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4); dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False); ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(50, B, generator=g) * 2 - 1, torch.rand(50, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)
img = None
for _ in range(5):
    img = ops.render_ego(cfg, dw, st, out=img)
torch.cuda.synchronize()
raw = img.reshape(B, -1)[:, :32].contiguous().cpu().numpy()
dur = raw[:, :8].copy().view(np.uint64)[:, 0].astype(np.float64)
st3 = raw[:, 8:20].copy().view(np.int32).astype(np.float64)              # listed 4x4 blocks, MIXED pixels, painted boxes
q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
print("per-view duration, s_memtime ticks:", " ".join(f"{x:.0f}" for x in q), "mean %.0f" % dur.mean())
for i, nm in enumerate(("listed 4x4 blocks", "MIXED-cell pixels", "painted boxes")):
    v = st3[:, i]
    print(f"  {nm:18s} mean {v.mean():6.1f}  p50 {np.percentile(v, 50):5.0f} p90 {np.percentile(v, 90):5.0f} max {v.max():5.0f}   corr with duration {np.corrcoef(v, dur)[0, 1]:.2f}")
X = np.column_stack([np.ones(B), st3])
coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("least squares: duration ~ %.0f + %.1f * blocks + %.1f * mixed_px + %.0f * boxes ticks" % tuple(coef))
slow = np.argsort(dur)[-8:]
print("slowest views:", [(int(dur[i]), int(st3[i, 0]), int(st3[i, 1]), int(st3[i, 2])) for i in slow])
