"""s_memtime stamps per pass of the rasteriser.
usage: python scripts/make_stamped_build.py render && TDE_HIP_LIB=$PWD/ab/libS.so python scripts/render_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
A = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B, K = 8192, 50
dev = torch.device("cuda:0"); lib = _lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4); dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False); ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, actions)
img = ops.render_ego(cfg, dw, st)
torch.cuda.synchronize()
out = (C.c_ulonglong * 24)(); lib.tde_debug_stamps(out, 1)
for _ in range(5): img = ops.render_ego(cfg, dw, st, out=img)
torch.cuda.synchronize(); lib.tde_debug_stamps(out, 0)
n = out[10]
names = {6: "entry + cull (pass 0)", 1: "blocks (pass 1)", 2: "queued pixels (cell word)", 3: "mixed pixels (triangles)", 4: "objects", 5: "stream out"}
print("views sampled", n, "queued px/view", out[11] / n, "mixed px/view", out[12] / n)
for i, nm in names.items(): print(f"  {nm:28s} {out[i] / n:9.0f} ticks")
print("  total", sum(out[:7]) / n)
