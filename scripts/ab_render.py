"""Interleaved same-process A/B of libtde_hip.so builds on the rasteriser (BASELINE configs[4]: 8192 envs x 32 agents,
64x64x3 ego birdview): every library renders the SAME state, launches alternate lib by lib.
    python scripts/ab_render.py [--lights] [--stack 3] libA.so libB.so ..."""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--agents", type=int, default=32)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--stack", type=int, default=1)
ap.add_argument("--lights", action="store_true")
ap.add_argument("--cell", type=float, default=0.25, help="grid cell edge of the world's offroad index [m]")
ap.add_argument("--town", action="store_true", help="the 1 km x 1 km town map (synth.synthetic_town) instead of the junction maps")
args = ap.parse_args()
B, A, ns = args.envs, args.agents, args.stack
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=256, A=A, seed=0, cell=args.cell) if args.town else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, cell=args.cell)
dw = world.to_device(dev)
flags = _abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if args.lights else 0)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=flags)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(50, B, generator=g) * 2 - 1, torch.rand(50, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)                      # a mid-episode state (in-tree library)
torch.cuda.synchronize()
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
handles = []
ref = None
dw_new, dw_tiled = dw, None
for path in args.libs:
    L = C.CDLL(os.path.abspath(path))
    L.tde_abi_version.restype = C.c_int
    dw = dw_new
    if L.tde_abi_version() <= 5:                          # 8x8-tile cell order (see scripts/ab_rollout.py)
        if dw_tiled is None:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from ab_layout import tiled_world
            dw_tiled = tiled_world(world).to_device(dev)
        dw = dw_tiled
    L.tde_render_ego.argtypes = [C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState),
                                 C.POINTER(_abi.TdeRender), C.c_void_p]
    out = torch.zeros((B, 3 * ns, 64, 64), dtype=torch.uint8, device=dev)
    layers = torch.full((B, ns, 4096), _abi.LAYER_BLANK, dtype=torch.uint8, device=dev) if ns > 1 else None
    rd = _abi.TdeRender(out.data_ptr(), 64, 64, 35.0, ns, None if layers is None else layers.data_ptr(), 0, 0, None, None)
    for _ in range(3):
        assert L.tde_render_ego(C.byref(cfg), C.byref(dw.struct), C.byref(st.struct), C.byref(rd), stream) == 0
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    handles.append((path, L, rd, out, layers, [], same, dw))
for r in range(args.launches):
    for path, L, rd, out, layers, ts, same, dw in handles:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.tde_render_ego(C.byref(cfg), C.byref(dw.struct), C.byref(st.struct), C.byref(rd), stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
print(f"{B} views, {A} agents per env, n_stack {ns}, lights={int(args.lights)}, {args.launches} interleaved launches per library")
for path, L, rd, out, layers, ts, same, dw in handles:
    print(f"  {path:28s} median {statistics.median(ts):7.2f}  min {min(ts):7.2f}  mean {statistics.mean(ts):7.2f} us   "
          f"pixels equal to the first library: {same}")
