"""N launches of the rasteriser alone on a mid-episode state (for rocprofv3 --pmc / --kernel-trace passes).
usage: python3 scripts/run_render.py [launches] [agents] [envs] [n_stack] [lights]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world
TOWN = os.environ.get("TDE_WORLD") == "town"            # TDE_WORLD=town: the 1 km^2 town map instead of the junction maps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
A = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
ns = int(sys.argv[4]) if len(sys.argv) > 4 else 1
lights = len(sys.argv) > 5 and sys.argv[5] == "1"
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=256, A=A, seed=0) if TOWN else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=_abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0))
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
K = 50
acts = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)
stack = ops.FrameStack(B, ns, device=dev) if ns > 1 else None
img = None
for _ in range(n):
    img = stack.render(cfg, dw, st) if stack else ops.render_ego(cfg, dw, st, out=img)
torch.cuda.synchronize()
print("ok", n, "launches", B, "views")
