"""N closed-loop steps (one tde_env_step launch each) for rocprofv3 passes.
usage: python3 scripts/run_step.py [steps] [solo|trio] [agents] [envs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world
TOWN = os.environ.get("TDE_WORLD") == "town"            # TDE_WORLD=town: the 1 km^2 town map instead of the junction maps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kern = sys.argv[2] if len(sys.argv) > 2 else None
A = int(sys.argv[3]) if len(sys.argv) > 3 else 16
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
dev = torch.device("cuda:0")
CROWDED = os.environ.get("TDE_WORLD") == "crowded"      # TDE_WORLD=crowded: bench.py's 128-slot town (~122 agents present per env)
world = (synthetic_town(n_scn=32, A=A, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4) if CROWDED else
         synthetic_town(n_scn=256, A=A, seed=0) if TOWN else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4))
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
if os.environ.get("TDE_COAST") == "1":                  # the opt-out of TDE_F_NPC_FIRST_STEP: the NPCs coast through an episode's first step
    cfg.flags &= ~_abi.F_NPC_FIRST_STEP
_lib.kernel_override(step=kern if kern in ("solo", "trio") else None)
FULL = os.environ.get("TDE_STEP_OUTPUTS", "")             # "" bare, "full" = info / done bits / episode stats / obs, "mag" = + magnitudes
st = EnvState(B, A, device=dev, with_info=bool(FULL), with_obs=bool(FULL), with_magnitudes=(FULL == "mag"))
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, acts)                      # a steady-state mix of episode ages
for i in range(n):
    ops.env_step(cfg, dw, st, action=acts[i % 250])
torch.cuda.synchronize()
print("ok", n, "steps", kern, FULL)
