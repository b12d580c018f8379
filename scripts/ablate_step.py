"""One-launch-per-step kernel (closed-loop mode) under feature-flag subsets: fixed launch cost vs step arithmetic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

A, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 2000
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
F = _abi
sets = {"none(kin+coll)": 0, "npc": F.F_NPC, "offroad": F.F_OFFROAD, "reward+reset": F.F_REWARD | F.F_AUTORESET, "all-autoreset": F.F_ALL & ~F.F_AUTORESET, "all": F.F_ALL}
for B in (1024, 8192):
    act = torch.zeros(B, 2, device=dev)
    for name, fl in sets.items():
        cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=fl)
        st = EnvState(B, A, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        for _ in range(200):
            ops.env_step(cfg, dw, st, action=act)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(N):
            ops.env_step(cfg, dw, st, action=act)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B:5d} {name:16s} {e0.elapsed_time(e1) * 1e3 / N:8.2f} us/launch")
