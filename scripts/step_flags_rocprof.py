"""device time of the one-step kernels by feature subset (rocprofv3 --kernel-trace --stats around this script)
usage: python3 scripts/step_flags_rocprof.py [agents] [envs] [solo|trio]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
A = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
kern = sys.argv[3] if len(sys.argv) > 3 else "solo"
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
_lib.kernel_override(step=kern)
F = _abi
sets = [("kin+collision", 0), ("+npc", F.F_NPC | F.F_REPLAY), ("+offroad", F.F_NPC | F.F_REPLAY | F.F_OFFROAD),
        ("+reward", F.F_NPC | F.F_REPLAY | F.F_OFFROAD | F.F_REWARD), ("all", F.F_ALL)]
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
for name, flags in sets:
    cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=flags)
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(100):
        ops.env_step(cfg, dw, st, action=acts[i % 250])
    torch.cuda.synchronize()
    # device time from the gaps-free estimate: events around 500 launches (host-bound if the kernel is shorter than a call)
    ev[0].record()
    for i in range(500):
        ops.env_step(cfg, dw, st, action=acts[i % 250])
    ev[1].record(); torch.cuda.synchronize()
    print(f"{kern} A={A} B={B} {name:14s} {ev[0].elapsed_time(ev[1]) * 2:.2f} us per launch (wall between events)")
