"""Times the fused step kernel under feature-flag subsets (which part of the step costs what)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A, K = 8192, 16, 250
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, cell=float(os.environ.get("TDE_CELL", "0.25")))
dw = world.to_device(dev)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
F = _abi
sets = {"none(kin+coll)": 0, "npc": F.F_NPC, "offroad": F.F_OFFROAD, "reward": F.F_REWARD, "reward+reset": F.F_REWARD | F.F_AUTORESET,
        "npc+replay": F.F_NPC | F.F_REPLAY, "all-offroad": F.F_ALL & ~F.F_OFFROAD, "all-npc": F.F_ALL & ~F.F_NPC, "all": F.F_ALL}
for name, fl in sets.items():
    cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=fl)
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    for _ in range(2):
        ops.env_rollout(cfg, dw, st, actions, reward, done)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(4):
        ops.env_rollout(cfg, dw, st, actions, reward, done)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:16s} {e0.elapsed_time(e1) * 1e3 / (4 * K):8.2f} us/step")
