import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world
B, A, K = 8192, 16, 250
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
actions = torch.zeros(K, B, 2, device=dev); reward = torch.empty((K, B), device=dev); done = torch.empty((K, B), dtype=torch.uint8, device=dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
for _ in range(3): ops.env_rollout(cfg, dw, st, actions, reward, done)
torch.cuda.synchronize()
for n in (1, 4, 16):
    t0 = time.perf_counter()
    ts = []
    for _ in range(n):
        t1 = time.perf_counter(); ops.env_rollout(cfg, dw, st, actions, reward, done); ts.append(time.perf_counter() - t1)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(n, "issue ms", t_issue * 1e3, "total ms", t_all * 1e3, "per-call issue us", [round(t * 1e6) for t in ts[:6]])
