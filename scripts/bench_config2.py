"""BASELINE configs[1]: 1024 envs x 8 agents, bicycle kinematics + OBB collision only (tde_kin_collide_step)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torchdriveenv_amd import ops

B, A = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 8)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = B * A
cx = torch.rand(B, generator=g).repeat_interleave(A) * 400 - 200
x = (cx + torch.rand(n, generator=g) * 24 - 12).to(dev); y = (torch.rand(n, generator=g) * 24 - 12).to(dev)
psi = (torch.rand(n, generator=g) * 6.28 - 3.14).to(dev); v = (torch.rand(n, generator=g) * 10).to(dev)
lr = torch.full((n,), 1.83, device=dev); L = torch.full((n,), 4.8, device=dev); W = torch.full((n,), 2.07, device=dev)
present = torch.ones(n, dtype=torch.uint8, device=dev)
act = torch.stack([torch.rand(n, generator=g) * 2 - 1, torch.rand(n, generator=g) * 0.6 - 0.3], -1).contiguous().to(dev)
out = torch.empty(n, dtype=torch.uint8, device=dev)
for _ in range(50): ops.kin_collide_step(B, A, x, y, psi, v, lr, L, W, present, act, out=out)
K = 2000
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
for _ in range(K): ops.kin_collide_step(B, A, x, y, psi, v, lr, L, W, present, act, out=out)
e1.record(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
us = e0.elapsed_time(e1) * 1e3 / K
alg = 53 * n   # 53 B per agent-step for this config
print(json.dumps(dict(config=f"{B} envs x {A} agents, kinematics + collision", us_per_step_device=us, us_per_step_wall=wall / K * 1e6,
                      agent_steps_per_s=n / us * 1e6, achieved_GBps=alg / us / 1e3, frac_of_8TBps=alg / us / 1e3 / 8000,
                      note="one launch per step from Python: launch-bound at this size")))
