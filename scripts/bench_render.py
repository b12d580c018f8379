"""BASELINE configs[4]: 8192 envs x 32 agents, full step + 64x64x3 uint8 ego birdview per env (HIP rasteriser)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

A = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B, K = 8192, 50
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25)
st = EnvState(B, A, device=dev, with_info=False)
ops.env_reset(cfg, dw, st)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ops.env_rollout(cfg, dw, st, actions)
out = {}
for ns, ring in ((1, False), (3, False), (3, True)):
    img = None
    stack = ops.FrameStack(B, ns, device=dev) if ring else None

    def render(img):
        return stack.render(cfg, dw, st) if ring else ops.render_ego(cfg, dw, st, n_stack=ns, out=img)

    for _ in range(3):
        img = render(img)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 20
    for _ in range(n):
        img = render(img)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    wbytes = B * 12288 * ns               # every frame of the stack is (re)written: that is what the policy reads
    out[f"render_n_stack{ns}" + ("_layer_ring" if ring else "")] = dict(
        us=us, write_GBps=wbytes / us / 1e3, frac_of_8TBps=wbytes / us / 1e3 / 8000)
    del img, stack
# step + render per timestep (closed loop shape): one step launch + one render launch
st["action"].copy_(actions[0])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
img = ops.render_ego(cfg, dw, st)
torch.cuda.synchronize(); e0.record()
n = 200
for i in range(n):
    ops.env_step(cfg, dw, st)
    img = ops.render_ego(cfg, dw, st, out=img)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
alg = (46 * A + 16 * (A - 1) + 38 + 12288) * B
out["step_plus_render"] = dict(A=A, us_per_step=us, env_steps_per_s=B / us * 1e6, agent_steps_per_s=B * A / us * 1e6,
                               algorithmic_bytes=alg, achieved_GBps=alg / us / 1e3, frac_of_8TBps=alg / us / 1e3 / 8000)
print(json.dumps(out))
