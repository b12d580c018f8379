"""Closed loop at configs[2]'s shape as n contiguous sub-batches on n HIP streams (tde_env_step_render without the rasteriser): does overlapping
one sub-batch's launch boundary with another's kernel beat one launch per step?  us per timestep of the WHOLE batch, open-loop driver (the
streams are joined once at the end: an upper bound for a policy that consumes every step's outputs).
    python scripts/step_streams.py [envs] [agents]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
rows = [acts[i] for i in range(250)]
for n in (1, 2, 3, 4, 6, 8):
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    ops.env_rollout(cfg, dw, st, acts)
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
    ops.fork_streams(streams, dev)
    for i in range(200):
        ops.env_step_render(cfg, dw, st, streams, action=rows[i % 250], render=False)
    ops.join_streams(streams, dev)
    torch.cuda.synchronize()
    # the host out of the loop: K timesteps x n streams captured into ONE graph (fork / join inside the capture), replayed
    K = 50
    cap = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    cap.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(cap):
        with torch.cuda.graph(graph, stream=cap):
            ops.fork_streams(streams, dev)
            for i in range(K):
                ops.env_step_render(cfg, dw, st, streams, action=rows[i % 250], render=False)
            ops.join_streams(streams, dev)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
    print(f"{B} envs x {A} agents, {n} sub-batch(es) on {n} stream(s): {best:7.2f} us per timestep (graph of {K} timesteps)", flush=True)
