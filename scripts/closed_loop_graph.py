"""configs[4] shape in CLOSED loop as a HIP GRAPH: one timestep (fork of the sub-batch streams, tde_env_step_render, join) is captured
once and replayed per timestep - do the graph's internal dependencies cost less than stream events across hardware queues
(scripts/closed_loop_streams.py: 78 - 82 us per timestep with events against 51 on one stream)?
usage: python3 scripts/closed_loop_graph.py [envs] [agents] [streams ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = int(sys.argv[2]) if len(sys.argv) > 2 else 32
groups = [int(x) for x in sys.argv[3:]] or [1, 2, 3]
dev = torch.device("cuda:0")
_lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
for G in groups:
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    ops.env_rollout(cfg, dw, st, actions)
    img = ops.render_ego(cfg, dw, st)
    h = _ext.env_handle(cfg, dw, st)
    act = torch.zeros(B, 2, device=dev)                   # the graph's action buffer: the policy writes it before every replay
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    ptrs = [s.cuda_stream for s in streams]
    flags = int(cfg.flags)
    main = torch.cuda.Stream(device=dev)

    def timestep():
        cur = torch.cuda.current_stream(dev)
        if G == 1:
            h.step(act, flags)
            h.render(img, 64, 64, 35.0, 1, None, 0, 0, None, None)
        else:
            for s in streams: s.wait_stream(cur)
            h.step_render(act, flags, img, 64, 64, 35.0, 1, None, 0, 0, None, ptrs)
            for s in streams: cur.wait_stream(s)

    graph = torch.cuda.CUDAGraph()
    main.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(main):
        for _ in range(3): timestep()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=main):
            timestep()
    torch.cuda.synchronize()
    with torch.cuda.stream(main):
        for t in range(300):
            act.copy_(actions[t % 250], non_blocking=True)
            graph.replay()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for t in range(1000):
                act.copy_(actions[t % 250], non_blocking=True)     # (stands for the policy: a device-side write of the actions)
                graph.replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    print(f"closed loop, one graph replay per timestep, B={B} A={A} streams={G}: {best:.2f} us per timestep (action copy + fork + step + birdview + join)", flush=True)
