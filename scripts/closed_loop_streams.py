"""configs[4] shape in CLOSED loop: every timestep forks the sub-batch streams from the current stream (the policy's action is
ready), runs tde_env_step_render, and joins them back (the policy reads the observation) - overlap within a timestep only.
usage: python3 scripts/closed_loop_streams.py [envs] [agents] [streams ...]"""
import ctypes, os, sys, time
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
CH = 250
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
rows = [actions[i] for i in range(CH)]
for G in groups:
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    img = ops.render_ego(cfg, dw, st)
    h = _ext.env_handle(cfg, dw, st)
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    ptrs = [s.cuda_stream for s in streams]
    flags = int(cfg.flags)
    cur = torch.cuda.current_stream(dev)

    def run(T, t0=0):
        for t in range(t0, t0 + T):
            if G == 1:
                h.step(rows[t % CH], flags)
                h.render(img, 64, 64, 35.0, 1, None, 0, 0, None, None)
            else:
                for s in streams: s.wait_stream(cur)
                h.step_render(rows[t % CH], flags, img, 64, 64, 35.0, 1, None, 0, 0, None, ptrs)
                for s in streams: cur.wait_stream(s)
    run(500)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        run(1000, 500 + rep * 1000)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    print(f"closed loop B={B} A={A} streams={G}: {best:.2f} us per timestep (fork + step + birdview + join)", flush=True)
