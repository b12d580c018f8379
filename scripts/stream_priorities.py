"""configs[4] through tde_env_step_render on three streams with different stream priorities (torch: -1 = high, 0 = normal)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B, A = 8192, 32
dev = torch.device("cuda:0")
_lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
CH = 250
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
rows = [actions[i] for i in range(CH)]
for prios in ([0, 0, 0], [-1, 0, 0], [-1, -1, 0], [0, -1, 0], [-1, -1, -1], [0, 0], [-1, 0]):
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    img = ops.render_ego(cfg, dw, st)
    h = _ext.env_handle(cfg, dw, st)
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    ptrs = [s.cuda_stream for s in streams]
    ops.fork_streams(streams, dev)
    flags = int(cfg.flags)

    def run(T, t0=0):
        for t in range(t0, t0 + T):
            h.step_render(rows[t % CH], flags, img, 64, 64, 35.0, 1, None, 0, 0, None, ptrs)
    run(500)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        run(1000, 500 + rep * 1000)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    print(f"priorities {prios}: {best:.2f} us per timestep", flush=True)
