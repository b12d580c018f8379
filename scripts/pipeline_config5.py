"""configs[4] (8192 envs x 32 agents, step + birdview per timestep) as G independent sub-batches on G HIP streams: the step of
one group overlaps the rasteriser of another (both are latency-bound alone).  Same global batch (env_base per group), same
results.  usage: python3 scripts/pipeline_config5.py [envs] [agents] [groups ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.sharding import shard_config
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_world

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = int(sys.argv[2]) if len(sys.argv) > 2 else 32
groups = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
dev = torch.device("cuda:0")
_lib.load()
world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
dw = world.to_device(dev)
base = _abi.default_config(seed=1000, distance_cutoff=0.25)
CH = 250
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ext = _ext.load()
ref_img = None
for G in groups:
    nb = B // G
    parts = []
    for r in range(G):
        cfg, n = shard_config(base, r, G, B)
        assert n == nb
        st = EnvState(nb, A, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        img = ops.render_ego(cfg, dw, st)
        s = torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream()
        h = _ext.env_handle(cfg, dw, st)
        rows = [actions[i, r * nb:(r + 1) * nb].contiguous() for i in range(CH)]
        parts.append(dict(cfg=cfg, st=st, img=img, s=s, h=h, rows=rows, flags=int(cfg.flags)))
    torch.cuda.synchronize()

    STAGGER = os.environ.get("STAGGER", "0")
    evs = [torch.cuda.Event() for _ in range(64)]

    def run(T, t0=0):
        for t in range(t0, t0 + T):
            prev = None
            for i, p in enumerate(parts):
                with torch.cuda.stream(p["s"]):
                    if prev is not None and (STAGGER == "1" or (STAGGER == "once" and t == 0)):
                        p["s"].wait_event(prev)                      # this group's step starts when the previous group's has finished
                    p["h"].step(p["rows"][t % CH], p["flags"])
                    if STAGGER != "0" and G > 1:
                        prev = evs[(t * G + i) % 64]
                        prev.record(p["s"])
                    p["h"].render(p["img"], 64, 64, 35.0, 1, None, 0, 0, None, None)
    run(500)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        run(1000, 500 + rep * 1000)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    full = torch.cat([p["img"] for p in parts], 0)
    cols = {k: torch.cat([p["st"][k] for p in parts], 0).clone() for k in ("reward", "x", "psi", "episode", "steps")}
    same = None
    if ref_img is None: ref_img = full.clone(); ref_cols = cols
    else: same = {"img": bool(torch.equal(ref_img, full)), **{k: bool(torch.equal(ref_cols[k], cols[k])) for k in cols}}
    print(f"B={B} A={A} groups={G}: {best:.2f} us per timestep of the whole batch ({B / best * 1e6:.3e} env-steps/s); "
          f"identical to 1 group: {same}", flush=True)
