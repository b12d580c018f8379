"""us per one-launch step (tde_env_step through the extension, closed loop, device-resident action rows) with and without
tde_state.magnitudes (ABI 10: the ego's infraction magnitudes written by the step kernel itself), same process, interleaved:
    python scripts/step_magnitudes_cost.py [junctions|town|wide] ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world

dev = "cuda:0"
CH = 250


def world_of(which):
    if which == "town":
        return synthetic_town(n_scn=256, A=16, seed=0), 8192, 16
    if which == "wide":
        return synthetic_town(n_scn=32, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4), 1024, 128
    if which == "big":                         # configs[3]'s whole batch on one GPU: the one-role kernel (above 131 072 agent slots)
        return synthetic_world(n_scn=64, A=16, seed=0, n_maps=4), 65536, 16
    if which == "a32":
        return synthetic_world(n_scn=64, A=32, seed=0, n_maps=4), 8192, 32
    return synthetic_world(n_scn=64, A=16, seed=0, n_maps=4), 8192, 16


for which in (sys.argv[1:] or ["junctions", "town", "wide"]):
    w, B, A = world_of(which)
    dw = w.to_device(dev)
    cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=_abi.F_ALL)
    g = torch.Generator(device="cpu").manual_seed(0)
    actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
    rows = [actions[i] for i in range(CH)]
    variants = {}
    for name, kw in (("bare", dict(with_info=False)), ("full", dict(with_info=True, with_obs=True, with_magnitudes=False)),
                     ("full+mag", dict(with_info=True, with_obs=True, with_magnitudes=True))):
        st = EnvState(B, A, device=dev, **kw)
        ops.env_reset(cfg, dw, st)
        variants[name] = (st, _ext.env_handle(cfg, dw, st))
    res = {k: [] for k in variants}
    for rep in range(6):
        for name, (st, h) in variants.items():
            for i in range(CH):
                h.step(rows[i], int(cfg.flags))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                for i in range(CH):
                    h.step(rows[i], int(cfg.flags))
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / (4 * CH))
    st = variants["full+mag"][0]
    m = st["magnitudes"]
    print(f"{which:9s} {B} x {A}: " + "  ".join(f"{k} {min(v):.2f} (med {sorted(v)[len(v)//2]:.2f})" for k, v in res.items()) +
          f"   [flagged egos at the last step: off {int((m[:,0]>0).sum())} col {int((m[:,2]>0).sum())}]", flush=True)
