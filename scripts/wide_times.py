"""one-step launches and rollouts at 64 / 128 agent slots per env: us per step (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchdriveenv_amd import _abi, _ext, _lib, ops
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world

dev = torch.device("cuda:0")
_lib.load()
if os.environ.get("TDE_ROLLOUT_TEAM"): _lib.kernel_override(rollout=os.environ["TDE_ROLLOUT_TEAM"])   # solo | duo: force a rollout form
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25)
for A, kind in ((128, "town"),) if os.environ.get("TDE_HIP_LIB") else ((64, "junctions"), (128, "town")):
    # (town: ~100 of 128 slots present per scenario, the reference's assembled scene; junctions: what fits on the arms)
    world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4) if kind == "junctions" else \
        synthetic_town(n_scn=32, A=128, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
    dw = world.to_device(dev)
    for B in (256, 1024, 2048, 4096):
        g = torch.Generator().manual_seed(0)
        actions = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
        rows = [actions[i] for i in range(250)]
        st = EnvState(B, A, device=dev, with_info=False)
        ops.env_reset(cfg, dw, st)
        fl = int(cfg.flags)
        if os.environ.get("TDE_HIP_LIB"):                  # an A/B build: through ctypes (the extension links the in-tree library)
            class h:
                @staticmethod
                def step(a, _): ops.env_step(cfg, dw, st, action=a)
        else:
            h = _ext.env_handle(cfg, dw, st)
        for i in range(300): h.step(rows[i % 250], fl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(1000): h.step(rows[i % 250], fl)
        e1.record(); torch.cuda.synchronize()
        t_step = e0.elapsed_time(e1) * 1e3 / 1000
        ops.env_rollout(cfg, dw, st, actions)
        torch.cuda.synchronize(); e0.record()
        for _ in range(4): ops.env_rollout(cfg, dw, st, actions)
        e1.record(); torch.cuda.synchronize()
        t_roll = e0.elapsed_time(e1) * 1e3 / 1000
        live = float(st["present"].float().mean()) * A
        print(f"A={A:3d} {kind:9s} ({live:5.1f} present) B={B:5d}: step {t_step:7.2f} us, rollout {t_roll:7.2f} us per step", flush=True)
