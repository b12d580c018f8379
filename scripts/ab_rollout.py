"""Interleaved same-process A/B of libtde_hip.so builds on the headline rollout (8192 envs x 16 agents, 250 steps per
launch): every library is dlopen'ed in THIS process and the launches alternate lib by lib, so clocks, box and cache
state are shared.  Reports median / min / mean us per step over >= 40 launches per library.

    python scripts/ab_rollout.py [--envs 8192] [--agents 16] [--launches 40] [--lights] [--env NAME=VAL:lib] libA.so libB.so ...
A library argument may be prefixed with rollout kernel choice, e.g. duo:path.so (tde_kernel_override on that handle's
first call; the choice is latched per library at first use)."""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from torchdriveenv_amd import _abi
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--agents", type=int, default=16)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--steps", type=int, default=250)
ap.add_argument("--lights", action="store_true")
ap.add_argument("--signal-reach", type=int, default=1, help="0: a scenario sees only its own junction's lights")
ap.add_argument("--signals", type=int, default=0, help="--town: signalised junctions (light groups; up to 64)")
ap.add_argument("--cell", type=float, default=0.25, help="grid cell edge of the world's offroad index [m]")
ap.add_argument("--town", action="store_true", help="the 1 km x 1 km town map (synth.synthetic_town) instead of the junction maps")
ap.add_argument("--endless", action="store_true", help="episodes never end (no termination, no truncation): no re-spawns")
ap.add_argument("--truncate-only", type=int, default=0, metavar="N",
                help="no termination at infractions, truncation after N steps: every env re-spawns every N steps")
ap.add_argument("--dark-world", action="store_true", help="a world WITHOUT stop lines / light cycles (with --lights: the LIGHTS kernel variant with nothing to do: what its code alone costs)")
ap.add_argument("--coast", action="store_true", help="clear TDE_F_NPC_FIRST_STEP: the NPCs coast through an episode's first step (the rule of rounds 4 / 5)")
args = ap.parse_args()

B, A, K = args.envs, args.agents, args.steps
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=256, A=A, seed=0, cell=args.cell, n_signals=args.signals, signal_reach=args.signal_reach) if args.town else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, cell=args.cell, lights=not args.dark_world)
dw = world.to_device(dev)


from ab_layout import tiled_world  # noqa: E402

dw_tiled = None
g = torch.Generator().manual_seed(0)
actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1)
actions = actions.float().contiguous().to(dev)
reward = torch.empty((K, B), device=dev)
done = torch.empty((K, B), dtype=torch.uint8, device=dev)
flags = (_abi.F_ALL & ~(_abi.F_NPC_FIRST_STEP if args.coast else 0)) | (_abi.F_TRAFFIC_LIGHTS if args.lights else 0)
cfg = _abi.default_config(seed=1, distance_cutoff=0.25, flags=flags)
if args.endless:
    cfg.terminated_at_infraction = 0
    cfg.max_steps = 1 << 30
if args.truncate_only:
    cfg.terminated_at_infraction = 0
    cfg.max_steps = args.truncate_only
ro = _abi.TdeRollout(actions.data_ptr(), reward.data_ptr(), done.data_ptr(), K, 0)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

handles = []
for spec in args.libs:
    mode, _, path = spec.rpartition(":")
    path = os.path.abspath(path)
    L = C.CDLL(path)
    if mode and hasattr(L, "tde_kernel_override"):              # (per loaded library copy: its own choice)
        assert L.tde_kernel_override({"solo": 1, "duo": 2, "trio": 3}[mode], 0) == 0
    L.tde_env_rollout.argtypes = [C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState),
                                  C.POINTER(_abi.TdeRollout), C.c_void_p]
    L.tde_env_reset.argtypes = [C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState),
                                C.c_void_p, C.c_void_p]
    st = EnvState(B, A, device=dev, with_info=False)           # every library steps its own copy of the batch
    L.tde_abi_version.restype = C.c_int
    wd = dw
    if L.tde_abi_version() <= 5:
        dw_tiled = dw_tiled or tiled_world(world).to_device(dev)
        wd = dw_tiled
    assert L.tde_env_reset(C.byref(cfg), C.byref(wd.struct), C.byref(st.struct), None, stream) == 0
    for _ in range(2):                                          # warm-up; warm-up
        assert L.tde_env_rollout(C.byref(cfg), C.byref(wd.struct), C.byref(st.struct), C.byref(ro), stream) == 0
    torch.cuda.synchronize()
    handles.append((spec, L, st, [], wd))

for r in range(args.launches):
    for spec, L, st, ts, wd in handles:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.tde_env_rollout(C.byref(cfg), C.byref(wd.struct), C.byref(st.struct), C.byref(ro), stream)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
print(f"{B} envs x {A} agents, {K} steps per launch, {args.launches} interleaved launches per library, lights={int(args.lights)}")
for spec, L, st, ts, wd in handles:
    print(f"  {spec:40s} median {statistics.median(ts):6.3f}  min {min(ts):6.3f}  mean {statistics.mean(ts):6.3f} us/step")
