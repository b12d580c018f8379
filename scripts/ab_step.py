"""Interleaved same-process A/B of libtde_hip.so builds on the CLOSED-LOOP step (one tde_env_step launch per timestep,
8192 envs x 16 agents by default): every library steps its own copy of the batch; K consecutive launches are captured
into a HIP graph per library (no host in the loop: the period is the device's) and the graphs are replayed alternately.

    python scripts/ab_step.py [--envs 8192] [--agents 16] [--town] [--kernel trio|solo] [--outputs] libA.so libB.so ..."""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from torchdriveenv_amd import _abi
from torchdriveenv_amd.state import EnvState
from torchdriveenv_amd.synth import synthetic_town, synthetic_world

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--agents", type=int, default=16)
ap.add_argument("--steps", type=int, default=100, help="launches per graph")
ap.add_argument("--replays", type=int, default=30)
ap.add_argument("--signal-reach", type=int, default=1, help="0: a scenario sees only its own junction's lights")
ap.add_argument("--signals", type=int, default=0, help="--town: signalised junctions (light groups; up to 64)")
ap.add_argument("--cell", type=float, default=0.25, help="grid cell edge of the world's offroad index [m]")
ap.add_argument("--town", action="store_true")
ap.add_argument("--lights", action="store_true", help="with TDE_F_TRAFFIC_LIGHTS")
ap.add_argument("--kernel", default=None, choices=["solo", "trio"])
ap.add_argument("--endless", action="store_true", help="episodes never end (no termination, no truncation): no re-spawn tail")
ap.add_argument("--outputs", action="store_true", help="with info / done bits / episode statistics / compact observation")
ap.add_argument("--plain", action="store_true", help="time plain launches (K per block, through ctypes) instead of graph replays: the kernarg path of a real closed loop")
ap.add_argument("--coast", action="store_true", help="clear TDE_F_NPC_FIRST_STEP: the NPCs coast through an episode's first step (the rule of rounds 4 / 5)")
args = ap.parse_args()
B, A, K = args.envs, args.agents, args.steps
dev = torch.device("cuda:0")
world = synthetic_town(n_scn=256, A=A, seed=0, cell=args.cell, n_signals=args.signals, signal_reach=args.signal_reach) if args.town else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, cell=args.cell)
dw = world.to_device(dev)
cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=(_abi.F_ALL & ~(_abi.F_NPC_FIRST_STEP if args.coast else 0)) | (_abi.F_TRAFFIC_LIGHTS if args.lights else 0))
if args.endless:
    cfg.terminated_at_infraction = 0
    cfg.max_steps = 1 << 30
g = torch.Generator().manual_seed(0)
acts = torch.stack([torch.rand(250, B, generator=g) * 2 - 1, torch.rand(250, B, generator=g) * 0.6 - 0.3], -1).float().contiguous().to(dev)
ro = _abi.TdeRollout(acts.data_ptr(), None, None, 250, 0)
side = torch.cuda.Stream(device=dev)
handles = []
for path in args.libs:
    L = C.CDLL(os.path.abspath(path))
    sig = [C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState)]
    L.tde_env_step.argtypes = sig + [C.c_void_p]
    L.tde_env_reset.argtypes = sig + [C.c_void_p, C.c_void_p]
    L.tde_env_rollout.argtypes = sig + [C.POINTER(_abi.TdeRollout), C.c_void_p]
    if args.kernel:
        assert L.tde_kernel_override(0, {"solo": 1, "trio": 3}[args.kernel]) == 0
    st = EnvState(B, A, device=dev, with_info=args.outputs, with_obs=args.outputs)
    s0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert L.tde_env_reset(C.byref(cfg), C.byref(dw.struct), C.byref(st.struct), None, s0) == 0
    assert L.tde_env_rollout(C.byref(cfg), C.byref(dw.struct), C.byref(st.struct), C.byref(ro), s0) == 0   # a steady mix of episode ages
    structs = []
    for i in range(K):                                   # one tde_state per launch: its own row of the action buffer
        s = _abi.TdeState.from_buffer_copy(st.struct)
        s.action = acts[i % 250].data_ptr()
        structs.append(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        sp = C.c_void_p(side.cuda_stream)
        for s in structs[:3]:
            assert L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(s), sp) == 0
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for s in structs:
                assert L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(s), sp) == 0
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    handles.append((path, L, st, graph, [], structs))
for r in range(args.replays):
    for path, L, st, graph, ts, _ in handles:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if args.plain:
            s0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for s in _:
                L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(s), s0)
        else:
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / K)
print(("PLAIN LAUNCHES; " if args.plain else "") + f"closed loop, {B} envs x {A} agents, {'town' if args.town else 'junction maps'}, {K} launches per graph, {args.replays} "
      f"interleaved replays per library, kernel={args.kernel or 'auto'}, outputs={int(args.outputs)}, endless={int(args.endless)}")
for path, L, st, graph, ts, _ in handles:
    print(f"  {path:40s} median {statistics.median(ts):6.3f}  min {min(ts):6.3f}  mean {statistics.mean(ts):6.3f} us/step")
