#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched env step path (BASELINE.json metric: env-steps/sec and agent-steps/sec
at 8192 envs x 16 agents on 1/2/4/8 MI355X).

    python bench.py --gpus 1 --steps 40000 --warmup 2000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE timestep of the whole batch held by a GPU: configs[2] = 8192 envs x 16 agents, full step (bicycle
kinematics, heuristic NPC controller, replay NPCs, all-pairs OBB collision, drivable-mesh offroad, WaypointSuite
reward / termination / truncation, in-place auto-reset).  State, world tables and the ego-action buffer are resident in
HBM when the timed region starts; steps are driven through the C-ABI (tde_env_rollout) with no host round trip.
Multi-GPU: envs are independent -> every rank owns its own 8192-env shard (weak scaling), no data-path collective; the
only collective is the MAX over ranks of the timed region.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ENVS, A_AGENTS = 8192, 16
# SURVEY §8(d) / BASELINE.md §4 algorithmic bytes: per agent-step 46 B (state r/w, attrs r, 2 flag bytes), per NPC
# agent-step 16 B (controller target + index), per env-step 38 B (action, target, counters, reward, done flags)
BYTES_PER_ENV_STEP = 46 * A_AGENTS + 16 * (A_AGENTS - 1) + 38   # = 1014
HBM_PEAK_GBPS = 8000.0                                          # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(world, cfg, budget_s=12.0):
    """The CPU oracle (a scalar C port of the same step, OpenMP over envs) timed on this box's host cores on a bounded
    sample of the same workload."""
    import numpy as np

    from oracle import oracle
    from torchdriveenv_amd.state import EnvState

    cores = os.cpu_count() or 1
    oracle.set_num_threads(cores)
    B = 2048 if cores >= 64 else 512
    hs = EnvState(B, A_AGENTS)
    oracle.env_reset(cfg, world, hs)
    rng = np.random.default_rng(0)

    def acts(K):
        return np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)

    t0 = time.perf_counter()
    oracle.env_rollout(cfg, world, hs, acts(4))
    per_step = (time.perf_counter() - t0) / 4
    K = int(max(8, min(400, budget_s / max(per_step, 1e-6))))
    a = acts(K)
    t0 = time.perf_counter()
    oracle.env_rollout(cfg, world, hs, a)
    dt = time.perf_counter() - t0
    # the reference's own operating point: ONE env stepped call by call on one thread
    oracle.set_num_threads(1)
    h1 = EnvState(1, A_AGENTS)
    oracle.env_reset(cfg, world, h1)
    h1["action"][:] = 0.0
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(50):
            oracle.env_step(cfg, world, h1)
        n1 += 50
    b1 = n1 / (time.perf_counter() - t0)
    oracle.set_num_threads(cores)
    return {"value": B * K / dt, "unit": "env-steps/s", "agent_steps_per_s": B * A_AGENTS * K / dt,
            "cores": oracle.num_threads(), "kind": "port", "b1_single_thread_env_steps_per_s": b1,
            "sample": f"{B} envs x {A_AGENTS} agents x {K} steps of the same workload, oracle/tde_oracle.c "
                      f"(brute-force mesh distance, OpenMP over envs), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--envs", type=int, default=B_ENVS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the N>1 timing reduce (nccl = RCCL; gloo for tests that put "
                         "several ranks on one GPU)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"],
                    help="rollout: K steps per C-ABI call (default); step: one C-ABI call per step from Python")
    args = ap.parse_args()

    import numpy as np
    import torch

    from torchdriveenv_amd import _abi, _lib, ops
    from torchdriveenv_amd.state import EnvState
    from torchdriveenv_amd.synth import synthetic_world

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world_size > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                        device_id=torch.device(f"cuda:{local_rank}"))
            except Exception as exc:      # e.g. several ranks on one GPU: the data path needs no collective, so the
                print(f"[bench] RCCL init failed ({type(exc).__name__}); timing reduce over gloo", file=sys.stderr)
                args.backend = "gloo"     # barrier / MAX reduce may as well run over gloo
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    _lib.load()

    B, A = args.envs, A_AGENTS
    world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)          # same tables on every GPU (replicated)
    cfg = _abi.default_config(seed=1000 + rank, distance_cutoff=0.25)  # shipped reward constants; per-shard seed
    dw = world.to_device(dev)
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)

    CH = 250                                                           # steps per C-ABI call (action buffer rows)
    g = torch.Generator(device="cpu").manual_seed(rank)
    actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1)
    actions = actions.to(torch.float32).contiguous().to(dev)
    reward = torch.empty((CH, B), dtype=torch.float32, device=dev)
    done = torch.empty((CH, B), dtype=torch.uint8, device=dev)

    def run(nsteps):
        left = nsteps
        while left > 0:
            k = min(CH, left)
            if args.mode == "rollout":
                ops.env_rollout(cfg, dw, st, actions[:k], reward[:k], done[:k])
            else:
                for i in range(k):
                    ops.env_step(cfg, dw, st, action=actions[i])
            left -= k

    run(args.warmup)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()                       # on torch's current stream == the stream the kernels are launched on
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(tt[0]), float(tt[1])

    # sanity of what was just computed (not timed): finite rewards, episodes progressing
    chk = dict(reward_sum=float(reward.double().sum()), done_frac=float((done > 0).float().mean()),
               episodes=int(st["episode"].max()))
    assert np.isfinite(chk["reward_sum"])
    if dist is not None:
        # the "host gather" of per-shard results: three numbers per rank through all_gather (no pickling, so it works
        # the same over RCCL and gloo), assembled on the host
        mine = torch.tensor([chk["reward_sum"], chk["done_frac"], float(chk["episodes"])], dtype=torch.float64,
                            device=dev if args.backend == "nccl" else "cpu")
        parts = [torch.empty_like(mine) for _ in range(world_size)]
        dist.all_gather(parts, mine)
        chk = [dict(rank=i, reward_sum=float(p[0]), done_frac=float(p[1]), episodes=int(p[2]))
               for i, p in enumerate(parts)]
    if rank == 0:
        n = world_size
        env_steps = B * n * args.steps
        # dominant kernel: rollout mode = tde::env_rollout_trio_kernel<16> (drive + two judge wavefronts per 64 agent
        # slots; TDE_ROLLOUT=duo|solo select the two- / one-wavefront forms), ONE launch per CH timesteps;
        # step mode = tde::env_step_kernel<16>, one launch per timestep
        steps_per_launch = CH if args.mode == "rollout" else 1
        launches = -(-args.steps // steps_per_launch)
        kern_us = dev_ms * 1e3 / launches                  # HIP-event time of the region / launches
        alg_bytes = BYTES_PER_ENV_STEP * B * (args.steps / launches)
        achieved = alg_bytes / (kern_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.mode == "rollout":
            try:
                tj = json.load(open(tpath))
                if tj.get("steps_per_launch") == steps_per_launch and tj.get("envs") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec", "value": env_steps / wall, "unit": "env-steps/s",
            "agent_steps_per_sec": env_steps * A / wall,
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[2]: {B} envs x {A} agents per GPU, full step (kinematics + NPC + replay + "
                                   "OBB collision + offroad mesh + waypoint reward + auto-reset)",
                       "envs_per_gpu": B, "agents_per_env": A, "mode": args.mode, "steps_per_call": CH,
                       "sharding": f"{n} independent shard(s), no data-path collective",
                       "timing_backend": (args.backend if n > 1 else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": {"solo": f"tde::env_rollout_kernel<{A}, false>",
                                    "duo": f"tde::env_rollout_duo_kernel<{A}, false>"}.get(
                             os.environ.get("TDE_ROLLOUT", ""), f"tde::env_rollout_trio_kernel<{A}, false>")
                         if args.mode == "rollout"
                         else f"tde::env_step_kernel<{A}, false>",
                         "kernel_avg_us": kern_us, "launches": launches, "steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes, "us_per_step": dev_ms * 1e3 / args.steps},
            "check": chk,
        }
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, cfg)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
