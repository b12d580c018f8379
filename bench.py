#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched env step path (BASELINE.json metric: env-steps/sec and agent-steps/sec
at 8192 envs x 16 agents on 1/2/4/8 MI355X).

    python bench.py                                   # N=1, configs[2] (8192 envs x 16 agents, full step)
    python bench.py --gpus N --steps K --warmup W     # N>1: starts one rank per GPU itself (no torchrun needed) ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # ... or runs as a rank of an existing launcher
    python bench.py --config 5                        # configs[4]: 8192 envs x 32 agents + 64x64x3 birdview raster
    python bench.py --config 2                        # configs[1]: 1024 envs x 8 agents, kinematics + collision only

A "step" is ONE timestep of the whole batch held by a GPU.  State, world tables and the ego-action buffer are resident
in HBM when the timed region starts; steps are driven through the C-ABI (tde_env_rollout: up to 250 consecutive
timesteps per call, no host round trip in between).

Timing protocol (what makes a short driver invocation such as `--steps 20 --warmup 5` report the steady state):
  * un-timed: `--warmup` steps, and in any case at least one full 250-step launch and >= 50 ms of launches;
  * timed: `repeats` back-to-back copies of the `--steps` block, repeats chosen so that the region lasts >= 0.25 s;
    the steps*repeats consecutive timesteps are issued in launches of up to 250 steps (the rollout call's operating
    point); `steps` / `warmup` echo the command line, `ms_per_step` = wall / (steps * repeats);
  * the region is bracketed by barrier + torch.cuda.synchronize() on both sides, MAX over ranks;
  * every timed launch is also bracketed by HIP events on the launch stream: `roofline` is computed from those
    (algorithmic bytes of the launches actually made / their summed duration), with min / median per launch.

Multi-GPU: envs are independent, so the global batch (configs[3] = 65536 envs at N=8) is cut into contiguous shards of
8192 envs, one rank per GPU (weak scaling), with the reset RNG keyed by the GLOBAL env index (tde_config.env_base via
sharding.shard_config): the N shards together are exactly the unsharded batch.  No data-path collective; the only
collectives are the barrier, the MAX of the timed region and the host gather of per-shard check sums.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
CH = 250                          # timesteps per tde_env_rollout call (rows of the resident action buffer)
MIN_WARM_S, MIN_REGION_S = 0.05, 0.25

# SURVEY §8(d) algorithmic bytes: per agent-step 46 B (state r/w 32, attrs 12, 2 flag bytes), per NPC agent-step 16 B
# (controller target + index), per env-step 38 B (action, target, counters, reward, done flags); config 2 (kinematics +
# collision only) 53 B per agent-step; config 5 adds 12 288 B of pixels per env-step.
CONFIGS = {
    3: dict(name="configs[2]", envs=8192, agents=16,
            what="full step (kinematics + NPC + replay + OBB collision + offroad mesh + waypoint reward + auto-reset)"),
    2: dict(name="configs[1]", envs=1024, agents=8, what="bicycle kinematics + OBB collision only"),
    5: dict(name="configs[4]", envs=8192, agents=32,
            what="full step + 64x64x3 uint8 ego birdview per env (HIP rasteriser), one step launch + one raster "
                 "launch per timestep and sub-batch (--streams sub-batches, each on its own HIP stream)"),
}


def bytes_per_env_step(config, A):
    if config == 2:
        return 53 * A
    full = 46 * A + 16 * (A - 1) + 38
    return full + (12288 if config == 5 else 0)


def cpu_baseline(world, cfg, A, budget_s=12.0):
    """The CPU oracle (a scalar C port of the same step, OpenMP over envs) timed on this box's host cores on a bounded
    sample of the same workload."""
    import numpy as np

    from oracle import oracle
    from torchdriveenv_amd.state import EnvState

    cores = os.cpu_count() or 1
    oracle.set_num_threads(cores)
    B = 8192                                       # the headline's own batch (round 4: the oracle's bounding-box shortcut makes it affordable)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    rng = np.random.default_rng(0)

    def acts(K):
        return np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)

    oracle.env_rollout(cfg, world, hs, acts(2))                        # (first touch of the tables)
    t0 = time.perf_counter()
    oracle.env_rollout(cfg, world, hs, acts(4))
    per_step = (time.perf_counter() - t0) / 4
    K = int(max(8, min(1000, budget_s / max(per_step, 1e-6))))
    a = acts(K)
    t0 = time.perf_counter()
    oracle.env_rollout(cfg, world, hs, a)
    dt = time.perf_counter() - t0
    # the reference's own operating point: ONE env stepped call by call on one thread
    oracle.set_num_threads(1)
    h1 = EnvState(1, A)
    oracle.env_reset(cfg, world, h1)
    h1["action"][:] = 0.0
    n1, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(50):
            oracle.env_step(cfg, world, h1)
        n1 += 50
    b1 = n1 / (time.perf_counter() - t0)
    oracle.set_num_threads(cores)
    # (a) of SURVEY 8d: the same step as batched B x A tensor ops on the host (oracle/torch_step.py), all cores
    bt = None
    if cfg.flags & 1:                                                  # (the full-step workload)
        try:
            import torch

            from oracle.torch_step import TorchWorld, torch_env_step
            nthr = min(cores, 16)                                      # many small ops: more threads only add contention
            torch.set_num_threads(nthr)
            Bt = 256
            tw = TorchWorld(world)
            ht = EnvState(Bt, A)
            oracle.env_reset(cfg, world, ht)
            rt = np.random.default_rng(1)

            def tact():
                return np.stack([rt.uniform(-1, 1, Bt), rt.uniform(-0.3, 0.3, Bt)], -1).astype(np.float32)

            ht["action"][...] = tact()
            t0 = time.perf_counter()
            torch_env_step(cfg, world, tw, ht, oracle_reset=oracle.env_reset)          # warm-up, also sizes the sample
            first = time.perf_counter() - t0
            nt, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 3.0 and first < 20.0:
                ht["action"][...] = tact()
                torch_env_step(cfg, world, tw, ht, oracle_reset=oracle.env_reset)
                nt += 1
            dtt = time.perf_counter() - t0
            if nt == 0:                                               # (a pathologically slow host: the warm-up step is the sample)
                nt, dtt = 1, first
            bt = {"value": Bt * nt / dtt, "unit": "env-steps/s", "threads": nthr,
                  "sample": f"{Bt} envs x {A} agents x {nt} steps as batched torch ops on the host (oracle/torch_step.py), {dtt:.1f} s"}
        except Exception as exc:                                       # pragma: no cover
            bt = {"error": repr(exc)}
    return {"batched_torch": bt, "value": B * K / dt, "unit": "env-steps/s", "agent_steps_per_s": B * A * K / dt,
            "cores": oracle.num_threads(), "kind": "port", "b1_single_thread_env_steps_per_s": b1,
            "sample": f"{B} envs x {A} agents x {K} steps of the same workload, oracle/tde_oracle.c "
                      f"(every triangle of the map per box corner behind a bounding-box reject, OpenMP over envs), {dt:.1f} s"}


def secondary(dev, region_s=0.3):
    """The other operating points of the path, timed in the same process after the headline region (rank 0, N = 1), each for
    >= `region_s` of back-to-back launches bracketed by HIP events on the launch stream: BASELINE configs[4] (full step +
    birdview raster, two launches per timestep), the closed loop at the headline shape (one launch per timestep), configs[1]
    (kinematics + collision only), the headline with the traffic-light term, and the headline with episodes that only
    end by truncation at 200 steps (the long-episode regime of SURVEY 8d; the headline's uniformly random ego leaves the
    road after ~50 steps)."""
    import ctypes

    import torch

    from torchdriveenv_amd import _abi, _ext, ops
    from torchdriveenv_amd.state import EnvState
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world

    worlds = {}

    def world_of(A, town=False):
        if (A, town) not in worlds:
            worlds.clear()                             # one world resident at a time (a town's tables are ~140 MB)
            if town == "crowded":                      # ~122 agents in 128 slots per scenario: the reference's assembled scenes (gym_env.py:216-237)
                w = synthetic_town(n_scn=32, A=A, seed=5, n_streets=4, spacing=100.0, ext=160.0, min_gap=3.4)
            else:
                w = synthetic_town(n_scn=256, A=A, seed=0) if town else synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
            worlds[(A, town)] = (w, w.to_device(dev))
        return worlds[(A, town)]

    def run(name, config, B, A, stepwise, flags, render=False, n_streams=1, with_info=False, town=False, magnitudes=None, **cfg_over):
        w, dw = world_of(A, town)
        cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=flags, **cfg_over)
        st = EnvState(B, A, device=dev, with_info=with_info, with_obs=with_info, with_magnitudes=magnitudes)
        ops.env_reset(cfg, dw, st)
        g = torch.Generator(device="cpu").manual_seed(0)
        actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1)
        actions = actions.to(torch.float32).contiguous().to(dev)
        reward = torch.empty((CH, B), dtype=torch.float32, device=dev)
        done = torch.empty((CH, B), dtype=torch.uint8, device=dev)
        rows = [actions[i] for i in range(CH)]
        img = ops.render_ego(cfg, dw, st) if render else None
        h = None
        if stepwise:
            h = _ext.env_handle(cfg, dw, st)

        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else []
        ptrs = [s_.cuda_stream for s_ in streams]
        if streams:
            ops.fork_streams(streams, dev)

        def block():                                   # CH consecutive timesteps
            if not stepwise:
                ops.env_rollout(cfg, dw, st, actions, reward, done)
                return
            if streams:
                for i in range(CH):
                    h.step_render(rows[i], int(cfg.flags), img, 64, 64, 35.0, 1, None, 0, 0, None, ptrs)
                return
            for i in range(CH):
                h.step(rows[i], int(cfg.flags))
                if render:
                    h.render(img, 64, 64, 35.0, 1, None, 0, 0, None, None)

        t0 = time.perf_counter()
        block()
        torch.cuda.synchronize()
        per_block = max(time.perf_counter() - t0, 1e-6)
        block()                                        # (second warm block: clocks, caches)
        n = max(1, int(-(-region_s // per_block)))
        # two timed regions of n blocks each, the faster one reported: a closed-loop point is driven by this Python loop, and one
        # disturbance of the host (another process, a collector pause) would otherwise be booked as device time
        us = float("inf")
        for _rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(streams[0] if streams else None)
            for _ in range(n):
                block()
            if streams:
                ops.join_streams(streams, dev)         # the region ends when every sub-batch has finished
            e1.record()
            torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / (n * CH))
            if streams:
                ops.fork_streams(streams, dev)
        bpes = bytes_per_env_step(config, A)
        ach = bpes * B / (us * 1e-6) / 1e9
        return {"workload": name, "envs": B, "agents_per_env": A, "us_per_step": us, "env_steps_per_s": B / us * 1e6,
                "agent_steps_per_s": B * A / us * 1e6, "timed_steps": n * CH, "launches_per_step": ((2 if render else 1) * max(1, n_streams)) if stepwise else 1.0 / CH, "streams": max(1, n_streams),
                "roofline": {"bound": "hbm", "bytes_per_env_step": bpes, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBPS},
                "done_frac_last_step": float((st["terminated"] | st["truncated"]).float().mean())}

    F = _abi.F_ALL
    out = {}
    for key, kw in (
        ("config5", dict(name="configs[4]: 8192 envs x 32 agents, full step + 64x64x3 birdview; three sub-batches on three HIP "
                              "streams (tde_env_step_render: step launch + raster launch per sub-batch and timestep)",
                         config=5, B=8192, A=32, stepwise=True, flags=F, render=True, n_streams=3)),
        ("config5_one_stream", dict(name="configs[4] on one stream: step launch + raster launch per timestep",
                                    config=5, B=8192, A=32, stepwise=True, flags=F, render=True)),
        ("config5_16384", dict(name="configs[4] shape at 16 384 envs x 32 agents: three sub-batches on three HIP streams",
                               config=5, B=16384, A=32, stepwise=True, flags=F, render=True, n_streams=3)),
        ("closed_loop", dict(name="configs[2] shape, closed loop: one tde_env_step launch per timestep through the extension",
                             config=3, B=8192, A=16, stepwise=True, flags=F)),
        ("closed_loop_full_outputs", dict(name="the same with everything BatchedWaypointEnv.step asks of the kernel: float64 info terms, done bits, Monitor "
                                               "episode statistics, the compact observation AND the magnitudes of the ego's infractions (the reference's "
                                               "info['offroad' | 'collision'], gym_env.py:427-428), all written by the one step launch",
                                          config=3, B=8192, A=16, stepwise=True, flags=F, with_info=True, magnitudes=True)),
        ("closed_loop_full_outputs_indicators", dict(name="the same with 0 / 1 infraction indicators instead of the magnitudes (BatchedWaypointEnv(info_magnitudes=False): "
                                                          "round 4's closed_loop_full_outputs)",
                                                     config=3, B=8192, A=16, stepwise=True, flags=F, with_info=True, magnitudes=False)),
        ("coast_rule", dict(name="configs[2] with TDE_F_NPC_FIRST_STEP cleared (the opt-out: NPCs coast through an episode's first step - the rule "
                                 "rounds 4 / 5 were measured under), rollout", config=3, B=8192, A=16, stepwise=False, flags=F & ~_abi.F_NPC_FIRST_STEP)),
        ("coast_rule_closed_loop", dict(name="the same in closed loop", config=3, B=8192, A=16, stepwise=True, flags=F & ~_abi.F_NPC_FIRST_STEP)),
        ("config2", dict(name="configs[1]: 1024 envs x 8 agents, kinematics + collision only, 250 steps per launch",
                         config=2, B=1024, A=8, stepwise=False, flags=0)),
        ("lights", dict(name="configs[2] + traffic-light / stop-line term (TDE_F_TRAFFIC_LIGHTS), rollout",
                        config=3, B=8192, A=16, stepwise=False, flags=F | _abi.F_TRAFFIC_LIGHTS)),
        ("lights_closed_loop", dict(name="the same in closed loop: what BatchedWaypointEnv.step launches on maps that carry traffic lights",
                                    config=3, B=8192, A=16, stepwise=True, flags=F | _abi.F_TRAFFIC_LIGHTS)),
        ("lights_closed_loop_full_outputs", dict(name="the reference's operating point in closed loop: traffic lights + NPCs acting from step one + every output of "
                                                      "BatchedWaypointEnv.step (info terms, done bits, episode statistics, observation, infraction magnitudes)",
                                                 config=3, B=8192, A=16, stepwise=True, flags=F | _abi.F_TRAFFIC_LIGHTS, with_info=True, magnitudes=True)),
        ("agents_128", dict(name="1024 envs x 128 agent slots (~122 present per env: the reference's ~100-agent scenes), rollout",
                            config=3, B=1024, A=128, stepwise=False, flags=F, town="crowded")),
        ("agents_128_closed_loop", dict(name="the same in closed loop: one tde_env_step launch per timestep",
                                        config=3, B=1024, A=128, stepwise=True, flags=F, town="crowded")),
        ("agents_128_256", dict(name="256 envs x 128 agent slots, rollout (eight wavefronts per env up to half a residency round)",
                                config=3, B=256, A=128, stepwise=False, flags=F, town="crowded")),
        ("agents_128_closed_loop_256", dict(name="256 envs x 128 agent slots in closed loop (half a residency round and below: eight wavefronts per env - "
                                                 "drive, judge, sweep helper, offroad helper for each half of the slots)",
                                            config=3, B=256, A=128, stepwise=True, flags=F, town="crowded")),
        ("long_episodes", dict(name="configs[2] with episodes that end by truncation at 200 steps only (terminated_at_infraction = 0), rollout",
                               config=3, B=8192, A=16, stepwise=False, flags=F, terminated_at_infraction=0)),
        # the reference's map size (a CARLA town's drivable mesh): ONE 1 km x 1 km map of 5.7e4 triangles, 100
        # junctions, 256 scenarios spread over it (synth.synthetic_town) instead of four 200-triangle junction maps
        ("town", dict(name="configs[2] on the town map (1 km^2, 5.7e4 triangles, 256 scenarios), rollout",
                      config=3, B=8192, A=16, stepwise=False, flags=F, town=True)),
        ("town_closed_loop", dict(name="configs[2] shape on the town map, closed loop: one tde_env_step launch per timestep",
                                  config=3, B=8192, A=16, stepwise=True, flags=F, town=True)),
        ("town_config5", dict(name="configs[4] on the town map: 8192 envs x 32 agents, full step + birdview, three sub-batches on three streams",
                              config=5, B=8192, A=32, stepwise=True, flags=F, render=True, n_streams=3, town=True)),
    ):
        try:
            out[key] = run(**kw)
        except Exception as exc:                       # pragma: no cover
            out[key] = {"error": repr(exc)}
    return out


def python_boundary(B=8192, A=16):
    """SURVEY 8(d) items (ii) and (iii), in the driver's own run: one timestep through the Python boundary of the reference's
    interface at the headline shape - (ii) BatchedWaypointEnv.step with device-resident outputs (compact state observation;
    64x64x3 birdview), (iii) the SB3-shaped numpy path (WaypointVecEnv.step: one D2H copy of observations / rewards / dones /
    infos per step; `views` = copy_obs=False, the arrays are views of a ring of pinned buffers).  Wall clock, host included."""
    import numpy as np
    import torch

    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv
    from torchdriveenv_amd.synth import synthetic_world

    world = synthetic_world(n_scn=64, A=A, seed=0, n_maps=4)
    out = {"envs": B, "agents_per_env": A, "timer": "time.perf_counter around N calls, torch.cuda.synchronize at both ends; best of 3 passes"}
    for mode in ("state", "birdview"):
        env = BatchedWaypointEnv(EnvConfig(seed=3, distance_cutoff=0.25), world, num_envs=B, obs_mode=mode, with_info=True)
        env.reset()
        # the action rows of `secondary.closed_loop` (uniform acceleration in [-1, 1], steering in [-0.3, 0.3]): the step time
        # depends on what the agents do (a constant push sends every ego off the road within 30 steps: 12.2 us instead of 9.4)
        g = torch.Generator(device="cpu").manual_seed(0)
        rows_t = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1).float().contiguous()
        rows = list(rows_t.to(env.torch_device))
        rows_np = list(rows_t.numpy())
        it = [0]

        def act_dev():
            it[0] += 1
            return rows[it[0] % CH]

        def act_host():
            it[0] += 1
            return rows_np[it[0] % CH]

        def timed(fn, n):
            best = float("inf")
            for rep in range(3):                                       # (best of three: the first pass also ramps the clocks)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / n * 1e6)
            return best

        timed(lambda: env.step(act_dev()), 500)                       # (a steady mix of episode ages first)
        r = {"device_outputs_us_per_step": timed(lambda: env.step(act_dev()), 2000)}
        r["numpy_copies_us_per_step"] = timed(lambda: env.vec_step(act_host()), 200 if mode == "state" else 10)
        env._vec = None
        venv = env.as_vec_env(copy_obs=False)
        r["numpy_views_us_per_step"] = timed(lambda: venv.step(act_host()), 200 if mode == "state" else 30)
        r["env_steps_per_s"] = {k[:-12]: B / v * 1e6 for k, v in r.items()}
        out[mode] = r
        del env, venv
    # The env above steps WITH the traffic-light term (the maps have lights, as the reference's do: the closed-loop kernel then takes
    # 11.1 us instead of 8.2, profiles/r04_z_town_lights.txt); the same call on maps without lights, for comparison with
    # `secondary.closed_loop`:
    out["traffic_lights"] = True
    env = BatchedWaypointEnv(EnvConfig(seed=3, distance_cutoff=0.25), synthetic_world(n_scn=64, A=A, seed=0, n_maps=4, lights=False),
                             num_envs=B, obs_mode="state", with_info=True)
    env.reset()
    timed(lambda: env.step(act_dev()), 500)
    out["state_no_lights_device_outputs_us_per_step"] = timed(lambda: env.step(act_dev()), 2000)
    return out


FIXED_MOD = 1 << 48                  # check sums are exact integers below 2^53: they travel as float64 through all_gather


def fixed_check(cfg, dw, B, A, actions, dev):
    """fresh state, reset, ONE rollout over the CH action rows: (sum of the reward bit patterns, sum of the done bytes, sum of the
    final x bit patterns), each modulo 2^48 - exact and independent of the order of summation, so a shard's values can be compared
    with the same sums over its columns of the unsharded batch"""
    import torch

    from torchdriveenv_amd import ops
    from torchdriveenv_amd.state import EnvState

    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    reward, done = ops.env_rollout(cfg, dw, st, actions)
    torch.cuda.synchronize()
    return fixed_sums(reward, done, st["x"])


def fixed_sums(reward, done, x):
    import torch

    bits = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) % FIXED_MOD   # noqa: E731
    return [bits(reward), int(done.to(torch.int64).sum().item()) % FIXED_MOD, bits(x)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def bench_world(kind, A):
    """the benchmark's static tables, built ONCE per machine (world.cached_world: the first caller builds and saves them, the other
    ranks load the file) - N ranks that each build the grid index of a town side by side on the same host cores spend seconds
    there before their first launch.  Returns (world, "built" | "loaded", seconds)."""
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world
    from torchdriveenv_amd.world import cached_world

    t0 = time.perf_counter()
    if kind == "town":
        w, how = cached_world(f"bench-town-A{A}", lambda: synthetic_town(n_scn=256, A=A, seed=0))
    else:
        w, how = cached_world(f"bench-junctions-A{A}", lambda: synthetic_world(n_scn=64, A=A, seed=0, n_maps=4))
    return w, how, time.perf_counter() - t0


def spawn_ranks(n, world_kind=None, A=None):
    """`python bench.py --gpus N` without a launcher: start one fresh child process per rank.  This process has not
    touched the GPU (no HIP call: the grid-index build it does for the ranks - bench_world - is host code), so the
    children are ordinary new processes; rank 0's JSON line passes through on stdout."""
    if world_kind is not None:
        # in a child of its own: THIS process then still has not loaded the HIP runtime (nor torch) when it starts the ranks
        code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; _, how, dt = bench.bench_world({world_kind!r}, {int(A)}); "
                "print(f'[bench] parent: world tables {how} in {dt:.2f} s', file=sys.stderr, flush=True)")
        subprocess.run([sys.executable, "-c", code], check=False)     # (a failure only means that the first rank builds them)
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that dies (no GPU for it, RCCL initialisation error ...) leaves the others waiting in the rendezvous or in
    # the barrier: poll, and when one has failed stop the rest (the exact processes started above) instead of hanging
    import time
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
        if rc and live:
            time.sleep(5.0)                                      # let the failing rank's siblings report, then stop them
            for p in live:
                if p.poll() is None:
                    p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    return rc


class _StdoutToStderr:
    """gloo announces its connections on stdout; stdout must carry the ONE JSON line only"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json config, 1-based: 3 = the headline (default), 2 = kinematics + collision only, "
                         "5 = 32 agents + birdview raster")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary operating points (config 5, closed loop, config 2, lights, long episodes) that a "
                         "default N = 1 headline run reports under `secondary`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process group of the N>1 barrier / timing reduce (nccl = RCCL, one rank per GPU; gloo for "
                         "tests that put several ranks on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still create the process group (world size 1) and run the barrier, the MAX reduce and "
                         "the host gather through it - how the RCCL path (--backend nccl) is exercised on a one-GPU box")
    ap.add_argument("--check-fixed", action="store_true",
                    help="after the timed region (not timed): a fresh reset of this rank's shard and ONE 250-step rollout from its "
                         "action rows, reported as exact integer check sums per rank under check[].fixed - a function of (seed, global "
                         "env index, actions) alone, so the N shards can be held against the unsharded batch (tests/test_gpu_sharded_bench.py)")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"],
                    help="rollout: up to 250 steps per C-ABI call (default); step: one C-ABI call per step from Python")
    ap.add_argument("--binding", default="ext", choices=["ext", "ctypes"],
                    help="step mode / config 5: launches through the PyTorch-ROCm C++ extension (default) or through "
                         "the ctypes binding of the same C-ABI")
    ap.add_argument("--streams", type=int, default=None,
                    help="config 5 only: run each timestep as this many sub-batches on their own HIP streams "
                         "(tde_env_step_render: the step of one sub-batch overlaps the rasteriser of another); default 3 "
                         "(us per timestep by stream count: profiles/r03_f_config5_streams_matrix.txt; more than 3 needs GPU_MAX_HW_QUEUES > 4), 1 = one stream")
    ap.add_argument("--world", default="junctions", choices=["junctions", "town"],
                    help="junctions: four ~200-triangle junction maps, 64 scenarios (the SURVEY 8d recipe; default); town: one "
                         "1 km x 1 km map of 5.7e4 triangles with 256 scenarios (the reference's map size, SURVEY R10)")
    ap.add_argument("--rollout-kernel", default=None, choices=["solo", "duo", "trio"],
                    help="force a form of the rollout kernel (A/B runs; default: the library's choice by group shape)")
    ap.add_argument("--step-kernel", default=None, choices=["solo", "trio"],
                    help="force a form of the one-step kernel (A/B runs; default: the library's choice by batch size)")
    args = ap.parse_args()

    t_start = time.perf_counter()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size == 1 and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, args.world, CONFIGS[args.config]["agents"]))   # before anything touches the GPU
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")

    import numpy as np
    import torch

    from torchdriveenv_amd import _abi, _lib, ops
    from torchdriveenv_amd.sharding import shard_config, shard_range
    from torchdriveenv_amd.state import EnvState
    from torchdriveenv_amd.synth import synthetic_town, synthetic_world

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device: there is no CPU path")
    dist = None
    if world_size > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world_size == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.backend == "nccl" and ndev < world_size:
            raise SystemExit(f"--backend nccl needs one GPU per rank ({world_size} ranks, {ndev} device(s)); several "
                             "ranks on one GPU only time over --backend gloo")
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":                     # no silent fallback: an RCCL failure fails the run
            import datetime
            dist.init_process_group("nccl", rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=300),
                                    device_id=torch.device(f"cuda:{local_rank}"))
        else:
            with _StdoutToStderr():
                import datetime
                dist.init_process_group("gloo", rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=300))
                dist.barrier()
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    _lib.load()
    _lib.kernel_override(rollout=args.rollout_kernel, step=args.step_kernel)

    C = CONFIGS[args.config]
    B, A = (args.envs or C["envs"]), C["agents"]
    n = world_size
    world, world_how, world_s = bench_world(args.world, A)            # same tables on every GPU (replicated); built once per machine
    flags = 0 if args.config == 2 else _abi.F_ALL
    base_cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=flags)   # shipped reward constants
    cfg, nb = shard_config(base_cfg, rank, n, B * n)                  # env_base = rank * B: shard of the global batch
    assert nb == B and shard_range(rank, n, B * n) == (rank * B, (rank + 1) * B)
    dw = world.to_device(dev)
    st = EnvState(B, A, device=dev, with_info=False)
    ops.env_reset(cfg, dw, st)
    stepwise = args.mode == "step" or args.config == 5
    spl = 1 if stepwise else CH                                       # steps per launch of the dominant kernel

    # columns [rank*B, (rank+1)*B) of the global [CH, n*B, 2] action tensor (defined shard by shard from seed = rank)
    g = torch.Generator(device="cpu").manual_seed(rank)
    actions = torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1)
    actions = actions.to(torch.float32).contiguous().to(dev)
    reward = torch.empty((CH, B), dtype=torch.float32, device=dev)
    done = torch.empty((CH, B), dtype=torch.uint8, device=dev)
    img = ops.render_ego(cfg, dw, st) if args.config == 5 else None
    handle = None
    if args.binding == "ext":                       # every launch of the timed region goes through the PyTorch-ROCm extension
        from torchdriveenv_amd import _ext
        handle = _ext.env_handle(cfg, dw, st)
    cfg_flags = int(cfg.flags)
    act_rows = [actions[i] for i in range(CH)]      # views made once: slicing a tensor costs microseconds of host time
    if args.streams is not None and (args.config != 5 or not 1 <= args.streams <= 16):
        raise SystemExit("--streams applies to --config 5 only and must be in [1, 16]")
    n_streams = (args.streams or 3) if args.config == 5 else 1
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else []
    stream_ptrs = [s_.cuda_stream for s_ in streams]
    ev_stream = streams[0] if streams else None     # the stream the per-launch events are recorded on
    if streams:
        ops.fork_streams(streams, dev)              # (the reset and the first birdview above ran on the current stream)

    def launch(k, row):
        """k consecutive timesteps (k <= CH); `row` = first row of the action buffer to use"""
        if not stepwise:
            if handle is not None:
                handle.rollout(actions[:k], reward[:k], done[:k], cfg_flags)
            else:
                ops.env_rollout(cfg, dw, st, actions[:k], reward[:k], done[:k])
            return
        if streams:                                 # config 5 as sub-batches on their own streams: one call per timestep
            for i in range(k):
                if handle is not None:
                    handle.step_render(act_rows[(row + i) % CH], cfg_flags, img, 64, 64, 35.0, 1, None, 0, 0, None, stream_ptrs)
                else:
                    ops.env_step_render(cfg, dw, st, streams, action=act_rows[(row + i) % CH], out=img)
            return
        if handle is not None:
            for i in range(k):
                handle.step(act_rows[(row + i) % CH], cfg_flags)
                if img is not None:
                    handle.render(img, 64, 64, 35.0, 1, None, 0, 0, None, None)
            return
        for i in range(k):
            ops.env_step(cfg, dw, st, action=act_rows[(row + i) % CH])
            if img is not None:
                ops.render_ego(cfg, dw, st, out=img)

    def run(nsteps, events=None):
        left, row = nsteps, 0
        while left > 0:
            k = min(CH, left)
            if events is not None and len(events) < 8192:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(ev_stream)               # on the stream the kernels are launched on (torch's current one; with
                                                   # sub-batch streams: the first sub-batch's)
                events.append((ev, k))
            launch(k, row)
            left -= k
            row += k

    # ---- un-timed: the CLI's warm-up, at least one full-length launch, at least MIN_WARM_S of launches --------------
    t0 = time.perf_counter()
    startup_s = t0 - t_start                        # process start (after argument parsing) -> first launch: imports, world, uploads
    run(max(args.warmup, CH))
    torch.cuda.synchronize()
    warm_steps = max(args.warmup, CH)
    while time.perf_counter() - t0 < MIN_WARM_S:
        run(CH)
        torch.cuda.synchronize()
        warm_steps += CH
    t1 = time.perf_counter()
    run(CH)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t1) / CH                             # s per step, warm
    repeats = max(1, int(-(-MIN_REGION_S // max(est * args.steps, 1e-9))))
    if dist is not None:                                              # every rank must time the same number of steps
        rr = torch.tensor([repeats], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        repeats = int(rr[0])
    total = args.steps * repeats

    # ---- timed region ---------------------------------------------------------------------------------------------
    events = []
    ev_end = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(total, events)
    ev_last = torch.cuda.Event(enable_timing=True)      # end of the last launch on the launch stream (per-launch statistics)
    ev_last.record(ev_stream)
    if streams:
        ops.join_streams(streams, dev)                  # the region ends when EVERY sub-batch has finished
    ev_end.record(None if streams else ev_stream)       # (after the join: on the current stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # per-launch durations from the HIP events (launch i lasts from its event to the next one)
    marks = [e for e, _ in events] + [ev_last]
    lens = [k for _, k in events]
    dur_us = [marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(len(events))]
    dev_ms = marks[0].elapsed_time(ev_end)
    if streams:
        ops.fork_streams(streams, dev)
    built = 1.0 if world_how == "built" else 0.0
    if dist is not None:
        tt = torch.tensor([wall, dev_ms, startup_s, world_s, built], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, dev_ms, startup_s, world_s = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])
        ts = torch.tensor([built], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        built = float(ts[0])

    # sanity of what was just computed (not timed): finite rewards, episodes progressing
    chk = dict(reward_sum=float((st["reward"] if stepwise else reward).double().sum()),
               done_frac=float(((st["terminated"] | st["truncated"]) if stepwise else (done > 0)).float().mean()),
               episodes=int(st["episode"].max()))
    assert np.isfinite(chk["reward_sum"])
    fixed = [0.0, 0.0, 0.0]
    if args.check_fixed:
        fixed = fixed_check(cfg, dw, B, A, actions, dev)
        chk["fixed"] = dict(reward_bits=int(fixed[0]), done_sum=int(fixed[1]), x_bits=int(fixed[2]))
    if dist is not None:
        # the "host gather" of per-shard results: a few numbers per rank through all_gather (no pickling, so it works
        # the same over RCCL and gloo), assembled on the host
        mine = torch.tensor([chk["reward_sum"], chk["done_frac"], float(chk["episodes"]), float(cfg.env_base)] + [float(v) for v in fixed],
                            dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        parts = [torch.empty_like(mine) for _ in range(world_size)]
        dist.all_gather(parts, mine)
        chk = [dict(rank=i, env_base=int(p[3]), reward_sum=float(p[0]), done_frac=float(p[1]), episodes=int(p[2]),
                    **({"fixed": dict(reward_bits=int(p[4]), done_sum=int(p[5]), x_bits=int(p[6]))} if args.check_fixed else {}))
               for i, p in enumerate(parts)]
    if rank == 0:
        bpes = bytes_per_env_step(args.config, A)
        env_steps = B * n * total
        # dominant kernel: rollout = tde::env_rollout_trio_kernel<A> (drive + two judge wavefronts per 64 agent slots;
        # --rollout-kernel duo|solo force the two- / one-wavefront forms), one launch per <= CH timesteps;
        # step mode = tde::env_step_kernel<A>; config 5 = step + tde::render_views_kernel per timestep
        if args.config == 5:
            kernel = f"tde::env_step_kernel<{A}> + tde::render_views_kernel<64>"
            if streams:
                kernel += f" as {n_streams} sub-batches on {n_streams} streams (tde_env_step_render)"
        elif stepwise:
            trio = st["slot_cache"] is not None and A in (8, 16, 32) and (
                args.step_kernel == "trio" or (args.step_kernel is None and B * A <= 131072))
            kernel = f"tde::env_step_trio_kernel<{A}, false, false, false>" if trio else f"tde::env_step_kernel<{A}, false, false>"
        else:
            team = {"solo": "", "duo": "_duo", "trio": "_trio"}.get(
                args.rollout_kernel, "_trio" if A in (8, 16, 32) else "_duo")
            # template arguments: <A, LIGHTS, BIG> for the two- / three-role kernels (BIG = large-grid world), <A, LIGHTS> for the one-role one
            big = "true" if args.world == "town" else "false"
            kernel = f"tde::env_rollout{team}_kernel<{A}, false>" if team == "" else f"tde::env_rollout{team}_kernel<{A}, false, {big}>"
        # per-step-equivalent durations of the timed launches (a launch of k steps: its duration / k * spl)
        per = sorted(d / k * spl for d, k in zip(dur_us, lens)) if not stepwise else sorted(d / k for d, k in zip(dur_us, lens))
        kern_us = sum(dur_us) / max(1, sum(lens)) * spl               # average duration per launch of `spl` steps
        alg_launch = float(bpes * B * spl)
        achieved = alg_launch / (kern_us * 1e-6) / 1e9
        # HBM traffic per launch: NOT measured in this run (PMC counters need rocprofv3) - the figure of the builder's own
        # separate --pmc passes over the same shape (profiles/traffic.json), labelled as such
        traffic = traffic_source = None
        issue = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.config == 3 and not stepwise and args.world == "junctions":
            try:
                tj = json.load(open(tpath))
                full = [k for k in lens if k == CH]
                if tj.get("steps_per_launch") == CH and tj.get("envs") == B and len(full) >= len(lens) - 1 and full:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/traffic.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)"
                    # the bound that describes this kernel: VALU issue.  insts per 64-slot group and step from the
                    # builder's SQ_INSTS_VALU pass; a SIMD issues one wave64 VALU instruction per 2 cycles
                    vi = tj.get("valu_insts_per_group_step")
                    if vi:
                        groups_per_simd = B * A / 64.0 / 1024.0
                        floor_us = vi * groups_per_simd * 2.0 / (tj.get("clock_ghz", 2.4) * 1e3)
                        issue = {"insts_per_group_step": vi, "groups_per_simd": groups_per_simd, "cycles_per_inst": 2,
                                 "clock_ghz": tj.get("clock_ghz", 2.4), "floor_us_per_step": floor_us,
                                 "frac": floor_us / (kern_us / spl),
                                 "source": "profiles/traffic.json (builder's SQ_INSTS_VALU pass)"}
            except Exception:
                traffic = None
        t5path = os.path.join(ROOT, "profiles", "traffic_config5.json")
        if args.config == 5 and os.path.exists(t5path) and args.world == "junctions":
            try:
                t5 = json.load(open(t5path))
                if t5.get("envs") == B and t5.get("agents") == A:
                    traffic = t5.get("hbm_bytes_per_timestep")
                    traffic_source = ("profiles/traffic_config5.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the two "
                                      "kernels at this shape, summed per timestep; not this run)")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec", "value": env_steps / wall, "unit": "env-steps/s",
            "agent_steps_per_sec": env_steps * A / wall,
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "repeats": repeats,
            "ms_per_step": wall * 1e3 / total,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{C['name']}: {B} envs x {A} agents per GPU, {C['what']}" +
                                   (" [town map: 1 km^2, 5.7e4 triangles, 256 scenarios]" if args.world == "town" else ""),
                       "world": args.world,
                       "envs_per_gpu": B, "agents_per_env": A, "global_envs": B * n, "mode": args.mode,
                       "binding": args.binding,
                       "streams": max(1, n_streams), "steps_per_call": spl, "timed_steps": total, "untimed_warmup_steps": warm_steps + CH,
                       "sharding": f"{n} contiguous shard(s) of one global batch (env_base = rank * {B}), "
                                   "no data-path collective",
                       "timing_backend": (args.backend if dist is not None else None)},
            # `achieved` / `peak` / `frac` are the contract's figures: ALGORITHMIC bytes over the kernel's duration against the HBM
            # peak.  For the register-resident rollout kernel the bytes that really cross the HBM interface are a small fraction
            # of the algorithmic ones (`traffic`), so what bounds it is VALU issue (`valu_issue`): the label says so.
            "roofline": {"bound": ("valu_issue" if issue else "hbm"), "bound_of_achieved": "hbm (algorithmic bytes / duration vs the 8 TB/s peak)",
                         "achieved": achieved, "algorithmic_GBps": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "measured_hbm_GBps": (traffic / (kern_us * 1e-6) / 1e9 if traffic else None),
                         "valu_issue": issue, "kernel": kernel,
                         "kernel_avg_us": kern_us, "kernel_min_us": per[0] if per else None,
                         "kernel_median_us": per[len(per) // 2] if per else None,
                         "launches": (sum(lens) if stepwise else len(lens)), "steps_per_launch": spl,
                         "launch_lengths": ([1] if stepwise else sorted(set(lens))), "algorithmic_bytes_per_launch": alg_launch,
                         "bytes_per_env_step": bpes, "us_per_step": dev_ms * 1e3 / total,
                         "streams": max(1, n_streams),
                         "timer": ("HIP events on the launch stream around every timed launch (rank 0)" if not streams else
                                   "HIP events on the first sub-batch's stream around every timed timestep (rank 0): the "
                                   "sub-batches' kernels overlap, so a timestep's duration is the period of that stream; "
                                   "us_per_step is taken from the first event to an event recorded after all sub-batch "
                                   "streams were joined")},
            "check": chk,
            # rank start-up (MAX over ranks): process start -> first launch, and the part of it spent on the world tables; the
            # tables are built once per machine (by the parent of `--gpus N`, else by the first rank to ask) and loaded by the rest
            "startup": {"to_first_launch_s": startup_s, "world_tables_s": world_s, "ranks_that_built_the_world": int(built)},
        }
        if n == 1 and args.config == 3 and not stepwise and not args.no_secondary and not args.force_dist:
            out["secondary"] = secondary(dev)
            # the other operating points, compact, INSIDE `roofline` (what the driver's record keeps): us per timestep and the
            # algorithmic-bytes fraction of the 8 TB/s peak, same definition as roofline.frac.  All with the default flags (NPCs
            # acting from an episode's first step) unless named otherwise; `secondary` holds the full entries.
            sec = out["secondary"]

            def pt(key):
                e = sec.get(key) or {}
                return ({"us": round(e["us_per_step"], 3), "frac": round(e["roofline"]["frac"], 4)} if "us_per_step" in e else {"error": e.get("error", "missing")})
            out["roofline"]["points"] = {
                "headline": {"us": round(kern_us / spl, 3), "frac": round(achieved / HBM_PEAK_GBPS, 4)},
                "coast_rule": pt("coast_rule"), "coast_rule_closed_loop": pt("coast_rule_closed_loop"),
                "closed_loop": pt("closed_loop"), "closed_loop_full": pt("closed_loop_full_outputs"),
                "config5": pt("config5"), "config5_one_stream": pt("config5_one_stream"), "town": pt("town"),
                "town_closed_loop": pt("town_closed_loop"), "town_config5": pt("town_config5"),
                "lights": pt("lights"),                                   # rollout, lights + first step
                "reference_semantics": pt("lights_closed_loop"),          # closed loop, lights + first step: the reference's step()
                "reference_semantics_full": pt("lights_closed_loop_full_outputs"),
                "agents_128": pt("agents_128"), "agents_128_closed_loop": pt("agents_128_closed_loop"),
                "agents_128_256_envs": pt("agents_128_256"), "agents_128_closed_loop_256_envs": pt("agents_128_closed_loop_256"),
                "note": "us per timestep / algorithmic bytes over the 8 TB/s peak; default flags = TDE_F_ALL (NPCs act from step one); "
                        "reference_semantics = traffic lights + first step, closed loop"}
            try:
                out["secondary"]["python_boundary"] = python_boundary()
            except Exception as exc:                                   # pragma: no cover
                out["secondary"]["python_boundary"] = {"error": repr(exc)}
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, base_cfg, A)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
