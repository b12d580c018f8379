"""The N = 8 shape of the multi-GPU path rehearsed on ONE GPU (BASELINE configs[3]: 65 536 envs x 16 agents as eight
shards of 8192; the reference's parallelism is one process per env, examples/rl_training.py:159):

  * `python bench.py --gpus 8 --backend gloo`: eight rank processes (here sharing the one device), the barrier, the MAX reduce and
    the host gather of the per-shard check sums - every rank's fixed check sums (a fresh reset + one 250-step rollout of its shard,
    env_base = rank * 8192) equal the sums over its columns of the UNSHARDED 65 536-env batch run in this process;
  * the world tables are built once (by the parent) and loaded by the ranks: no rank builds them, start-up is reported;
  * `ShardedBatchedEnv(n_shards=8)`: eight worker processes, birdview observations gathered in shared host memory, equal to the
    unsharded env's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.mark.timeout(900)
def test_bench_eight_ranks_on_one_gpu_equal_the_unsharded_batch(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    from torchdriveenv_amd import _abi, ops
    from torchdriveenv_amd.state import EnvState

    N, B, A, CH = 8, 8192, 16, bench.CH
    env = dict(os.environ, TDE_WORLD_CACHE=str(tmp_path / "worlds"))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(N), "--backend", "gloo", "--envs", str(B), "--steps", "250",
           "--warmup", "0", "--no-cpu-baseline", "--check-fixed"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == N and out["config"]["global_envs"] == N * B and out["config"]["timing_backend"] == "gloo"
    assert [c["env_base"] for c in out["check"]] == [r * B for r in range(N)]
    # the world tables: built once by the parent, loaded by every rank
    assert "world tables built" in p.stderr and out["startup"]["ranks_that_built_the_world"] == 0
    assert out["startup"]["world_tables_s"] < 5.0, out["startup"]           # (typically 0.03 s: a load; the bound only guards against a build)
    # the unsharded batch: 65 536 envs in this process, the action columns of shard r drawn as rank r draws them (seed = r)
    os.environ["TDE_WORLD_CACHE"] = str(tmp_path / "worlds")
    world, how, _ = bench.bench_world("junctions", A)
    assert how == "loaded"
    cfg = _abi.default_config(seed=1000, distance_cutoff=0.25, flags=_abi.F_ALL)
    dw = world.to_device(DEV)
    cols = []
    for r in range(N):
        g = torch.Generator(device="cpu").manual_seed(r)
        cols.append(torch.stack([torch.rand(CH, B, generator=g) * 2 - 1, torch.rand(CH, B, generator=g) * 0.6 - 0.3], -1))
    actions = torch.cat(cols, 1).to(torch.float32).contiguous().to(DEV)
    st = EnvState(N * B, A, device=DEV, with_info=False)
    ops.env_reset(cfg, dw, st)
    reward, done = ops.env_rollout(cfg, dw, st, actions)
    torch.cuda.synchronize()
    x = st["x"].view(N * B, A)
    for r in range(N):
        want = bench.fixed_sums(reward[:, r * B:(r + 1) * B], done[:, r * B:(r + 1) * B], x[r * B:(r + 1) * B])
        got = out["check"][r]["fixed"]
        assert [got["reward_bits"], got["done_sum"], got["x_bits"]] == want, (r, got, want)
    assert out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.timeout(900)
def test_sharded_env_eight_workers_birdview_equals_unsharded(small_world):
    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv
    from torchdriveenv_amd.sharding import ShardedBatchedEnv

    total = 8192
    cfg = EnvConfig(seed=4, distance_cutoff=0.25)
    kw = dict(agents_per_env=16, obs_mode="birdview", frame_stack=1)
    sh = ShardedBatchedEnv(cfg, small_world, total_envs=total, n_shards=8, devices=[0] * 8, copy_obs=False, **kw)
    try:
        assert sh.n_shards == 8 and [hi - lo for lo, hi in sh.ranges] == [1024] * 8
        assert sh.startup["world_load_s_max"] < 5.0, sh.startup          # workers load the tables the parent saved once (typically 0.05 s)
        ref = BatchedWaypointEnv(cfg, small_world, num_envs=total, **kw).as_vec_env()
        o0, o1 = sh.reset(), ref.reset()
        assert np.array_equal(o0, o1)
        rng = np.random.default_rng(0)
        for t in range(12):
            act = np.stack([rng.uniform(-1, 1, total), rng.uniform(-0.3, 0.3, total)], -1).astype(np.float32)
            a0, r0, d0, i0 = sh.step(act)
            a1, r1, d1, i1 = ref.step(act)
            assert np.array_equal(a0, a1) and np.array_equal(r0, r1) and np.array_equal(d0, d1), t
            assert np.array_equal(i0.column("offroad"), i1.column("offroad")) and np.array_equal(i0.column("collision"), i1.column("collision"))
        tm = sh.last_step_timing
        assert tm["gather_s"] >= 0 and tm["step_s"] > 0 and tm["obs_bytes"] == total * 3 * 64 * 64
    finally:
        sh.close()
