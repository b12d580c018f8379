"""Edge cases through the C-ABI on a GPU: degenerate sizes, absent agents, argument errors, NaN propagation,
graph capture (the entry points allocate nothing and only touch the stream they are given)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle  # noqa: E402
from torchdriveenv_amd import _abi, _lib, ops  # noqa: E402
from torchdriveenv_amd.state import EnvState  # noqa: E402
from torchdriveenv_amd.synth import synthetic_world  # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_single_env_single_agent_and_empty_batch():
    world = synthetic_world(n_scn=2, A=1, seed=0, n_maps=1)
    cfg = _abi.default_config(seed=4)
    dw = world.to_device(DEV)
    hs, ds = EnvState(1, 1), EnvState(1, 1, device=DEV)
    oracle.env_reset(cfg, world, hs)
    ops.env_reset(cfg, dw, ds)
    for t in range(40):
        hs["action"][...] = [0.8, 0.01]
        ds["action"].copy_(dev(hs["action"]))
        oracle.env_step(cfg, world, hs)
        ops.env_step(cfg, dw, ds)
    h, d = hs.host(), ds.host()
    for k in h:
        if k != "action":
            assert np.array_equal(h[k].view(np.uint8), d[k].view(np.uint8)), k
    # B = 0: every entry point is a no-op that reports success
    L = _lib.load()
    st0 = _abi.TdeState()
    st0.B, st0.A = 0, 1
    assert L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(st0), None) == 0
    assert L.tde_env_reset(C.byref(cfg), C.byref(dw.struct), C.byref(st0), None, None) == 0
    assert L.tde_compute_collision(0, 4, None, None, None, None, None, None, None, None) == 0


def test_all_npcs_absent_matches_oracle(small_world):
    cfg = _abi.default_config(seed=6)
    dw = small_world.to_device(DEV)
    B, A = 37, 16
    hs, ds = EnvState(B, A), EnvState(B, A, device=DEV)
    oracle.env_reset(cfg, small_world, hs)
    ops.env_reset(cfg, dw, ds)
    mask = np.zeros(B * A, np.uint8)
    mask[::A] = 1                                       # only the egos stay
    hs["present"][...] = mask
    ds["present"].copy_(dev(mask))
    cfg2 = _abi.default_config(seed=6, flags=_abi.F_ALL & ~_abi.F_AUTORESET)
    rng = np.random.default_rng(0)
    for t in range(30):
        act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(dev(act))
        oracle.env_step(cfg2, small_world, hs)
        ops.env_step(cfg2, dw, ds)
    h, d = hs.host(), ds.host()
    for k in h:
        if k != "action":
            assert np.array_equal(h[k].view(np.uint8), d[k].view(np.uint8)), k
    assert h["collided"].sum() == 0


def test_argument_errors_are_reported_not_fatal(small_world):
    L = _lib.load()
    cfg = _abi.default_config()
    dw = small_world.to_device(DEV)
    st = EnvState(4, 16, device=DEV)
    bad = _abi.TdeState.from_buffer_copy(st.struct)
    bad.A = 12                                           # not a power of two
    rc = L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(bad), None)
    assert rc != 0 and b"power of two" in L.tde_last_error()
    bad.A = 8                                            # does not match the world tables
    rc = L.tde_env_step(C.byref(cfg), C.byref(dw.struct), C.byref(bad), None)
    assert rc != 0 and b"world.A" in L.tde_last_error()
    assert L.tde_env_step(None, C.byref(dw.struct), C.byref(st.struct), None) != 0
    with pytest.raises(_lib.TdeError):
        _lib.check(rc, "tde_env_step")
    with pytest.raises(ValueError):
        ops.env_rollout(cfg, dw, st, torch.zeros(3, 5, 2, device=DEV))      # wrong env count
    with pytest.raises(ValueError, match="contiguous"):
        ops.kinematics_step(*(torch.zeros(8, 2, device=DEV)[:, 0] for _ in range(5)), torch.zeros(8, 2, device=DEV))


def test_nan_action_propagates_to_that_env_only(small_world):
    cfg = _abi.default_config(seed=2, flags=_abi.F_ALL & ~_abi.F_AUTORESET)
    dw = small_world.to_device(DEV)
    B, A = 8, 16
    ds = EnvState(B, A, device=DEV)
    ops.env_reset(cfg, dw, ds)
    act = torch.zeros(B, 2, device=DEV)
    act[3, 0] = float("nan")
    ds["action"].copy_(act)
    ops.env_step(cfg, dw, ds)
    x = ds["x"].cpu().numpy().reshape(B, A)
    assert np.isnan(x[3, 0]) and np.isfinite(np.delete(x, 3, 0)).all() and np.isfinite(x[3, 1:]).all()


def test_step_and_render_are_graph_capturable(small_world):
    """hipGraph capture of a closed-loop timestep (step + render): nothing allocates or syncs inside the calls"""
    cfg = _abi.default_config(seed=9)
    dw = small_world.to_device(DEV)
    B, A = 256, 16
    ds, ref = EnvState(B, A, device=DEV), EnvState(B, A, device=DEV)
    ops.env_reset(cfg, dw, ds)
    ops.env_reset(cfg, dw, ref)
    img = torch.zeros(B, 3, 64, 64, dtype=torch.uint8, device=DEV)
    img_ref = torch.zeros_like(img)
    ds["action"][:, 0] = 0.5
    ref["action"][:, 0] = 0.5
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.env_step(cfg, dw, ds)                        # warm-up on the side stream
        ops.render_ego(cfg, dw, ds, out=img)
    torch.cuda.current_stream().wait_stream(s)
    ops.env_step(cfg, dw, ref)
    ops.render_ego(cfg, dw, ref, out=img_ref)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.env_step(cfg, dw, ds)
        ops.render_ego(cfg, dw, ds, out=img)
    for _ in range(20):
        g.replay()
        ops.env_step(cfg, dw, ref)
        ops.render_ego(cfg, dw, ref, out=img_ref)
    torch.cuda.synchronize()
    assert torch.equal(ds["x"], ref["x"]) and torch.equal(ds["steps"], ref["steps"]) and torch.equal(img, img_ref)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_on_one_gpu(launcher):
    """the N > 1 path of bench.py: started as plain `python bench.py --gpus 2` (bench.py spawns its own ranks, the
    driver's command shape) and under torch.distributed.run; barrier, MAX reduce, host gather, with two ranks sharing
    this box's single GPU over gloo; the JSON contract is checked on the one line rank 0 prints"""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "500", "--warmup", "250", "--backend", "gloo",
            "--envs", "2048"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 500 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["metric"] == "env-steps/sec" and j["value"] > 0 and len(j["check"]) == 2
    timed = j["steps"] * j["repeats"]
    assert j["config"]["timed_steps"] == timed
    assert abs(j["value"] - 2 * 2048 * timed / (j["ms_per_step"] * 1e-3 * timed)) / j["value"] < 1e-6
    # the two ranks are shards of ONE global batch: global env index bases 0 and 2048
    assert [c["env_base"] for c in j["check"]] == [0, 2048] and j["config"]["global_envs"] == 4096
    rf = j["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_min_us", "kernel_median_us"} <= set(rf)
    # self-consistent roofline block: bytes per launch = bytes per env-step x envs x steps per launch
    assert rf["algorithmic_bytes_per_launch"] == rf["bytes_per_env_step"] * 2048 * rf["steps_per_launch"]
    assert rf["bytes_per_env_step"] == 1014 and rf["steps_per_launch"] == 250 and rf["launch_lengths"] == [250]
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / rf["kernel_avg_us"] / 1e3) / rf["achieved"] < 1e-6


def test_bench_one_rank_over_rccl():
    """the RCCL leg of bench.py's N > 1 path on a one-GPU box: `--gpus 1 --force-dist --backend nccl` creates the process
    group (init_process_group("nccl", device_id=...)), and the barrier, the MAX all_reduce and the all_gather of the
    per-shard check sums run through RCCL with device tensors - once, here, before an 8-GPU node sees them.  A fresh
    child process (nothing in it has touched the GPU before the rendezvous)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl",
                          "--steps", "500", "--warmup", "250", "--envs", "2048", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["timing_backend"] == "nccl" and j["value"] > 0
    assert isinstance(j["check"], list) and len(j["check"]) == 1 and j["check"][0]["env_base"] == 0     # the gathered form
    assert "secondary" not in j


def test_bench_nccl_refuses_ranks_sharing_a_gpu():
    """--backend nccl (the default) must fail, not fall back to gloo, when the ranks cannot each have a GPU"""
    import os
    import subprocess
    import sys

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                          "--envs", "256"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode != 0 and "--backend gloo" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_sharded_hip_batch_equals_unsharded(small_world):
    """tde_config.env_base on the HIP path: two shards (env_base 0 and B/2) stepped by the kernels reproduce the
    unsharded HIP batch and the oracle bit for bit (contiguous env shards, reset RNG keyed by global env)"""
    from torchdriveenv_amd.sharding import shard_config

    B, A, K = 192, 16, 230
    base = _abi.default_config(seed=21, distance_cutoff=0.25)
    dw = small_world.to_device(DEV)
    rng = np.random.default_rng(4)
    actions = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
    full = EnvState(B, A, device=DEV)
    ops.env_reset(base, dw, full)
    fr, fd = ops.env_rollout(base, dw, full, dev(actions))
    hs = EnvState(B, A)
    oracle.env_reset(base, small_world, hs)
    hr, hd = oracle.env_rollout(base, small_world, hs, actions)
    assert np.array_equal(fr.cpu().numpy().view(np.uint32), hr.view(np.uint32)) and np.array_equal(fd.cpu().numpy(), hd)
    parts = []
    for rank in range(2):
        cfg, n = shard_config(base, rank, 2, B)
        assert cfg.env_base == rank * (B // 2) and n == B // 2
        st = EnvState(n, A, device=DEV)
        ops.env_reset(cfg, dw, st)
        r, d = ops.env_rollout(cfg, dw, st, dev(actions[:, rank * n:(rank + 1) * n]))
        parts.append((r.cpu().numpy(), d.cpu().numpy(), st.host()))
    assert np.array_equal(np.concatenate([p[0] for p in parts], 1).view(np.uint32), hr.view(np.uint32))
    assert np.array_equal(np.concatenate([p[1] for p in parts], 1), hd)
    want = full.host()
    for k in ("x", "y", "psi", "v", "episode", "scn", "steps", "target_idx", "collided", "offroad"):
        got = np.concatenate([p[2][k] for p in parts])
        assert np.array_equal(got.view(np.uint8), want[k].view(np.uint8)), k
    assert want["episode"].max() > 1


def test_render_other_sizes_and_bad_sizes(small_world):
    cfg = _abi.default_config(seed=1)
    dw = small_world.to_device(DEV)
    hs, ds = EnvState(5, 16), EnvState(5, 16, device=DEV)
    oracle.env_reset(cfg, small_world, hs)
    ops.env_reset(cfg, dw, ds)
    for (H, W, fov) in ((32, 32, 35.0), (64, 32, 20.0), (48, 64, 50.0), (36, 36, 35.0), (60, 64, 28.0), (8, 256, 90.0)):
        want = oracle.render_ego(cfg, small_world, hs, H=H, W=W, fov=fov)
        got = ops.render_ego(cfg, dw, ds, H=H, W=W, fov=fov).cpu().numpy()
        assert np.array_equal(got, want), (H, W, int((got != want).sum()))
    with pytest.raises(_lib.TdeError, match="multiples of 4"):
        ops.render_ego(cfg, dw, ds, H=30, W=30)
    with pytest.raises(_lib.TdeError, match="4096"):
        ops.render_ego(cfg, dw, ds, H=128, W=64)


def test_step_render_with_a_bad_render_request_advances_nothing(small_world):
    """tde_env_step_render validates the render request BEFORE its first launch: a failing call leaves every sub-batch at
    the timestep it was (round 3 launched sub-batch 0's step first and only then looked at H / W)"""
    cfg = _abi.default_config(seed=4)
    dw = small_world.to_device(DEV)
    B = 192
    ds = EnvState(B, 16, device=DEV)
    ops.env_reset(cfg, dw, ds)
    ds["action"][:, 0] = 0.7
    before = ds.host()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(3)]
    ops.fork_streams(streams, torch.device(DEV))
    for kw, msg in ((dict(H=84, W=84), "4096"), (dict(H=30, W=30), "multiples of 4"), (dict(phase=-1), "phase")):
        with pytest.raises(_lib.TdeError, match=msg):
            img = torch.zeros(B, 3, kw.get("H", 64), kw.get("W", 64), dtype=torch.uint8, device=DEV)
            ops.env_step_render(cfg, dw, ds, streams, out=img, **kw)
    ops.join_streams(streams, torch.device(DEV))
    torch.cuda.synchronize()
    after = ds.host()
    for k in ("x", "y", "psi", "v", "steps", "episode", "target_idx"):
        assert np.array_equal(before[k].view(np.uint8), after[k].view(np.uint8)), k


@pytest.mark.parametrize("H,W,n_stack", [(32, 32, 2), (48, 64, 4), (64, 64, 5)])
def test_frame_stack_other_sizes_ring_and_in_place(small_world, H, W, n_stack):
    """frame stack of other depths / image sizes, both ways (ring of layer planes, shifted in place) against the
    oracle's memmove; a cleared view starts again from blank (all-zero) frames like VecFrameStack after a reset"""
    cfg = _abi.default_config(seed=2)
    dw = small_world.to_device(DEV)
    B = 6
    hs, ds = EnvState(B, 16), EnvState(B, 16, device=DEV)
    oracle.env_reset(cfg, small_world, hs)
    ops.env_reset(cfg, dw, ds)
    stack = ops.FrameStack(B, n_stack, H, W, device=DEV)
    want = inplace = None
    rng = np.random.default_rng(0)
    for t in range(n_stack + 3):
        act = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.3, 0.3, B)], -1).astype(np.float32)
        hs["action"][...] = act
        ds["action"].copy_(torch.from_numpy(act).to(DEV))
        oracle.env_step(cfg, small_world, hs)
        ops.env_step(cfg, dw, ds)
        want = oracle.render_ego(cfg, small_world, hs, H=H, W=W, n_stack=n_stack, out=want)
        inplace = ops.render_ego(cfg, dw, ds, H=H, W=W, n_stack=n_stack, out=inplace)
        ring = stack.render(cfg, dw, ds)
        assert np.array_equal(inplace.cpu().numpy(), want), ("in place", t)
        assert np.array_equal(ring.cpu().numpy(), want), ("ring", t)
        if t == n_stack + 1:                               # one frame after the clear: only the newest is drawn
            assert (want[1, :-3] == 0).all() and want[1, -3:].any() and want[0, :3].any()
        if t == n_stack:                                   # clear views 1 and 4 on both sides
            m = torch.zeros(B, dtype=torch.bool, device=DEV)
            m[[1, 4]] = True
            stack.clear(m)
            want[[1, 4]] = 0
            inplace[m] = 0


@pytest.mark.parametrize("mode", ["solo", "duo", "trio"])
def test_every_rollout_kernel_matches_the_oracle(mode, small_world):
    """tde_env_rollout picks a one-, two- or three-wavefront kernel by group shape; tde_kernel_override forces one.  Each
    of them must reproduce the oracle bit for bit (the parity module's rollout tests, run under the forced form)."""
    from tests import test_gpu_parity as P

    _lib.kernel_override(rollout=mode)
    try:
        P.test_env_rollout_matches_oracle(small_world)
        for (A, K, lights) in ((8, 37, False), (16, 50, True), (32, 21, False)):
            P.test_env_rollout_other_agent_counts_and_window_lengths(A, K, lights)
    finally:
        _lib.kernel_override()


def test_configs3_partitioning_at_full_size_on_one_gpu():
    """BASELINE configs[3] is 65 536 envs x 16 agents sharded 8 192 per GPU over 8 GPUs.  One GPU can hold the whole
    batch, so the partitioning is checked at FULL size here: the eight shards (env_base = r * 8192, stepped one after the
    other on this GPU exactly as rank r would step them) reproduce the unsharded 65 536-env batch bit for bit, and a
    slice of the LAST shard equals the oracle run with the same global env indices."""
    from torchdriveenv_amd.sharding import shard_config

    world = synthetic_world(n_scn=64, A=16, seed=0, n_maps=4)
    base = _abi.default_config(seed=1000, distance_cutoff=0.25)            # bench.py's config
    N, Bs, A, K = 8, 8192, 16, 48
    B = N * Bs
    dw = world.to_device(DEV)
    g = torch.Generator().manual_seed(5)
    actions = torch.stack([torch.rand(K, B, generator=g) * 2 - 1, torch.rand(K, B, generator=g) * 0.6 - 0.3], -1).float()
    full = EnvState(B, A, device=DEV, with_info=False)
    ops.env_reset(base, dw, full)
    fr, fd = ops.env_rollout(base, dw, full, actions.contiguous().to(DEV))
    fr, fd = fr.cpu().numpy(), fd.cpu().numpy()
    fx, fep = full["x"].cpu().numpy(), full["episode"].cpu().numpy()
    for r in range(N):
        cfg, n = shard_config(base, r, N, B)
        assert n == Bs and cfg.env_base == r * Bs
        st = EnvState(Bs, A, device=DEV, with_info=False)
        ops.env_reset(cfg, dw, st)
        a = actions[:, r * Bs:(r + 1) * Bs].contiguous().to(DEV)
        sr, sd = ops.env_rollout(cfg, dw, st, a)
        lo, hi = r * Bs, (r + 1) * Bs
        assert np.array_equal(sr.cpu().numpy().view(np.uint32), fr[:, lo:hi].view(np.uint32)), r
        assert np.array_equal(sd.cpu().numpy(), fd[:, lo:hi]), r
        assert np.array_equal(st["x"].cpu().numpy().view(np.uint32), fx[lo * A:hi * A].view(np.uint32)), r
        assert np.array_equal(st["episode"].cpu().numpy(), fep[lo:hi]), r
    # the oracle on the first 96 envs of the last shard (global env indices 57344 ...)
    cfg, _ = shard_config(base, N - 1, N, B)
    SUB = 96
    hs = EnvState(SUB, A)
    oracle.env_reset(cfg, world, hs)
    lo = (N - 1) * Bs
    hr, hd = oracle.env_rollout(cfg, world, hs, np.ascontiguousarray(actions[:, lo:lo + SUB].numpy()))
    assert np.array_equal(hr.view(np.uint32), fr[:, lo:lo + SUB].view(np.uint32)) and np.array_equal(hd, fd[:, lo:lo + SUB])
    assert (fd & 1).sum() > 0 and fep.max() >= 2


@pytest.mark.parametrize("A,lights", [(16, True), (8, False), (32, True), (128, True), (128, False)])
def test_first_step_gap_cache_filled_and_missing_give_the_oracles_results(A, lights):
    """tde_world.first_gap (ABI 11): with the cache filled (tde_first_gaps; tde_env_step / tde_env_rollout fill it on first use) a
    re-spawned env's first NPC actions come from min(cached gap, exact test against the ego) - with its entries missing (zeroed behind
    the library's back, or never filled: another configuration's keys) from the whole controller.  Both equal the oracle bit for bit,
    in the three-role step kernel and the two- / three-role rollout kernels (128 slots: the two-role step kernel's `early` path, eight
    wavefronts per env at this batch size); the table itself: a keyed entry per NPC slot, 1e30 for a slot without a route."""
    from tests.test_gpu_parity import assert_state_equal, dev

    world = synthetic_world(n_scn=8, A=A, seed=5, n_maps=2)
    dw = world.to_device(DEV)
    B, K = (128 if A < 128 else 20), 70
    cfg = _abi.default_config(seed=21, flags=_abi.F_ALL | (_abi.F_TRAFFIC_LIGHTS if lights else 0), max_steps=25, distance_cutoff=0.25)
    rng = np.random.default_rng(2)
    acts = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    for t in range(K):
        hs["action"][...] = acts[t]
        oracle.env_step(cfg, world, hs)
    hr = EnvState(B, A)
    oracle.env_reset(cfg, world, hr)
    rew_o, done_o = oracle.env_rollout(cfg, world, hr, acts)
    assert int(hs["episode"].max()) > 2
    fg = dw.tensors["first_gap"]
    assert int(fg.to(torch.int64).sum()) == 0                              # uploaded empty
    ops.first_gaps(cfg, dw)
    tab = fg.cpu().numpy().reshape(world.n_scn, A, 2)
    assert (tab[:, 1:, 1] == tab[0, 1, 1]).all() and tab[0, 1, 1] & 1 and (tab[:, 0] == 0).all()      # one key, odd; the ego's slot unused
    gaps = tab[..., 0].copy().view(np.float32)
    routed = (world.arrays["spawn"].reshape(world.n_scn, A)["route"] >= 0) & (world.arrays["spawn"].reshape(world.n_scn, A)["present"] != 0)
    assert (gaps[:, 1:][~routed[:, 1:]] == np.float32(1e30)).all() and (gaps[:, 1:][routed[:, 1:]] < np.float32(1e30)).any()
    filled = fg.clone()
    try:
        for what in ("filled", "missing", "foreign keys"):
            if what == "missing":
                fg.zero_()                                       # (the library's memo says "filled": it does not fill again)
            elif what == "foreign keys":
                fg.copy_(filled.view(torch.int32).bitwise_xor(torch.tensor([0, 0x10], dtype=torch.int32, device=DEV)).view(torch.uint32))   # entries of "another configuration"
            _lib.kernel_override()
            ds = EnvState(B, A, device=DEV)
            ops.env_reset(cfg, dw, ds)
            for t in range(K):
                ops.env_step(cfg, dw, ds, action=dev(acts[t]))
            assert_state_equal(hs.host(), ds.host(), f"closed loop, first-step gap cache {what}")
            for form in ("duo", "trio"):
                _lib.kernel_override(rollout=form)
                dr = EnvState(B, A, device=DEV)
                ops.env_reset(cfg, dw, dr)
                rew_d, done_d = ops.env_rollout(cfg, dw, dr, dev(acts))
                assert np.array_equal(rew_d.cpu().numpy().view(np.uint32), rew_o.view(np.uint32)) and np.array_equal(done_d.cpu().numpy(), done_o), (what, form)
                assert_state_equal(hr.host(), dr.host(), f"rollout {form}, first-step gap cache {what}")
            if what != "filled":
                assert not torch.equal(fg, filled)               # nothing re-filled it behind the test's back
    finally:
        _lib.kernel_override()


@pytest.mark.parametrize("rule", ["acts", "coasts"])
@pytest.mark.parametrize("A,world_kw,lights", [(16, dict(n_scn=8, A=16, seed=0, n_maps=2), False), (8, dict(n_scn=8, A=8, seed=1, n_maps=2), True),
                                               (32, dict(n_scn=6, A=32, seed=2, n_maps=2), False), (4, dict(n_scn=6, A=4, seed=3, n_maps=2), False)])
def test_npc_first_step_flag_step_and_rollout_match_the_oracle(A, world_kw, lights, rule):
    """TDE_F_NPC_FIRST_STEP - the default since round 6: the NPC controller acts from the first step of an episode, as the reference's
    NPCs do (gym_env.py:285-294) - and its opt-out (the NPCs coast through step one), in EVERY kernel form: the three-role step kernel
    (a re-spawned env's first actions computed in the next launch's prologue) and the one-role one, the one- / two- / three-role
    rollout kernels (the role-split drivers run the controller again on the spawn rows of an env they re-spawned).  Bit for bit the
    oracle under the same flag through many re-spawns; and the two rules give different trajectories (the flag is not a no-op)."""
    from tests.test_gpu_parity import assert_state_equal, dev

    world = synthetic_world(**world_kw)
    dw = world.to_device(DEV)
    B, K = 192, 90
    base = (_abi.F_ALL & ~_abi.F_NPC_FIRST_STEP) | (_abi.F_TRAFFIC_LIGHTS if lights else 0)
    flags = base | (_abi.F_NPC_FIRST_STEP if rule == "acts" else 0)
    cfg = _abi.default_config(seed=14, flags=flags, max_steps=30, distance_cutoff=0.25)
    rng = np.random.default_rng(1)
    acts = np.stack([rng.uniform(-1, 1, (K, B)), rng.uniform(-0.3, 0.3, (K, B))], -1).astype(np.float32)
    hs = EnvState(B, A)
    oracle.env_reset(cfg, world, hs)
    for t in range(K):
        hs["action"][...] = acts[t]
        oracle.env_step(cfg, world, hs)
    assert int(hs["episode"].max()) > 2
    hr = EnvState(B, A)
    oracle.env_reset(cfg, world, hr)
    rew_o, done_o = oracle.env_rollout(cfg, world, hr, acts)
    try:
        # closed loop: the library's own choice (three roles at 8 / 16 / 32 slots with the caches) and the one-role kernel
        for form in (None, "solo"):
            _lib.kernel_override(step=form)
            ds = EnvState(B, A, device=DEV)
            ops.env_reset(cfg, dw, ds)
            for t in range(K):
                ops.env_step(cfg, dw, ds, action=dev(acts[t]))
            assert_state_equal(hs.host(), ds.host(), f"first-step rule {rule}, closed loop, step form {form}")
        _lib.kernel_override()
        # rollout: every form that exists at this slot count
        for form in (None, "solo", "duo") + (("trio",) if A in (8, 16, 32) else ()):
            _lib.kernel_override(rollout=form)
            dr = EnvState(B, A, device=DEV)
            ops.env_reset(cfg, dw, dr)
            rew_d, done_d = ops.env_rollout(cfg, dw, dr, dev(acts))
            assert np.array_equal(rew_d.cpu().numpy().view(np.uint32), rew_o.view(np.uint32)) and np.array_equal(done_d.cpu().numpy(), done_o), (rule, form)
            assert_state_equal(hr.host(), dr.host(), f"first-step rule {rule}, rollout form {form}")
    finally:
        _lib.kernel_override()
    # the other rule gives other trajectories (the flag is not a no-op)
    cfg0 = _abi.default_config(seed=14, flags=flags ^ _abi.F_NPC_FIRST_STEP, max_steps=30, distance_cutoff=0.25)
    d0 = EnvState(B, A, device=DEV)
    ops.env_reset(cfg0, dw, d0)
    ops.env_rollout(cfg0, dw, d0, dev(acts))
    assert not torch.equal(d0["x"], dr["x"])
