"""SURVEY §8(d) metric (ii) / (iii): end-to-end wall time of one step through the Python boundary,
(ii) with device-resident outputs (BatchedWaypointEnv.step), (iii) with numpy outputs (the SB3-shaped vec_step path,
one D2H copy of obs/reward/done/info per step).  obs_mode "state" (8 floats per env) and "birdview" (3x64x64 u8)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from torchdriveenv_amd.config import EnvConfig
from torchdriveenv_amd.env import BatchedWaypointEnv
from torchdriveenv_amd.synth import synthetic_world

if __name__ == "__main__":     # (ShardedBatchedEnv's spawned workers import this module again)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    world = synthetic_world(n_scn=64, A=16, seed=0, n_maps=4)
    out = {"envs": B, "agents": 16, "steps": N}
    for mode, fs in (("state", 1), ("birdview", 1), ("birdview", 3)):
        env = BatchedWaypointEnv(EnvConfig(seed=3), world, num_envs=B, obs_mode=mode, with_info=True, frame_stack=fs)
        env.reset()
        act = torch.zeros(B, 2, device=env.torch_device)
        for _ in range(20):
            env.step(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            env.step(act)
        torch.cuda.synchronize()
        dt_dev = (time.perf_counter() - t0) / N
        act_np = np.zeros((B, 2), np.float32)
        n2 = max(10, N // 10)
        for _ in range(3):
            env.vec_step(act_np)
        t0 = time.perf_counter()
        for _ in range(n2):
            env.vec_step(act_np)
        dt_np = (time.perf_counter() - t0) / n2
        # numpy outputs without the fresh-array copy: views of a ring of three pinned buffers (WaypointVecEnv(copy_obs=False))
        env._vec = None
        venv = env.as_vec_env(copy_obs=False)
        for _ in range(3):
            venv.step(act_np)
        t0 = time.perf_counter()
        for _ in range(n2):
            venv.step(act_np)
        dt_np_view = (time.perf_counter() - t0) / n2
        env._vec = None
        # the same through the ctypes binding (closed-loop host floor)
        env_c = BatchedWaypointEnv(EnvConfig(seed=3), world, num_envs=B, obs_mode=mode, with_info=True, frame_stack=fs,
                                   binding="ctypes")
        env_c.reset()
        for _ in range(20):
            env_c.step(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            env_c.step(act)
        torch.cuda.synchronize()
        dt_ct = (time.perf_counter() - t0) / N
        del env_c
        # SB3 consumer pattern: every info entry is looked at (Monitor / _update_info_buffer read .get("episode"))
        t0 = time.perf_counter()
        for _ in range(3):
            o, r, d, infos = env.vec_step(act_np)
            n_ep = sum(1 for i in np.nonzero(d)[0] if infos[i].get("episode") is not None)
        dt_ep = (time.perf_counter() - t0) / 3
        out[mode + (f"_stack{fs}" if fs > 1 else "")] = {"device_outputs_us_per_step_ctypes": dt_ct * 1e6, "numpy_plus_done_infos_us_per_step": dt_ep * 1e6, "device_outputs_us_per_step": dt_dev * 1e6, "device_outputs_env_steps_per_s": B / dt_dev,
                     "numpy_outputs_us_per_step": dt_np * 1e6, "numpy_outputs_env_steps_per_s": B / dt_np,
                     "numpy_views_us_per_step": dt_np_view * 1e6, "numpy_views_env_steps_per_s": B / dt_np_view}
    # the batch as two shard processes on this GPU, gathered in page-locked shared host buffers (sharding.ShardedBatchedEnv)
    try:
        from torchdriveenv_amd.sharding import ShardedBatchedEnv

        for copy in (True, False):
            sh = ShardedBatchedEnv(EnvConfig(seed=3), world, total_envs=B, n_shards=2, devices=[0, 0], copy_obs=copy,
                                   obs_mode="birdview", with_info=True)
            sh.reset()
            act_np = np.zeros((B, 2), np.float32)
            for _ in range(3):
                sh.step(act_np)
            t0 = time.perf_counter()
            for _ in range(10):
                sh.step(act_np)
            dt = (time.perf_counter() - t0) / 10
            out[f"sharded_2proc_birdview_copy{int(copy)}"] = {"us_per_step": dt * 1e6, "env_steps_per_s": B / dt, "pinned": sh.pinned}
            sh.close()
    except Exception as exc:                                   # pragma: no cover
        out["sharded_2proc_birdview"] = {"error": repr(exc)}
    print(json.dumps(out))
