"""Multi-GPU layout of the batched env: envs are independent (each reference env owns its simulator, ref
gym_env.py:341-347), so a batch is cut into contiguous shards, one process per GPU, with NO data-path collective.
Static world tables are replicated; the reset RNG is keyed by the GLOBAL env index (tde_config.env_base), so the
sharded batch replays exactly the episodes of the unsharded one.  Results are gathered on the host."""


def shard_range(rank, world_size, total_envs):
    """contiguous [lo, hi) env range of `rank`; remainders go to the first ranks"""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_config(cfg, rank, world_size, total_envs):
    """copy of a tde_config for this rank's shard + its env count"""
    import copy

    lo, hi = shard_range(rank, world_size, total_envs)
    c = copy.copy(cfg)
    c.env_base = int(cfg.env_base) + lo
    return c, hi - lo


def host_gather(obj, dst=0):
    """gather one picklable per-shard result on rank `dst` (list ordered by rank); plain return when not distributed"""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


# ------------------------------------------------------------------------------------------------------------------
# ShardedBatchedEnv: the batch cut into contiguous shards, one worker PROCESS per shard / GPU, host gather of the
# results (reference analogue: SubprocVecEnv with one process per ENV, examples/rl_training.py:159).
# ------------------------------------------------------------------------------------------------------------------
def _shard_worker(rank, n_shards, device, cfg, world, total, kw, conn, shm):
    """one shard: its own process, its own GPU context, its own BatchedWaypointEnv with env_base = first global env"""
    import time

    import numpy as np
    import torch

    from .env import BatchedWaypointEnv

    try:
        t0 = time.perf_counter()
        if isinstance(world, str):                        # the tables the parent saved once (World.save): a read of the page cache
            from .world import World
            world = World.load(world)
        t_world = time.perf_counter() - t0
        torch.cuda.set_device(device)
        lo, hi = shard_range(rank, n_shards, total)
        env = BatchedWaypointEnv(cfg, world, num_envs=hi - lo, device=f"cuda:{device}", env_base=lo, **kw)
        # The shard's observations land directly in its slice of the gathered buffer: the shared segment is page-locked
        # in this process (hipHostRegister), so the device-to-host copy is ONE pass over the bytes (a pageable target
        # costs a staging copy inside the runtime, and the former pinned ring + memcpy a second pass over 100 MB per
        # shard and step for birdview observations).  Registration failing (rlimit) only loses the speed.
        mine = shm["obs"][lo:hi]
        pinned = page_lock(mine)
        venv = env.as_vec_env(copy_obs=False, obs_buffers=[mine])
        rew_all, done_all = shm["reward"].numpy(), shm["done"].numpy()
        conn.send(("ready", lo, hi, pinned, t_world, time.perf_counter() - t0))
        while True:
            cmd, arg = conn.recv()
            if cmd == "reset":
                venv.reset()                              # (written in place: the ring is the shared slice)
                conn.send(("ok", None))
            elif cmd == "step":
                obs, rew, done, infos = venv.step(shm["action"].numpy()[lo:hi])
                rew_all[lo:hi] = rew
                done_all[lo:hi] = done
                cols = {k: np.asarray(v) for k, v in infos.columns.items()}
                conn.send(("ok", (cols, {lo + i: ex for i, ex in infos.terminal.items()})))
            elif cmd == "state":
                conn.send(("ok", {k: v.cpu().numpy() for k, v in env.state.arrays.items() if v is not None}))
            elif cmd == "close":
                conn.send(("ok", None))
                break
    except Exception as exc:                              # pragma: no cover
        import traceback

        conn.send(("error", f"shard {rank}: {exc}\n{traceback.format_exc()}"))


def page_lock(t):
    """hipHostRegister the storage of a host tensor (so that copies from a device into it are true DMA); False if the
    runtime refuses"""
    import torch

    try:
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel() * t.element_size(), 0)
        return int(rc) == 0
    except Exception:
        return False


class ShardedBatchedEnv:
    """`total_envs` envs cut into `n_shards` contiguous shards, one process per shard (one per GPU; several shards may
    share a GPU), no data-path collective: the reset RNG is keyed by the GLOBAL env index (tde_config.env_base) and the seed is
    resolved once, here, so the sharded batch is bit for bit the unsharded one with that seed (`.config.seed`).  Every step the shards write their slices of the shared host
    buffers (observations, rewards, dones) - the host gather - and the caller gets VecEnv-shaped numpy results for the
    whole batch: step(actions [total, 2]) -> (obs, rewards, dones, infos)."""

    def __init__(self, cfg, world, total_envs, n_shards=None, devices=None, copy_obs=True, **env_kw):
        """copy_obs=False: reset / step hand out VIEWS of the gathered host buffers (valid until the next step) instead
        of fresh arrays - 100 MB per 8192 birdview envs and step that nobody has to copy"""
        import dataclasses

        import numpy as np
        import torch
        import torch.multiprocessing as mp

        from .env import BatchedWaypointEnv  # noqa: F401  (fail early if the package cannot load)

        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise RuntimeError("ShardedBatchedEnv needs at least one HIP device")
        self.n_shards = int(n_shards or ndev)
        if self.n_shards < 1 or self.n_shards > int(total_envs):
            raise ValueError(f"n_shards must be in [1, total_envs]: {self.n_shards} shards for {total_envs} envs")
        if cfg.seed is None:
            # one seed for the whole batch, drawn here: shards that each drew their own would neither reproduce the
            # unsharded batch nor each other's runs (ref helpers.py:39-41 draws one per process)
            cfg = dataclasses.replace(cfg, seed=int(np.random.randint(0, 2**31 - 1)))
        self.config = cfg
        self.copy_obs = bool(copy_obs)
        self.devices = list(devices) if devices is not None else [r % ndev for r in range(self.n_shards)]
        self.num_envs = int(total_envs)
        obs_mode = env_kw.get("obs_mode", "birdview")
        fs = max(1, int(env_kw.get("frame_stack", 1)))
        r = cfg.simulator.renderer
        oshape = (3 * fs, int(r.res), int(r.res)) if obs_mode == "birdview" else (8,)
        odt = torch.uint8 if obs_mode == "birdview" else torch.float32
        self._shm = {"obs": torch.zeros((self.num_envs,) + oshape, dtype=odt).share_memory_(),
                     "reward": torch.zeros(self.num_envs, dtype=torch.float32).share_memory_(),
                     "done": torch.zeros(self.num_envs, dtype=torch.bool).share_memory_(),
                     "action": torch.zeros((self.num_envs, 2), dtype=torch.float32).share_memory_()}
        # The static tables reach the workers through ONE file the parent writes (World.save) instead of a pickle per worker: a
        # town's tables are ~170 MB, and a World that every shard assembles for itself costs seconds of grid-index build per
        # process on the same host cores before the first launch
        import os
        import tempfile
        import time

        from .world import World
        t0 = time.perf_counter()
        world_arg, tmp = world, None
        if isinstance(world, World):
            shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
            fd, tmp = tempfile.mkstemp(prefix="tde_world_", suffix=".npz", dir=shm_dir)
            os.close(fd)
            world.save(tmp)
            world_arg = tmp
        t_save = time.perf_counter() - t0
        ctx = mp.get_context("spawn")
        self._conns, self._procs, self.ranges = [], [], []
        try:
            for rank in range(self.n_shards):
                parent, child = ctx.Pipe()
                p = ctx.Process(target=_shard_worker, args=(rank, self.n_shards, self.devices[rank], cfg, world_arg,
                                                            self.num_envs, env_kw, child, self._shm), daemon=True)
                p.start()
                self._conns.append(parent)
                self._procs.append(p)
            self.pinned = []
            loads, readies = [], []
            for c in self._conns:
                tag, lo, hi, pinned, t_world, t_ready = self._recv(c)
                self.ranges.append((lo, hi))
                self.pinned.append(bool(pinned))
                loads.append(t_world)
                readies.append(t_ready)
        finally:
            if tmp is not None:
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
        # start-up, seconds: the parent's one save, the slowest worker's load of it, the slowest worker from its first line to "ready"
        self.startup = {"world_save_s": t_save, "world_load_s_max": max(loads), "worker_ready_s_max": max(readies),
                        "total_s": time.perf_counter() - t0}
        self.last_step_timing = None
        self._np = np

    @staticmethod
    def _recv(conn):
        msg = conn.recv()
        if msg[0] == "error":
            raise RuntimeError(msg[1])
        return msg

    def _all(self, cmd, arg=None):
        for c in self._conns:
            c.send((cmd, arg))
        # every reply is read before an error is raised: a reply left in a pipe would be taken for the answer to the
        # next command (close() reading a stale "ok")
        msgs = [c.recv() for c in self._conns]
        errs = [m[1] for m in msgs if m[0] == "error"]
        if errs:
            raise RuntimeError("\n".join(errs))
        return [m[1] for m in msgs]

    def _obs(self):
        o = self._shm["obs"].numpy()
        return o.copy() if self.copy_obs else o

    def reset(self):
        self._all("reset")
        return self._obs()

    def step(self, actions):
        from .env import LazyInfos

        import time

        np = self._np
        t0 = time.perf_counter()
        self._shm["action"].numpy()[...] = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)
        parts = self._all("step")
        t1 = time.perf_counter()
        cols = {k: np.concatenate([p[0][k] for p in parts]) for k in parts[0][0]}
        terminal = {}
        for p in parts:
            terminal.update(p[1])
        out = (self._obs(), self._shm["reward"].numpy().copy(), self._shm["done"].numpy().copy(),
               LazyInfos(self.num_envs, cols, terminal))
        t2 = time.perf_counter()
        # step_s: until every shard has stepped and written its slices of the shared buffers (the observations' device-to-host
        # copies land there directly); gather_s: what the parent then does on the host - info columns, reward / done copies and,
        # with copy_obs, the copy of the gathered observations
        self.last_step_timing = {"step_s": t1 - t0, "gather_s": t2 - t1, "obs_bytes": int(self._shm["obs"].numel() * self._shm["obs"].element_size())}
        return out

    def gather_state(self):
        """every shard's state arrays concatenated in global env order (tests / checkpoints)"""
        np = self._np
        parts = self._all("state")
        return {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}

    def close(self):
        try:
            self._all("close")
        except Exception:
            pass
        for p in self._procs:
            p.join(timeout=10)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
