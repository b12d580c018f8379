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
