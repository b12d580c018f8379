"""Mutable env/agent state buffers (struct-of-arrays, env-major) and the `tde_state` struct that points at them.

Reference counterpart: the tensors a torchdrivesim `Simulator` holds after `build_simulator`
(ref gym_env.py:241-247: agent_states (B,A,4), agent_attributes (B,A,3)) plus the Python-side counters of
`WaypointSuiteEnv` (ref gym_env.py:325-339: current_target_idx, reached_waypoint_num, environment_steps).
"""
import numpy as np

from . import _abi


# per-env OUTPUTS of a step: on a device they share one byte arena, so the SB3-shaped numpy path fetches all of them with
# ONE device-to-host copy into a pinned buffer (EnvState.fetch_outputs) instead of one synchronising .cpu() per array
OUTPUT_KEYS = ["reward", "terminated", "truncated", "tl_violation", "done_bits", "info", "info_reached", "ep_final",
               "ep_final_len", "magnitudes"]


class EnvState:
    """`arrays[name]` are numpy arrays (host; used with the CPU oracle in tests) or torch tensors (device)."""

    def __init__(self, B, A, device=None, with_info=True, with_obs=False, with_episode=None, with_cache=None,
                 with_magnitudes=None):
        assert A >= 1 and (A & (A - 1)) == 0 and A <= _abi.TDE_MAX_AGENTS, f"A must be a power of two <= {_abi.TDE_MAX_AGENTS}"
        self.B, self.A, self.device = int(B), int(A), device
        with_episode = with_info if with_episode is None else with_episode
        shapes = _abi.state_shapes(B, A)
        off_keys = set()
        if not with_info:
            off_keys |= {"info", "info_reached", "done_bits"}
        if not with_obs:
            off_keys.add("obs")
        if not with_episode:
            off_keys |= {"ep_return", "ep_final", "ep_final_len"}
        # the magnitudes of the ego's infractions (tde_state.magnitudes: what the reference's info["offroad" | "collision"] hold)
        if not (with_info if with_magnitudes is None else with_magnitudes):
            off_keys.add("magnitudes")
        # the step's lookup caches (tde_state.slot_cache / env_cache): device-side only (the oracle has no use for them)
        if not (with_cache if with_cache is not None else device is not None):
            off_keys |= {"slot_cache", "env_cache", "act_cache"}
        self.arrays = {}
        self._arena = self._pinned = None
        self._slots = {}
        if device is None:
            for n, sh in shapes.items():
                self.arrays[n] = None if n in off_keys else np.zeros(sh, dtype=_abi.STATE_DTYPES[n])
        else:
            import torch

            off = 0
            for n in OUTPUT_KEYS:
                if n in off_keys:
                    continue
                nbytes = int(np.prod(shapes[n])) * np.dtype(_abi.STATE_DTYPES[n]).itemsize
                self._slots[n] = (off, nbytes)
                off += (nbytes + 15) // 16 * 16
            self._arena = torch.zeros(max(off, 16), dtype=torch.uint8, device=device)
            for n, sh in shapes.items():
                if n in off_keys:
                    self.arrays[n] = None
                    continue
                dt = getattr(torch, np.dtype(_abi.STATE_DTYPES[n]).name)
                if n in self._slots:
                    o, nb = self._slots[n]
                    self.arrays[n] = self._arena[o:o + nb].view(dt).view(sh)
                else:
                    self.arrays[n] = torch.zeros(sh, dtype=dt, device=device)
        self._invalidate_act_keys()
        self.struct = _abi.fill_state_struct(self.arrays, B, A)

    def fetch_outputs(self):
        """dict of numpy views of every per-env output after ONE device-to-host copy (pinned staging buffer, one
        stream synchronisation).  The views alias the staging buffer: they are overwritten by the next fetch."""
        import torch

        self.copy_outputs_async(ring=1)
        torch.cuda.current_stream(self._arena.device).synchronize()
        return self.outputs_views()

    def copy_outputs_async(self, ring=1):
        """queue the ONE device-to-host copy of the output arena on the current stream, into the next of `ring` pinned staging
        buffers, WITHOUT synchronising: the caller queues what else the step needs (the re-spawn of the finished envs, the
        observation copies) behind it and synchronises once; outputs_views() then gives the numpy views.  With ring > 1 the views of
        a fetch stay valid until ring - 1 further fetches have been made."""
        import torch

        if self._pinned is None or len(self._pinned) != ring:
            self._pinned = [torch.empty(self._arena.shape, dtype=torch.uint8, pin_memory=True) for _ in range(ring)]
            self._hosts = [p.numpy() for p in self._pinned]
            self._turn = 0
            shapes = _abi.state_shapes(self.B, self.A)
            # (the views are formed once per staging buffer: a dict comprehension of numpy views per step was 6 us of the step)
            self._views = [{n: h[o:o + nb].view(_abi.STATE_DTYPES[n]).reshape(shapes[n]) for n, (o, nb) in self._slots.items()}
                           for h in self._hosts]
        self._turn = (self._turn + 1) % ring
        self._pinned[self._turn].copy_(self._arena, non_blocking=True)

    def outputs_views(self):
        """numpy views of the staging buffer the last copy_outputs_async filled (after the caller's synchronisation)"""
        return self._views[self._turn]

    def __getitem__(self, k):
        return self.arrays[k]

    def host(self):
        """dict of numpy copies"""
        out = {}
        for n, a in self.arrays.items():
            if a is None:
                continue
            out[n] = a.copy() if isinstance(a, np.ndarray) else a.detach().cpu().numpy()
        return out

    def _invalidate_act_keys(self):
        """the per-env key entries of the action cache (tde_act_cache: entry A of every group of A + 1): episode < 0 = invalid"""
        a = self.arrays.get("act_cache")
        if a is not None:
            a.reshape(self.B, self.A + 1, 2)[:, self.A, 0] = -1

    def invalidate_caches(self):
        """forget the step's lookup / action caches (call after editing state arrays by hand: the action cache is keyed by
        the episode / step counters only)"""
        for n in ("slot_cache", "env_cache", "act_cache"):
            a = self.arrays.get(n)
            if a is not None:
                a.fill(0) if isinstance(a, np.ndarray) else a.zero_()
        self._invalidate_act_keys()

    def load(self, host_arrays):
        """overwrite from a dict of numpy arrays (e.g. another state's .host()); the caches are invalidated"""
        self.invalidate_caches()
        for n, a in host_arrays.items():
            dst = self.arrays.get(n)
            if dst is None or n in ("slot_cache", "env_cache", "act_cache"):
                continue       # (a dict taken from a device state's .host() carries its caches: they are keyed by counters
                               #  only and would come back "valid" for the poses the caller has just edited)
            if isinstance(dst, np.ndarray):
                dst[...] = a
            else:
                import torch

                dst.copy_(torch.from_numpy(np.ascontiguousarray(a)))
