"""Mutable env/agent state buffers (struct-of-arrays, env-major) and the `tde_state` struct that points at them.

Reference counterpart: the tensors a torchdrivesim `Simulator` holds after `build_simulator`
(ref gym_env.py:241-247: agent_states (B,A,4), agent_attributes (B,A,3)) plus the Python-side counters of
`WaypointSuiteEnv` (ref gym_env.py:325-339: current_target_idx, reached_waypoint_num, environment_steps).
"""
import numpy as np

from . import _abi


class EnvState:
    """`arrays[name]` are numpy arrays (host; used with the CPU oracle in tests) or torch tensors (device)."""

    def __init__(self, B, A, device=None, with_info=True, with_obs=False):
        assert A >= 1 and (A & (A - 1)) == 0 and A <= _abi.TDE_MAX_AGENTS, "A must be a power of two <= 64"
        self.B, self.A, self.device = int(B), int(A), device
        shapes = _abi.state_shapes(B, A)
        self.arrays = {}
        if device is None:
            for n, sh in shapes.items():
                self.arrays[n] = np.zeros(sh, dtype=_abi.STATE_DTYPES[n])
        else:
            import torch

            for n, sh in shapes.items():
                dt = getattr(torch, np.dtype(_abi.STATE_DTYPES[n]).name)
                self.arrays[n] = torch.zeros(sh, dtype=dt, device=device)
        if not with_info:
            self.arrays["info"] = None
            self.arrays["info_reached"] = None
            self.arrays["done_bits"] = None
        if not with_obs:
            self.arrays["obs"] = None
        self.struct = _abi.fill_state_struct(self.arrays, B, A)

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

    def load(self, host_arrays):
        """overwrite from a dict of numpy arrays (e.g. another state's .host())"""
        for n, a in host_arrays.items():
            dst = self.arrays.get(n)
            if dst is None:
                continue
            if isinstance(dst, np.ndarray):
                dst[...] = a
            else:
                import torch

                dst.copy_(torch.from_numpy(np.ascontiguousarray(a)))
