"""Python mirror of the operator-level interface the reference env calls on its simulator
(`SimulatorInterface` methods used at ref gym_env.py:117,127,142-144) plus the fused env-level entry points, over
libtde_hip.so.  Inputs are torch tensors on a HIP device (PyTorch is only the allocator / stream owner); every call
is asynchronous on torch's current stream.  Argument errors raise ValueError (the reference does no validation, ref
gym_env.py:115-120; we check dtype/shape/device/contiguity because a bad pointer is fatal on a GPU).
"""
import ctypes as C

import numpy as np
import torch

from . import _abi, _lib


def _call(dev, fn, *args):
    """invoke a C-ABI entry point with `dev` as the current HIP device (a stream is only valid on its own device);
    the guard costs nothing in the usual single-device-per-process layout"""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if torch.cuda.current_device() != idx:
        with torch.cuda.device(idx):
            return fn(*args)
    return fn(*args)


def _chk(t, dtype, n, name, device=None, optional=False):
    if t is None:
        if optional:
            return None
        raise ValueError(f"{name} is required")
    if not torch.is_tensor(t):
        raise ValueError(f"{name} must be a torch tensor")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on a HIP device (there is no CPU path)")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if n is not None and t.numel() != n:
        raise ValueError(f"{name} must have {n} elements, got {t.numel()}")
    return t.data_ptr()


def kinematics_step(x, y, psi, v, lr, action, present=None, dt=0.1):
    """KinematicBicycle.step for every agent, in place (ref gym_env.py:117; model :245-247). action [...,2]."""
    L = _lib.load()
    n, dev = x.numel(), x.device
    args = [_chk(t, torch.float32, n, nm, dev) for t, nm in ((x, "x"), (y, "y"), (psi, "psi"), (v, "v"), (lr, "lr"))]
    pp = _chk(present, torch.uint8, n, "present", dev, optional=True)
    pa = _chk(action, torch.float32, 2 * n, "action", dev)
    _lib.check(_call(dev, L.tde_kinematics_step, n, *args, pp, pa, dt, _lib.current_stream(dev)), "tde_kinematics_step")


def compute_collision(B, A, x, y, psi, length, width, present, out=None):
    """compute_collision() > 0 per agent (ref gym_env.py:143) -> uint8 [B*A]"""
    L = _lib.load()
    n, dev = B * A, x.device
    ptrs = [_chk(t, torch.float32, n, nm, dev) for t, nm in ((x, "x"), (y, "y"), (psi, "psi"), (length, "length"),
                                                            (width, "width"))]
    pp = _chk(present, torch.uint8, n, "present", dev)
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=dev)
    po = _chk(out, torch.uint8, n, "out", dev)
    _lib.check(_call(dev, L.tde_compute_collision, B, A, *ptrs, pp, po, _lib.current_stream(dev)), "tde_compute_collision")
    return out


def compute_offroad(B, A, x, y, psi, length, width, present, dworld, map_of_env, threshold=0.5, out=None):
    """compute_offroad() > 0 per agent (ref gym_env.py:142) -> uint8 [B*A]"""
    L = _lib.load()
    n, dev = B * A, x.device
    ptrs = [_chk(t, torch.float32, n, nm, dev) for t, nm in ((x, "x"), (y, "y"), (psi, "psi"), (length, "length"),
                                                            (width, "width"))]
    pp = _chk(present, torch.uint8, n, "present", dev)
    pm = _chk(map_of_env, torch.int32, B, "map_of_env", dev)
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=dev)
    po = _chk(out, torch.uint8, n, "out", dev)
    _lib.check(_call(dev, L.tde_compute_offroad, B, A, *ptrs, pp, C.byref(dworld.struct), pm, threshold, po,
                                     _lib.current_stream(dev)), "tde_compute_offroad")
    return out


def kin_collide_step(B, A, x, y, psi, v, lr, length, width, present, action, dt=0.1, out=None):
    """fused kinematics + collision for all agents (BASELINE configs[1]); action [B*A,2]"""
    L = _lib.load()
    n, dev = B * A, x.device
    ptrs = [_chk(t, torch.float32, n, nm, dev) for t, nm in ((x, "x"), (y, "y"), (psi, "psi"), (v, "v"), (lr, "lr"),
                                                            (length, "length"), (width, "width"))]
    pp = _chk(present, torch.uint8, n, "present", dev)
    pa = _chk(action, torch.float32, 2 * n, "action", dev)
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=dev)
    po = _chk(out, torch.uint8, n, "out", dev)
    _lib.check(_call(dev, L.tde_kin_collide_step, B, A, *ptrs, pp, pa, dt, po, _lib.current_stream(dev)), "tde_kin_collide_step")
    return out


def waypoint_reward(cfg, pre, post, offroad, collided, tl, wp_xy, wp_n, scn, steps, target_idx, reached,
                    with_info=True):
    """Reference-owned reward/termination logic (ref gym_env.py:391-437) for n envs.
    pre/post: tuples of 4 float32 tensors [n] (x,y,psi,v).  steps/target_idx/reached int32 [n], updated in place."""
    L = _lib.load()
    n, dev = pre[0].numel(), pre[0].device
    p_pre = [_chk(t, torch.float32, n, f"pre[{i}]", dev) for i, t in enumerate(pre)]
    p_post = [_chk(t, torch.float32, n, f"post[{i}]", dev) for i, t in enumerate(post)]
    po = _chk(offroad, torch.uint8, n, "offroad", dev)
    pc = _chk(collided, torch.uint8, n, "collided", dev)
    pt = _chk(tl, torch.uint8, n, "tl", dev, optional=True)
    S, NW = wp_xy.shape[0], wp_xy.shape[1]
    pw = _chk(wp_xy, torch.float64, S * NW * 2, "wp_xy", dev)
    pn = _chk(wp_n, torch.int32, S, "wp_n", dev)
    ps = _chk(scn, torch.int32, n, "scn", dev)
    pst, pti, prc = (_chk(t, torch.int32, n, nm, dev) for t, nm in ((steps, "steps"), (target_idx, "target_idx"),
                                                                    (reached, "reached")))
    out = dict(reward=torch.empty(n, dtype=torch.float32, device=dev),
               terminated=torch.empty(n, dtype=torch.uint8, device=dev),
               truncated=torch.empty(n, dtype=torch.uint8, device=dev),
               info=torch.empty((n, 4), dtype=torch.float64, device=dev) if with_info else None,
               info_reached=torch.empty(n, dtype=torch.int32, device=dev) if with_info else None)
    _lib.check(_call(dev, L.tde_waypoint_reward, C.byref(cfg), n, *p_pre, *p_post, po, pc, pt, pw, pn, NW, ps, pst, pti, prc,
                                     out["reward"].data_ptr(), out["terminated"].data_ptr(),
                                     out["truncated"].data_ptr(),
                                     None if out["info"] is None else out["info"].data_ptr(),
                                     None if out["info_reached"] is None else out["info_reached"].data_ptr(),
                                     _lib.current_stream(dev)), "tde_waypoint_reward")
    return out


def env_reset(cfg, dworld, state, mask=None):
    L = _lib.load()
    pm = _chk(mask, torch.uint8, state.B, "mask", optional=True)
    _lib.check(_call(state.device, L.tde_env_reset, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct), pm,
                               _lib.current_stream(state.device)), "tde_env_reset")


def first_gaps(cfg, dworld):
    """tde_first_gaps: fill the device world's first-step gap cache for `cfg` now (tde_env_step / tde_env_rollout do it themselves
    on first use: only needed to choose when the launch happens)"""
    L = _lib.load()
    dev = next(iter(dworld.tensors.values())).device
    _lib.check(_call(dev, L.tde_first_gaps, C.byref(cfg), C.byref(dworld.struct), _lib.current_stream(dev)), "tde_first_gaps")


def env_step(cfg, dworld, state, action=None):
    """one timestep of every env.  `action` (float32 [B,2] on the device) is read in place if given, else
    state["action"] is used."""
    L = _lib.load()
    st = state.struct
    if action is not None:
        st = _abi.TdeState.from_buffer_copy(state.struct)
        st.action = _chk(action, torch.float32, 2 * state.B, "action", torch.device(state.device))
    _lib.check(_call(state.device, L.tde_env_step, C.byref(cfg), C.byref(dworld.struct), C.byref(st),
                              _lib.current_stream(state.device)), "tde_env_step")


def env_rollout(cfg, dworld, state, actions, reward=None, done=None):
    """actions float32 [K,B,2] on device -> (reward [K,B] f32, done [K,B] u8)"""
    L = _lib.load()
    K, B = actions.shape[0], actions.shape[1]
    dev = actions.device
    if B != state.B:
        raise ValueError(f"actions are for {B} envs, state has {state.B}")
    pa = _chk(actions, torch.float32, K * B * 2, "actions", dev)
    if reward is None:
        reward = torch.empty((K, B), dtype=torch.float32, device=dev)
    if done is None:
        done = torch.empty((K, B), dtype=torch.uint8, device=dev)
    ro = _abi.TdeRollout(pa, _chk(reward, torch.float32, K * B, "reward", dev), _chk(done, torch.uint8, K * B, "done", dev),
                         K, 0)
    _lib.check(_call(dev, L.tde_env_rollout, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct), C.byref(ro),
                                 _lib.current_stream(dev)), "tde_env_rollout")
    return reward, done


def ego_infractions(cfg, dworld, state, out=None):
    """tde_ego_infractions: float32 [B, 4] = the ego's (offroad, collision = sum of IoUs, number of overlapping agents, 0)
    MAGNITUDES of the current state - what the reference's info dict holds (ref gym_env.py:427-428) where the step path only
    needs `> 0`.  Call it after a step made
    WITHOUT TDE_F_AUTORESET and before the finished envs are re-spawned."""
    L = _lib.load()
    dev = state.device
    if out is None:
        out = torch.empty((state.B, 4), dtype=torch.float32, device=dev)
    _lib.check(_call(dev, L.tde_ego_infractions, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct),
                     _chk(out, torch.float32, 4 * state.B, "out", torch.device(dev)), _lib.current_stream(dev)), "tde_ego_infractions")
    return out


def env_post_step(cfg, dworld, state, magnitudes=None):
    """tde_env_post_step, after a step made WITHOUT TDE_F_AUTORESET: `magnitudes` (float32 [B, 4], optional) = ego_infractions of the
    state that step left, computed only for the envs it flagged; with TDE_F_AUTORESET in cfg.flags the envs it finished are
    re-spawned (their compact observation refreshed when the state carries one).  One launch.  Returns `magnitudes`."""
    L = _lib.load()
    dev = state.device
    p = None if magnitudes is None else _chk(magnitudes, torch.float32, 4 * state.B, "magnitudes", torch.device(dev))
    _lib.check(_call(dev, L.tde_env_post_step, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct), p, _lib.current_stream(dev)),
               "tde_env_post_step")
    return magnitudes


def state_obs(dworld, state, out=None):
    """compact kinematic observation of every ego -> float32 [B, 8] on device: x, y, psi, v, target waypoint offset in
    the ego frame (forward, left), target-exists flag, environment_steps"""
    L = _lib.load()
    dev = state.device
    if out is None:
        out = torch.empty((state.B, 8), dtype=torch.float32, device=dev)
    po = _chk(out, torch.float32, state.B * 8, "out", torch.device(dev))
    _lib.check(_call(dev, L.tde_state_obs, C.byref(dworld.struct), C.byref(state.struct), po, _lib.current_stream(dev)),
               "tde_state_obs")
    return out


def render_ego(cfg, dworld, state, H=64, W=64, fov=35.0, n_stack=1, out=None, layers=None, phase=0, flags=0,
               fresh=None, only=None):
    """render_egocentric() of every env's ego -> uint8 [B, 3*n_stack, H, W] on device (ref gym_env.py:122-124).
    Frame stack (n_stack > 1): `out` is the stack of the previous call.  With `layers` (uint8 [B, n_stack, H*W], see
    FrameStack) nothing is shifted: the ring of layer planes is expanded into all frames of `out`; without it the older
    frames are shifted in place by a launch of their own.
    flags: _abi.RENDER_LEFT_HANDED | _abi.RENDER_PLAIN_EGO; fresh / only: optional uint8 [B] device masks
    (tde_render.fresh: views whose episode just started get blank older frames; tde_render.only: render these views
    only, re-rendering their newest frame in place)."""
    L = _lib.load()
    ns = max(1, n_stack)
    dev = state.device
    if out is None:
        out = torch.zeros((state.B, 3 * ns, H, W), dtype=torch.uint8, device=dev)
    pl = _chk(layers, torch.uint8, state.B * ns * H * W, "layers", optional=True) if ns > 1 else None
    pf = _chk(fresh, torch.uint8, state.B, "fresh", optional=True)
    po = _chk(only, torch.uint8, state.B, "only", optional=True)
    rd = _abi.TdeRender(_chk(out, torch.uint8, state.B * 3 * ns * H * W, "out"), H, W, fov, n_stack, pl, int(phase),
                        int(flags), pf, po)
    _lib.check(_call(dev, L.tde_render_ego, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct), C.byref(rd),
                     _lib.current_stream(dev)), "tde_render_ego")
    return out


def fork_streams(streams, device=None):
    """order `streams` (torch.cuda.Stream) after the work already queued on the current stream (before the first
    env_step_render call / after the action tensor was produced)"""
    cur = torch.cuda.current_stream(device)
    for s in streams:
        s.wait_stream(cur)


def join_streams(streams, device=None):
    """order the current stream after the work queued on `streams` (before the outputs are consumed)"""
    cur = torch.cuda.current_stream(device)
    for s in streams:
        cur.wait_stream(s)


def env_reset_render(cfg, dworld, state, mask, out, H=64, W=64, fov=35.0, n_stack=1, layers=None, phase=0, flags=0):
    """tde_env_reset_render: masked reset + the re-spawned views' first observation in ONE call (their newest frame rendered in
    place, their older stack frames blanked); `phase` = the phase of the last full render.  Returns `out`."""
    L = _lib.load()
    dev = state.device
    ns = max(1, n_stack)
    pl = _chk(layers, torch.uint8, state.B * ns * H * W, "layers", optional=True) if ns > 1 else None
    rd = _abi.TdeRender(_chk(out, torch.uint8, state.B * 3 * ns * H * W, "out"), H, W, fov, n_stack, pl, int(phase), int(flags), None, None)
    _lib.check(_call(dev, L.tde_env_reset_render, C.byref(cfg), C.byref(dworld.struct), C.byref(state.struct),
                     _chk(mask, torch.uint8, state.B, "mask", torch.device(dev)), C.byref(rd), _lib.current_stream(dev)),
               "tde_env_reset_render")
    return out


def env_step_render(cfg, dworld, state, streams, action=None, out=None, H=64, W=64, fov=35.0, n_stack=1, layers=None, phase=0,
                    flags=0, fresh=None, render=True):
    """tde_env_step_render: one timestep + (render) the birdview of every env as len(streams) contiguous sub-batches, each on
    its own HIP stream, so that the step of one sub-batch overlaps the rasteriser of another.  Same results as env_step +
    render_ego.  `streams`: torch.cuda.Stream objects; the caller orders them against the current stream (fork_streams /
    join_streams) - open-loop drivers join once, at the end.  Returns `out` (None without render)."""
    L = _lib.load()
    dev = state.device
    st = state.struct
    if action is not None:
        st = _abi.TdeState.from_buffer_copy(state.struct)
        st.action = _chk(action, torch.float32, 2 * state.B, "action", torch.device(dev))
    rdp = None
    if render:
        ns = max(1, n_stack)
        if out is None:
            # allocated (and zero-filled) on the CURRENT stream: the side streams must see the fill before they write, and
            # the caching allocator must not hand the block out again while their kernels are pending.  Callers that pass
            # `out` / `layers` / `action` own that ordering (fork_streams before, join_streams or record_stream after).
            out = torch.zeros((state.B, 3 * ns, H, W), dtype=torch.uint8, device=dev)
            cur = torch.cuda.current_stream(torch.device(dev))
            for s_ in streams:
                s_.wait_stream(cur)
                out.record_stream(s_)
        pl = _chk(layers, torch.uint8, state.B * ns * H * W, "layers", optional=True) if ns > 1 else None
        pf = _chk(fresh, torch.uint8, state.B, "fresh", optional=True)
        rd = _abi.TdeRender(_chk(out, torch.uint8, state.B * 3 * ns * H * W, "out"), H, W, fov, n_stack, pl, int(phase),
                            int(flags), pf, None)
        rdp = C.byref(rd)
    arr = (C.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
    _lib.check(_call(dev, L.tde_env_step_render, C.byref(cfg), C.byref(dworld.struct), C.byref(st), rdp, arr,
                     len(streams)), "tde_env_step_render")
    return out


class FrameStack:
    """Device-side VecFrameStack(n_stack, channels_order="first") (ref examples/rl_training.py:160) kept as a ring of
    one-byte-per-pixel layer planes: every call writes all n_stack frames of `obs` (oldest first) from the ring, so no
    pixels are moved between calls.  `phase` (the ring slot of the next frame) stays reduced modulo n_stack."""

    def __init__(self, B, n_stack, H=64, W=64, device="cuda", flags=0, handle=None):
        self.n_stack, self.H, self.W, self.flags = int(n_stack), H, W, int(flags)
        self.handle = handle                      # optional _ext.EnvHandle: launches go through the C++ extension
        self.obs = torch.zeros((B, 3 * self.n_stack, H, W), dtype=torch.uint8, device=device)
        self.layers = torch.full((B, self.n_stack, H * W), _abi.LAYER_BLANK, dtype=torch.uint8, device=device)
        self.phase = 0

    def render(self, cfg, dworld, state, fov=35.0, fresh=None):
        """append the current frame of every view; `fresh` (uint8 [B]): views whose episode just (re)started - their
        older frames become blank, as VecFrameStack shows them after a reset"""
        if self.handle is not None:
            self.handle.render(self.obs, self.H, self.W, fov, self.n_stack, self.layers, self.phase, self.flags, fresh, None)
        else:
            render_ego(cfg, dworld, state, self.H, self.W, fov, self.n_stack, self.obs, self.layers, self.phase,
                       self.flags, fresh=fresh)
        self.phase = (self.phase + 1) % self.n_stack
        return self.obs

    def reset_rerender(self, cfg, dworld, state, mask, fov=35.0):
        """masked reset + rerender() of the same views as one C-ABI call (tde_env_reset_render)"""
        last = (self.phase - 1) % self.n_stack
        if self.handle is not None:
            self.handle.reset_render(mask, int(cfg.flags), self.obs, self.H, self.W, fov, self.n_stack, self.layers, last, self.flags)
        else:
            env_reset_render(cfg, dworld, state, mask, self.obs, self.H, self.W, fov, self.n_stack, self.layers, last, self.flags)
        return self.obs

    def rerender(self, cfg, dworld, state, mask, fov=35.0):
        """re-render the NEWEST frame of the masked views in place (they were re-spawned after the last `render`) and
        blank their older frames; the other views and the ring position are untouched"""
        last = (self.phase - 1) % self.n_stack
        if self.handle is not None:
            self.handle.render(self.obs, self.H, self.W, fov, self.n_stack, self.layers, last, self.flags, mask, mask)
        else:
            render_ego(cfg, dworld, state, self.H, self.W, fov, self.n_stack, self.obs, self.layers, last, self.flags,
                       fresh=mask, only=mask)
        return self.obs

    def clear(self, mask=None):
        """blank the stack of the masked views (all views without a mask), as VecFrameStack does on reset"""
        if mask is None:
            self.layers.fill_(_abi.LAYER_BLANK)
        else:
            self.layers[mask] = _abi.LAYER_BLANK

    def state_dict(self):
        return {"layers": self.layers.clone(), "obs": self.obs.clone(), "phase": self.phase}

    def load_state_dict(self, sd):
        self.layers.copy_(sd["layers"])
        self.obs.copy_(sd["obs"])
        self.phase = int(sd["phase"]) % self.n_stack
