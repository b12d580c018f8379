"""ctypes front of the CPU oracle (oracle/tde_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
torchdriveenv_amd/ does (tests/test_boundary.py greps for it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from torchdriveenv_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtde_oracle.so")


def build(force=False):
    """(re)build libtde_oracle.so when it is older than its sources.  Safe against concurrent callers (pytest-xdist,
    the 2-rank gloo test): the build runs under a file lock and the library is replaced atomically."""
    import fcntl

    src = os.path.join(_HERE, "tde_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "tde_abi.h")

    def stale():
        return (not os.path.exists(_LIB_PATH) or
                os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))

    if force or stale():
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                tmp = _LIB_PATH + f".tmp{os.getpid()}"
                # -mfma: the explicit fmaf() calls of the shared sin/cos and raster specifications become one instruction
                # instead of a libm call (same value: fmaf is correctly rounded either way; -ffp-contract=off still forbids
                # the compiler to fuse anything that is not written as fmaf)
                subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-ffp-contract=off", "-fno-fast-math"] + _fma_flag() +
                               ["-fvisibility=hidden", "-Wall", "-Wextra", "-fopenmp", "-shared", "-o", tmp, src, "-lm"],
                               check=True, capture_output=True)        # same flags as oracle/Makefile
                os.replace(tmp, _LIB_PATH)
    return _LIB_PATH


def _fma_flag():
    try:
        with open("/proc/cpuinfo") as f:
            return ["-mfma"] if " fma " in f.read() else []
    except OSError:
        return []


_lib = None


def lib():
    global _lib
    if _lib is None:
        # TDE_ORACLE_LIB: a sanitizer build of the same source (`make -C oracle san`; tests/test_sanitizers.py)
        path = os.environ.get("TDE_ORACLE_LIB")
        if not path:
            build()
            path = _LIB_PATH
        L = C.CDLL(path)
        f32p = C.POINTER(C.c_float)
        L.tde_oracle_sincosf.argtypes = [C.c_float, f32p, f32p]
        L.tde_oracle_sincosf.restype = None
        L.tde_oracle_sincosf_array.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tde_oracle_sincosf_array.restype = None
        L.tde_oracle_bicycle.argtypes = [f32p, f32p, f32p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.tde_oracle_bicycle.restype = None
        L.tde_oracle_kinematics_step.argtypes = [C.c_int64] + [C.c_void_p] * 7 + [C.c_float]
        L.tde_oracle_kinematics_step.restype = None
        L.tde_oracle_obb_overlap.argtypes = [C.c_float] * 12
        L.tde_oracle_obb_overlap.restype = C.c_int
        L.tde_oracle_compute_collision.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 7
        L.tde_oracle_compute_collision.restype = None
        L.tde_oracle_point_tri_d2.argtypes = [C.c_float, C.c_float, C.c_void_p]
        L.tde_oracle_point_tri_d2.restype = C.c_float
        L.tde_oracle_point_mesh_d2.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int32]
        L.tde_oracle_point_mesh_d2.restype = C.c_float
        L.tde_oracle_point_near_mesh.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_float]
        L.tde_oracle_point_near_mesh.restype = C.c_int
        L.tde_oracle_compute_offroad.argtypes = ([C.c_int32, C.c_int32] + [C.c_void_p] * 6 +
                                                 [C.POINTER(_abi.TdeWorld), C.c_void_p, C.c_float, C.c_void_p])
        L.tde_oracle_compute_offroad.restype = None
        L.tde_oracle_philox.argtypes = [C.c_uint64] + [C.c_uint32] * 4 + [C.POINTER(C.c_uint32 * 4)]
        L.tde_oracle_philox.restype = None
        cfgp, wp, sp = C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState)
        L.tde_oracle_env_reset.argtypes = [cfgp, wp, sp, C.c_void_p]
        L.tde_oracle_env_reset.restype = C.c_int
        L.tde_oracle_env_step.argtypes = [cfgp, wp, sp]
        L.tde_oracle_env_step.restype = C.c_int
        L.tde_oracle_env_rollout.argtypes = [cfgp, wp, sp, C.POINTER(_abi.TdeRollout)]
        L.tde_oracle_env_rollout.restype = C.c_int
        L.tde_oracle_ego_infractions.argtypes = [cfgp, wp, sp, C.c_void_p]
        L.tde_oracle_ego_infractions.restype = C.c_int
        L.tde_oracle_num_threads.restype = C.c_int
        L.tde_oracle_set_num_threads.argtypes = [C.c_int]
        L.tde_oracle_abi_version.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


def sincosf(x):
    x = np.ascontiguousarray(x, np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().tde_oracle_sincosf_array(x.size, _p(x), _p(s), _p(c))
    return s, c


def bicycle(x, y, psi, v, lr, a, beta, dt=0.1):
    """scalar KinematicBicycle.step; returns the new (x, y, psi, v) as python floats (exact fp32 values)"""
    fx, fy, fp, fv = C.c_float(x), C.c_float(y), C.c_float(psi), C.c_float(v)
    lib().tde_oracle_bicycle(C.byref(fx), C.byref(fy), C.byref(fp), C.byref(fv), lr, a, beta, dt)
    return fx.value, fy.value, fp.value, fv.value


def kinematics_step(x, y, psi, v, lr, present, action, dt=0.1):
    """in place on float32 arrays of n agents; action [n,2]"""
    lib().tde_oracle_kinematics_step(x.size, _p(x), _p(y), _p(psi), _p(v), _p(lr), _p(present), _p(action), dt)


def obb_overlap(bi, bj):
    """bi, bj = (x, y, cos, sin, half_len, half_wid)"""
    return lib().tde_oracle_obb_overlap(*[float(np.float32(t)) for t in (*bi, *bj)])


def compute_collision(B, A, x, y, psi, length, width, present):
    out = np.zeros(B * A, np.uint8)
    lib().tde_oracle_compute_collision(B, A, _p(x), _p(y), _p(psi), _p(length), _p(width), _p(present), _p(out))
    return out


def compute_offroad(B, A, x, y, psi, length, width, present, world, map_of_env, threshold=0.5):
    out = np.zeros(B * A, np.uint8)
    moe = np.ascontiguousarray(map_of_env, np.int32)
    lib().tde_oracle_compute_offroad(B, A, _p(x), _p(y), _p(psi), _p(length), _p(width), _p(present),
                                     C.byref(world.host_struct()), _p(moe), threshold, _p(out))
    return out


def point_mesh_d2(px, py, tri):
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 6)
    return lib().tde_oracle_point_mesh_d2(float(px), float(py), _p(tri), len(tri))


def point_near_mesh(px, py, tri, thr2):
    """the predicate the oracle's offroad test and raster use (bounding-box reject + exact distance): == point_mesh_d2 <= thr2"""
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 6)
    return bool(lib().tde_oracle_point_near_mesh(float(px), float(py), _p(tri), len(tri), float(thr2)))


def philox(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().tde_oracle_philox(seed, c0, c1, c2, c3, C.byref(out))
    return list(out)


def env_reset(cfg, world, state, mask=None):
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    return lib().tde_oracle_env_reset(C.byref(cfg), C.byref(world.host_struct()), C.byref(state.struct), _p(m))


def env_step(cfg, world, state):
    return lib().tde_oracle_env_step(C.byref(cfg), C.byref(world.host_struct()), C.byref(state.struct))


def ego_infractions(cfg, world, state):
    """float32 [B, 4]: the ego's offroad magnitude, collision magnitude (sum of IoUs), number of overlapping agents, 0 - of the
    current state (brute force over every triangle)"""
    out = np.zeros((state.B, 4), np.float32)
    assert lib().tde_oracle_ego_infractions(C.byref(cfg), C.byref(world.host_struct()), C.byref(state.struct), _p(out)) == 0
    return out


def env_rollout(cfg, world, state, actions):
    """actions [K,B,2] float32 -> (reward [K,B] f32, done [K,B] u8)"""
    actions = np.ascontiguousarray(actions, np.float32)
    K, B = actions.shape[0], actions.shape[1]
    reward = np.zeros((K, B), np.float32)
    done = np.zeros((K, B), np.uint8)
    ro = _abi.TdeRollout(_p(actions), _p(reward), _p(done), K, 0)
    lib().tde_oracle_env_rollout(C.byref(cfg), C.byref(world.host_struct()), C.byref(state.struct), C.byref(ro))
    return reward, done


def num_threads():
    return lib().tde_oracle_num_threads()


def set_num_threads(n):
    lib().tde_oracle_set_num_threads(int(n))


def waypoint_reward(cfg, pre, post, offroad, collided, tl, wp_xy, wp_n, scn, steps, target_idx, reached,
                    with_info=True):
    """Batched reference-owned reward/termination operator.  pre/post: [n,4] float32 ego states.
    steps/target_idx/reached are updated in place (int32 arrays).  Returns dict of outputs."""
    L = lib()
    if not getattr(L, "_wr_bound", False):
        L.tde_oracle_waypoint_reward.argtypes = ([C.POINTER(_abi.TdeConfig), C.c_int32] + [C.c_void_p] * 13 +
                                                 [C.c_int32] + [C.c_void_p] * 9)
        L.tde_oracle_waypoint_reward.restype = C.c_int
        L._wr_bound = True
    n = len(pre)
    pre = np.ascontiguousarray(pre, np.float32)
    post = np.ascontiguousarray(post, np.float32)
    cols = [np.ascontiguousarray(pre[:, k]) for k in range(4)] + [np.ascontiguousarray(post[:, k]) for k in range(4)]
    offroad = np.ascontiguousarray(offroad, np.uint8)
    collided = np.ascontiguousarray(collided, np.uint8)
    tl = None if tl is None else np.ascontiguousarray(tl, np.uint8)
    wp_xy = np.ascontiguousarray(wp_xy, np.float64)
    wp_n = np.ascontiguousarray(wp_n, np.int32)
    scn = np.ascontiguousarray(scn, np.int32)
    out = dict(reward=np.zeros(n, np.float32), terminated=np.zeros(n, np.uint8), truncated=np.zeros(n, np.uint8),
               info=np.zeros((n, 4), np.float64) if with_info else None,
               info_reached=np.zeros(n, np.int32) if with_info else None)
    L.tde_oracle_waypoint_reward(C.byref(cfg), n, *[_p(c) for c in cols], _p(offroad), _p(collided), _p(tl),
                                 _p(wp_xy), _p(wp_n), wp_xy.shape[1], _p(scn), _p(steps), _p(target_idx),
                                 _p(reached), _p(out["reward"]), _p(out["terminated"]), _p(out["truncated"]),
                                 _p(out["info"]), _p(out["info_reached"]))
    return out


def render_ego(cfg, world, state, H=64, W=64, fov=35.0, n_stack=1, out=None, flags=0, fresh=None, only=None):
    """ego-centred birdview [B, 3*n_stack, H, W] uint8 of the current state (frame-stack semantics if n_stack > 1).
    flags: _abi.RENDER_*; fresh / only: optional uint8 [B] masks (tde_render.fresh / .only)"""
    L = lib()
    if not getattr(L, "_rd_bound", False):
        L.tde_oracle_render_ego.argtypes = [C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld),
                                            C.POINTER(_abi.TdeState), C.POINTER(_abi.TdeRender)]
        L.tde_oracle_render_ego.restype = C.c_int
        L._rd_bound = True
    ns = max(1, n_stack)
    if out is None:
        out = np.zeros((state.B, 3 * ns, H, W), np.uint8)
    fresh = None if fresh is None else np.ascontiguousarray(fresh, np.uint8)
    only = None if only is None else np.ascontiguousarray(only, np.uint8)
    rd = _abi.TdeRender(_p(out), H, W, fov, n_stack, None, 0, int(flags), _p(fresh), _p(only))   # (stack shifted in place)
    L.tde_oracle_render_ego(C.byref(cfg), C.byref(world.host_struct()), C.byref(state.struct), C.byref(rd))
    return out
