"""Loader of libtde_hip.so (the C-ABI of include/tde_hip.h).  Fails loudly: there is no CPU or eager fallback."""
import ctypes as C
import os

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDE_HIP_LIB") or os.path.join(_PKG, "libtde_hip.so")   # override: A/B builds only

# every symbol include/tde_hip.h declares
SYMBOLS = ["tde_abi_version", "tde_last_error", "tde_kernel_override", "tde_kinematics_step", "tde_compute_collision",
           "tde_compute_offroad", "tde_kin_collide_step", "tde_waypoint_reward", "tde_env_reset", "tde_env_step",
           "tde_env_rollout", "tde_render_ego", "tde_env_reset_render", "tde_env_step_render", "tde_state_obs", "tde_ego_infractions", "tde_env_post_step", "tde_first_gaps", "tde_grid_build", "tde_grid_free"]

_lib = None


class TdeError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library.  `torch` is imported first so that its bundled HIP runtime (libamdhip64.so.7) is
    the one already mapped when the loader resolves our NEEDED entry: one HIP runtime per process, so torch's
    device pointers and streams are valid in our launches."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TdeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    import torch  # noqa: F401  (side effect: maps torch/lib/libamdhip64.so)

    L = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise TdeError(f"{LIB_PATH} lacks symbols {missing}: stale build?")
    if L.tde_abi_version() != _abi.TDE_ABI_VERSION:
        raise TdeError(f"ABI mismatch: library {L.tde_abi_version()} vs python {_abi.TDE_ABI_VERSION}")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    cfgp, wp, sp = C.POINTER(_abi.TdeConfig), C.POINTER(_abi.TdeWorld), C.POINTER(_abi.TdeState)
    L.tde_last_error.restype = C.c_char_p
    L.tde_kinematics_step.argtypes = [i64] + [vp] * 7 + [f32, vp]
    L.tde_compute_collision.argtypes = [i32, i32] + [vp] * 7 + [vp]
    L.tde_compute_offroad.argtypes = [i32, i32] + [vp] * 6 + [wp, vp, f32, vp, vp]
    L.tde_kin_collide_step.argtypes = [i32, i32] + [vp] * 9 + [f32, vp, vp]
    L.tde_waypoint_reward.argtypes = [cfgp, i32] + [vp] * 13 + [i32] + [vp] * 9 + [vp]
    L.tde_env_reset.argtypes = [cfgp, wp, sp, vp, vp]
    L.tde_env_step.argtypes = [cfgp, wp, sp, vp]
    L.tde_env_rollout.argtypes = [cfgp, wp, sp, C.POINTER(_abi.TdeRollout), vp]
    L.tde_render_ego.argtypes = [cfgp, wp, sp, C.POINTER(_abi.TdeRender), vp]
    L.tde_env_step_render.argtypes = [cfgp, wp, sp, C.POINTER(_abi.TdeRender), C.POINTER(vp), i32]
    L.tde_env_reset_render.argtypes = [cfgp, wp, sp, vp, C.POINTER(_abi.TdeRender), vp]
    L.tde_state_obs.argtypes = [wp, sp, vp, vp]
    L.tde_ego_infractions.argtypes = [cfgp, wp, sp, vp, vp]
    L.tde_env_post_step.argtypes = [cfgp, wp, sp, vp, vp]
    L.tde_first_gaps.argtypes = [cfgp, wp, vp]
    L.tde_kernel_override.argtypes = [C.c_int, C.c_int]
    L.tde_grid_build.argtypes = [vp, i32, f32, f32, f32, f32, i32, C.POINTER(C.POINTER(_abi.TdeGrid))]
    L.tde_grid_free.argtypes = [C.POINTER(_abi.TdeGrid)]
    L.tde_grid_free.restype = None
    for s in SYMBOLS:
        if s not in ("tde_last_error", "tde_grid_free"):
            getattr(L, s).restype = C.c_int
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = load().tde_last_error().decode() or f"error {rc}"
        raise TdeError(f"{what}: {msg}" if what else msg)


TEAMS = {None: 0, "auto": 0, "solo": 1, "duo": 2, "trio": 3}


def kernel_override(rollout=None, step=None):
    """force a kernel form of tde_env_rollout ("solo" | "duo" | "trio") / tde_env_step ("solo" | "trio"); None = the
    library's own choice (tde_hip.h: tde_kernel_override).  Process-wide; tests and A/B scripts only."""
    check(load().tde_kernel_override(TEAMS[rollout], TEAMS[step]), "tde_kernel_override")


def current_stream(device):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
