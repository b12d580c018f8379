"""The PyTorch-ROCm C++ extension (csrc/tde_torch_ext.cpp): builds in-tree next to libtde_hip.so and binds the same C-ABI
symbols with torch tensors.  `build()` is called from __graft_entry__.build(); `load()` imports the built module and
fails loudly if it is missing (there is no silent fallback: callers choose the binding explicitly)."""
import importlib.util
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
NAME = "tde_torch_ext"
BUILD_DIR = os.path.join(_PKG, "_ext_build")
SO_PATH = os.path.join(BUILD_DIR, NAME + ".so")
SRC = os.path.join(_PKG, "csrc", "tde_torch_ext.cpp")
_mod = None


def build(force=False, verbose=False):
    """compile csrc/tde_torch_ext.cpp with the host compiler against torch's headers and link it to the in-tree
    libtde_hip.so (rpath $ORIGIN/..).  No device code and no hipify pass: the source is HIP-native host C++."""
    import torch
    from torch.utils import cpp_extension as ce

    from . import _lib
    from . import build as libbuild

    libbuild.build()
    _lib.load()                                   # mapped first: the fresh extension is test-loaded at the end of the build
    deps = [SRC, os.path.join(_PKG, "..", "include", "tde_hip.h"), os.path.join(_PKG, "..", "include", "tde_abi.h")]
    if not force and os.path.exists(SO_PATH) and os.path.getmtime(SO_PATH) >= max(os.path.getmtime(p) for p in deps):
        return SO_PATH
    os.makedirs(BUILD_DIR, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    ce.load(name=NAME, sources=[SRC], build_directory=BUILD_DIR, with_cuda=False, is_python_module=False,
            extra_cflags=["-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
            extra_include_paths=[os.path.join(_PKG, "..", "include"), "/opt/rocm/include"],
            extra_ldflags=[f"-L{tlib}", "-lc10_hip", "-ltorch_hip", f"-L{_PKG}", "-ltde_hip", "-Wl,-rpath,\\$$ORIGIN/..",
                           f"-Wl,-rpath,{tlib}"], verbose=verbose)
    assert os.path.exists(SO_PATH), SO_PATH
    return SO_PATH


def load():
    global _mod
    if _mod is not None:
        return _mod
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    import torch  # noqa: F401  (maps libtorch / libamdhip64 before the extension resolves its NEEDED entries)

    from . import _lib

    _lib.load()                                   # libtde_hip.so mapped (and ABI-checked) first
    spec = importlib.util.spec_from_file_location(NAME, SO_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[NAME] = mod
    _mod = mod
    return mod


# ---- typed construction of the extension's argument carriers (no raw addresses cross the boundary) --------------------
def config_of(cfg):
    """ext.Config from a ctypes tde_config (_abi.TdeConfig): its bytes, copied"""
    return load().Config(bytes(cfg))


def world_of(dworld):
    """ext.World over the device tensors of a world.DeviceWorld (cached on it; the ext object keeps the tensors alive)"""
    w = getattr(dworld, "_ext_world", None)
    if w is None:
        dev = next(iter(dworld.tensors.values())).device
        w = load().World(dict(dworld.tensors), dict(dworld.ints), dev.index or 0)
        dworld._ext_world = w
    return w


def env_handle(cfg, dworld, state):
    """ext.EnvHandle for (tde_config, DeviceWorld, device EnvState): named tensors in, validated in C++"""
    tens = {k: v for k, v in state.arrays.items() if v is not None}
    return load().EnvHandle(config_of(cfg), world_of(dworld), tens, state.B, state.A)
