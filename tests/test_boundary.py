"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/tde_hip.h
declares (no compute calls without a GPU); the product never touches the oracle; failures are loud."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "tde_hip.h")).read()
    return re.findall(r"TDE_API\s+[\w\s\*]+?\b(tde_\w+)\s*\(", src)


def test_library_exports_every_declared_symbol():
    from torchdriveenv_amd import _lib, build

    build.build()
    syms = declared_symbols()
    assert len(syms) >= 11 and "tde_env_step" in syms and "tde_render_ego" in syms
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/tde_hip.h but not exported"
    assert sorted(syms) == sorted(_lib.SYMBOLS), "python loader and header disagree on the C-ABI"
    assert L.tde_abi_version() == _lib._abi.TDE_ABI_VERSION
    assert _lib.load() is not None     # full loader: argtypes + ABI version check


def test_kernel_override_accepts_the_documented_teams_only():
    """tde_kernel_override (no GPU involved: two atomics): rollout team 0..3, step team 0..3 (2 = the four-wavefront 128-slot step kernel
    at any batch size); anything else is an argument error with a message, and the choice is left as it was"""
    from torchdriveenv_amd import _lib

    L = _lib.load()
    try:
        for r in range(4):
            for s in range(4):
                assert L.tde_kernel_override(r, s) == 0
        for bad in ((4, 0), (-1, 0), (0, 4), (0, -1)):
            assert L.tde_kernel_override(*bad) != 0
            assert b"tde_kernel_override" in L.tde_last_error()
    finally:
        assert L.tde_kernel_override(0, 0) == 0


def test_abi_struct_sizes_match_header():
    """compile a tiny C program against include/tde_abi.h and compare sizeof/offsetof with the ctypes mirror"""
    import subprocess
    import tempfile

    from torchdriveenv_amd import _abi

    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "tde_abi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(tde_config), sizeof(tde_map), sizeof(tde_world),
         sizeof(tde_state), sizeof(tde_rollout), sizeof(tde_render), sizeof(tde_spawn), sizeof(tde_scenario),
         offsetof(tde_config, flags), offsetof(tde_world, n_maps), offsetof(tde_state, B));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        got = [int(t) for t in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    C = ctypes
    want = [C.sizeof(_abi.TdeConfig), C.sizeof(_abi.TdeMap), C.sizeof(_abi.TdeWorld), C.sizeof(_abi.TdeState),
            C.sizeof(_abi.TdeRollout), C.sizeof(_abi.TdeRender), _abi.SPAWN_DTYPE.itemsize, _abi.SCN_DTYPE.itemsize,
            _abi.TdeConfig.flags.offset, _abi.TdeWorld.n_maps.offset, _abi.TdeState.B.offset]
    assert got == want


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "torchdriveenv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f"{f} imports the oracle"
                assert "libtde_oracle" not in txt and "tde_oracle_" not in txt, f"{f} references the oracle"


def test_missing_library_fails_loudly(monkeypatch):
    from torchdriveenv_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libtde_hip.so")
    with pytest.raises(_lib.TdeError, match="no fallback"):
        _lib.load()


def test_ops_reject_cpu_tensors_and_bad_shapes():
    import torch

    from torchdriveenv_amd import ops

    x = torch.zeros(8)
    with pytest.raises(ValueError, match="HIP device"):
        ops.kinematics_step(x, x, x, x, x, torch.zeros(8, 2))
    with pytest.raises(ValueError, match="float32"):
        ops.kinematics_step(x.double(), x, x, x, x, torch.zeros(8, 2))


def test_env_requires_gpu():
    import torch

    from torchdriveenv_amd.config import EnvConfig
    from torchdriveenv_amd.env import BatchedWaypointEnv

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        BatchedWaypointEnv(EnvConfig(), None, num_envs=1)


def test_torch_extension_builds_in_tree_and_binds_the_same_abi():
    """the PyTorch-ROCm C++ extension (north_star's boundary form) is built next to libtde_hip.so, links to it (NEEDED
    entry + $ORIGIN run path, no copy of the kernels), reports the same ABI version and exposes the env-level entry
    points as methods; no GPU is needed to load it"""
    import subprocess

    from torchdriveenv_amd import _abi, _ext, _lib

    path = _ext.build()
    assert os.path.dirname(path).startswith(os.path.join(ROOT, "torchdriveenv_amd"))
    dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "libtde_hip.so" in dyn and "$ORIGIN/.." in dyn
    m = _ext.load()
    assert m.abi_version() == _lib.load().tde_abi_version() == _abi.TDE_ABI_VERSION
    for meth in ("step", "reset", "rollout", "render", "step_render", "state_obs", "ego_infractions", "reset_render", "post_step"):
        assert hasattr(m.EnvHandle, meth)
    assert hasattr(m, "Config") and hasattr(m, "World")             # typed carriers: no raw addresses cross the boundary
    with pytest.raises(RuntimeError, match="bytes"):
        m.Config(b"short")
    m.Config(bytes(_abi.default_config()))
    # device code lives in libtde_hip.so only: the extension object defines no kernels
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    assert "tde_env_step" not in syms and "PyInit_tde_torch_ext" in syms


def test_stamp_patches_apply_to_the_current_source():
    """scripts/make_stamped_build.py instruments a COPY of the kernel source by text substitution: every anchor of every
    patch must still exist, or the stamp profiles DESIGN.md cites can no longer be regenerated from HEAD"""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_stamped_build as m

    for mode in m.PATCHES:
        src = m.patched_source(mode)
        assert "tde_debug_stamps" in src and ("tde_mark(" in src or "stp.mark(" in src or mode == "trips"), mode


def test_grid_build_argument_errors_and_shared_lists():
    """tde_grid_build (host side, no GPU): bad meshes are refused with a message, not a crash; cells with identical candidates
    share their records (every record offset + count stays inside the map's record range)"""
    import ctypes as C

    import numpy as np

    from torchdriveenv_amd import _abi, _lib
    from torchdriveenv_amd.world import build_grid_index, strip_mesh

    L = _lib.load()
    gp = C.POINTER(_abi.TdeGrid)()
    huge = np.array([[0, 0, 1e6, 0, 0, 1e6]], np.float32)
    assert L.tde_grid_build(huge.ctypes.data, 1, 0.5, 0.25, 0.05, 2.0, 0, C.byref(gp)) != 0 and b"larger cell" in L.tde_last_error()
    nan = np.array([[np.nan, 0, 1, 0, 0, 1]], np.float32)
    assert L.tde_grid_build(nan.ctypes.data, 1, 0.5, 0.25, 0.05, 2.0, 0, C.byref(gp)) != 0 and b"non-finite" in L.tde_last_error()
    assert L.tde_grid_build(nan.ctypes.data, 0, 0.5, 0.25, 0.05, 2.0, 0, C.byref(gp)) != 0
    assert L.tde_grid_build(huge.ctypes.data, 1, 0.5, 0.25, 0.6, 2.0, 0, C.byref(gp)) != 0          # margin >= threshold
    ok = np.array([[0, 0, 10, 0, 0, 10]], np.float32)
    assert L.tde_grid_build(ok.ctypes.data, 1, 0.5, 0.25, 0.05, -1.0, 0, C.byref(gp)) != 0 and b"near_range" in L.tde_last_error()
    g = build_grid_index(strip_mesh([(0.0, 0.0), (60.0, 0.0)], 7.0, 5.0).astype(np.float32), 0.5, 0.25)
    mixed = g["cell_class"] == _abi.CELL_MIXED
    assert mixed.sum() > 500 and 0 < g["n_lists"] < mixed.sum() // 4           # a straight road: few distinct lists
    assert (g["cell_first"][mixed].astype(np.int64) + g["cell_count"][mixed] <= len(g["rec_tri"])).all()
    assert g["nx"] % 8 == 0 and g["ny"] % 8 == 0 and float(g["ox"]).is_integer()
    one = build_grid_index(strip_mesh([(0.0, 0.0), (60.0, 0.0)], 7.0, 5.0).astype(np.float32), 0.5, 0.25, n_threads=1)
    assert all(np.array_equal(g[k], one[k]) for k in ("cell_class", "cell_count", "cell_first", "cell_sub", "rec_tri", "rec_len",
                                                       "tile_near"))   # thread count does not show
    # near lists (ABI 10): tiles along the road carry one, their records lie inside the table, the length sits at the first record
    tn = g["tile_near"]
    listed = (tn != 0) & (tn != 0xFFFFFFFF)
    assert listed.sum() > 100 and (tn == 0xFFFFFFFF).sum() > 100 and g["n_near_lists"] == listed.sum()
    first = tn[listed].astype(np.int64) - 1
    assert (g["rec_len"][first] > 0).all() and (first + g["rec_len"][first] <= len(g["rec_tri"])).all()
    none = build_grid_index(strip_mesh([(0.0, 0.0), (60.0, 0.0)], 7.0, 5.0).astype(np.float32), 0.5, 0.25, near_range=0.0)
    assert ((none["tile_near"] == 0) | (none["tile_near"] == 0)).all() and none["n_near_lists"] == 0


def test_experiment_switches_do_not_compile_into_a_product_build(tmp_path):
    """TDE_EXP_* (timing experiments that skip work: WRONG results) are fenced: a translation unit of the library with one of them set
    only compiles with -DTDE_EXPERIMENT_BUILD (csrc/tde_kernels.h); build.py passes neither"""
    import subprocess

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "torchdriveenv_amd", "csrc", "tde_rollout_solo.hip")
    base = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "--offload-device-only", "-x", "hip", src]
    bad = subprocess.run(base + ["-DTDE_EXP_NO_D_RESPAWN"], capture_output=True, text=True)
    assert bad.returncode != 0 and "TDE_EXPERIMENT_BUILD" in bad.stderr
    from torchdriveenv_amd import build as b
    assert not any("TDE_EXP" in f for f in b.CFLAGS + b.LDFLAGS)
