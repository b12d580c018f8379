"""Host code under sanitizers (CPU suite), built with gcc / g++:
  * `make -C oracle san`: the oracle (oracle/tde_oracle.c) as a shared library with AddressSanitizer + UndefinedBehaviorSanitizer;
  * `make -C torchdriveenv_amd/csrc san`: the library's HOST grid builder (csrc/tde_gridbuild.h through csrc/tde_grid_host.cpp - in
    libtde_hip.so it is compiled by hipcc inside tde_api.hip, where no host sanitizer reaches it) as a shared library with ASan +
    UBSan, and its multi-threaded passes as a stand-alone driver under ThreadSanitizer (and under ASan + UBSan).
The oracle / grid / loader tests then run in a child process with the ASan runtime preloaded and TDE_ORACLE_LIB / TDE_GRID_LIB
pointing at the sanitized libraries: they must pass with no sanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "torchdriveenv_amd", "_san")          # the library's host code
ORACLE_SAN = os.path.join(ROOT, "oracle", "_san")              # the checker


@pytest.fixture(scope="module")
def san_build():
    for d in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "torchdriveenv_amd", "csrc")):
        p = subprocess.run(["make", "-C", d, "san"], capture_output=True, text=True)
        if p.returncode != 0:
            pytest.fail(f"make -C {d} san failed:\n" + p.stderr[-2000:])
    return SAN


def test_grid_builder_threads_under_thread_sanitizer(san_build):
    """1, 3 and 8 host threads build identical tables; ThreadSanitizer sees no race in the shared row / tile buffers, the
    exception hand-over of run_pool or the atomics"""
    p = subprocess.run([os.path.join(san_build, "grid_driver_tsan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0 and "grid driver:" in p.stdout and p.stdout.strip().endswith("ok"), p.stdout + p.stderr[-3000:]
    assert "ThreadSanitizer" not in p.stderr, p.stderr[-3000:]


def test_grid_builder_under_address_and_ub_sanitizer(san_build):
    p = subprocess.run([os.path.join(san_build, "grid_driver_asan")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout + p.stderr[-3000:]
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]


@pytest.mark.timeout(1500)
def test_oracle_and_grid_tests_pass_on_the_sanitized_libraries(san_build):
    """the oracle's own tests, the grid-index tests and the world loaders, re-run in a child process on the ASan + UBSan builds"""
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no ASan runtime next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan_rt, TDE_ORACLE_LIB=os.path.join(ORACLE_SAN, "libtde_oracle_asan.so"),
               TDE_GRID_LIB=os.path.join(san_build, "libtde_grid_asan.so"),
               # (leaks: CPython and torch keep memory until exit by design; alloc_dealloc_mismatch: torch's operator new vs free)
               ASAN_OPTIONS="detect_leaks=0:alloc_dealloc_mismatch=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="4")
    tests = ["tests/test_oracle_math.py", "tests/test_oracle_golden.py", "tests/test_oracle_properties.py",
             "tests/test_world_loaders.py", "tests/test_second_opinions.py",
             "tests/test_boundary.py::test_grid_build_argument_errors_and_shared_lists"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1400)
    tail = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error:" not in p.stdout + p.stderr, tail
    assert " passed" in p.stdout
