"""Builds libtde_hip.so in-tree with hipcc for gfx950: one translation unit per kernel family (csrc/tde_*.hip), compiled side by
side on the host's cores (~40 s on 8 instead of the ~120 s of the single unit csrc/tde_kernels.hip), linked, and the linked code
object audited (isa_audit.py) before the library is put in place."""
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
# the units, longest first (the pool starts them in this order)
UNITS = ["tde_step_solo_mag.hip", "tde_step_wide.hip", "tde_step_wide8.hip", "tde_step_solo.hip", "tde_step_trio.hip", "tde_rollout_trio.hip",
         "tde_rollout_duo.hip", "tde_rollout_solo.hip", "tde_api.hip"]
SRC = [os.path.join(_CSRC, u) for u in UNITS]
HEADERS = ["tde_kernels.h", "tde_host.h", "tde_device.h", "tde_raster.h", "tde_gridbuild.h", "tde_magnitudes.h", "tde_magnitudes_kernels.h"]
DEPS = SRC + [os.path.join(_CSRC, h) for h in HEADERS] + [os.path.join(_PKG, "..", "include", "tde_abi.h"),
                                                          os.path.join(_PKG, "..", "include", "tde_hip.h")]
OUT = os.path.join(_PKG, "libtde_hip.so")

# -ffp-contract=off / no fast-math: one IEEE rounding per written operation — the floating-point contract shared
# with the oracle (bit-exact masks AND state).  fp32 divide/sqrt stay IEEE-correct (hipcc default).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-fvisibility=hidden",
          # packed fp32 (v_pk_*_f32) issues slower than two scalar ops on gfx950 for this mix and costs v_mov shuffles:
          # same-box A/B 7.2 -> 6.7 us/step without the SLP vectoriser
          "-fno-slp-vectorize",
          # the loop vectoriser does the same to short column loops (the rasteriser went from 38 to 129 VGPRs with it)
          "-fno-vectorize"]
# a SONAME lets the torch extension's NEEDED entry resolve to the copy _lib.load() has already mapped
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,-soname,libtde_hip.so"]
FLAGS = CFLAGS + ["-shared", "-Wl,-soname,libtde_hip.so"]      # (the single-unit form: scripts/build_variant.sh mirrors it)


def _jobs():
    n = os.environ.get("TDE_BUILD_JOBS")
    return max(1, int(n)) if n else max(1, min(len(UNITS), os.cpu_count() or 1))


def compile_units(objdir, extra=(), verbose=False):
    """every unit -> an object file in `objdir`, in parallel; returns the object paths"""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc] + CFLAGS + list(extra) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{r.stdout}")
        if verbose and r.stdout.strip():
            print(r.stdout)
        return obj

    with ThreadPoolExecutor(_jobs()) as pool:
        return list(pool.map(one, SRC))


def link(objs, out, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + LDFLAGS + ["-o", out] + list(objs)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force=False, verbose=False, extra=(), out=None):
    """`extra`: more compiler flags (-D switches of an A/B variant); `out`: where to put the library (default: in-tree, where
    _lib.load() finds it; a variant for scripts/ab_*.py goes elsewhere, e.g. ab/libX.so)"""
    OUT = out or globals()["OUT"]
    stale = not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(p) for p in DEPS)
    if force or stale:
        from . import isa_audit
        tmp = OUT + ".tmp"
        with tempfile.TemporaryDirectory(prefix="tde_build_") as objdir:
            link(compile_units(objdir, extra, verbose), tmp, verbose)
        if os.environ.get("TDE_SKIP_ISA_AUDIT") != "1":
            # MI355X computes a 64-bit VALU shift wrong when its amount sits in the wavefront's last allocated VGPR
            # (profiles/r05_a32_respawn_anomaly.md).  The kernels form their lane masks without such shifts (tde_device.h: lane-mask
            # helpers), so the audit is a regression test: a 64-bit shift by a VGPR amount anywhere in the library fails the build.
            total, nk, bad = isa_audit.audit(tmp)
            if verbose:
                print(f"ISA audit: {nk} kernels, {total} 64-bit shifts by a VGPR amount, {len(bad)} in the last allocated VGPR")
            if total:
                os.remove(tmp)
                raise RuntimeError(f"the ISA audit refuses this build: {total} 64-bit shift(s) by a VGPR amount in the library "
                                   "(python -m torchdriveenv_amd.isa_audit <lib> lists them): take lane bits of 64-bit masks with "
                                   "csrc/tde_device.h's mask_bit / mask_field / one_bit64 / lane_prefix instead")
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    import sys

    # python -m torchdriveenv_amd.build [OUT.so] [-DNAME=VAL ...]: the in-tree library, or a variant of it
    args = sys.argv[1:]
    outs = [a for a in args if not a.startswith("-")]
    print(build(force=True, verbose=not outs, extra=[a for a in args if a.startswith("-")], out=os.path.abspath(outs[0]) if outs else None))
