"""Builds libtde_hip.so in-tree with hipcc for gfx950 (one translation unit; ~80 s for its ~150 kernel instantiations)."""
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(_PKG, "csrc", "tde_kernels.hip")]
DEPS = SRC + [os.path.join(_PKG, "csrc", "tde_device.h"), os.path.join(_PKG, "csrc", "tde_raster.h"),
              os.path.join(_PKG, "csrc", "tde_gridbuild.h"), os.path.join(_PKG, "csrc", "tde_magnitudes.h"), os.path.join(_PKG, "csrc", "tde_magnitudes_kernels.h"), os.path.join(_PKG, "..", "include", "tde_abi.h"),
              os.path.join(_PKG, "..", "include", "tde_hip.h")]
OUT = os.path.join(_PKG, "libtde_hip.so")

# -ffp-contract=off / no fast-math: one IEEE rounding per written operation — the floating-point contract shared
# with the oracle (bit-exact masks AND state).  fp32 divide/sqrt stay IEEE-correct (hipcc default).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden",
         # a SONAME lets the torch extension's NEEDED entry resolve to the copy _lib.load() has already mapped
         "-Wl,-soname,libtde_hip.so",
         # packed fp32 (v_pk_*_f32) issues slower than two scalar ops on gfx950 for this mix and costs v_mov shuffles:
         # same-box A/B 7.2 -> 6.7 us/step without the SLP vectoriser
         "-fno-slp-vectorize",
         # the loop vectoriser does the same to short column loops (the rasteriser went from 38 to 129 VGPRs with it)
         "-fno-vectorize"]


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stale = not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(p) for p in DEPS)
    if force or stale:
        cmd = [hipcc] + FLAGS + ["-o", OUT] + SRC
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
