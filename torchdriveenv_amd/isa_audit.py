"""Audit of the built gfx950 code objects for the instruction that gives wrong results on MI355X:

    v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64  vdst, vN, src      with N = the wavefront's LAST allocated VGPR

(N & 7 == 7 and v(N+1) outside the kernel's allocation: the 64-bit shifter fetches the register after the shift amount too).
LLVM knows the erratum as `fixShift64HighRegBug` and works around it for gfx11 only; on this part a round-5 build of
`env_step_trio_kernel<32, false, true, true>` whose allocator had put `base = lane & 32` into v79 of 80 re-spawned the wrong lanes on
one env-finish in a few hundred (profiles/r05_a32_respawn_anomaly.md; scripts/ubench/shift64_last_vgpr.hip reproduces it stand-alone:
all three 64-bit shifts, not the 32-bit ones, not with the amount in an SGPR or as a literal).

Where the allocator puts a shift amount is not something the sources control, so since round 6 the kernels do not shift 64-bit
values by per-lane amounts AT ALL (csrc/tde_device.h: mask_bit / mask_field / one_bit64 / lane_prefix take a lane's bit from the
32-bit half that holds it) and this audit is the regression test: `build.build()` runs it on the library it has just linked and
refuses a library that contains ANY 64-bit shift by a VGPR amount (`strict`), whichever register holds it.

Host-side developer tooling: nothing here runs on the product path."""
import os
import re
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_SHIFT = re.compile(r"\b(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\b")
_SYM = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def disassemble(lib_path, workdir):
    """the gfx950 code objects of a fat library -> (objdump text, {kernel symbol: vgpr_count}).  A library linked from several
    translation units carries one offload bundle per unit, back to back in .hip_fatbin: each is unbundled and disassembled."""
    fat = os.path.join(workdir, "fat.bin")
    _run(f"{LLVM_BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(_MAGIC), blob)]
    if not starts:
        raise RuntimeError(f"{lib_path}: no offload bundle in .hip_fatbin")
    dis, counts = [], {}
    for i, a in enumerate(starts):
        part, co = os.path.join(workdir, f"fat{i}.bin"), os.path.join(workdir, f"dev{i}.co")
        with open(part, "wb") as f:
            f.write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        _run(f"{LLVM_BIN}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
             "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}")
        notes = _run(f"{LLVM_BIN}/llvm-readelf", "--notes", co)
        name = None
        for ln in notes.splitlines():
            m = re.search(r"\.name:\s+(\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"\.vgpr_count:\s+(\d+)", ln)
            if m and name:
                counts[name] = int(m.group(1))
                name = None
        dis.append(_run(f"{LLVM_BIN}/llvm-objdump", "-d", "--mcpu=gfx950", co))
    return "\n".join(dis), counts


def risky_shifts(dis, counts):
    """[(kernel, vgpr_count, instruction)] for 64-bit shifts whose amount sits in the kernel's last allocated VGPR"""
    out, cur = [], None
    for ln in dis.splitlines():
        m = _SYM.match(ln)
        if m:
            cur = m.group(1)
            continue
        m = _SHIFT.search(ln)
        if m and cur is not None:
            n = counts.get(cur)
            idx = int(m.group(2))
            # the allocation is vgpr_count rounded up to 8 registers: v(idx + 1) is outside it iff idx is its last register
            if n is not None and (idx & 7) == 7 and idx + 1 >= -(-n // 8) * 8:
                out.append((cur, n, ln.split("//")[0].strip()))
    return out


def shift_sites(dis):
    """[(kernel symbol, instruction)] of every 64-bit shift by a VGPR amount"""
    out, cur = [], None
    for ln in dis.splitlines():
        m = _SYM.match(ln)
        if m:
            cur = m.group(1)
        elif _SHIFT.search(ln):
            out.append((cur, ln.split("//")[0].strip()))
    return out


def audit(lib_path):
    """-> (number of 64-bit shifts by a VGPR amount, number of kernels, list of risky ones)"""
    with tempfile.TemporaryDirectory() as d:
        dis, counts = disassemble(lib_path, d)
    total = sum(1 for ln in dis.splitlines() if _SHIFT.search(ln))
    return total, len(counts), risky_shifts(dis, counts)


def main(argv=None):
    import sys
    path = (argv or sys.argv[1:] or [os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtde_hip.so")])[0]
    total, nk, bad = audit(path)
    print(f"{path}: {nk} kernels, {total} 64-bit shifts by a VGPR amount, {len(bad)} with the amount in the last allocated VGPR")
    for k, n, ins in bad:
        print(f"  {k} ({n} VGPRs): {ins}")
    return 1 if (bad or total) else 0


if __name__ == "__main__":
    raise SystemExit(main())
