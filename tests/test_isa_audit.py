"""The built gfx950 code objects are free of the instruction that computes wrong on MI355X: a 64-bit VALU shift whose amount
sits in the wavefront's last allocated VGPR (torchdriveenv_amd/isa_audit.py; profiles/r05_a32_respawn_anomaly.md;
scripts/ubench/shift64_last_vgpr.hip reproduces the erratum stand-alone) - since round 6 by construction: the kernels take lane
bits of 64-bit masks through csrc/tde_device.h's helpers and the library holds NO 64-bit shift by a VGPR amount at all.
Host-side: llvm-objdump on the library that ships."""
import os
import re
import tempfile

import pytest

from torchdriveenv_amd import isa_audit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_audit_recognises_the_pattern():
    dis = """
0000000000001000 <_Zkernel_a>:
	v_lshrrev_b64 v[16:17], v79, s[6:7]                        // 000000001000: D2900010 00000D4F
	v_lshrrev_b64 v[16:17], v78, s[6:7]                        // 000000001008: D2900010 00000D4E
	v_lshlrev_b64 v[2:3], 2, v[6:7]                            // 000000001010: D28F0002 00020C82
	v_lshrrev_b64 v[8:9], v47, v[18:19]                        // 000000001018: D2900008 0002252F
0000000000002000 <_Zkernel_b>:
	v_lshrrev_b64 v[16:17], v79, s[6:7]                        // 000000002000: D2900010 00000D4F
	v_ashrrev_i64 v[0:1], v71, v[2:3]                          // 000000002008: D2910000 00020547
0000000000003000 <_Zkernel_c>:
	v_lshlrev_b64 v[2:3], v15, -1                              // 000000003000: D28F0002 0001830F
"""
    counts = {"_Zkernel_a": 80, "_Zkernel_b": 96, "_Zkernel_c": 13}
    bad = isa_audit.risky_shifts(dis, counts)
    # a: v79 of 80 is the last register of the allocation (v78 and the mid-allocation v47 are not; a constant amount is not a VGPR);
    # b: 96 registers - v79 and v71 have allocated successors; c: 13 used = 16 allocated, v15 is the last one
    assert [(k, ins.split()[0], ins.split()[2].rstrip(",")) for k, n, ins in bad] == [("_Zkernel_a", "v_lshrrev_b64", "v79"), ("_Zkernel_c", "v_lshlrev_b64", "v15")]


def test_the_library_that_ships_has_no_64_bit_shift_by_a_vgpr_amount():
    lib = os.path.join(ROOT, "torchdriveenv_amd", "libtde_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    if not os.path.exists(os.path.join(isa_audit.LLVM_BIN, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    with tempfile.TemporaryDirectory() as d:
        dis, counts = isa_audit.disassemble(lib, d)
    # the disassembly was really read: every translation unit's code object (the library is linked from seven), hundreds of kernels,
    # and 64-bit shifts by CONSTANT amounts (index arithmetic) all over
    assert len(counts) > 200
    assert len(re.findall(r"\bv_lshlrev_b64\s+v\[\d+:\d+\],\s*\d+,", dis)) > 100
    assert {k for k in counts if "env_rollout_trio_kernel" in k} and {k for k in counts if "env_step_kernel" in k} and {k for k in counts if "render_views_kernel" in k}
    sites = isa_audit.shift_sites(dis)
    assert sites == [], sites[:8]
    assert isa_audit.risky_shifts(dis, counts) == []
