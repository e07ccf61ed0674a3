"""Code-generation guards for the inflate kernel (no GPU needed: hipcc cross-compiles gfx950).

The kernel's speed depends on properties the source cannot express and a refactor can silently
lose (profiles/r01_unfilter_tuning.md): four workgroups of 256 threads per CU (LDS <= 40 KiB), no
scratch memory, no scalar-register spills in the decode loops, LDS and global memory reached with
their own instructions (no generic `flat_` accesses)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def inflate_asm():
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(ROOT, "swift_png_amd", "csrc", "inflate.hip")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", os.path.join(tmp, "inflate.s"), src], check=True, capture_output=True, timeout=600)
        yield open(os.path.join(tmp, "inflate.s")).read()


def _meta(asm, key):
    m = re.search(r"^\s+\." + key + r":\s+(\d+)", asm, re.M)
    assert m, key
    return int(m.group(1))


def test_inflate_kernel_resources(inflate_asm):
    assert _meta(inflate_asm, "group_segment_fixed_size") <= 40960        # four streams per CU
    assert _meta(inflate_asm, "private_segment_fixed_size") == 0          # no scratch
    assert _meta(inflate_asm, "vgpr_spill_count") == 0
    assert _meta(inflate_asm, "sgpr_spill_count") <= 8                    # (0 today)
    assert _meta(inflate_asm, "vgpr_count") <= 128                        # >= 4 waves per SIMD
    assert _meta(inflate_asm, "max_flat_workgroup_size") == 256           # 192 is not placed evenly


def test_inflate_kernel_address_spaces(inflate_asm):
    # the far-reference path may read the output back through global_load; nothing goes through flat_*
    assert len(re.findall(r"^\s+flat_(load|store)", inflate_asm, re.M)) <= 2
    assert "scratch_" not in inflate_asm


def test_inflate_walk_is_straight_line(inflate_asm):
    # the chain walk: eight v_readlane hops and the mask updates in one block, no SALU between hops
    m = re.search(r"(v_readlane_b32 s\d+, v\d+, s\d+\n\s+v_readlane_b32 s\d+, v\d+, s\d+\n\s+s_nop 2\n\s+){3}", inflate_asm)
    assert m, "the double-hop walk was reordered or split"
