"""Code-generation guards for the hot kernels (no GPU needed: hipcc cross-compiles gfx950).

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


# ---- the kernels the decode step spends its time in: occupancy is set by LDS and registers ------
def _kernels(name):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(ROOT, "swift_png_amd", "csrc", name + ".hip")
        out = os.path.join(tmp, name + ".s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, src],
                       check=True, capture_output=True, timeout=900)
        asm = open(out).read()
    table = {}
    for blk in re.split(r"\n  - ", asm[asm.index("amdhsa.kernels:"):])[1:]:
        def get(key, blk=blk):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        table[get("name")] = {k: int(get(k)) for k in ("group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count",
                                                       "vgpr_spill_count", "max_flat_workgroup_size")}
    return asm, table


def _one(table, fragment):
    # (not the <RETRY = 1> instantiations, nor resolve's <_, MARK = true> unless asked for by its mangled arguments)
    hits = [v for k, v in table.items() if fragment in k and "ILj1E" not in k and ("Lb1E" not in k or "Lb1E" in fragment)]
    assert len(hits) == 1, (fragment, list(table))
    return hits[0]


def test_pipeline_kernel_resources():
    asm, table = _kernels("pinflate2")
    dec = _one(table, "pinf2_decode_kernel")
    assert dec["group_segment_fixed_size"] <= 10240          # 16 one-wave workgroups per CU: as many as 128 registers allow
    assert dec["vgpr_count"] <= 128                          # >= 4 waves per SIMD
    assert dec["private_segment_fixed_size"] == 0 and dec["vgpr_spill_count"] == 0
    # resolve<RETRY, MARK, BIG>: whole streams / first parts on 8 KiB tiles; the marker parts on 4 KiB tiles when a batch has more of
    # them than CUs (two workgroups per CU: 2 x 80 KB), on 8 KiB tiles otherwise (a 64 KiB ring of symbols + the big tile: one per CU)
    def rk(frag):
        hits = [v for k, v in table.items() if "pinf2_resolve_kernel" + frag in k]
        assert len(hits) == 1, (frag, [k for k in table if "resolve" in k])
        return hits[0]
    res = rk("ILj0ELb0ELb1E")
    assert res["group_segment_fixed_size"] <= 65536          # (static LDS) and two 512-thread workgroups per CU
    assert res["vgpr_count"] <= 128 and res["private_segment_fixed_size"] == 0 and res["max_flat_workgroup_size"] == 512
    small, big = rk("ILj0ELb1ELb0E"), rk("ILj0ELb1ELb1E")
    assert small["group_segment_fixed_size"] <= 81920 and small["vgpr_count"] <= 128 and small["private_segment_fixed_size"] == 0
    assert big["group_segment_fixed_size"] <= 98304 and big["private_segment_fixed_size"] == 0
    find = _one(table, "pinf2_find_kernel")
    assert find["private_segment_fixed_size"] == 0 and find["group_segment_fixed_size"] <= 16384
    # LDS and global memory are reached with their own instructions
    assert len(re.findall(r"^\s+flat_(load|store)", asm, re.M)) <= 8


def test_unfilter_kernel_resources():
    _, table = _kernels("unfilter")
    k4 = [v for k, v in table.items() if "unfilter_kernel" in k]
    assert k4
    for v in k4:
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0
        assert v["group_segment_fixed_size"] <= 81920        # >= 2 workgroups per CU (LDS tiles set the occupancy)
    # the line-aligned form (bpp 4 / 8): a ring of two 128-byte tiles per row, two 4-wave workgroups per CU
    pk = [v for k, v in table.items() if "unfilter_pk_kernel" in k]
    assert len(pk) == 2
    for v in pk:
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0
        assert v["group_segment_fixed_size"] <= 81920 and v["vgpr_count"] <= 256 and v["max_flat_workgroup_size"] == 256


def test_deflate_kernel_resources():
    _, table = _kernels("deflate")
    full = [v for k, v in table.items() if "deflate_full_kernel" in k]
    assert full
    for v in full:
        assert v["private_segment_fixed_size"] <= 64 and v["vgpr_spill_count"] <= 10     # (40 bytes today, outside the passes' inner loops)
        assert v["group_segment_fixed_size"] <= 81920           # (the helper-wave form: 76 KiB, one workgroup of four waves per stream)
    # the kernels of a round.  The search workgroup (every level) is a whole CU's worth of waves with its window in
    # LDS -- and must leave room for ONE level >= 8 parse wave beside it (batches of <= 256 streams: the search of round r + 1 runs
    # beside the parse of round r); four parse workgroups per CU at either kind of level: all 1024 streams of BASELINE configs[3]
    # resident
    search = [v for k, v in table.items() if "dfl3_search_kernel" in k or "dfl3_search_fast_kernel" in k]
    parse = [v for k, v in table.items() if "dfl2_parse_kernel" in k]
    parse3 = [v for k, v in table.items() if "dfl3_parse_kernel" in k]
    assert len(search) == 2 and len(parse) == 1 and len(parse3) == 1
    walk = [v for k, v in table.items() if "dfl4_walk_kernel" in k]
    block = [v for k, v in table.items() if "dfl4_block_kernel" in k]
    assert len(walk) == 1 and len(block) == 1
    for k, v in table.items():
        if "dfl3_search" not in k:
            continue
        assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 128 and v["max_flat_workgroup_size"] == 1024
        if "fast" in k:     # levels 0-7: beside it a walk wave and a block wave of the round before
            assert v["group_segment_fixed_size"] + walk[0]["group_segment_fixed_size"] + block[0]["group_segment_fixed_size"] <= 163840
        else:               # levels >= 8: beside it one parse wave
            assert v["group_segment_fixed_size"] + parse[0]["group_segment_fixed_size"] <= 163840
    assert block[0]["group_segment_fixed_size"] <= 10752 and block[0]["private_segment_fixed_size"] <= 64      # fifteen block waves per CU
    # (scratch: 384 bytes, the by-value argument structs of the non-inlined passes at their call sites -- a few per block -- and
    #  callee-saved registers; nothing inside the passes' loops)
    assert parse[0]["group_segment_fixed_size"] <= 40960 and parse[0]["private_segment_fixed_size"] <= 512 and parse[0]["max_flat_workgroup_size"] == 64
    assert parse3[0]["group_segment_fixed_size"] <= 40960 and parse3[0]["private_segment_fixed_size"] <= 64 and parse3[0]["max_flat_workgroup_size"] == 128
