"""The scanline kernels of csrc/unfilter.hip -- unfilter_pk_kernel<4 | 8> (the line-aligned wavefront of the 4- and 8-byte pixel
formats: packed arithmetic, the skew kept in an LDS ring, waves of a workgroup handing rows to each other) and the byte-wise
unfilter_kernel<1 | 2 | 3 | 6> -- run on the CPU by the wave emulator of tools/emu and compared with the oracle's rows.  Host
compiler: the ROCm clang++ (the kernel is written with clang's vector extensions); the source is a prepared copy (launches blanked,
compiler-only barriers turned into meetings of the wave: tools/emu/prep_deflate.py).  Timing and memory ordering are not modelled;
the `-m gpu` tests remain the parity tests proper."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))

import pnghelp as ph  # noqa: E402

CLANG = os.environ.get("SPNG_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("clang++ not available")
    import prep_deflate
    d = tmp_path_factory.mktemp("emu_unfilter")
    inc = d / "unfilter_emu.inc"
    inc.write_text(prep_deflate.prepare_plain(open(os.path.join(ROOT, "swift_png_amd", "csrc", "unfilter.hip")).read()))
    out = d / "emu_unfilter"
    subprocess.run([CLANG, "-O1", "-std=c++17", "-DSPNG_EMU", f'-DEMU_UNFILTER_SRC="{inc}"', "-I" + os.path.join(ROOT, "tools", "emu"),
                    "-I" + os.path.join(ROOT, "swift_png_amd", "csrc"), "-x", "c++", "-w", "-o", str(out),
                    os.path.join(ROOT, "tools", "emu", "emu_unfilter.cpp")], check=True, capture_output=True, timeout=600)
    return out


# (name, width, height, channels, depth, rows per piece, filter types by row -- None: the reference heuristic's choice)
def _f(seed, n):
    return [int(x) for x in np.random.default_rng(seed).integers(0, 5, n)]


CASES = [
    ("rgba8 none", 40, 20, 4, 8, 64, [0]), ("rgba8 sub", 40, 20, 4, 8, 64, [1]), ("rgba8 up", 40, 20, 4, 8, 64, [0, 2]),
    ("rgba8 average", 40, 20, 4, 8, 64, [0, 3]), ("rgba8 paeth", 40, 20, 4, 8, 64, [0, 4]),
    ("rgba8 heuristic", 1000, 300, 4, 8, 64, None), ("rgba8 mixed, 10 pieces", 1000, 300, 4, 8, 32, _f(1, 97)),
    ("rgba16 mixed", 333, 200, 4, 16, 32, _f(2, 89)), ("va16 mixed", 777, 150, 2, 16, 64, _f(3, 31)),
    ("rgba8 one pixel", 1, 1, 4, 8, 64, None), ("rgba8 narrow and tall", 3, 700, 4, 8, 32, [4, 4, 4, 3, 2, 1, 0, 4]),
    ("rgba8 one wide band", 5000, 40, 4, 8, 64, _f(4, 40)), ("rgba16 tall", 31, 1000, 4, 16, 128, _f(5, 1000)),
    ("rgba8 pieces only where None / Sub allow", 200, 400, 4, 8, 32, [0] + [4] * 150 + [1] + [3] * 90),
    ("gray8", 500, 200, 1, 8, 64, _f(6, 53)), ("va8", 500, 200, 2, 8, 64, _f(7, 53)), ("rgb8", 500, 200, 3, 8, 32, _f(8, 53)),
    ("rgb16", 301, 170, 3, 16, 64, _f(9, 53)),
    # the packed form of the 3- and 6-byte pixels (round 5): every filter alone, widths around the 48-byte steps and 192-byte tiles
    ("rgb8 sub", 70, 10, 3, 8, 64, [1]), ("rgb8 up", 70, 10, 3, 8, 64, [0, 2]), ("rgb8 average", 70, 10, 3, 8, 64, [0, 3]),
    ("rgb8 paeth", 70, 10, 3, 8, 64, [1, 4]), ("rgb16 paeth", 70, 10, 3, 16, 64, [1, 4]), ("rgb16 average", 33, 70, 3, 16, 64, [3, 3, 1]),
    ("rgb8 one pixel", 1, 1, 3, 8, 64, [4]), ("rgb8 sixteen pixels", 16, 3, 3, 8, 64, [4, 3, 4]), ("rgb8 seventeen pixels", 17, 130, 3, 8, 64, _f(10, 130)),
    ("rgb8 wide, three bands", 1500, 150, 3, 8, 64, _f(11, 150)), ("rgb16 wide, pieces", 900, 200, 3, 16, 32, _f(12, 61)),
    ("rgb8 no paeth row in a band", 300, 140, 3, 8, 128, [1, 2, 3, 0, 2] * 20 + [4] * 40),
    # pixels of 1 and 2 bytes in dword units (round 6): every filter alone, rows that end inside a dword, and rows of 2 KiB and more
    # (tiles of 32 units instead of 16)
    ("gray8 sub", 70, 10, 1, 8, 64, [1]), ("gray8 average", 70, 10, 1, 8, 64, [0, 3]), ("gray8 paeth", 71, 10, 1, 8, 64, [1, 4]),
    ("gray8 one pixel", 1, 1, 1, 8, 64, [4]), ("gray8 five pixels", 5, 130, 1, 8, 64, _f(13, 130)), ("va8 paeth", 33, 20, 2, 8, 64, [2, 4]),
    ("gray16 mixed", 301, 70, 1, 16, 64, _f(14, 70)), ("gray8 wide", 3001, 140, 1, 8, 64, _f(15, 140)), ("va8 wide, pieces", 1100, 200, 2, 8, 32, _f(16, 61)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_emulated_unfilter_matches_the_oracle(emu, tmp_path, case):
    name, w, h, channels, depth, piece_rows, filters = case
    bpp = channels * depth // 8
    pitch = w * bpp
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
    img = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    rows = bytearray(ph.orc_filter(img.reshape(-1), w, h, depth, channels, False))
    if filters is not None:                                 # (any filter byte over any payload is a valid input of the UNfilter)
        for y in range(h):
            rows[y * (pitch + 1)] = filters[y % len(filters)]
    rows = bytes(rows)
    st, want = ph.orc_unfilter(rows, w, h, depth, channels, False)
    assert st == 0
    (tmp_path / "in").write_bytes(rows)
    (tmp_path / "want").write_bytes(bytes(want)[:h * pitch])
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(pitch), str(h), str(bpp), str(piece_rows)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-300:])
