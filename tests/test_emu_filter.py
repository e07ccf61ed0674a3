"""filter_kernel of csrc/encode.hip (PNG.Encoder.filter: the five residuals of a scanline scored, first strict minimum kept) run on
the CPU by the wave emulator of tools/emu (host compiler: the ROCm clang++) against the oracle's filtered scanlines."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
sys.path.insert(0, ROOT)

import pnghelp as ph  # noqa: E402

CLANG = os.environ.get("SPNG_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("clang++ not available")
    import prep_deflate
    d = tmp_path_factory.mktemp("emu_filter")
    inc = d / "encode_emu.inc"
    inc.write_text(prep_deflate.prepare_plain(open(os.path.join(ROOT, "swift_png_amd", "csrc", "encode.hip")).read()))
    out = d / "emu_filter"
    subprocess.run([CLANG, "-O1", "-std=c++17", "-DSPNG_EMU", f'-DEMU_FILTER_SRC="{inc}"', "-I" + os.path.join(ROOT, "tools", "emu"),
                    "-I" + os.path.join(ROOT, "swift_png_amd", "csrc"), "-x", "c++", "-w", "-o", str(out),
                    os.path.join(ROOT, "tools", "emu", "emu_filter.cpp")], check=True, capture_output=True, timeout=600)
    return out


CASES = [("rgba8 noise", 64, 20, 8, 4, "noise"), ("rgba8 photograph", 256, 64, 8, 4, "synth"), ("rgba8 flat", 128, 16, 8, 4, "flat"),
         ("rgba8 gradient", 200, 33, 8, 4, "gradient"), ("rgb8", 100, 30, 8, 3, "noise"), ("rgba16", 60, 40, 16, 4, "noise"),
         ("gray8", 333, 17, 8, 1, "gradient"), ("va8", 50, 50, 8, 2, "noise"), ("gray4", 99, 12, 4, 1, "noise"), ("gray1", 77, 9, 1, 1, "noise"),
         ("rgb16 one row", 19, 1, 16, 3, "noise"),
         # samples below a byte: rows packed once into LDS (round 6) -- several steps of 64 bytes, and a row too long for the buffer
         ("gray2", 130, 11, 2, 1, "noise"), ("gray1 wide", 3001, 7, 1, 1, "noise"), ("gray4 wide", 1500, 5, 4, 1, "gradient"),
         ("gray1 beyond the buffer", 20000, 3, 1, 1, "noise"),
         # rows wider than a wave's step of 1 KiB: the aligned store takes the bytes in front of lane 0 from lane 63 of the step before
         ("rgba8 wide", 600, 40, 8, 4, "synth"), ("rgba16 wide", 260, 18, 16, 4, "noise"), ("rgb8 wide", 1024, 19, 8, 3, "gradient")]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_emulated_filter_matches_the_oracle(emu, tmp_path, case):
    name, w, h, depth, channels, kind = case
    pitch = (w * depth * channels + 7) // 8
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
    if depth < 8:
        img = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)         # (PNG.Image.storage: one byte per sample below 8 bits)
    elif kind == "noise":
        img = rng.integers(0, 256, (h, pitch), dtype=np.uint8)
    elif kind == "flat":
        img = np.full((h, pitch), 77, dtype=np.uint8)
    elif kind == "gradient":
        y, x = np.mgrid[0:h, 0:pitch]
        img = ((x * 3 + y * 5) % 256).astype(np.uint8)
    else:
        from swift_png_amd import synth
        img = synth.image(1, w, h, channels, depth).reshape(h, -1)
    want = ph.orc_filter(img.reshape(-1), w, h, depth, channels, False)
    (tmp_path / "in").write_bytes(img.tobytes())
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(w), str(h), str(depth), str(channels)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-300:])
