"""The kernels of csrc/chunks.hip and the wave-parallel CRC-32 of csrc/crc32.hpp (chunk lexing + CRC-32 + IDAT assembly /
emission) run on the CPU by the wave emulator of tools/emu: the CRC against zlib.crc32 over lengths and alignments around every piece
size, the lexer against the test-side lexer on PngSuite fixtures and damaged files, the IDAT writer against a restatement."""
import os
import shutil
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))

import pnghelp as ph  # noqa: E402


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    import prep_deflate
    d = tmp_path_factory.mktemp("emu_chunks")
    inc = d / "chunks_emu.inc"
    inc.write_text(prep_deflate.prepare_plain(open(os.path.join(ROOT, "swift_png_amd", "csrc", "chunks.hip")).read()))
    out = d / "emu_chunks"
    subprocess.run(["g++", "-O1", "-std=c++17", "-DSPNG_EMU", f'-DEMU_CHUNKS_SRC="{inc}"', "-I" + os.path.join(ROOT, "tools", "emu"),
                    "-I" + os.path.join(ROOT, "swift_png_amd", "csrc"), "-x", "c++", "-fpermissive", "-Wno-attributes", "-w", "-o", str(out),
                    os.path.join(ROOT, "tools", "emu", "emu_chunks.cpp")], check=True, capture_output=True, timeout=600)
    return out


def test_emulated_crc32_over_lengths_and_alignments(emu, tmp_path):
    """pieces are a power of two long and right-aligned: lengths around 64 x 16, 64 x 32, ... (where the piece length doubles),
    lengths that leave the first lanes empty, odd offsets (unaligned 16-byte loads), a running CRC carried in"""
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    (tmp_path / "d").write_bytes(data)
    lengths = [0, 1, 2, 15, 16, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 1040, 2047, 2048, 2049, 4096, 4100, 8191, 8192, 8196, 8193, 16384,
               16385, 65535, 65536, 65537, 100000, 262144, 262145, 299990]
    for k, n in enumerate(lengths):
        for off, running in ((0, 0), (1 + k % 7, 0), (3, zlib.crc32(b"IDAT"))):
            if off + n > len(data):
                continue
            r = subprocess.run([str(emu), "crc", str(tmp_path / "d"), str(off), str(n), f"{running:x}"], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-300:]
            assert int(r.stdout.strip(), 16) == zlib.crc32(data[off:off + n], running), (n, off, running)


@pytest.mark.parametrize("name", ["common/basn6a08.png", "common/basi6a16.png", "common/oi9n2c16.png", "common/tbbn3p08.png", "ios/PngSuite.png",
                                  "common/ct1n0g04.png", "common/ch2n3p08.png"])
@pytest.mark.parametrize("waves,cap", [(1, 4096), (4, 4096), (3, 2)])
def test_emulated_lexer_on_fixtures(emu, tmp_path, name, waves, cap):
    """walk -> chunks -> finish: status, IHDR fields and the concatenated IDAT payloads equal the test-side lexer's; with a list of two
    entries the walk wave checks and copies the rest itself"""
    f = (ph.GOLDEN / "pngsuite" / name).read_bytes()
    png = ph.parse_png(f)
    (tmp_path / "f.png").write_bytes(f)
    r = subprocess.run([str(emu), "lex", str(tmp_path / "f.png"), str(tmp_path / "idat"), str(cap), str(waves)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-300:]
    v = r.stdout.split()
    assert int(v[0]) == 0, r.stdout
    assert (int(v[4]), int(v[5]), int(v[6]), int(v[7]), int(v[8]), int(v[9])) == (png.width, png.height, png.depth, png.color, int(png.interlaced), int(png.ios))
    assert int(v[10]) == len(png.idat) and (tmp_path / "idat").read_bytes() == png.idat


def test_emulated_lexer_reports_the_first_bad_checksum(emu, tmp_path):
    """a flipped payload byte in the third IDAT of oi9n2c16: invalidChunkChecksum(declared, computed) of THAT chunk, the IDAT bytes in
    front of it, whatever the later chunks say"""
    f = bytearray((ph.GOLDEN / "pngsuite" / "common" / "oi9n2c16.png").read_bytes())
    pos, idats = 8, []
    while pos < len(f):
        (n,) = struct.unpack(">I", f[pos:pos + 4])
        if f[pos + 4:pos + 8] == b"IDAT":
            idats.append((pos, n))
        pos += 12 + n
    at, n = idats[2]
    f[at + 8] ^= 0x40
    declared = struct.unpack(">I", f[at + 8 + n:at + 12 + n])[0]
    computed = zlib.crc32(bytes(f[at + 4:at + 8 + n]))
    (tmp_path / "f.png").write_bytes(bytes(f))
    for waves in (1, 4):
        r = subprocess.run([str(emu), "lex", str(tmp_path / "f.png"), str(tmp_path / "idat"), "4096", str(waves)], capture_output=True, text=True, timeout=300)
        v = r.stdout.split()
        assert int(v[0]) != 0 and (int(v[2], 16), int(v[3], 16)) == (declared, computed), r.stdout
        assert int(v[10]) == sum(k for _, k in idats[:2])


def test_emulated_idat_writer(emu, tmp_path):
    rng = np.random.default_rng(4)
    z = zlib.compress(rng.integers(0, 7, 100000, dtype=np.uint8).tobytes(), 6)
    (tmp_path / "z").write_bytes(z)
    for piece in (7, 8192, 65536, len(z), len(z) + 5):
        zz = z[:300] if piece == 7 else z
        (tmp_path / "z").write_bytes(zz)
        r = subprocess.run([str(emu), "write", str(tmp_path / "z"), str(piece), str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr[-200:])
        want = b"".join(struct.pack(">I", len(zz[i:i + piece])) + b"IDAT" + zz[i:i + piece] + struct.pack(">I", zlib.crc32(b"IDAT" + zz[i:i + piece]))
                        for i in range(0, len(zz), piece))
        assert (tmp_path / "out").read_bytes() == want, piece
