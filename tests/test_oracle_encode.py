"""CPU oracle, encode side: filter selection + DEFLATE restatement.

Pins: the 28 streams swift-png itself committed (Tests/Outputs, level 9: filter heuristic + full
shortest-path search + Huffman construction, bit for bit), the match-finder known-answer test of
LZ77Tests/Bitstreams.swift:97-185, and round trips mirroring LZ77Tests/Compression.swift:7-27 and
CompressionMicro.swift through the (independently pinned) inflate oracle and zlib."""
import ctypes
import hashlib
import json
import zlib

import numpy as np
import pytest

import pnghelp as ph

ENC = json.loads((ph.GOLDEN / "encode.json").read_text())
LOCAL = sorted(p.name[:-len(".baseline.png")] for p in (ph.GOLDEN / "encode").glob("*.baseline.png"))


def _encode9(png, storage):
    lib = ph.oracle()
    cap = lib.orc_deflate_bound(len(storage)) + (1 << 20)
    dst = np.empty(cap, np.uint8)
    w = ctypes.c_size_t(0)
    st = lib.orc_encode(ph._ptr(storage), png.width, png.height, png.depth, png.channels, int(png.interlaced),
                        0, 9, ph._ptr(dst), cap, ctypes.byref(w))
    assert st == 0
    return dst[:w.value].tobytes()


@pytest.mark.parametrize("name", LOCAL)
def test_level9_stream_is_bit_exact_local(name):
    """All 28 inputs of Tests/Baselines travel with the repo; the stream the oracle makes of each must have the digest of
    the IDAT data swift-png committed under Tests/Outputs (and equal it byte for byte where that file was copied too)."""
    src = ph.parse_png((ph.GOLDEN / "encode" / (name + ".baseline.png")).read_bytes())
    st, storage, _ = ph.orc_decode(src)
    assert st == 0
    mine = _encode9(src, storage)
    assert len(mine) == ENC[name]["idat_len"] and hashlib.sha256(mine).hexdigest() == ENC[name]["idat_sha256"]
    out = ph.GOLDEN / "encode" / (name + ".swiftpng9.png")
    if out.exists():
        gold = ph.parse_png(out.read_bytes())
        assert mine == gold.idat
        # and swift-png's output decodes back to the same raster (Compression.swift:56-84)
        assert (ph.orc_decode(gold)[1] == storage).all()


def test_all_reference_level9_goldens_travel():
    assert len(LOCAL) == 28 == len(ENC)


def test_stored_tail_departure_of_the_row_pushed_reference():
    """DESIGN section 1 (iii).  PNG.Encoder pushes one scanline at a time and finishes with push([], last: true);
    when the last non-final compress() leaves one or two bytes queued, compressBlocks(final:) takes the stored-tail
    path and drops the pending terms (LZ77.DeflatorBuffers.Stream.swift:45-60): the reference's own stream is corrupt.
    A flat 1033 x 4 RGBA8 raster at level 6 is such an input (rows of 4133 bytes: every push compresses).  The
    row-pushed restatement reproduces the bug; the one-shot stream -- what the device emits (test_gpu_decode.py::
    test_encode_departs_from_the_stored_tail_bug) -- is the correct one."""
    w, h = 1033, 4
    storage = np.tile(np.array([9, 200, 31, 255], np.uint8), w * h)
    rows = ph.orc_filter(storage, w, h, 8, 4, False)
    lib = ph.oracle()
    cap = len(storage) * 2 + 4096
    dst = np.empty(cap, np.uint8)
    wr = ctypes.c_size_t(0)
    assert lib.orc_encode(ph._ptr(storage), w, h, 8, 4, 0, 0, 6, ph._ptr(dst), cap, ctypes.byref(wr)) == 0
    pushed = dst[:wr.value].tobytes()
    oneshot = ph.orc_deflate(bytes(rows), 6)
    assert pushed != oneshot and len(pushed) < len(oneshot)
    assert zlib.decompress(oneshot) == bytes(rows)
    with pytest.raises(zlib.error):
        zlib.decompress(pushed)
    assert ph.orc_inflate(pushed, 0, cap=len(rows) + 16)[0] != 0          # (the pinned inflate oracle rejects it as well)


@pytest.mark.skipif(not ph.have_reference(), reason="reference checkout not mounted")
@pytest.mark.parametrize("name", sorted(ENC))
def test_level9_stream_is_bit_exact_reference(name):
    base = ph.REFERENCE / "Tests"
    src = ph.parse_png((base / "Baselines" / (name + ".png")).read_bytes())
    gold = ph.parse_png((base / "Outputs" / (name + ".png")).read_bytes())
    assert hashlib.sha256(gold.idat).hexdigest() == ENC[name]["idat_sha256"]
    st, storage, _ = ph.orc_decode(src)
    assert st == 0
    assert _encode9(src, storage) == gold.idat


def test_matching_kat():
    """LZ77Tests/Bitstreams.swift:97-185, window exponent 4, attempts = goal = .max."""
    segments = [
        [1, 2, 3, 3, 1, 2, 3, 3, 1, 2, 3, 1, 2, 2, 2, 2, 2, 2, 0, 1, 2],
        [2, 2, 2, 2, 0, 1, 2, 2, 0, 0, 0, 0, 2, 3, 2, 1, 2, 3, 3, 1, 5],
        [1, 1, 3, 3, 1, 2, 3, 1, 2, 4, 4, 2, 1],
    ]
    want = [[1], [2], [3], [3], [1, 2, 3, 3, 1, 2, 3], [1], [2], [2], [2], [2], [2], [2], [0], [1, 2, 2, 2, 2, 2],
            [0], [1], [2], [2], [0], [0], [0], [0], [2], [3], [2], [1], [2], [3], [3], [1], [5], [1], [1], [3], [3],
            [1], [2], [3], [1], [2], [4], [4], [2], [1]]
    lib = ph.oracle()
    data = np.array([b for s in segments for b in s], dtype=np.uint8)
    lens = (ctypes.c_int * 3)(*[len(s) for s in segments])
    out = np.zeros(512, np.uint8)
    lib.orc_kat_matching.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    n = lib.orc_kat_matching(ph._ptr(data), lens, 3, 4, 10, ph._ptr(out), 512)
    got, i = [], 0
    while i < n:
        k = int(out[i]); got.append(out[i + 1:i + 1 + k].tolist()); i += 1 + k
    assert got == want


@pytest.mark.parametrize("count", [5, 15, 100, 200, 2000, 5000])
@pytest.mark.parametrize("level", [4, 7, 9])
def test_roundtrip_small_window(level, count):
    """LZ77Tests/Compression.swift:7-27: random bytes, exponent 8 (256-byte window)."""
    rng = np.random.default_rng(level * 10000 + count)
    data = rng.integers(0, 256, count, dtype=np.uint8).tobytes()
    z = ph.orc_deflate(data, level, exponent=8)
    assert z[0] == 0x08                                      # CINFO = 0
    assert zlib.decompress(z) == data
    assert ph.orc_inflate(z)[:2] == (0, data)


@pytest.mark.parametrize("level", range(0, 14))
def test_roundtrip_all_levels(level):
    rng = np.random.default_rng(level)
    for n in (0, 1, 2, 3, 4, 300, 40000):
        data = (rng.integers(0, 256, n, dtype=np.uint8) * (rng.random(n) < 0.4)).astype(np.uint8).tobytes()
        for fmt in (0, 1):
            z = ph.orc_deflate(data, level, fmt)
            st, out, consumed, _ = ph.orc_inflate(z, fmt, cap=n + 16)
            assert (st, out, consumed) == (0, data, len(z))
        assert z if n else True


def test_stream_shape_matches_reference_probe():
    """SURVEY section 4 probe of Tests/Outputs: header 78 01, dynamic blocks only, level-9 block
    payloads covering 2047, 4095, 8191 ... input positions (graph limit doubling)."""
    rng = np.random.default_rng(5)
    data = rng.integers(0, 4, 40000, dtype=np.uint8).tobytes()
    z = ph.orc_deflate(data, 9)
    assert z[:2] == b"\x78\x01"
    st, out, _, _ = ph.orc_inflate(z)
    assert st == 0 and out == data
    # greedy/lazy: a block closes after 2047 (2046/2047) terms; terms <= bytes, so >= n/2047/258 blocks exist
    z6 = ph.orc_deflate(rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(), 6)
    assert z6[:2] == b"\x78\x01"


def test_filter_roundtrip_all_delays():
    """PNGTests/Filtering.swift:9-64: Encoder.filter -> Decoder.defilter identity, delay 1...8."""
    lib = ph.oracle()
    for delay in range(1, 9):
        rng = np.random.default_rng(delay)
        last = np.zeros(1 + 24 * delay, np.uint8)
        seen = set()
        for _ in range(16):
            line = np.concatenate([[0], rng.integers(0, 256, 24 * delay)]).astype(np.uint8)
            if rng.random() < 0.5:                           # smooth rows make the other filters win
                line[1:] = (np.arange(24 * delay) * 3 + rng.integers(0, 3, 24 * delay)).astype(np.uint8)
            out = np.zeros_like(line)
            seen.add(lib.orc_filter_row(ph._ptr(line), ph._ptr(last), len(line), delay, ph._ptr(out)))
            back = out.copy()
            lib.orc_defilter(ph._ptr(back), ph._ptr(last), len(line), delay)
            assert (back[1:] == line[1:]).all()
            last = line
        assert len(seen) >= 2


def test_levels_other_than_9_follow_the_published_size_curve():
    """A SOFT pin for the levels the reference holds no golden stream of ("parity unpinned" beyond round trips).
    Benchmarks/README.md:72-206 publishes, for Tests/Baselines/rgb8-color-photographic.png at commit 89aa614, swift-png's file
    size as a percentage of libpng's at the same level (levels 10-13: of libpng's level 9).  The libpng side is Pillow 12 /
    zlib here -- an independent encoder with libpng's filter heuristic, ~1 % larger than libpng itself -- so the comparison is
    of the CURVE: level 0 absolutely (libpng stores at level 0: its size is the raster's), every other level relative to level 9,
    whose stream is pinned bit for bit (test_level9_reference_outputs).  A restatement that searched less, closed blocks
    elsewhere or costed its trees differently at some level would leave the curve there.  Tolerances: 0.35 percentage points
    for levels 4-13 (measured: <= 0.25), 1.2 for levels 1-3 (zlib's deflate_fast levels, where Pillow and libpng differ more)."""
    import io
    Image = pytest.importorskip("PIL.Image")
    published = {0: 58.4, 1: 99.64, 2: 99.78, 3: 99.99, 4: 100.93, 5: 101.1, 6: 101.71, 7: 101.84, 8: 99.13, 9: 98.27,
                 10: 98.04, 11: 97.89, 12: 97.82, 13: 97.74}
    data = (ph.GOLDEN / "encode" / "rgb8-color-photographic.baseline.png").read_bytes()
    png = ph.parse_png(data)
    st, storage, _ = ph.orc_decode(png)
    assert st == 0
    rows = ph.orc_filter(storage, png.width, png.height, png.depth, png.channels, False)
    im = Image.open(io.BytesIO(data))
    im.load()

    def libpng_like(level):
        b = io.BytesIO()
        im.save(b, "PNG", compress_level=level)
        return len(b.getvalue())

    ours = {}
    for level in range(14):
        z = ph.orc_deflate(rows, level)
        size = 8 + 25 + 12 + len(z) + 12 * ((len(z) + 32767) // 32768)         # signature, IHDR, IEND, IDAT chunks of 32 KiB
        ours[level] = 100.0 * size / libpng_like(min(level, 9))
    assert abs(ours[0] - published[0]) <= 0.3, ours[0]
    gap = ours[9] - published[9]                                                 # (Pillow vs libpng, at the pinned level)
    assert -2.0 <= gap <= 0.0, gap
    for level in range(1, 14):
        tol = 1.2 if level <= 3 else 0.35
        assert abs(ours[level] - published[level] - gap) <= tol, (level, ours[level], published[level], gap)
