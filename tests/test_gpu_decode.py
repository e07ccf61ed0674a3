"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs, against
the committed PngSuite goldens, and through the reference-shaped Python mirror.  Bit-exact."""
import hashlib
import json
import zlib

import numpy as np
import pytest

import pnghelp as ph
import swift_png_amd as spng
from test_oracle_decode import _dynamic_header, _payloads

pytestmark = pytest.mark.gpu
TABLE = json.loads((ph.GOLDEN / "pngsuite.json").read_text())


def _same_inflate(gpu, data, fmt=0, cap=None):
    cap = cap if cap is not None else max(1 << 16, 1100 * len(data))
    want = ph.orc_inflate(data, fmt, cap)
    got = gpu.inflate(data, fmt, cap)
    assert got[0] == want[0], (got[0], want[0])
    assert got[1] == want[1]
    assert got[3] == want[3]
    if want[0] == 0:
        assert got[2] == want[2]
    return got


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("kind", sorted(_payloads()))
def test_inflate_vs_oracle(gpu, kind, level):
    s = gpu.load()
    data = _payloads()[kind]
    z = zlib.compress(data, level)
    st, out, consumed, _ = _same_inflate(s, z, 0, len(data) + 16)
    assert (st, out, consumed) == (0, data, len(z))
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    st, out, consumed, _ = _same_inflate(s, raw, 1, len(data) + 16)
    assert (st, out, consumed) == (0, data, len(raw))


def test_inflate_fixed_and_window_sizes(gpu):
    s = gpu.load()
    data = b"abcabcabcabc" * 50 + bytes(range(200))
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    _same_inflate(s, co.compress(data) + co.flush())
    rng = np.random.default_rng(11)
    # long-distance matches (beyond the 32 KiB - 258 LDS window) and every wbits
    block = rng.integers(0, 256, 40000, dtype=np.uint8).tobytes()
    big = block[:32600] + block[:300] + block[100:33000] + block[:5000]
    for wbits in (9, 12, 15):
        co = zlib.compressobj(9, zlib.DEFLATED, wbits)
        z = co.compress(big) + co.flush()
        st, out, _, _ = _same_inflate(s, z, 0, len(big) + 16)
        assert st == 0 and out == big


def test_inflate_truncated_and_capacity(gpu):
    s = gpu.load()
    data = _payloads()["text"]
    z = zlib.compress(data, 6)
    for cut in (0, 1, 2, 3, 10, len(z) // 2, len(z) - 5, len(z) - 1):
        st, out, _, _ = _same_inflate(s, z[:cut], 0, len(data) + 16)
        assert st == 1 and data.startswith(out)
    for cap in (0, 1, 100, len(data) - 1):
        got = s.inflate(z, 0, cap)
        want = ph.orc_inflate(z, 0, cap)
        assert got[0] == want[0] == 64
    z0 = zlib.compress(_payloads()["noise"], 0)            # stored blocks, truncated mid-block
    for cut in (7, 100, 65535 + 20, len(z0) - 3):
        _same_inflate(s, z0[:cut], 0, 80000)


def test_inflate_error_vocabulary(gpu):
    s = gpu.load()
    good = zlib.compress(b"hello hello hello hello", 9)
    bad = bytearray(good); bad[-1] ^= 1
    cases = [
        (b"\x77\x01" + good[2:], 0), (b"\x88\x01" + good[2:], 0), (b"\x78\x02" + good[2:], 0),
        (b"\x78\x20" + good[2:], 0), (bytes(bad), 0), (b"\x78\x01\x07", 0),
        (b"\x78\x01\x01\x03\x00\xfc\xfe\x00\x00\x00", 0),
        (bytes([0x05 | (31 << 3) & 0xff, (31 >> 5), 0, 0, 0, 0, 0, 0]), 1),
        (bytes([0x05, 0, 0, 0, 0, 0, 0, 0, 0, 0]), 1),
        (_dynamic_header(257, 1, [1, 0, 0, 1], "1" + "00"), 1),
    ]
    clens = [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]
    cases.append((_dynamic_header(257, 1, clens, "1" + "1111111" + "1" + "1111111"), 1))
    cases.append((_dynamic_header(257, 1, clens, "1" + "1111111" + "1" + format(120 - 11, "07b")[::-1]), 1))
    bits = "1" + "10" + "0000001" + "00000"
    cases.append((bytes(int("".join(reversed(bits[i:i + 8].ljust(8, "0"))), 2) for i in range(0, len(bits), 8)) + b"\0\0", 1))
    seen = set()
    for data, fmt in cases:
        seen.add(_same_inflate(s, data, fmt, 4096)[0])
    assert seen == {16, 17, 18, 19, 32, 33, 34, 35, 36, 37, 38, 39}


def test_inflate_batch_many_streams(gpu):
    s = gpu.load()
    rng = np.random.default_rng(5)
    datas, zs = [], []
    for i in range(70):
        n = int(rng.integers(0, 200000))
        d = (rng.integers(0, 256, n, dtype=np.uint8) * (rng.random(n) < rng.random())).astype(np.uint8).tobytes()
        datas.append(d)
        zs.append(zlib.compress(d, int(rng.integers(0, 10))))
    outs, res = s.inflate_batch([s.to_device(z) for z in zs], [len(d) + 32 for d in datas])
    for d, z, o, r in zip(datas, zs, outs, res):
        assert r.status == 0 and r.written == len(d) and r.consumed == len(z)
        assert bytes(o[:len(d)].cpu().numpy()) == d


FORMATS = [(1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (8, 3), (16, 3), (8, 4), (16, 4)]


@pytest.mark.parametrize("interlaced", [False, True])
@pytest.mark.parametrize("depth,channels", FORMATS)
def test_unfilter_vs_oracle(gpu, depth, channels, interlaced):
    s = gpu.load()
    rng = np.random.default_rng(depth * 100 + channels * 10 + interlaced)
    for (w, h) in [(1, 1), (3, 2), (8, 8), (33, 70), (130, 131), (257, 65)]:
        u = gpu.inflated_size(w, h, depth, channels, interlaced)
        rows = rng.integers(0, 256, u, dtype=np.uint8)
        # put a valid-ish filter byte at the start of every row (plus a few invalid ones: treated as None)
        off = 0
        for p in _passes(w, h, depth * channels, interlaced):
            for y in range(p[1]):
                rows[off] = rng.integers(0, 6) if rng.random() < 0.9 else rng.integers(5, 256)
                off += p[0] + 1
        st_o, want = ph.orc_unfilter(rows.tobytes(), w, h, depth, channels, interlaced)
        st_g, got = s.unfilter(rows.tobytes(), w, h, depth, channels, interlaced)
        assert st_g == st_o == 0
        assert got == want.tobytes(), (w, h)


def _passes(w, h, volume, interlaced):
    if not interlaced:
        return [((w * volume + 7) >> 3, h)]
    out = []
    for bx, by, ex, ey in [(0, 0, 3, 3), (4, 0, 3, 3), (0, 4, 2, 3), (2, 0, 2, 2), (0, 2, 1, 2), (1, 0, 1, 1), (0, 1, 0, 1)]:
        sw, sh = (w + (1 << ex) - bx - 1) >> ex, (h + (1 << ey) - by - 1) >> ey
        if sw > 0 and sh > 0:
            out.append(((sw * volume + 7) >> 3, sh))
    return out


@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4])
def test_unfilter_single_filter_wide(gpu, ft):
    """Rows wider than several tiles and taller than several 64-row bands, one filter type."""
    s = gpu.load()
    rng = np.random.default_rng(ft)
    w, h = 1000, 200
    rows = rng.integers(0, 256, (h, 1 + 4 * w), dtype=np.uint8)
    rows[:, 0] = ft
    st, want = ph.orc_unfilter(rows.tobytes(), w, h, 8, 4, False)
    st2, got = s.unfilter(rows.tobytes(), w, h, 8, 4, False)
    assert st == st2 == 0 and got == want.tobytes()


def test_unfilter_short_and_extraneous(gpu):
    s = gpu.load()
    rng = np.random.default_rng(9)
    w, h = 40, 100
    rows = rng.integers(0, 256, (h, 1 + 4 * w), dtype=np.uint8)
    rows[:, 0] = rng.integers(0, 5, h)
    full = rows.tobytes()
    sentinel = bytes([0xAB]) * (w * h * 4)
    for n in (0, 1, 161, 160, 161 * 70 + 5, len(full) - 1):
        st_o, want = ph.orc_unfilter(full[:n], w, h, 8, 4, False)
        st_g, got = s.unfilter(full[:n], w, h, 8, 4, False, storage=sentinel)
        assert st_g == st_o == 0
        done = (n // 161) * 160
        assert got[:done] == want.tobytes()[:done]
        assert got[done:] == sentinel[done:]                 # undecoded rows are left untouched
    st_g, got = s.unfilter(full + b"\0", w, h, 8, 4, False)
    assert st_g == 48 and got == ph.orc_unfilter(full, w, h, 8, 4, False)[1].tobytes()


@pytest.mark.parametrize("name", sorted(TABLE))
def test_pngsuite_golden(gpu, name):
    s = gpu.load()
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    st, storage, _ = s.decode(png.idat, png.width, png.height, png.depth, png.channels, png.interlaced, png.fmt)
    assert st == 0
    assert hashlib.sha256(storage).hexdigest() == TABLE[name]["storage_sha256"]
    rgba = ph.unpack_rgba16(np.frombuffer(storage, np.uint8), png).astype("<u2")
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == TABLE[name]["rgba16_sha256"]


def test_decode_batch_mixed(gpu):
    """One spng_decode_batch call over every PngSuite fixture at once (mixed formats, Adam7, CgBI)."""
    s = gpu.load()
    names = sorted(TABLE)
    pngs = [ph.parse_png((ph.GOLDEN / "pngsuite" / n).read_bytes()) for n in names]
    keep, descs = [], []
    for p in pngs:
        u = gpu.inflated_size(p.width, p.height, p.depth, p.channels, p.interlaced)
        idat, rows = s.to_device(p.idat), s.empty(u + 4096)
        storage = s.empty(gpu.storage_size(p.width, p.height, p.depth, p.channels))
        keep.append((idat, rows, storage))
        descs.append(s.image_desc(idat, rows, storage, p.width, p.height, p.depth, p.channels, p.interlaced, p.fmt))
    res = s.decode_batch(descs)
    for n, p, (_, _, storage), r in zip(names, pngs, keep, res):
        assert r.status == 0, n
        size = gpu.storage_size(p.width, p.height, p.depth, p.channels)
        assert hashlib.sha256(bytes(storage[:size].cpu().numpy())).hexdigest() == TABLE[n]["storage_sha256"], n
    # asynchronous form: results stay on the device
    s.decode_batch(descs, wait=False)
    s.sync()
    assert [r.status for r in s.fetch_results(len(descs))] == [0] * len(descs)


def test_decode_errors_and_partial(gpu):
    s = gpu.load()
    rng = np.random.default_rng(2)
    rows = rng.integers(0, 256, (8, 1 + 32), dtype=np.uint8)
    rows[:, 0] = rng.integers(0, 5, 8)
    cases = {
        "extra": zlib.compress(rows.tobytes() + b"\x00", 6),
        "short": zlib.compress(rows.tobytes()[:-40], 6),
        "trunc": zlib.compress(rows.tobytes(), 6)[:-6],
        "badsum": zlib.compress(rows.tobytes(), 6)[:-1] + b"\x00",
        "ok": zlib.compress(rows.tobytes(), 6),
    }
    for k, z in cases.items():
        png = ph.Png(8, 8, 8, 6, False, False, z)
        st_o, want, aux_o = ph.orc_decode(png)
        st_g, got, aux_g = s.decode(z, 8, 8, 8, 4, False, 0, storage=bytes(256))
        assert (st_g, aux_g) == (st_o, aux_o), k
        assert got == want.tobytes(), k


def test_config1_stored_blocks(gpu):
    """BASELINE config 1: 256x256 RGBA8, filter=Sub, level 0 (stored blocks)."""
    s = gpu.load()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (256, 1024), dtype=np.uint8)
    rows = np.zeros((256, 1025), dtype=np.uint8)
    rows[:, 0] = 1
    rows[:, 1:5] = img[:, :4]
    rows[:, 5:] = img[:, 4:] - img[:, :-4]
    z = zlib.compress(rows.tobytes(), 0)
    st, storage, _ = s.decode(z, 256, 256, 8, 4, False)
    assert st == 0 and storage == img.tobytes()


def test_mirror_inflator_and_context(gpu):
    """The reference-shaped API: LZ77.Inflator push/pull (Snippets/LZ77/StreamingZlib.swift) and
    PNG.Context.push per IDAT chunk (PNG.Image.swift:385-389)."""
    data = _payloads()["text"]
    z = zlib.compress(data, 9)
    inf = gpu.LZ77.Inflator()
    assert inf.push(z[:100]) == ()
    assert inf.pull(len(data)) is None
    assert inf.push(z[100:]) is None
    assert inf.pull(10) == data[:10]
    assert inf.pull() == data[10:]
    with pytest.raises(gpu.DecompressionError) as e:
        gpu.LZ77.Inflator().push(z[:-1] + bytes([z[-1] ^ 1]))
    assert e.value.status == 32
    # multi-IDAT image (oi9n2c16: 229 chunks) pushed chunk by chunk
    name = "common/oi4n2c16.png"
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    ctx = gpu.PNG.Context((png.width, png.height), png.depth, png.channels, png.interlaced, png.fmt)
    pos = 0
    for n in png.idat_chunks:
        ctx.push(png.idat[pos:pos + n]); pos += n
    ctx.push_ancillary_iend()
    assert hashlib.sha256(ctx.storage).hexdigest() == TABLE[name]["storage_sha256"]
    with pytest.raises(gpu.DecodingError):
        ctx.push(b"\0")                                        # extraneousImageDataCompressedData


@pytest.mark.parametrize("level,count", [(4, 5), (4, 5000), (7, 50), (7, 5000), (9, 5), (9, 500), (9, 5000)])
def test_mirror_deflator_roundtrip(gpu, level, count):
    """LZ77Tests/Compression.swift:7-27 through the mirrored seam: Deflator(level:exponent: 8, hint: 16), pull until
    nil, push everything into an Inflator, identity -- plus the stream equals the oracle's at that exponent."""
    rng = np.random.default_rng(level * 10007 + count)
    data = rng.integers(0, 256, count, dtype=np.uint8).tobytes()
    deflator = gpu.LZ77.Deflator(level=level, exponent=8, hint=16)
    half = count // 2
    deflator.push(data[:half])
    chunks = []
    while (c := deflator.pop()) is not None:                          # whole chunks as the first half yields them (every push compresses)
        chunks.append(c)
    deflator.push(data[half:], last=True)
    while True:
        c = deflator.pull()
        if c is None:
            break
        chunks.append(c)
    assert all(len(c) == 32 for c in chunks[:-1]) and 0 < len(chunks[-1]) <= 32
    stream = b"".join(chunks)
    assert stream == ph.orc_deflate(data, level, 0, 8)
    inflator = gpu.LZ77.Inflator()
    status = ()
    for c in chunks:
        status = inflator.push(c)
    assert status is None and inflator.pull() == data


def test_encode_departs_from_the_stored_tail_bug(gpu):
    """on an input that trips the reference's stored-tail bug (a flat 1033 x 4 RGBA8 raster at
    level 6, LZ77.DeflatorBuffers.Stream.swift:45-60; tests/test_oracle_encode.py shows the row-pushed restatement
    producing the corrupt stream) the device emits the one-shot stream: equal to the oracle's one-shot deflate of the
    oracle's filtered rows, inflating to them, and different from the row-pushed reference output."""
    import ctypes
    w, h = 1033, 4
    storage = np.tile(np.array([9, 200, 31, 255], np.uint8), w * h)
    rows = ph.orc_filter(storage, w, h, 8, 4, False)
    enc = gpu.PNG.ImageEncoder(storage.tobytes(), (w, h), 8, 4, interlaced=False, level=6, hint=1 << 15)
    payloads = []
    while True:
        p = enc.pull()
        if p is None:
            break
        payloads.append(p)
    idat = b"".join(payloads)
    assert idat == ph.orc_deflate(bytes(rows), 6)
    assert zlib.decompress(idat) == bytes(rows)
    lib = ph.oracle()
    cap = len(storage) * 2 + 4096
    dst = np.empty(cap, np.uint8)
    wr = ctypes.c_size_t(0)
    assert lib.orc_encode(ph._ptr(storage), w, h, 8, 4, 0, 0, 6, ph._ptr(dst), cap, ctypes.byref(wr)) == 0
    assert dst[:wr.value].tobytes() != idat


def test_mirror_image_encoder(gpu):
    """PNG.Image.compress's loop over PNG.Encoder.pull (PNG.Image.swift:658-665): the IDAT payloads concatenate
    to the oracle's PNG.Encoder output at the reference's default level 9, and decode back."""
    rng = np.random.default_rng(77)
    w, h = 37, 23
    storage = ((np.arange(w * h * 4) * 7 + rng.integers(0, 2, w * h * 4)) % 256).astype(np.uint8).tobytes()
    enc = gpu.PNG.ImageEncoder(storage, (w, h), 8, 4, interlaced=True, level=9, hint=64)
    payloads = []
    while True:
        p = enc.pull()
        if p is None:
            break
        payloads.append(p)
    assert len(payloads) > 1 and all(len(p) == 128 for p in payloads[:-1])
    idat = b"".join(payloads)
    assert idat == ph.orc_deflate(ph.orc_filter(np.frombuffer(storage, np.uint8), w, h, 8, 4, True), 9)
    ctx = gpu.PNG.Context((w, h), 8, 4, True)
    for p in payloads:
        ctx.push(p)
    ctx.push_ancillary_iend()
    assert ctx.storage == storage


@pytest.mark.parametrize("delay", [1, 2, 3, 4, 6, 8])
def test_mirror_filter_defilter_roundtrip(gpu, delay):
    """PNGTests/Filtering.swift:9-64: Encoder.filter -> Decoder.defilter identity, and both agree
    with the oracle row functions."""
    rng = np.random.default_rng(delay)
    pitch = 24 * delay
    last = bytes([0]) + bytes(pitch)
    orc = ph.oracle()
    for _ in range(6):
        line = bytes([0]) + rng.integers(0, 256, pitch, dtype=np.uint8).tobytes()
        filtered = gpu.PNG.Encoder.filter(line, last, delay)
        a, b = np.frombuffer(line, np.uint8).copy(), np.frombuffer(last, np.uint8).copy()
        out = np.zeros(pitch + 1, np.uint8)
        orc.orc_filter_row(ph._ptr(a), ph._ptr(b), pitch + 1, delay, ph._ptr(out))
        assert filtered == out.tobytes()
        restored = gpu.PNG.Decoder.defilter(filtered, last, delay)
        assert restored[1:] == line[1:]
        last = line


@pytest.mark.parametrize("interlaced", [False, True])
@pytest.mark.parametrize("depth,channels", FORMATS)
def test_filter_vs_oracle(gpu, depth, channels, interlaced):
    s = gpu.load()
    rng = np.random.default_rng(depth + channels)
    for (w, h) in [(1, 1), (9, 5), (64, 33), (131, 17)]:
        n = gpu.storage_size(w, h, depth, channels)
        hi = (1 << depth) if depth < 8 else 256
        if rng.random() < 0.5:
            storage = rng.integers(0, hi, n, dtype=np.uint8)
        else:                                                 # smooth content so that all filters get chosen
            storage = ((np.arange(n) // max(1, (depth * channels + 7) // 8) * 3) % hi).astype(np.uint8)
        want = ph.orc_filter(storage, w, h, depth, channels, interlaced)
        got = s.filter(storage.tobytes(), w, h, depth, channels, interlaced)
        assert got == want, (w, h)


def test_adler32(gpu):
    s = gpu.load()
    rng = np.random.default_rng(3)
    for n in (0, 1, 5552, 65536, 65537, 300000):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert s.adler32(d) == zlib.adler32(d)


def test_synthetic_4k_image_roundtrip(gpu):
    """One image of the benchmark workload, end to end: GPU filter == oracle filter; zlib level 6;
    GPU decode == the original raster (size-independent property for the full-size config)."""
    from swift_png_amd import synth
    s = gpu.load()
    img = synth.image(3, 4096, 4096)
    rows = s.filter(img.tobytes(), 4096, 4096, 8, 4, False)
    assert hashlib.sha256(rows).digest() == hashlib.sha256(ph.orc_filter(img.reshape(-1), 4096, 4096, 8, 4, False)).digest()
    hist = np.bincount(np.frombuffer(rows, np.uint8).reshape(4096, 16385)[:, 0], minlength=5)
    assert (hist > 0).sum() >= 3                              # "mixed filters"
    z = zlib.compress(rows, 6)
    st, storage, _ = s.decode(z, 4096, 4096, 8, 4, False)
    assert st == 0 and storage == img.tobytes()


# ------------------------------------------------------------------------------------------ unpack
def _palette_quads(png):
    if png.palette is None:
        return None
    pal = np.frombuffer(png.palette, dtype=np.uint8).reshape(-1, 3)
    quads = np.full((len(pal), 4), 255, dtype=np.uint8)
    quads[:, :3] = pal
    if png.trns:
        t = np.frombuffer(png.trns, dtype=np.uint8)[:len(pal)]
        quads[:len(t), 3] = t
    return quads.tobytes()


@pytest.mark.parametrize("name", sorted(TABLE))
def test_unpack_rgba_vs_reference_goldens(gpu, name):
    """PNG.Image.unpack(as: PNG.RGBA<UInt16>.self) on the device equals the reference's own
    .rgba golden of every PngSuite fixture (digests in pngsuite.json; CgBI goldens premultiplied as
    Roundtripping.swift:206-215 does); the UInt8 target is the same pixels at 8 bits."""
    import struct
    s = gpu.load()
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    st, storage, _ = s.decode(png.idat, png.width, png.height, png.depth, png.channels, png.interlaced, png.fmt)
    assert st == 0
    key = None
    if png.trns and png.color in (0, 2):
        key = struct.unpack(">" + "H" * (1 if png.color == 0 else 3), png.trns[:2 if png.color == 0 else 6])
    kw = dict(indexed=png.color == 3, bgr=png.ios and png.color in (2, 6), palette=_palette_quads(png) if png.color == 3 else None,
              key=key)
    got16 = s.unpack(storage, png.width, png.height, png.depth, png.channels, target=16, **kw)
    assert hashlib.sha256(got16).hexdigest() == TABLE[name]["rgba16_sha256"], name
    got8 = s.unpack(storage, png.width, png.height, png.depth, png.channels, target=8, **kw)
    assert got8 == (np.frombuffer(got16, dtype="<u2") >> 8).astype(np.uint8).tobytes()


IOS_NAMES = sorted(n.split("/", 1)[1] for n in TABLE if n.startswith("ios/"))


@pytest.mark.parametrize("name", IOS_NAMES)
def test_unpack_premultiplied_vs_the_ios_goldens(gpu, name):
    """f3: RGBA<UInt16>.premultiplied(as: UInt8.self) (PNG.RGBA.swift:146-158) on the device, pinned on the reference's
    goldens: the iOS set holds CgBI versions of PngSuite images, whose pixels are premultiplied, and the reference compares
    them with its straight golden premultiplied in 8 bits (Roundtripping.swift:206-215) -- so the COMMON version of the
    same image, unpacked and premultiplied by the device, must have the digest pngsuite.json keeps for the iOS one."""
    import struct
    s = gpu.load()
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / "common" / name).read_bytes())
    st, storage, _ = s.decode(png.idat, png.width, png.height, png.depth, png.channels, png.interlaced, png.fmt)
    assert st == 0
    key = None
    if png.trns and png.color in (0, 2):
        key = struct.unpack(">" + "H" * (1 if png.color == 0 else 3), png.trns[:2 if png.color == 0 else 6])
    kw = dict(indexed=png.color == 3, palette=_palette_quads(png) if png.color == 3 else None, key=key)
    got = s.unpack(storage, png.width, png.height, png.depth, png.channels, target=16, premultiply=spng.PREMULTIPLY_AS_U8, **kw)
    assert hashlib.sha256(got).hexdigest() == TABLE["ios/" + name]["rgba16_sha256"], name


@pytest.mark.parametrize("name", sorted(TABLE)[::7])
def test_unpack_va_and_premultiplied_targets(gpu, name):
    """f3: PNG.VA<T>.unpack (PNG.VA.swift:184-290: the grey value or the red channel, and alpha) and the .premultiplied
    forms (PNG.swift:55-66: (c * a + (T.max >> 1)) / T.max) of both targets, against the RGBA<T> pixels the goldens pin."""
    import struct
    s = gpu.load()
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    st, storage, _ = s.decode(png.idat, png.width, png.height, png.depth, png.channels, png.interlaced, png.fmt)
    assert st == 0
    key = None
    if png.trns and png.color in (0, 2):
        key = struct.unpack(">" + "H" * (1 if png.color == 0 else 3), png.trns[:2 if png.color == 0 else 6])
    kw = dict(indexed=png.color == 3, bgr=png.ios and png.color in (2, 6), palette=_palette_quads(png) if png.color == 3 else None,
              key=key)
    args = (storage, png.width, png.height, png.depth, png.channels)
    for target, dt, tmax in ((16, "<u2", 65535), (8, "u1", 255)):
        rgba = np.frombuffer(s.unpack(*args, target=target, **kw), dtype=dt).reshape(-1, 4).astype(np.uint64)
        va = np.frombuffer(s.unpack(*args, target=target, layout=spng.TARGET_VA, **kw), dtype=dt).reshape(-1, 2)
        assert (va[:, 0] == rgba[:, 0]).all() and (va[:, 1] == rgba[:, 3]).all()
        pre = np.frombuffer(s.unpack(*args, target=target, premultiply=spng.PREMULTIPLY, **kw), dtype=dt).reshape(-1, 4)
        want = rgba.copy()
        want[:, :3] = (rgba[:, :3] * rgba[:, 3:4] + (tmax >> 1)) // tmax
        assert (pre == want).all()
        vap = np.frombuffer(s.unpack(*args, target=target, layout=spng.TARGET_VA, premultiply=spng.PREMULTIPLY, **kw), dtype=dt).reshape(-1, 2)
        assert (vap[:, 0] == want[:, 0]).all() and (vap[:, 1] == want[:, 3]).all()


# ------------------------------------------------------------------------------------------ deflate
def _deflate_payloads():
    rng = np.random.default_rng(21)
    p = dict(_payloads())
    p["two"] = b"ab"; p["three"] = b"abc"; p["four"] = b"abcd"; p["five"] = b"abcde"
    p["runs"] = b"".join(bytes([i % 7]) * int(rng.integers(1, 600)) for i in range(300))
    p["periodic"] = (bytes(range(37)) * 3000)[:100001]
    p["end-in-match"] = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes() + b"\x05" * 300
    return p


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_deflate_vs_oracle(gpu, level):
    """Identical DEFLATE bitstream at the same level (BASELINE north star): every row of
    LZ77.DeflatorSearch.swift:17-35 -- greedy, lazy and the shortest-path search."""
    s = gpu.load()
    for kind, data in sorted(_deflate_payloads().items()):
        for fmt in (0, 1):
            want = ph.orc_deflate(data, level, fmt)
            got = s.deflate(data, level, fmt)
            assert got == want, (kind, fmt, len(got), len(want))
        assert zlib.decompress(got if fmt == 0 else want) == data if fmt == 0 else True


def test_deflate_one_kernel_mode(gpu):
    """SPNG_CFG_DEFLATE_MODE = SPNG_DEFLATE_ONE_KERNEL: a wave per stream does everything (`deflate_kernel` at levels 0-7,
    `deflate_full_kernel` from 8 on) -- the form the rounds of search + parse kernels fall back to when their records find no
    memory: the same bytes"""
    s = gpu.load()
    payloads = _deflate_payloads()
    try:
        s.configure(gpu.CFG_DEFLATE_MODE, gpu.DEFLATE_ONE_KERNEL)
        for level in (1, 6, 9):
            for kind in ("text", "noise", "sparse", "ramp"):
                assert s.deflate(payloads[kind], level) == ph.orc_deflate(payloads[kind], level), (kind, level)
    finally:
        s.configure(gpu.CFG_DEFLATE_MODE, gpu.DEFLATE_AUTO)


def test_deflate_inserter_forms(gpu):
    """The match search's inserter exchanges bucket heads with one ds_mskor_rtn_b32 per lane where the device's LDS serves the
    lanes of an address in ascending order (probed when the context is created: spng_lds_exchange_ordered), and reads its store
    back where it does not (SPNG_D3_READBACK forces that form): both give the oracle's bytes -- hash chains with many positions
    of one bucket in a batch (runs, a short period) and none (noise)"""
    import ctypes
    import hashlib
    import os
    import subprocess
    import sys
    s = gpu.load()
    v = ctypes.c_int32(-1)
    assert s.lib.spng_lds_exchange_ordered(s.ctx, ctypes.byref(v)) == 0 and v.value in (0, 1)
    assert s.lib.spng_lds_exchange_ordered(s.ctx, None) == gpu.E_ARGUMENT
    code = ("import sys, hashlib, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r); import swift_png_amd as spng; import test_gpu_decode as t\n"
            "s = spng.load(); v = ctypes.c_int32(-1); s.lib.spng_lds_exchange_ordered(s.ctx, ctypes.byref(v)); print('ordered', v.value)\n"
            "for kind, data in sorted(t._deflate_payloads().items()):\n"
            "    for level in (4, 6, 9): print(kind, level, hashlib.sha256(s.deflate(data, level)).hexdigest())\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for form in ("probed", "readback"):
        env = dict(os.environ)
        env.pop("SPNG_D3_READBACK", None)
        if form == "readback":
            env["SPNG_D3_READBACK"] = "1"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-400:]
        out[form] = r.stdout.strip().splitlines()
    assert out["readback"][0] == "ordered 0"
    assert out["probed"][0] == "ordered %d" % v.value
    assert out["probed"][1:] == out["readback"][1:]
    for line in out["probed"][1:]:
        kind, level, digest = line.split()
        assert digest == hashlib.sha256(ph.orc_deflate(_deflate_payloads()[kind], int(level))).hexdigest(), (kind, level)


@pytest.mark.parametrize("level", [1, 6])
def test_deflate_greedy_lazy_over_several_rounds(gpu, level):
    """levels 0-7 go through in rounds of 2^21 positions (search of round r + 1 beside the parse of round r, parse position, queued
    terms and bit writer kept in the D1State, a lazy look at the position behind a round's last): 5 MiB of mixed content -- three
    rounds, matches across the round boundaries, windows that wrap the search's LDS ring many times -- equal the oracle's stream"""
    s = gpu.load()
    rng = np.random.default_rng(31 + level)
    a = rng.integers(-3, 4, 3 << 20).astype(np.int16)
    a[rng.random(len(a)) < 0.6] = 0
    part = a.astype(np.uint8).tobytes()
    blk = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    data = part[:(2 << 20) - 1000] + blk + blk + bytes(70000) + part[(2 << 20):] + (b"abcdefgh" * 40000) + blk[:20000] + rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    data = data + part[:(5 << 20) - len(data)] if len(data) < (5 << 20) else data[:5 << 20]
    assert len(data) > 2 * (1 << 21)
    assert s.deflate(data, level) == ph.orc_deflate(data, level)


@pytest.mark.parametrize("level", [1, 6])
def test_deflate_batch_of_streams_of_different_lengths(gpu, level):
    """in one call of spng_deflate_batch at levels 0-7 the block-placement kernels index every stream's block
    start bits by the LAUNCH's largest block count; a short stream beside a long one had its staged bits overwritten and still
    reported SPNG_DONE.  10 KB, 200 KB and 3 MB (and the short ones once more behind the long one) in one call: every stream is
    the oracle's."""
    import torch
    s = gpu.load()
    rng = np.random.default_rng(77 + level)

    def content(n, seed):
        r = np.random.default_rng(seed)
        a = r.integers(-4, 5, n).astype(np.int16)
        a[r.random(n) < 0.5] = 0
        return np.cumsum(a).astype(np.uint8).tobytes()

    sizes = [10_000, 200_000, 3 << 20, 10_007, 65_000, 1, 0, 2046 * 3]
    datas = [content(n, 1000 + i) if n else b"" for i, n in enumerate(sizes)]
    tens = [torch.frombuffer(bytearray(d or b"\0"), dtype=torch.uint8).cuda()[:len(d)] for d in datas]
    outs, res = s.deflate_batch(tens, level)
    for i, (d, o, r) in enumerate(zip(datas, outs, res)):
        assert r.status == gpu.DONE, (i, r.status)
        got = bytes(o[:r.written].cpu().numpy())
        assert got == ph.orc_deflate(d, level), (i, sizes[i], r.written)


def test_deflate_block_boundaries(gpu):
    """Blocks close after 2047 terms (greedy) / 2046-2047 (lazy); literal-only inputs around the
    boundary exercise the term-buffer guards (DeflatorBuffers.Stream.swift:219,277)."""
    s = gpu.load()
    rng = np.random.default_rng(8)
    for n in (2046, 2047, 2048, 2049, 2050, 4094, 4095, 4096, 6141):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for level in (0, 6):
            assert s.deflate(data, level) == ph.orc_deflate(data, level), (n, level)


def test_deflate_level9_reference_goldens(gpu):
    """BASELINE configs[3] parity pin: the device's filter-select + level-9 DEFLATE reproduces, bit for bit,
    the IDAT streams swift-png itself committed under Tests/Outputs -- all 28 of them (inputs copied to
    tests/golden/encode by make_golden.py, digests in encode.json), the 16-bit photographic ones with their
    deep limit-doubling blocks included (Sources/PNGCompressionTests/Compression.swift:7-84)."""
    s = gpu.load()
    table = json.loads((ph.GOLDEN / "encode.json").read_text())
    seen = 0
    for base in sorted((ph.GOLDEN / "encode").glob("*.baseline.png")):
        name = base.name[:-len(".baseline.png")]
        src = ph.parse_png(base.read_bytes())
        st, storage, _ = s.decode(src.idat, src.width, src.height, src.depth, src.channels, src.interlaced)
        assert st == 0
        rows = s.filter(storage, src.width, src.height, src.depth, src.channels, False)
        got = s.deflate(rows, 9)
        assert len(got) == table[name]["idat_len"] and hashlib.sha256(got).hexdigest() == table[name]["idat_sha256"], name
        out = ph.GOLDEN / "encode" / f"{name}.swiftpng9.png"
        if out.exists():
            assert got == ph.parse_png(out.read_bytes()).idat, name
        seen += 1
    assert seen == 28 == len(table)


def test_deflate_full_search_block_growth(gpu):
    """Levels >= 8 close a block after limit - 1 vertices, the limit doubling per block (2047, 4095, 8191 ...
    vertices, LZ77.DeflatorMatches.swift:229); runs > 100 make the following vertices edgeless
    (DeflatorBuffers.Stream.swift:376-380); the first block runs 2 x iterations passes."""
    s = gpu.load()
    rng = np.random.default_rng(31)
    noisy = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    runs = b"".join(bytes([int(rng.integers(0, 4))]) * int(rng.integers(90, 700)) for _ in range(120))
    texty = (b"lorem ipsum dolor sit amet, consectetur adipiscing elit " * 900)[:40001]
    mixed = noisy[:7000] + runs[:20000] + texty[:9000] + noisy[7000:9000]
    for kind, data in (("noisy", noisy), ("runs", runs), ("text", texty), ("mixed", mixed)):
        for level in (8, 9, 12):
            got = s.deflate(data, level)
            assert got == ph.orc_deflate(data, level), (kind, level)
            assert zlib.decompress(got) == data
    for n in (2046, 2047, 2048, 2049, 2050, 6141, 6142, 6143):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert s.deflate(data, 9) == ph.orc_deflate(data, 9), n


def test_deflate_full_search_sparse_matches(gpu):
    """Levels >= 8 on (nearly) incompressible input, BASELINE configs[3]'s kind: batches of 64 vertices without a
    single edge take the device's scan-only forward pass and literal-only back-trace; a sprinkling of repeated
    snippets (some with runs > 100: the skip rule) mixes those with the batches that have edges.  Blocks up to
    131071 vertices."""
    s = gpu.load()
    rng = np.random.default_rng(77)
    noise = rng.integers(0, 256, 300000, dtype=np.uint8)
    sprinkled = noise.copy()
    for _ in range(400):
        n = int(rng.choice([4, 5, 8, 16, 40, 130, 300]))
        src = int(rng.integers(0, 300000 - 400)); dst = int(rng.integers(src + 1, min(src + 40000, 300000 - n)))
        sprinkled[dst:dst + n] = sprinkled[src:src + n]
    few = rng.integers(0, 4, 120000, dtype=np.uint8)            # 2-bit alphabet: short matches everywhere, Huffman-heavy
    for kind, data in (("noise", noise.tobytes()), ("sprinkled", sprinkled.tobytes()), ("few", few.tobytes())):
        for level in (8, 9, 13):
            got = s.deflate(data, level)
            assert got == ph.orc_deflate(data, level), (kind, level)
        assert zlib.decompress(got) == data


def _mixed_content(n, seed, kinds=(0, 1, 2, 3)):
    """noise, runs longer than 100 (the skip rule of DeflatorBuffers.Stream.swift:376-380), text, a 6-letter alphabet"""
    rng = np.random.default_rng(seed)
    parts, total = [], 0
    text = b"lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor " * 64
    while total < n:
        k = int(rng.choice(kinds))
        if k == 0:
            p = rng.integers(0, 256, int(rng.integers(20000, 200000)), dtype=np.uint8).tobytes()
        elif k == 1:
            p = b"".join(bytes([int(rng.integers(0, 8))]) * int(rng.integers(101, 900)) for _ in range(200))
        elif k == 2:
            o = int(rng.integers(0, 4000))
            p = text[o:o + int(rng.integers(3000, 60000))]
        else:
            p = rng.integers(0, 6, int(rng.integers(10000, 80000)), dtype=np.uint8).tobytes()
        parts.append(p)
        total += len(p)
    return b"".join(parts)[:n]


def test_deflate_full_search_at_the_vertex_cap(gpu):
    """BASELINE configs[3] at its real stream length.  The block limit of levels >= 8 doubles per block -- 2047, 4095, ...
    vertices -- up to 2^21 - 1 (LZ77.DeflatorMatches.swift:229; capacity DeflatorBuffers.swift:33): the cap is reached 2.1 MB
    into a stream and first APPLIES (a block that would have doubled again) past 4.2 MB.  Whole streams against the oracle:
    9 MiB of mixed content at level 9 (three capped blocks), 6.5 MiB (two) at level 13, whose search never gives up."""
    s = gpu.load()
    nine = _mixed_content(9 << 20, 5)
    got = s.deflate(nine, 9)
    want = ph.orc_deflate(nine, 9)
    assert len(got) == len(want) and got == want
    assert zlib.decompress(got) == nine
    thirteen = _mixed_content(13 << 19, 6, kinds=(0, 0, 0, 2, 3))
    got = s.deflate(thirteen, 13)
    assert got == ph.orc_deflate(thirteen, 13)
    assert zlib.decompress(got) == thirteen


def test_encode_random_4k_raster_whole_stream(gpu):
    """One full-size unit of BASELINE configs[3]: a 4096 x 4096 RGBA8 raster of uniform random bytes, filter-select +
    level 9, the WHOLE 64 MiB stream (32 blocks at the vertex cap) equal to the oracle's."""
    s = gpu.load()
    rng = np.random.default_rng(2024)
    raster = rng.integers(0, 256, 4096 * 4096 * 4, dtype=np.uint8)
    rows = s.filter(raster.tobytes(), 4096, 4096, 8, 4, False)
    assert rows == ph.orc_filter(raster, 4096, 4096, 8, 4, False)
    got = s.deflate(rows, 9)
    want = ph.orc_deflate(rows, 9)
    assert len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()


@pytest.mark.parametrize("exponent", [8, 11, 15])
def test_deflate_window_exponent(gpu, exponent):
    """LZ77.Deflator(format:level:exponent:hint:) with a small window (LZ77Tests/Compression.swift:12 uses 8):
    zlib header, candidate reach and stream equal the oracle's at every search kind."""
    s = gpu.load()
    rng = np.random.default_rng(exponent)
    data = (rng.integers(0, 6, 3000, dtype=np.uint8).tobytes() + b"abcdefgh" * 200) * 3
    for level in (4, 7, 9, 10, 13):
        got = s.deflate(data, level, 0, exponent)
        assert got == ph.orc_deflate(data, level, 0, exponent), (level, exponent)
        assert got[0] == ((exponent - 8) << 4 | 8) and zlib.decompress(got) == data
    for count in (5, 50, 500, 5000):                            # the reference's own property test, levels 4 / 7 / 9
        blob = rng.integers(0, 256, count, dtype=np.uint8).tobytes()
        for level in (4, 7, 9):
            got = s.deflate(blob, level, 0, exponent)
            assert got == ph.orc_deflate(blob, level, 0, exponent)
            st, out, _, _ = s.inflate(got)
            assert st == 0 and out == blob


def test_deflate_4k_rows_level6(gpu):
    """The benchmark's own stream: filtered scanlines of a synthetic 4096x4096 RGBA8 image at
    level 6; GPU stream == oracle stream, and the GPU inflates it back."""
    from swift_png_amd import synth
    s = gpu.load()
    img = synth.image(5, 4096, 4096)
    rows = s.filter(img.tobytes(), 4096, 4096, 8, 4, False)
    got = s.deflate(rows, 6)
    want = ph.orc_deflate(rows, 6)
    assert hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
    st, storage, _ = s.decode(got, 4096, 4096, 8, 4, False)
    assert st == 0 and storage == img.tobytes()


def test_encode_batch_end_to_end(gpu):
    """spng_encode_batch = PNG.Encoder.pull end to end (filter-select + deflate) for a mixed batch;
    streams equal the oracle's and decode back to the rasters."""
    import ctypes
    s = gpu.load()
    rng = np.random.default_rng(17)
    cases = [(64, 33, 8, 4, False), (31, 17, 8, 3, True), (50, 20, 16, 4, True), (200, 9, 4, 1, False), (9, 9, 1, 1, True)]
    keep, descs, raws = [], [], []
    for (w, h, depth, ch, il) in cases:
        n = gpu.storage_size(w, h, depth, ch)
        hi = (1 << depth) if depth < 8 else 256
        raw = ((np.arange(n) * 5 + rng.integers(0, 3, n)) % hi).astype(np.uint8)
        u = gpu.inflated_size(w, h, depth, ch, il)
        cap = s.lib.spng_deflate_bound(u)
        st_t, rows_t, out_t = s.to_device(raw), s.empty(u), s.empty(cap)
        keep.append((st_t, rows_t, out_t)); raws.append(raw)
        d = s.image_desc(out_t, rows_t, st_t, w, h, depth, ch, il, 0, rows_cap=u)
        d.idat_len = cap
        descs.append(d)
    n = len(descs)
    arr = (gpu.ImageDesc * n)(*descs)
    res = (gpu.Result * n)()
    assert s.lib.spng_encode_batch(s.ctx, arr, 6, n, None, res) == 0
    lib = ph.oracle()
    for (w, h, depth, ch, il), (st_t, rows_t, out_t), raw, r in zip(cases, keep, raws, res):
        assert r.status == 0
        got = bytes(out_t[:r.written].cpu().numpy())
        u = gpu.inflated_size(w, h, depth, ch, il)
        want = ph.orc_deflate(ph.orc_filter(raw, w, h, depth, ch, il), 6)
        assert got == want, (w, h, depth, ch, il)
        st, storage, _ = s.decode(got, w, h, depth, ch, il)
        assert st == 0 and storage == raw.tobytes()


def test_config5_shape_rgba16_adam7_multi_idat(gpu):
    """BASELINE config 5 at reduced size (1024x1024, same code paths: 16-bit RGBA, Adam7, stream split
    into 65,536-byte IDATs and pushed chunk by chunk); the full 8192x8192 case is a manual run
    (one serial 512 MiB stream)."""
    from swift_png_amd import synth
    s = gpu.load()
    w = h = 1024
    img = synth.image(9, w, h, 4, 16)
    rows = s.filter(img.tobytes(), w, h, 16, 4, True)
    assert rows == ph.orc_filter(img.reshape(-1), w, h, 16, 4, True)
    z = s.deflate(rows, 6)
    st, storage, _ = s.decode(z, w, h, 16, 4, True)
    assert st == 0 and storage == img.tobytes()
    png = ph.Png(w, h, 16, 6, True, False, z)
    assert (ph.orc_decode(png)[1] == img.reshape(-1)).all()


def test_config5_full_size_8192_rgba16_adam7(gpu):
    """BASELINE configs[4] at its real size: one 8192x8192 16-bit RGBA Adam7 image (536,886,272 inflated bytes,
    seven sub-images), the stream cut into 65,536-byte IDATs and concatenated again as the host does
    (PNG.Image.swift:385-389); device raster == source raster == the oracle's, and the inflate ran on the
    parallel pipeline (a single serial 512 MiB stream would take the serial kernel ~13 s)."""
    import time
    from swift_png_amd import synth
    s = gpu.load()
    w = h = 8192
    img = synth.image(11, w, h, 4, 16)
    raw = img.tobytes()
    rows = s.filter(raw, w, h, 16, 4, True)
    assert len(rows) == 536886272
    co = zlib.compressobj(6)
    z = co.compress(rows) + co.flush()
    chunks = [z[i:i + 65536] for i in range(0, len(z), 65536)]
    assert len(chunks) > 1000
    idat = b"".join(chunks)
    d_idat, d_rows, d_out = s.to_device(idat), s.empty(len(rows) + 4096), s.empty(len(raw))
    desc = s.image_desc(d_idat, d_rows, d_out, w, h, 16, 4, True, 0, rows_cap=len(rows) + 4096)
    res = s.decode_batch([desc])
    t0 = time.perf_counter()
    res = s.decode_batch([desc])
    dt = time.perf_counter() - t0
    assert res[0].status == 0 and res[0].written == len(rows) and res[0].reserved == 1
    got = bytes(d_out.cpu().numpy())
    assert hashlib.sha256(got).digest() == hashlib.sha256(raw).digest()
    png = ph.Png(w, h, 16, 6, True, False, idat)
    st, storage, _ = ph.orc_decode(png)
    assert st == 0 and hashlib.sha256(storage.tobytes()).digest() == hashlib.sha256(raw).digest()
    print(f"config 5: device decode {dt * 1e3:.0f} ms")
    assert dt < 2.0, dt                                       # ("config 5's single stream <= 2 s")


def test_inflate_fuzz_mutated_streams(gpu):
    """Differential fuzz: valid streams with a few bytes flipped, truncated, or spliced.  The device
    path and the oracle must agree on status, output bytes and error payload for every one of them
    (malformed input is where the reference's accept/reject rules differ from zlib's)."""
    s = gpu.load()
    rng = np.random.default_rng(1234)
    seeds = []
    for level in (1, 6, 9):
        n = int(rng.integers(200, 6000))
        data = (rng.integers(0, 256, n, dtype=np.uint8) * (rng.random(n) < 0.5)).astype(np.uint8).tobytes()
        seeds.append(zlib.compress(data, level))
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        seeds.append(co.compress(data) + co.flush())
    seeds.append(zlib.compress(rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), 0))
    cases = []
    for z in seeds:
        for _ in range(40):
            m = bytearray(z)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(0, len(m)))
                m[i] ^= 1 << int(rng.integers(0, 8))
            if rng.random() < 0.3:
                m = m[:int(rng.integers(0, len(m)))]
            cases.append(bytes(m))
    n = len(cases)
    cap = 1 << 16
    outs, res = s.inflate_batch([s.to_device(c) for c in cases], [cap] * n)
    statuses = set()
    for c, o, r in zip(cases, outs, res):
        st, out, consumed, aux = ph.orc_inflate(c, 0, cap)
        statuses.add(st)
        assert r.status == st, (r.status, st, c[:16].hex())
        assert r.written == len(out) and bytes(o[:r.written].cpu().numpy()) == out
        assert (r.aux[0], r.aux[1]) == aux
        if st == 0:
            assert r.consumed == consumed
    assert len(statuses) >= 5                                 # the fuzz really reaches the error paths


def test_unfilter_fuzz_shapes(gpu):
    """Random geometry sweep (every format, both interlacing modes, widths around the 64-unit tile
    and 64-row band boundaries), one batched call, against the oracle."""
    s = gpu.load()
    rng = np.random.default_rng(99)
    descs, keep, wants = [], [], []
    for _ in range(60):
        depth, ch = FORMATS[int(rng.integers(0, len(FORMATS)))]
        il = bool(rng.integers(0, 2))
        w = int(rng.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 200, int(rng.integers(1, 300))]))
        h = int(rng.choice([1, 2, 63, 64, 65, 128, 129, int(rng.integers(1, 200))]))
        u = gpu.inflated_size(w, h, depth, ch, il)
        rows = rng.integers(0, 256, u, dtype=np.uint8)
        off = 0
        for pitch, ph_rows in _passes(w, h, depth * ch, il):
            for _y in range(ph_rows):
                rows[off] = rng.integers(0, 5)
                off += pitch + 1
        st, want = ph.orc_unfilter(rows.tobytes(), w, h, depth, ch, il)
        assert st == 0
        rt, stt = s.to_device(rows), s.empty(gpu.storage_size(w, h, depth, ch))
        keep.append((rt, stt)); wants.append(want)
        descs.append(s.image_desc(None, rt, stt, w, h, depth, ch, il, rows_cap=u))
    res = s.unfilter_batch(descs)
    for (rt, stt), want, r in zip(keep, wants, res):
        assert r.status == 0
        assert bytes(stt[:len(want)].cpu().numpy()) == want.tobytes()
