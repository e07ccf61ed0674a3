"""GPU parity of spng_pack_batch / spng_pack_as (f3, the encode half: PNG.Image.init(packing:size:layout:)) against
oracle/pixels.py, which tests/test_oracle_pack.py pins on the reference's golden pixels; and of the scalar unpack target."""
import json
import sys

import numpy as np
import pytest

import pnghelp as ph

sys.path.insert(0, str(ph.ROOT / "oracle"))
import pixels as orc_pixels  # noqa: E402
from test_oracle_pack import golden_case, repeats_a_colour  # noqa: E402

pytestmark = pytest.mark.gpu
TABLE = json.loads((ph.GOLDEN / "pngsuite.json").read_text())


@pytest.mark.parametrize("name", sorted(TABLE))
def test_pack_golden_pixels_vs_oracle(gpu, name):
    """every PngSuite image: its golden RGBA<UInt16> pixels (and their UInt8 / VA / scalar views) packed as the image's own
    format on the device = the oracle's storage = (T = UInt16, RGBA) the storage the decode produced."""
    s = gpu.load()
    png, storage, rgba, kw = golden_case(name)
    args = (png.width, png.height, png.depth, png.channels)
    dkw = dict(indexed=kw["indexed"], bgr=kw["bgr"], palette=kw["palette"])
    if kw["indexed"] and repeats_a_colour(kw["palette"]):
        # the reference traps here; the device's documented answer: the lowest index of a repeated colour
        got = np.frombuffer(s.pack(rgba.astype("<u2").tobytes(), *args, source=16, **dkw), dtype=np.uint8)
        pal = np.frombuffer(kw["palette"], dtype=np.uint8).reshape(-1, 4)
        first = {}
        for i, e in enumerate(pal):
            first.setdefault(tuple(e), i)
        want = np.array([first[tuple(pal[i])] for i in storage], dtype=np.uint8)
        assert (got == want).all()
        return
    for dt, src in (("<u2", rgba), ("u1", (rgba >> 8).astype(np.uint8))):
        bits = 16 if dt == "<u2" else 8
        got = s.pack(src.astype(dt).tobytes(), *args, source=bits, **dkw)
        assert got == orc_pixels.pack(src, **kw), (name, bits)
        if bits == 16:
            assert got == storage.tobytes()
        va = np.ascontiguousarray(src[:, [0, 3]])
        assert s.pack(va.astype(dt).tobytes(), *args, source=bits, layout=gpu.TARGET_VA, **dkw) == \
            orc_pixels.pack(va, layout=orc_pixels.VA, **kw)
        v = np.ascontiguousarray(src[:, 0])
        assert s.pack(v.astype(dt).tobytes(), *args, source=bits, layout=gpu.TARGET_SCALAR, **dkw) == \
            orc_pixels.pack(v, layout=orc_pixels.SCALAR, **kw)


FORMATS = [(1, 1), (2, 1), (4, 1), (8, 1), (16, 1), (8, 2), (16, 2), (8, 3), (16, 3), (8, 4), (16, 4)]


@pytest.mark.parametrize("depth,channels", FORMATS)
@pytest.mark.parametrize("bits", [8, 16])
def test_pack_random_pixels_every_format(gpu, depth, channels, bits):
    """random pixels of every colour target into every format (bgr too), sizes that leave 0-3 pixels behind the last quad"""
    s = gpu.load()
    rng = np.random.default_rng(depth * 100 + channels * 10 + bits)
    dt = np.uint8 if bits == 8 else np.uint16
    for (w, h) in ((1, 1), (3, 1), (5, 3), (64, 9), (257, 31), (1023, 17)):
        n = w * h
        for layout, k in ((orc_pixels.RGBA, 4), (orc_pixels.VA, 2), (orc_pixels.SCALAR, 1)):
            px = rng.integers(0, 1 << bits, (n, k), dtype=np.uint32).astype(dt)
            for bgr in ((False, True) if (channels >= 3 and depth == 8) else (False,)):
                want = orc_pixels.pack(px if k > 1 else px[:, 0], depth, channels, bgr=bgr, layout=layout)
                got = s.pack(px.astype("<u%d" % (bits // 8)).tobytes(), w, h, depth, channels, bgr=bgr, source=bits, layout=layout)
                assert got == want, (w, h, layout, bgr)


@pytest.mark.parametrize("depth", [1, 2, 4, 8])
@pytest.mark.parametrize("bits", [8, 16])
def test_pack_indexed_with_strangers(gpu, depth, bits):
    """indexed formats: the default indexer finds the entry equal to the pixel (reduced to 8 bits) and gives entry 0 to colours
    the palette does not hold (PNG.Color.swift:182-190)"""
    s = gpu.load()
    rng = np.random.default_rng(depth + bits)
    count = min(1 << depth, 256)
    pal = np.unique(rng.integers(0, 256, (600, 4), dtype=np.uint8), axis=0)
    pal = pal[rng.permutation(len(pal))[:count]]
    dt = np.uint8 if bits == 8 else np.uint16
    w, h = 331, 47
    idx = rng.integers(0, count, w * h)
    px = pal[idx].astype(np.uint32)
    strangers = rng.random(w * h) < 0.2
    px[strangers] = rng.integers(0, 256, (int(strangers.sum()), 4))
    if bits == 16:
        px = px << 8 | rng.integers(0, 256, px.shape)       # the low byte is shifted away
    px = px.astype(dt)
    for layout, view in ((orc_pixels.RGBA, px), (orc_pixels.VA, np.ascontiguousarray(px[:, [0, 3]])),
                         (orc_pixels.SCALAR, np.ascontiguousarray(px[:, 0]))):
        want = orc_pixels.pack(view, depth, 1, indexed=True, palette=pal.tobytes(), layout=layout)
        got = s.pack(view.astype("<u%d" % (bits // 8)).tobytes(), w, h, depth, 1, indexed=True, source=bits, palette=pal.tobytes(),
                     layout=layout)
        assert got == want, layout
    # grey palettes, so that the (v, v, v, a) and (v, v, v, 255) indexers find something
    g = rng.permutation(256)[:count].astype(np.uint8)
    gpal = np.stack([g, g, g, np.full(count, 255, dtype=np.uint8)], axis=1)
    v = g[rng.integers(0, count, w * h)].astype(dt)
    if bits == 16:
        v = (v.astype(np.uint32) << 8 | 0x5a).astype(dt)
    want = orc_pixels.pack(v, depth, 1, indexed=True, palette=gpal.tobytes(), layout=orc_pixels.SCALAR)
    assert np.frombuffer(want, dtype=np.uint8).any()
    assert s.pack(v.astype("<u%d" % (bits // 8)).tobytes(), w, h, depth, 1, indexed=True, source=bits, palette=gpal.tobytes(),
                  layout=gpu.TARGET_SCALAR) == want


@pytest.mark.parametrize("name", sorted(TABLE)[::5])
def test_unpack_scalar_target(gpu, name):
    """PNG.Image.unpack<T>(as:) (PNG.Image.swift:682-760, 1030-1040): the grey value / the red channel / palette[i].r, keys
    ignored -- the r component of the RGBA<T> pixels the goldens pin"""
    import struct
    s = gpu.load()
    png, storage, rgba, kw = golden_case(name)
    key = None
    if png.trns and png.color in (0, 2):
        key = struct.unpack(">" + "H" * (1 if png.color == 0 else 3), png.trns[:2 if png.color == 0 else 6])
    args = (storage.tobytes(), png.width, png.height, png.depth, png.channels)
    ukw = dict(indexed=kw["indexed"], bgr=kw["bgr"], palette=kw["palette"], key=key)
    got16 = np.frombuffer(s.unpack(*args, target=16, layout=gpu.TARGET_SCALAR, **ukw), dtype="<u2")
    assert (got16 == rgba[:, 0]).all()
    got8 = np.frombuffer(s.unpack(*args, target=8, layout=gpu.TARGET_SCALAR, **ukw), dtype=np.uint8)
    assert (got8 == (rgba[:, 0] >> 8)).all()


def test_pack_then_encode_then_decode_round_trip(gpu):
    """pixels -> storage -> PNG stream -> storage -> pixels on the device: what Image(packing:).compress / decompress.unpack do"""
    import zlib
    s = gpu.load()
    rng = np.random.default_rng(5)
    w, h = 200, 120
    px = rng.integers(0, 256, (w * h, 4), dtype=np.uint8)
    px[:, :3] &= 0xf0
    storage = s.pack(px.tobytes(), w, h, 8, 3, source=8)
    assert storage == px[:, :3].tobytes()
    rows = s.filter(storage, w, h, 8, 3, False)
    z = s.deflate(rows, 9)
    assert zlib.decompress(z) == rows
    st, back, _ = s.decode(z, w, h, 8, 3, False)
    assert st == 0 and back == storage
    out = np.frombuffer(s.unpack(back, w, h, 8, 3, target=8), dtype=np.uint8).reshape(-1, 4)
    assert (out[:, :3] == px[:, :3]).all() and (out[:, 3] == 255).all()
