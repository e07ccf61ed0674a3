"""CPU oracle of `pack` (oracle/pixels.py), pinned on the reference's own goldens.

The reference holds, per PngSuite input, the RGBA<UInt16> pixels it must decode to (Sources/PNGIntegrationTests/RGBA/*.rgba;
digests in tests/golden/pngsuite.json, checked against the reference's files by test_oracle_decode.py).  `unpack` is injective
for T = UInt16 at every depth, so those pixels packed as the image's own format must give back PNG.Image.storage -- which the
pinned decode oracle produces: the golden pixels are the INPUT of the restatement here, the storage its expected output."""
import hashlib
import json
import sys

import numpy as np
import pytest

import pnghelp as ph

sys.path.insert(0, str(ph.ROOT / "oracle"))
import pixels as orc_pixels  # noqa: E402

TABLE = json.loads((ph.GOLDEN / "pngsuite.json").read_text())


def palette_quads(png):
    pal = np.frombuffer(png.palette, dtype=np.uint8).reshape(-1, 3)
    alpha = np.full(len(pal), 255, dtype=np.uint8)
    if png.trns:
        t = np.frombuffer(png.trns, dtype=np.uint8)
        alpha[:len(t)] = t
    return np.concatenate([pal, alpha[:, None]], axis=1).tobytes()


def repeats_a_colour(quads: bytes) -> bool:
    q = np.frombuffer(quads, dtype=np.uint8).reshape(-1, 4)
    return len({tuple(e) for e in q}) != len(q)


def golden_case(name):
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    st, storage, _ = ph.orc_decode(png)
    assert st == 0
    rgba = ph.unpack_rgba16(storage, png)
    kw = dict(depth=png.depth, channels=png.channels, indexed=png.color == 3, bgr=png.ios and png.color in (2, 6),
              palette=palette_quads(png) if png.color == 3 else None)
    return png, storage, rgba, kw


@pytest.mark.parametrize("name", sorted(TABLE))
def test_pack_of_the_golden_pixels_is_the_storage(name):
    png, storage, rgba, kw = golden_case(name)
    if not png.ios:                                    # (the iOS goldens are premultiplied: their straight pixels come from unpack)
        assert hashlib.sha256(rgba.astype("<u2").tobytes()).hexdigest() == TABLE[name]["rgba16_sha256"]
    if kw["indexed"] and repeats_a_colour(kw["palette"]):
        with pytest.raises(ValueError):
            orc_pixels.pack(rgba, **kw)
        return
    assert orc_pixels.pack(rgba, **kw) == storage.tobytes()
    # T = UInt8: the same pixels seen through >> 8; equal storage wherever the format is at most 8 bits deep
    rgba8 = (rgba >> 8).astype(np.uint8)
    got = orc_pixels.pack(rgba8, **kw)
    if png.depth <= 8:
        assert got == storage.tobytes()
    else:                                              # 8 -> 16: times the quantum 257 (PNG.swift:255-261)
        s16 = np.frombuffer(storage.tobytes(), dtype=">u2").astype(np.uint32)
        assert got == ((s16 >> 8) * 257).astype(">u2").tobytes()
    # the other two colour targets: (v, a) and v carry the red channel
    va = rgba[:, [0, 3]]
    grey = rgba.copy(); grey[:, 1] = grey[:, 2] = grey[:, 0]
    assert orc_pixels.pack(va, layout=orc_pixels.VA, **kw) == orc_pixels.pack(grey, **kw)
    opaque = grey.copy(); opaque[:, 3] = 65535
    assert orc_pixels.pack(rgba[:, 0].copy(), layout=orc_pixels.SCALAR, **kw) == orc_pixels.pack(opaque, **kw)


def test_pack_strangers_index_zero_and_sub_byte_shifts():
    pal = bytes([10, 20, 30, 255, 1, 2, 3, 4, 200, 200, 200, 255])
    px = np.array([[1, 2, 3, 4], [9, 9, 9, 9], [200, 200, 200, 255], [10, 20, 30, 255]], dtype=np.uint8)
    assert orc_pixels.pack(px, 8, 1, indexed=True, palette=pal) == bytes([1, 0, 2, 0])
    assert orc_pixels.pack((px.astype(np.uint16) * 257), 2, 1, indexed=True, palette=pal) == bytes([1, 0, 2, 0])
    v = np.array([0, 0x55, 0xAA, 0xFF, 0x80], dtype=np.uint8)
    assert orc_pixels.pack(v, 2, 1, layout=orc_pixels.SCALAR) == bytes([0, 1, 2, 3, 2])       # v >> (8 - 2)
    assert orc_pixels.pack(v, 16, 2, layout=orc_pixels.SCALAR) == b"".join(
        (int(x) * 257).to_bytes(2, "big") + b"\xff\xff" for x in v)                                # (v, .max), quantum 257
