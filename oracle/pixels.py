"""oracle/pixels.py -- TEST INFRASTRUCTURE, not product code: a numpy restatement of the reference's `pack` (pixels ->
PNG.Image.storage), the checker of `spng_pack_batch`.  Only tests/, smoke() and bench legs' checkers may import it.

Follows, function by function:
  PNG.RGBA<T>.pack(_:as:indexer:)      Sources/PNG/ColorTargets/PNG.RGBA.swift:409-478
  PNG.VA<T>.pack(_:as:indexer:)        Sources/PNG/ColorTargets/PNG.VA.swift:334-403
  PNG.Image.pack<T>(_:as:indexer:)     Sources/PNG/PNG.Image.swift:767-834        (scalar pixels)
  the default indexers                 Sources/PNG/ColorTargets/PNG.Color.swift:158-226, PNG.Image.swift:1043-1062
  PNG.deconvolve(_:as:depth:kernel:)   Sources/PNG/PNG.swift:1064-1285, :699-745  (big-endian stores)
  PNG.deconvolve(_:reference:kernel:)  Sources/PNG/PNG.swift:819-1062, :748-793   (one byte per pixel)
  PNG.quantum(source:destination:)     Sources/PNG/PNG.swift:255-261

Pinned (tests/test_oracle_pack.py) on the reference's own goldens: every `.rgba` file of Sources/PNGIntegrationTests/RGBA
(RGBA<UInt16> pixels) packed as its PNG's format gives back the storage the pinned decode oracle produces for that PNG, and
pack(unpack(storage)) == storage for T = UInt8 wherever the unpack is injective.
"""
from __future__ import annotations

import numpy as np

RGBA, VA, SCALAR = 0, 1, 2


def _transform(v: np.ndarray, tbits: int, depth: int) -> np.ndarray:
    """T -> A at the format's depth (PNG.swift:1076-1095): equal widths as is, narrower T times the quantum, wider T shifted."""
    v = v.astype(np.uint32)
    if tbits == depth:
        return v
    if tbits < depth:
        quantum = ((1 << depth) - 1) // ((1 << tbits) - 1)             # T.max >> (T.bitWidth - destination) / T.max >> (T.bitWidth - source)
        return (quantum * v) & ((1 << depth) - 1)
    return v >> (tbits - depth)


def pack(pixels: np.ndarray, depth: int, channels: int, indexed: bool = False, bgr: bool = False,
         palette: bytes | None = None, layout: int = RGBA) -> bytes:
    """pixels: (n, 4) r g b a | (n, 2) v a | (n,) v, dtype uint8 / uint16  ->  storage bytes."""
    tbits = pixels.dtype.itemsize * 8
    tmax = (1 << tbits) - 1
    px = pixels.reshape(len(pixels), -1).astype(np.uint32)
    n = len(px)
    if layout == RGBA:
        r, g, b, a = px[:, 0], px[:, 1], px[:, 2], px[:, 3]
    elif layout == VA:
        r = g = b = px[:, 0]
        a = px[:, 1]
    else:
        r = g = b = px[:, 0]
        a = np.full(n, tmax, dtype=np.uint32)                          # (v, .max), (v, v, v, .max); indexer: (v, v, v, 255)
    if indexed:
        # components as UInt8 atoms (A == UInt8: T wider -> shift), then the default indexer's dictionary; entry 0 for strangers.
        # Dictionary(uniqueKeysWithValues:) traps on a repeated colour (PNG.Color.swift:182-183): refused here as well.
        pal = np.frombuffer(palette, dtype=np.uint8).reshape(-1, 4)
        lookup = {}
        for i, e in enumerate(pal):
            if tuple(e) in lookup:
                raise ValueError("the reference traps: Dictionary(uniqueKeysWithValues:) on a palette that repeats a colour")
            lookup[tuple(int(x) for x in e)] = i
        s = tbits - 8
        keys = np.stack([r >> s, g >> s, b >> s, a >> s], axis=1)
        out = np.fromiter((lookup.get(tuple(int(x) for x in k), 0) for k in keys), dtype=np.uint8, count=n)
        return out.tobytes()
    if channels == 1:
        comps = [r]
    elif channels == 2:
        comps = [r, a]
    elif channels == 3:
        comps = [b, g, r] if bgr else [r, g, b]
    else:
        comps = [b, g, r, a] if bgr else [r, g, b, a]
    atoms = np.stack([_transform(c, tbits, depth) for c in comps], axis=1)
    if depth == 16:
        return atoms.astype(">u2").tobytes()                           # samples[i] = transform(...).bigEndian
    return atoms.astype(np.uint8).tobytes()                            # depth < 8: one unscaled byte per sample in storage


# ---- PNG.Context.push(data:overdraw: true): progressive display of an unfinished Adam7 image --------------------------------
ADAM7 = [((0, 0), (3, 3)), ((4, 0), (3, 3)), ((0, 4), (2, 3)), ((2, 0), (2, 2)), ((0, 2), (1, 2)), ((1, 0), (1, 1)),
         ((0, 1), (0, 1))]                                                # PNG.adam7 (base, exponent), PNG.Decoder.swift:6-15


def overdrawn(final: np.ndarray, scanlines: int, fill: int = 0) -> np.ndarray:
    """What PNG.Image.storage holds after the first `scanlines` scanlines of an interlaced image went through
    PNG.Context.push(data:overdraw: true) -- scanline by scanline as the reference does it (PNG.Decoder.swift:59-110 order,
    PNG.Context.swift:92-97 brush, PNG.Image.swift:134-183 overdraw; assign = the finished image's pixel).
    final: (H, W, elem) uint8, the finished storage; pixels nothing has reached yet hold `fill`."""
    H, W, _ = final.shape
    img = np.full_like(final, fill)
    left = scanlines
    for (bx, by), (ex, ey) in ADAM7:
        sx, sy = 1 << ex, 1 << ey
        sub_w, sub_h = (W + sx - bx - 1) >> ex, (H + sy - by - 1) >> ey
        if sub_w <= 0 or sub_h <= 0:
            continue
        for y in range(sub_h):
            if left == 0:
                return img
            left -= 1
            base_x, base_y = bx, by + y * sy
            img[base_y, base_x::sx] = final[base_y, base_x::sx]          # assign(scanline:at:stride:)
            s_x, s_y = (0 if base_x == 0 else 1), (0 if base_y & 7 == 0 else 1)
            brush_x, brush_y = sx >> s_x, sy >> s_y
            if brush_x * brush_y <= 1:
                continue
            for yy in range(base_y, min(base_y + brush_y, H)):
                for x in range(base_x, W, brush_x):
                    for xx in range(x, min(x + brush_x, W)):
                        img[yy, xx] = img[base_y, x]
    return img
