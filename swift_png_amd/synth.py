"""Deterministic synthetic images for the benchmark / parity tests.

A mix that exercises every filter and both LZ77 regimes: smooth 2-D gradients with low-amplitude
noise (Paeth / Average / Up win), flat and tiled regions (long matches, None / Sub win) and ~10 %
pure-noise tiles (literal-dominated).  Seed = image index.  Pure numpy; no GPU, no oracle.
"""
from __future__ import annotations

import numpy as np


def image(seed: int, width: int = 4096, height: int = 4096, channels: int = 4, depth: int = 8,
          tile: int = 256) -> np.ndarray:
    """-> uint8 array (height, width * channels * depth/8): PNG.Image.storage row-major bytes
    (16-bit samples big-endian)."""
    rng = np.random.default_rng(seed)
    bps = depth // 8
    maxv = (1 << depth) - 1
    out = np.empty((height, width, channels), dtype=np.uint16 if depth == 16 else np.uint8)
    yy = np.arange(height, dtype=np.float32)[:, None]
    xx = np.arange(width, dtype=np.float32)[None, :]
    for c in range(channels):
        fx, fy = rng.uniform(0.2, 3.0, 2)
        ph = rng.uniform(0, 6.28)
        g = 0.5 + 0.25 * np.sin(xx * (fx * 6.28 / width) + ph) + 0.25 * np.cos(yy * (fy * 6.28 / height))
        out[:, :, c] = (g * (maxv * 0.9)).astype(out.dtype)
    noise = rng.integers(-2, 3, size=out.shape, dtype=np.int32)
    if depth == 16:
        noise *= 64
    out[:] = np.clip(out.astype(np.int32) + noise, 0, maxv).astype(out.dtype)
    ty, tx = (height + tile - 1) // tile, (width + tile - 1) // tile
    kinds = rng.random((ty, tx))
    for j in range(ty):
        for i in range(tx):
            k = kinds[j, i]
            y0, x0 = j * tile, i * tile
            y1, x1 = min(y0 + tile, height), min(x0 + tile, width)
            if k < 0.10:        # pure noise
                out[y0:y1, x0:x1] = rng.integers(0, maxv + 1, size=(y1 - y0, x1 - x0, channels)).astype(out.dtype)
            elif k < 0.25:      # flat colour
                out[y0:y1, x0:x1] = rng.integers(0, maxv + 1, size=(1, 1, channels)).astype(out.dtype)
            elif k < 0.35:      # 16x16 pattern repeated
                pat = rng.integers(0, maxv + 1, size=(16, 16, channels)).astype(out.dtype)
                reps = (-(-(y1 - y0) // 16), -(-(x1 - x0) // 16), 1)
                out[y0:y1, x0:x1] = np.tile(pat, reps)[:y1 - y0, :x1 - x0]
    if depth == 16:
        out = out.astype(">u2")
    return out.view(np.uint8).reshape(height, width * channels * bps)


def forced_filter_rows(storage: np.ndarray, bpp: int, filters) -> np.ndarray:
    """Filters `storage` (H, pitch) with the given per-row filter ids (numpy, for tests/bench
    variants that force a filter mix).  Plain PNG filter arithmetic (RFC 2083 section 6)."""
    h, pitch = storage.shape
    rows = np.zeros((h, pitch + 1), dtype=np.uint8)
    prev = np.zeros(pitch, dtype=np.int32)
    for y in range(h):
        f = int(filters[y % len(filters)])
        x = storage[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), x[:-bpp]])
        b = prev
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0:
            r = x
        elif f == 1:
            r = x - a
        elif f == 2:
            r = x - b
        elif f == 3:
            r = x - ((a + b) >> 1)
        else:
            pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            r = x - pred
        rows[y, 0] = f
        rows[y, 1:] = (r & 255).astype(np.uint8)
        prev = x
    return rows
