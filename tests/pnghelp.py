"""Test-side helpers: PNG chunk lexing, the ctypes binding of the CPU oracle, and the
storage -> RGBA16 "unpack" needed to compare with the reference's .rgba goldens.

None of this is product code.  The chunk lexer restates just enough of
Sources/PNG/Lexing/PNG.BytestreamSource.swift:44-108 (signature, length/type/data/CRC-32)
and Sources/PNG/Parsing/PNG.Header.swift:73-129 (IHDR fields) to feed the hot path; the
unpack restates Sources/PNG/ColorTargets/PNG.RGBA.swift:259-365 for T == UInt16.
"""
from __future__ import annotations

import ctypes
import os
import struct
import subprocess
import zlib
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
REFERENCE = Path("/root/reference")

SIGNATURE = bytes([137, 80, 78, 71, 13, 10, 26, 10])
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


@dataclass
class Png:
    width: int
    height: int
    depth: int
    color: int
    interlaced: bool
    ios: bool
    idat: bytes
    idat_chunks: list = field(default_factory=list)
    palette: bytes | None = None
    trns: bytes | None = None

    @property
    def channels(self) -> int:
        return CHANNELS[self.color]

    @property
    def fmt(self) -> int:
        return 1 if self.ios else 0


def parse_png(data: bytes) -> Png:
    if data[:8] != SIGNATURE:
        raise ValueError("invalid signature")
    pos, hdr, ios = 8, None, False
    idat, chunks, palette, trns = [], [], None, None
    while pos < len(data):
        (length,) = struct.unpack(">I", data[pos:pos + 4])
        ctype = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        (crc,) = struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])
        if zlib.crc32(ctype + body) != crc:
            raise ValueError("invalid chunk checksum")
        pos += 12 + length
        if ctype == b"CgBI":
            ios = True
        elif ctype == b"IHDR":
            w, h, depth, color, _, _, inter = struct.unpack(">IIBBBBB", body)
            hdr = (w, h, depth, color, bool(inter))
        elif ctype == b"PLTE":
            palette = body
        elif ctype == b"tRNS":
            trns = body
        elif ctype == b"IDAT":
            idat.append(body)
            chunks.append(len(body))
        elif ctype == b"IEND":
            break
    w, h, depth, color, inter = hdr
    return Png(w, h, depth, color, inter, ios, b"".join(idat), chunks, palette, trns)


def unpack_rgba16(storage: np.ndarray, png: Png) -> np.ndarray:
    """PNG.Image.storage -> (H*W, 4) uint16 RGBA, as PNG.RGBA<UInt16>.unpack does."""
    n = png.width * png.height
    out = np.empty((n, 4), dtype=np.uint32)
    d, c = png.depth, png.color
    if c == 3:
        pal = np.frombuffer(png.palette, dtype=np.uint8).reshape(-1, 3).astype(np.uint32)
        alpha = np.full(len(pal), 255, dtype=np.uint32)
        if png.trns:
            t = np.frombuffer(png.trns, dtype=np.uint8)
            alpha[:len(t)] = t
        idx = storage.astype(np.int64)
        out[:, :3] = pal[idx] * 257
        out[:, 3] = alpha[idx] * 257
    elif c == 0:
        if d == 16:
            v = storage.reshape(n, 2).astype(np.uint32)
            raw = v[:, 0] << 8 | v[:, 1]
            val = raw
        else:
            raw = storage.astype(np.uint32)
            val = raw * (65535 // ((1 << d) - 1))
        out[:, 0] = out[:, 1] = out[:, 2] = val
        out[:, 3] = 65535
        if png.trns:
            key = struct.unpack(">H", png.trns[:2])[0]
            out[raw == key, 3] = 0
    elif c == 4:
        if d == 16:
            v = storage.reshape(n, 2, 2).astype(np.uint32)
            s = v[:, :, 0] << 8 | v[:, :, 1]
        else:
            s = storage.reshape(n, 2).astype(np.uint32) * 257
        out[:, 0] = out[:, 1] = out[:, 2] = s[:, 0]
        out[:, 3] = s[:, 1]
    else:
        ch = 3 if c == 2 else 4
        if d == 16:
            v = storage.reshape(n, ch, 2).astype(np.uint32)
            raw = v[:, :, 0] << 8 | v[:, :, 1]
            s = raw
        else:
            raw = storage.reshape(n, ch).astype(np.uint32)
            s = raw * 257
        if png.ios:                                   # bgr8 / bgra8 (PNG.RGBA.swift:313,348)
            s = s.copy(); raw = raw.copy()
            s[:, [0, 2]] = s[:, [2, 0]]
            raw[:, [0, 2]] = raw[:, [2, 0]]
        out[:, :3] = s[:, :3]
        out[:, 3] = s[:, 3] if ch == 4 else 65535
        if ch == 3 and png.trns:
            key = np.array(struct.unpack(">HHH", png.trns[:6]), dtype=np.uint32)
            out[(raw[:, :3] == key).all(axis=1), 3] = 0
    return out.astype(np.uint16)


def premultiply8(rgba16: np.ndarray) -> np.ndarray:
    """PNG.RGBA<UInt16>.premultiplied(as: UInt8.self) (PNG.RGBA.swift:146-158)."""
    v = (rgba16 >> 8).astype(np.uint32)
    a = v[:, 3:4]
    rgb = (v[:, :3] * a + 127) // 255                 # PNG.premultiply, PNG.swift:55-67
    return (np.concatenate([rgb, a], axis=1) * 257).astype(np.uint16)


# ------------------------------------------------------------------ oracle binding

_oracle = None


def oracle():
    """Builds (if needed) and loads oracle/liboracle.so."""
    global _oracle
    if _oracle is not None:
        return _oracle
    so = ROOT / "oracle" / "liboracle.so"
    srcs = [ROOT / "oracle" / f for f in ("inflate.c", "png_rows.c", "deflate.c", "spng_oracle.h")]
    if not so.exists() or any(s.stat().st_mtime > so.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    lib = ctypes.CDLL(str(so))
    u8p, szp, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint64)
    lib.orc_adler32.restype = ctypes.c_uint32
    lib.orc_adler32.argtypes = [ctypes.c_uint32, u8p, ctypes.c_size_t]
    lib.orc_inflate.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p, ctypes.c_size_t, szp, szp, u64p]
    lib.orc_inflated_size.restype = ctypes.c_size_t
    lib.orc_inflated_size.argtypes = [ctypes.c_int] * 5
    lib.orc_storage_size.restype = ctypes.c_size_t
    lib.orc_storage_size.argtypes = [ctypes.c_int] * 4
    lib.orc_unfilter.argtypes = [u8p, ctypes.c_size_t] + [ctypes.c_int] * 5 + [u8p]
    lib.orc_decode.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int] + [ctypes.c_int] * 5 + [u8p, u64p]
    lib.orc_defilter.restype = None
    lib.orc_defilter.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_int]
    lib.orc_filter_row.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_int, u8p]
    lib.orc_filter.argtypes = [u8p] + [ctypes.c_int] * 5 + [u8p]
    if hasattr(lib, "orc_deflate"):
        lib.orc_deflate.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    u8p, ctypes.c_size_t, szp]
        lib.orc_deflate_bound.restype = ctypes.c_size_t
        lib.orc_deflate_bound.argtypes = [ctypes.c_size_t]
        lib.orc_encode.argtypes = [u8p] + [ctypes.c_int] * 7 + [u8p, ctypes.c_size_t, szp]
    _oracle = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def orc_inflate(data: bytes, fmt: int = 0, cap: int | None = None):
    """-> (status, output bytes, consumed, (aux0, aux1))"""
    lib = oracle()
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    cap = cap if cap is not None else max(1 << 16, 1100 * len(data))
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    written, consumed = ctypes.c_size_t(0), ctypes.c_size_t(0)
    aux = (ctypes.c_uint64 * 2)()
    st = lib.orc_inflate(_ptr(src), len(data), fmt, _ptr(dst), cap,
                         ctypes.byref(written), ctypes.byref(consumed), aux)
    return st, dst[:written.value].tobytes(), consumed.value, (aux[0], aux[1])


def orc_sizes(png: Png):
    lib = oracle()
    u = lib.orc_inflated_size(png.width, png.height, png.depth, png.channels, int(png.interlaced))
    s = lib.orc_storage_size(png.width, png.height, png.depth, png.channels)
    return u, s


def orc_decode(png: Png, idat: bytes | None = None):
    """-> (status, storage ndarray, aux)"""
    lib = oracle()
    idat = png.idat if idat is None else idat
    _, s = orc_sizes(png)
    storage = np.zeros(max(s, 1), dtype=np.uint8)
    src = np.frombuffer(idat, dtype=np.uint8) if len(idat) else np.zeros(1, np.uint8)
    aux = (ctypes.c_uint64 * 2)()
    st = lib.orc_decode(_ptr(src), len(idat), png.fmt, png.width, png.height, png.depth,
                        png.channels, int(png.interlaced), _ptr(storage), aux)
    return st, storage[:s], (aux[0], aux[1])


def orc_unfilter(rows: bytes, w, h, depth, channels, interlaced):
    lib = oracle()
    s = lib.orc_storage_size(w, h, depth, channels)
    storage = np.zeros(max(s, 1), dtype=np.uint8)
    src = np.frombuffer(rows, dtype=np.uint8) if len(rows) else np.zeros(1, np.uint8)
    st = lib.orc_unfilter(_ptr(src), len(rows), w, h, depth, channels, int(interlaced), _ptr(storage))
    return st, storage[:s]


def orc_filter(storage: np.ndarray, w, h, depth, channels, interlaced) -> bytes:
    lib = oracle()
    u = lib.orc_inflated_size(w, h, depth, channels, int(interlaced))
    rows = np.zeros(max(u, 1), dtype=np.uint8)
    storage = np.ascontiguousarray(storage, dtype=np.uint8)
    lib.orc_filter(_ptr(storage), w, h, depth, channels, int(interlaced), _ptr(rows))
    return rows[:u].tobytes()


def orc_deflate(data: bytes, level: int, fmt: int = 0, exponent: int = 15) -> bytes:
    lib = oracle()
    src = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, np.uint8)
    cap = lib.orc_deflate_bound(len(data))
    dst = np.empty(cap, dtype=np.uint8)
    written = ctypes.c_size_t(0)
    st = lib.orc_deflate(_ptr(src), len(data), fmt, level, exponent, _ptr(dst), cap, ctypes.byref(written))
    assert st == 0, st
    return dst[:written.value].tobytes()


def have_reference() -> bool:
    return (REFERENCE / "Sources" / "PNGIntegrationTests" / "Inputs" / "Common").is_dir()
