"""The gzip restatement (oracle/gzip_wrap.py) pinned against Python's gzip module -- an independent RFC 1952
implementation -- in both directions, and on the header rules the reference states
(Gzip.StreamHeader.swift:17-97)."""
import gzip
import importlib.util
import io
import zlib

import numpy as np

import pnghelp as ph

_spec = importlib.util.spec_from_file_location("gzip_wrap", ph.ROOT / "oracle" / "gzip_wrap.py")
gw = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gw)


def raw_inflate(payload, cap=None):
    return ph.orc_inflate(payload, 1, cap if cap is not None else max(1 << 16, 1100 * len(payload)))


def raw_deflate(level):
    return lambda data: ph.orc_deflate(data, level, 1)


def members():
    rng = np.random.default_rng(3)
    text = b"swift-png on the MI355X " * 300
    noise = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    out = {"plain": (gzip.compress(text, 6, mtime=0), text), "noise": (gzip.compress(noise, 9, mtime=12345), noise),
           "empty": (gzip.compress(b"", 6, mtime=0), b"")}
    buf = io.BytesIO()
    with gzip.GzipFile(filename="name.bin", mode="wb", fileobj=buf, mtime=7) as f:      # FNAME
        f.write(text)
    out["fname"] = (buf.getvalue(), text)
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = body.compress(noise) + body.flush()
    trailer = (zlib.crc32(noise) & 0xffffffff).to_bytes(4, "little") + len(noise).to_bytes(4, "little")
    # FEXTRA + FNAME + FCOMMENT + FTEXT
    out["all-fields"] = (bytes([0x1f, 0x8b, 8, 0x1d, 1, 2, 3, 4, 2, 3]) + (5).to_bytes(2, "little") + b"extra" + b"file\0" +
                         b"a comment\0" + raw + trailer, noise)
    out["bad-isize"] = (gzip.compress(text, 6, mtime=0)[:-4] + b"\1\2\3\4", text)     # ISIZE is read, not checked
    return out


def test_inflate_matches_python_gzip():
    for name, (z, want) in members().items():
        st, out, consumed, aux = gw.inflate(z, raw_inflate)
        assert (st, out, consumed) == (0, want, len(z)), name
        if name != "bad-isize":
            assert gzip.decompress(z) == want


def test_deflate_is_read_by_python_gzip():
    rng = np.random.default_rng(4)
    for data in (b"", b"\1", b"abc" * 1000, rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()):
        for level in (0, 7, 10):
            z = gw.deflate(data, raw_deflate(level))
            assert z[:10] == bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff])
            assert gzip.decompress(z) == data
            assert gw.inflate(z, raw_inflate)[:3] == (0, data, len(z))


def test_header_rules():
    z, want = members()["plain"]
    bad = bytearray(z); bad[0] = 0x1e
    assert gw.inflate(bytes(bad), raw_inflate)[0] == gw.E_GZIP_SIGIL
    bad = bytearray(z); bad[2] = 7
    assert gw.inflate(bytes(bad), raw_inflate)[::3] == (gw.E_GZIP_METHOD, (7, 0))
    bad = bytearray(z); bad[3] = 0x40
    assert gw.inflate(bytes(bad), raw_inflate)[::3] == (gw.E_GZIP_FLAG_BITS, (0x40, 0))
    bad = bytearray(z); bad[3] = 0x02
    assert gw.inflate(bytes(bad), raw_inflate)[0] == gw.E_GZIP_HEADER_CHECKSUM
    bad = bytearray(z); bad[-8] ^= 1
    st, out, _, aux = gw.inflate(bytes(bad), raw_inflate)
    assert st == gw.E_STREAM_CHECKSUM and out == want and aux == (int.from_bytes(bad[-8:-4], "little"), zlib.crc32(want))
    for cut in (0, 5, 9):                                     # header incomplete: wants more input
        assert gw.inflate(z[:cut], raw_inflate)[0] == gw.NEED_MORE_INPUT
    assert gw.inflate(z[:-8], raw_inflate)[:2] == (gw.NEED_MORE_INPUT, want)      # no trailer yet
    assert gw.inflate(z[:-2], raw_inflate)[:2] == (gw.NEED_MORE_INPUT, want)      # CRC fine, ISIZE incomplete
    full, _ = members()["all-fields"]
    assert gw.inflate(full[:11], raw_inflate)[0] == gw.NEED_MORE_INPUT            # XLEN incomplete
    assert gw.inflate(full[:15], raw_inflate)[0] == gw.NEED_MORE_INPUT            # FEXTRA bytes incomplete
    assert gw.inflate(full[:20], raw_inflate)[0] == gw.NEED_MORE_INPUT            # FNAME unterminated
