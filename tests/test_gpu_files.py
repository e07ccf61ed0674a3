"""GPU parity for the rows either side of the hot path: chunk lexing with CRC-32 + IDAT assembly
(PNG.BytestreamSource.chunk, PNG.Image.decompress's IDAT loop), IDAT emission (PNG.BytestreamDestination.format),
checked against the test-side lexer (zlib.crc32) on every PngSuite fixture, against the reference's negative
fixtures with the two checksums its own tests pin, and end to end: file bytes -> pixels on the device."""
import hashlib
import json
import struct
import zlib

import numpy as np
import pytest

import pnghelp as ph

pytestmark = pytest.mark.gpu
TABLE = json.loads((ph.GOLDEN / "pngsuite.json").read_text())


def test_crc32_vs_zlib(gpu):
    s = gpu.load()
    rng = np.random.default_rng(3)
    for n in (0, 1, 3, 63, 64, 65, 4095, 70000, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 3 * (1 << 20) + 12345):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert s.crc32(data) == zlib.crc32(data), n
    assert s.crc32(b"IEND") == 0xAE426082


def test_lex_every_fixture(gpu):
    """signature, chunk walk, CRC-32 of every chunk, IHDR fields, concatenated IDAT: equal to the host lexer."""
    s = gpu.load()
    names = sorted(TABLE)
    files = [(ph.GOLDEN / "pngsuite" / n).read_bytes() for n in names]
    infos, idats = s.lex_batch(files)
    for name, f, r, idat in zip(names, files, infos, idats):
        png = ph.parse_png(f)
        assert r.status == 0, (name, r.status)
        assert (r.width, r.height, r.depth, r.color, bool(r.interlace), bool(r.ios)) == \
               (png.width, png.height, png.depth, png.color, png.interlaced, png.ios), name
        assert idat == png.idat and r.idat_len == len(png.idat), name
        if png.palette is not None:
            assert f[r.plte_off:r.plte_off + r.plte_len] == png.palette
        if png.trns:
            assert f[r.trns_off:r.trns_off + r.trns_len] == png.trns
        assert r.consumed <= len(f) and f[r.consumed - 12 + 4:r.consumed - 4] == b"IEND"


def test_lex_reference_negative_fixtures(gpu):
    """Sources/PNGIntegrationTests/ErrorHandling.swift: six bad signatures, the two pinned chunk checksums."""
    s = gpu.load()
    inv = ph.GOLDEN / "invalid"
    def lex(name):
        return s.lex_batch([(inv / f"{name}.png").read_bytes()])[0][0]
    for name in ("xs1n0g01", "xs2n0g01", "xs4n0g01", "xs7n0g01", "xcrn0g04", "xlfn0g04"):
        assert lex(name).status == gpu.E_SIGNATURE, name
    r = lex("xhdn0g08")
    assert (r.status, r.aux[0], r.aux[1]) == (gpu.E_CHUNK_CHECKSUM, 1129534797, 1443964200)
    r = lex("xcsn0g01")
    assert (r.status, r.aux[0], r.aux[1]) == (gpu.E_CHUNK_CHECKSUM, 1129534797, 3492746441)
    for name in ("xc1n0g08", "xc9n2c08", "xd0n2c08", "xd3n2c08", "xd9n2c08", "xdtn0g01"):
        assert lex(name).status == 0, name                      # (parsing / decoding errors: above the lexer)
    assert lex("xdtn0g01").idat_len == 0
    # truncation
    f = (ph.GOLDEN / "pngsuite" / "common" / "basn6a08.png").read_bytes()
    assert s.lex_batch([f[:5]])[0][0].status == gpu.E_TRUNCATED_SIGNATURE
    assert s.lex_batch([f[:12]])[0][0].status == gpu.E_TRUNCATED_CHUNK_HEADER
    assert s.lex_batch([f[:30]])[0][0].status == gpu.E_TRUNCATED_CHUNK_BODY             # inside IHDR's body
    assert s.lex_batch([f[:-12]])[0][0].status == gpu.E_TRUNCATED_CHUNK_HEADER      # no IEND
    bad = bytearray(f); bad[8 + 6] = ord("d")                                          # "IHdR": reserved bit set
    r = s.lex_batch([bytes(bad)])[0][0]
    assert r.status == gpu.E_CHUNK_TYPE and r.aux[0] == struct.unpack(">I", bytes(bad[12:16]))[0]


def test_write_idat_roundtrip(gpu):
    s = gpu.load()
    rng = np.random.default_rng(4)
    z = zlib.compress(rng.integers(0, 7, 300000, dtype=np.uint8).tobytes(), 6)
    for piece in (1, 7, 8192, 65536, len(z), len(z) + 5):
        if piece == 1:
            zz = z[:300]
        else:
            zz = z
        out = s.write_idat(zz, piece)
        pos, got = 0, []
        while pos < len(out):
            (n,) = struct.unpack(">I", out[pos:pos + 4])
            assert out[pos + 4:pos + 8] == b"IDAT" and n <= piece
            body = out[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", out[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(b"IDAT" + body)
            got.append(body)
            pos += 12 + n
        assert b"".join(got) == zz and len(got) == -(-len(zz) // piece)


def test_file_to_pixels_on_device(gpu):
    """The whole chain the reference's decode benchmark times (Benchmarks/Decompression/Swift/Main.swift:103-109):
    file bytes -> lex + CRC -> inflate -> defilter -> storage -> unpack(as: RGBA<UInt16>), all kernels."""
    s = gpu.load()
    for name in ("common/basi6a16.png", "common/basn3p08.png", "common/oi9n2c16.png", "common/tbbn3p08.png", "ios/PngSuite.png"):
        f = (ph.GOLDEN / "pngsuite" / name).read_bytes()
        (r,), (idat,) = s.lex_batch([f])
        assert r.status == 0
        channels = ph.CHANNELS[r.color]
        st, storage, _ = s.decode(idat, r.width, r.height, r.depth, channels, r.interlace, int(r.ios))
        assert st == 0
        pal = None
        if r.color == 3:
            q = np.full((r.plte_len // 3, 4), 255, np.uint8)
            q[:, :3] = np.frombuffer(f[r.plte_off:r.plte_off + r.plte_len], np.uint8).reshape(-1, 3)
            t = np.frombuffer(f[r.trns_off:r.trns_off + r.trns_len], np.uint8)[:len(q)]
            q[:len(t), 3] = t
            pal = q.tobytes()
        key = None
        if r.trns_len and r.color in (0, 2):
            key = struct.unpack(">" + "H" * (r.trns_len // 2), f[r.trns_off:r.trns_off + r.trns_len])
        px = s.unpack(storage, r.width, r.height, r.depth, channels, indexed=r.color == 3, bgr=bool(r.ios) and r.color in (2, 6),
                      target=16, palette=pal, key=key)
        assert hashlib.sha256(px).hexdigest() == TABLE[name]["rgba16_sha256"], name
