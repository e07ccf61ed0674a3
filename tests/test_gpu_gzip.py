"""SPNG_FORMAT_GZIP on the device (csrc/gzip.hip around the DEFLATE kernels) against the gzip restatement
(oracle/gzip_wrap.py, itself pinned on Python's gzip module) -- status, bytes, consumed and error payloads --
and through the mirrored Gzip.Inflator / Gzip.Deflator with the reference's own round-trip tests
(Sources/LZ77Tests/Compression.swift:29-51, CompressionMicro.swift:6-28)."""
import gzip
import zlib

import numpy as np
import pytest

import pnghelp as ph
import swift_png_amd as spng
from test_oracle_gzip import gw, members, raw_deflate, raw_inflate

pytestmark = pytest.mark.gpu


def same(s, z, cap=None):
    cap = cap if cap is not None else max(1 << 16, 1100 * len(z))
    want = gw.inflate(z, lambda p, c: raw_inflate(p, cap), cap)
    got = s.inflate(z, spng.FORMAT_GZIP, cap)
    assert got[0] == want[0], (got[0], want[0])
    assert got[1] == want[1]
    assert got[3] == tuple(want[3])
    if want[0] == 0:
        assert got[2] == want[2]
    return got


def test_gzip_inflate_members(gpu):
    s = gpu.load()
    for name, (z, data) in members().items():
        got = same(s, z)
        assert got[0] == 0 and got[1] == data and got[2] == len(z), name


def test_gzip_inflate_header_and_trailer_errors(gpu):
    s = gpu.load()
    z, data = members()["plain"]
    full, _ = members()["all-fields"]
    cases = []
    for at, val in ((0, 0x1e), (1, 0x8a), (2, 7), (3, 0x40), (3, 0x80), (3, 0x02), (len(z) - 8, z[-8] ^ 1), (len(z) // 2, z[len(z) // 2] ^ 0x55)):
        bad = bytearray(z); bad[at] = val
        cases.append(bytes(bad))
    cases += [z[:cut] for cut in (0, 1, 5, 9, 10, 40, len(z) - 9, len(z) - 8, len(z) - 5, len(z) - 4, len(z) - 1)]
    cases += [full[:cut] for cut in (11, 12, 15, 17, 20, 22, 30, 32)]
    seen = set()
    for c in cases:
        seen.add(same(s, c)[0])
    assert {spng.E_GZIP_SIGIL, spng.E_GZIP_METHOD, spng.E_GZIP_FLAG_BITS, spng.E_GZIP_HEADER_CHECKSUM, spng.E_STREAM_CHECKSUM,
            spng.NEED_MORE_INPUT} <= seen


@pytest.mark.parametrize("level", [0, 3, 7, 9, 10])
def test_gzip_deflate_vs_oracle(gpu, level):
    s = gpu.load()
    rng = np.random.default_rng(level)
    for data in (b"", b"\1", b"\1\2", b"abc" * 2000, rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(),
                 (rng.integers(0, 5, 30000, dtype=np.uint8)).tobytes()):
        got = s.deflate(data, level, spng.FORMAT_GZIP)
        assert got == gw.deflate(data, raw_deflate(level)), (level, len(data))
        assert gzip.decompress(got) == data


def test_gzip_large_stream_takes_the_pipeline_and_checks_the_crc(gpu):
    """8 MiB inflated: the parallel inflate pipeline decodes the payload (reserved == 1), the CRC-32 of the output is
    folded from 256 wave-parallel pieces; a flipped output-side bit (in the trailer) is reported with both sums."""
    s = gpu.load()
    rng = np.random.default_rng(11)
    a = rng.integers(-3, 4, 8 << 20).astype(np.int16); a[rng.random(8 << 20) < 0.6] = 0
    data = a.astype(np.uint8).tobytes()
    z = gzip.compress(data, 6, mtime=0)
    outs, res = s.inflate_batch([s.to_device(z)], [len(data) + 64], spng.FORMAT_GZIP)
    assert res[0].status == 0 and res[0].written == len(data) and res[0].consumed == len(z) and res[0].reserved == 1
    assert bytes(outs[0][:len(data)].cpu().numpy()) == data
    bad = bytearray(z); bad[-6] ^= 0x10
    outs, res = s.inflate_batch([s.to_device(bytes(bad))], [len(data) + 64], spng.FORMAT_GZIP)
    assert res[0].status == spng.E_STREAM_CHECKSUM
    assert (res[0].aux[0], res[0].aux[1]) == (int.from_bytes(bad[-8:-4], "little"), zlib.crc32(data))
    # and the way out: 8 MiB through spng_deflate_batch, trailer appended on the device
    got = s.deflate(data, 1, spng.FORMAT_GZIP)
    assert got[-8:] == z[-8:] and gzip.decompress(got) == data


def test_gzip_mixed_formats_in_one_batch(gpu):
    s = gpu.load()
    rng = np.random.default_rng(2)
    datas = [bytes(rng.integers(0, 7, int(rng.integers(10, 90000)), dtype=np.uint8)) for _ in range(12)]
    fmts = [(spng.FORMAT_ZLIB, spng.FORMAT_IOS, spng.FORMAT_GZIP)[i % 3] for i in range(12)]
    zs = []
    for d, f in zip(datas, fmts):
        if f == spng.FORMAT_ZLIB:
            zs.append(zlib.compress(d, 6))
        elif f == spng.FORMAT_IOS:
            c = zlib.compressobj(6, zlib.DEFLATED, -15); zs.append(c.compress(d) + c.flush())
        else:
            zs.append(gzip.compress(d, 6))
    bad = bytearray(zs[5]); bad[2] = 9; zs[5] = bytes(bad)                       # a gzip member with a bad method
    outs, res = s.inflate_batch([s.to_device(z) for z in zs], [len(d) + 16 for d in datas], fmts)
    for i, (d, z) in enumerate(zip(datas, zs)):
        if i == 5:
            assert res[i].status == spng.E_GZIP_METHOD and res[i].aux[0] == 9
            continue
        assert res[i].status == 0 and res[i].consumed == len(z) and bytes(outs[i][:len(d)].cpu().numpy()) == d, i


@pytest.mark.parametrize("count", [5, 15, 100, 200, 2000, 5000])
def test_mirror_gzip_roundtrip(gpu, count):
    """Compression.Gzip (Compression.swift:29-51)"""
    from swift_png_amd.mirror import Gzip
    s = gpu.load()
    data = np.random.default_rng(count).integers(0, 256, count, dtype=np.uint8).tobytes()
    deflator = Gzip.Deflator(level=7, exponent=15, hint=64 << 10, session=s)
    deflator.push(data, last=True)
    compressed = b""
    while (part := deflator.pull()) is not None:
        compressed += part
    inflator = Gzip.Inflator(session=s)
    assert inflator.push(compressed) is None
    assert inflator.pull() == data
    assert gzip.decompress(compressed) == data


def test_mirror_gzip_micro(gpu):
    """CompressionMicro.Roundtrip / InParts (CompressionMicro.swift:6-28)"""
    from swift_png_amd.mirror import Gzip
    s = gpu.load()
    for data in (b"", bytes([1]), bytes([2, 3]), bytes([4, 5, 6])):
        archive = Gzip.archive(data, level=10, session=s)
        assert Gzip.extract(archive, session=s) == data
    deflator = Gzip.Deflator(level=13, exponent=15, session=s)
    deflator.push(bytes([1]), last=False)
    deflator.push(bytes([2]), last=True)
    archive = b""
    while (part := deflator.pull()) is not None:
        archive += part
    assert Gzip.extract(archive, session=s) == bytes([1, 2])
    with pytest.raises(spng.GzipStreamHeaderError):
        Gzip.extract(b"\x1f\x8c" + archive[2:], session=s)
