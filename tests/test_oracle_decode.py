"""CPU oracle, decode side, pinned against the reference's own goldens and against zlib.

Mirrors the reference's decode suites: PNGIntegrationTests/Roundtripping.swift:7-77 (golden
decode of every PngSuite input), LZ77Tests/Bitstreams.swift:10-59 (bit order), and the error
vocabulary of LZ77.DecompressionError / LZ77.StreamHeaderError.
"""
import hashlib
import json
import struct
import zlib

import numpy as np
import pytest

import pnghelp as ph

TABLE = json.loads((ph.GOLDEN / "pngsuite.json").read_text())


@pytest.mark.parametrize("name", sorted(TABLE))
def test_pngsuite_golden(name):
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    st, storage, _ = ph.orc_decode(png)
    assert st == 0
    assert hashlib.sha256(storage.tobytes()).hexdigest() == TABLE[name]["storage_sha256"]
    rgba = ph.unpack_rgba16(storage, png).astype("<u2")
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == TABLE[name]["rgba16_sha256"]


@pytest.mark.skipif(not ph.have_reference(), reason="reference checkout not mounted")
def test_fixtures_match_reference_files():
    """The committed fixtures are byte-identical to the reference's inputs and goldens."""
    base = ph.REFERENCE / "Sources" / "PNGIntegrationTests"
    for name, rec in TABLE.items():
        sub, fn = name.split("/")
        src = base / "Inputs" / {"common": "Common", "ios": "iOS"}[sub] / fn
        assert src.read_bytes() == (ph.GOLDEN / "pngsuite" / name).read_bytes()
        gold = np.frombuffer((base / "RGBA" / (fn + ".rgba")).read_bytes(), dtype="<u2").reshape(-1, 4)
        if sub == "ios":
            gold = ph.premultiply8(gold)
        assert hashlib.sha256(gold.astype("<u2").tobytes()).hexdigest() == rec["rgba16_sha256"]


def _payloads():
    rng = np.random.default_rng(7)
    text = (b"the quick brown fox jumps over the lazy dog. " * 400)
    ramp = bytes(range(256)) * 64
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    sparse = (rng.integers(0, 256, 50000, dtype=np.uint8) * (rng.random(50000) < 0.05)).astype(np.uint8).tobytes()
    return {"empty": b"", "one": b"a", "text": text, "ramp": ramp, "noise": noise, "sparse": sparse,
            "zeros": bytes(100000)}


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("kind", sorted(_payloads()))
def test_differential_vs_zlib(kind, level):
    data = _payloads()[kind]
    z = zlib.compress(data, level)
    st, out, consumed, _ = ph.orc_inflate(z, 0, cap=len(data) + 16)
    assert (st, out, consumed) == (0, data, len(z))
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    raw = co.compress(data) + co.flush()
    st, out, consumed, _ = ph.orc_inflate(raw, 1, cap=len(data) + 16)
    assert (st, out, consumed) == (0, data, len(raw))


def test_fixed_huffman_blocks():
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    data = b"abcabcabcabc" * 50 + bytes(range(200))
    z = co.compress(data) + co.flush()
    assert ph.orc_inflate(z)[:2] == (0, data)


def test_adler32():
    rng = np.random.default_rng(3)
    for n in (0, 1, 5551, 5552, 5553, 100000):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert ph.oracle().orc_adler32(1, ph._ptr(d) if n else None, n) == zlib.adler32(d.tobytes())


def test_bit_order_kat():
    """LZ77Tests/Bitstreams.swift:12-38: bytes 9e f6 23 read LSB-first; restated through a stored
    block whose header sits at a non-zero bit offset is not possible, so pin bit order with the
    smallest hand-assembled streams instead."""
    # final stored block, LEN=3: bits 1,00 then pad; 03 00 fc ff; payload
    z = bytes([0x78, 0x01, 0x01, 0x03, 0x00, 0xfc, 0xff, 0x9e, 0xf6, 0x23]) + struct.pack(">I", zlib.adler32(b"\x9e\xf6\x23"))
    assert ph.orc_inflate(z)[:2] == (0, b"\x9e\xf6\x23")
    # final fixed block holding literal 'a' (0x61 -> code 0x30+0x61 = 0x91, 8 bits MSB-first) + EOB
    bits = "1" + "10" + format(0x91, "08b") + "0000000"
    by = bytes(int("".join(reversed(bits[i:i + 8].ljust(8, "0"))), 2) for i in range(0, len(bits), 8))
    assert ph.orc_inflate(by, 1)[:2] == (0, b"a")


def test_truncation_wants_more_input():
    data = _payloads()["text"]
    z = zlib.compress(data, 6)
    for cut in (0, 1, 2, 3, 10, len(z) // 2, len(z) - 5, len(z) - 1):
        st, out, _, _ = ph.orc_inflate(z[:cut], 0, cap=len(data) + 16)
        assert st == 1
        assert data.startswith(out)


def test_error_vocabulary():
    good = zlib.compress(b"hello hello hello hello", 9)
    def st(b, fmt=0):
        r = ph.orc_inflate(bytes(b), fmt, cap=4096)
        return r[0], r[3]
    assert st(b"\x77\x01" + good[2:]) == (16, (7, 0))            # invalidCompressionMethod(7)
    assert st(b"\x88\x01" + good[2:]) == (17, (16, 0))           # invalidWindowSize(exponent: 16)
    assert st(b"\x78\x02" + good[2:])[0] == 18                   # invalidCheckBits
    assert st(b"\x78\x20" + good[2:])[0] == 19                   # unexpectedDictionary (0x7820 % 31 == 0)
    bad = bytearray(good); bad[-1] ^= 1
    code, aux = st(bad)
    assert code == 32 and aux == (zlib.adler32(b"hello hello hello hello") ^ 1, zlib.adler32(b"hello hello hello hello"))
    assert st(b"\x78\x01\x07") == (33, (3, 0))                   # invalidBlockTypeCode(3)
    assert st(b"\x78\x01\x01\x03\x00\xfc\xfe\x00\x00\x00") == (34, (3, 0xfefc))   # LEN/NLEN parity
    # dynamic block, HLIT = 31 -> 288 literals
    assert st(bytes([0x05 | (31 << 3) & 0xff, (31 >> 5) | 0, 0, 0, 0, 0, 0, 0]), 1) == (35, (288, 0))
    # dynamic block whose code-length code is empty (all 3-bit lengths zero) -> incomplete tree
    assert st(bytes([0x05, 0, 0, 0, 0, 0, 0, 0, 0, 0]), 1)[0] == 36
    # back-reference before the start of the output: fixed block, length 3 distance 1 as first token
    bits = "1" + "10" + "0000001" + "00000"
    by = bytes(int("".join(reversed(bits[i:i + 8].ljust(8, "0"))), 2) for i in range(0, len(bits), 8)) + b"\0\0"
    assert st(by, 1)[0] == 39


def _dynamic_header(hlit, hdist, clens, seq_bits):
    """Assembles a BFINAL dynamic block header: clens = 19 code-length-code lengths in zigzag
    storage order, seq_bits = already-encoded code length sequence as a bit string."""
    bits = "1" + "01"                                            # BFINAL=1, BTYPE=2 (LSB first: 0,1)
    bits += format(hlit - 257, "05b")[::-1] + format(hdist - 1, "05b")[::-1] + format(len(clens) - 4, "04b")[::-1]
    for c in clens:
        bits += format(c, "03b")[::-1]
    bits += seq_bits
    return bytes(int("".join(reversed(bits[i:i + 8].ljust(8, "0"))), 2) for i in range(0, len(bits), 8)) + bytes(8)


def test_codelength_sequence_errors():
    # code-length code: symbols 16 and 0 both 1 bit (complete): order 16,17,18,0 -> lengths 1,0,0,1
    # canonical: sym 0 -> code '0', sym 16 -> code '1'
    # sequence starting with 16 (repeat previous, nothing to repeat) -> invalidHuffmanCodelengthSequence
    st = ph.orc_inflate(_dynamic_header(257, 1, [1, 0, 0, 1], "1" + "00"), 1, cap=64)
    assert st[0] == 37
    # code-length code: symbols 18 and 1: order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1 -> 18 entries
    clens = [0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]
    # canonical: sym 1 -> '0', sym 18 -> '1'.  258 lengths wanted; 18 with 7 extra bits = 138 zeros, twice = 276 > 258
    seq = "1" + "1111111" + "1" + "1111111"
    st = ph.orc_inflate(_dynamic_header(257, 1, clens, seq), 1, cap=64)
    assert st[0] == 37
    # all 258 lengths zero (138 + 120): literal tree has no codes -> invalidHuffmanTable
    seq = "1" + "1111111" + "1" + format(120 - 11, "07b")[::-1]
    st = ph.orc_inflate(_dynamic_header(257, 1, clens, seq), 1, cap=64)
    assert st[0] == 38


def test_stored_blocks_config1():
    """BASELINE config 1: 256x256 RGBA8, filter=Sub, level 0 (stored blocks)."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (256, 256, 4), dtype=np.uint8)
    rows = np.zeros((256, 1 + 1024), dtype=np.uint8)
    rows[:, 0] = 1
    flat = img.reshape(256, 1024)
    rows[:, 1:5] = flat[:, :4]
    rows[:, 5:] = flat[:, 4:] - flat[:, :-4]
    z = zlib.compress(rows.tobytes(), 0)
    png = ph.Png(256, 256, 8, 6, False, False, z)
    st, storage, _ = ph.orc_decode(png)
    assert st == 0 and (storage == img.reshape(-1)).all()


def test_extraneous_and_incomplete():
    rng = np.random.default_rng(2)
    rows = rng.integers(0, 256, (8, 1 + 32), dtype=np.uint8)
    rows[:, 0] = rng.integers(0, 5, 8)
    png = ph.Png(8, 8, 8, 6, False, False, zlib.compress(rows.tobytes() + b"\x00", 6))
    assert ph.orc_decode(png)[0] == 48                           # extraneousImageData
    short = ph.Png(8, 8, 8, 6, False, False, zlib.compress(rows.tobytes()[:-40], 6))
    st, storage, _ = ph.orc_decode(short)                        # short stream: no error, partial image
    full = ph.orc_decode(ph.Png(8, 8, 8, 6, False, False, zlib.compress(rows.tobytes(), 6)))[1]
    assert st == 0 and (storage[:6 * 32] == full[:6 * 32]).all()
