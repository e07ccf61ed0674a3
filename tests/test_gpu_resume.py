"""spng_inflate_resume_batch: a stream pushed piece by piece (LZ77.Inflator.push, LZ77.Inflator.swift:30-61).
After EVERY push the device reports what the oracle reports for the same prefix of the stream -- status, number of
inflated bytes available, the bytes, error payloads -- while decoding nothing twice but the block a push ends in
(complete blocks go through the parallel pipeline once; the state handed from call to call is a block boundary)."""
import hashlib
import json
import zlib

import numpy as np
import pytest

import pnghelp as ph
import swift_png_amd as spng
from test_gpu_pinflate import make, scanlines

pytestmark = pytest.mark.gpu


class Pusher:
    def __init__(self, s, fmt=spng.FORMAT_ZLIB, cap=1 << 16):
        self.s, self.fmt = s, fmt
        self.d_in, self.n = s.empty(1 << 16), 0
        self.d_out = s.empty(cap)
        self.state = (0, 0, 0, 0)
        self.states = [self.state]

    def push(self, piece):
        s = self.s
        need = self.n + len(piece)
        if self.d_in.numel() < need:
            grown = s.empty(2 * need); grown[:self.n] = self.d_in[:self.n]; self.d_in = grown
        if piece:
            self.d_in[self.n:need] = s.to_device(piece)
        self.n = need
        while True:
            res, state = s.inflate_resume(self.d_in, self.n, self.d_out, self.fmt, self.state)
            if res.status != spng.E_OUTPUT_CAPACITY:
                break
            grown = s.empty(4 * self.d_out.numel()); grown[:self.d_out.numel()] = self.d_out; self.d_out = grown
        self.state = state
        self.states.append(state)
        return res

    def out(self, n):
        return bytes(self.d_out[:n].cpu().numpy())


def pieces(z, sizes):
    at, k = 0, 0
    while at < len(z):
        n = sizes[k % len(sizes)]; k += 1
        yield z[at:at + n]
        at += n


def check_prefixes(s, z, sizes, fmt=spng.FORMAT_ZLIB, every=1):
    p = Pusher(s, fmt)
    seen = b""
    last = None
    for k, piece in enumerate(pieces(z, sizes)):
        res = p.push(piece)
        seen += piece
        if k % every == 0 or len(seen) == len(z):
            st, out, consumed, aux = ph.orc_inflate(seen, fmt, cap=max(1 << 16, 1100 * len(seen)))
            assert res.status == st, (k, len(seen), res.status, st)
            assert res.written == len(out) and p.out(res.written) == out, (k, len(seen))
            if st not in (0, 1):
                assert (res.aux[0], res.aux[1]) == tuple(aux)
            if st == 0:
                assert res.consumed == consumed
        last = res
        if res.status not in (0, 1):
            break
    # the resume point only ever moves forward, in both coordinates
    for a, b in zip(p.states, p.states[1:]):
        assert b[0] >= a[0] and b[1] >= a[1]
    return last, p


@pytest.mark.parametrize("kind", ["zlib1", "zlib6", "zlib9", "noise", "huffonly", "fixed", "flushes", "stored_mix", "zeros"])
def test_resume_matches_oracle_after_every_push(gpu, kind):
    s = gpu.load()
    z = make(kind, 73 * 4096)
    want = zlib.decompress(z)
    last, p = check_prefixes(s, z, [1, 7, 4096, 33333, 100, 65536], every=1)
    assert last.status == 0 and last.written == len(want) and p.out(len(want)) == want
    last, p = check_prefixes(s, z, [997], every=25)
    assert last.status == 0 and p.out(len(want)) == want


def test_resume_swiftpng_made_stream_and_ios_format(gpu):
    s = gpu.load()
    d = scanlines(6, 98 * 4096)
    z = s.deflate(d, 6)
    last, p = check_prefixes(s, z, [5000, 1, 20000], every=1)
    assert last.status == 0 and p.out(len(d)) == d
    raw = s.deflate(d, 4, spng.FORMAT_IOS)
    last, p = check_prefixes(s, raw, [3000], spng.FORMAT_IOS, every=7)
    assert last.status == 0 and p.out(len(d)) == d


def test_resume_errors_surface_at_the_push_that_brings_them(gpu):
    s = gpu.load()
    d = scanlines(7, 49 * 4096)
    z = bytearray(zlib.compress(d, 6))
    z[len(z) // 2] ^= 0x5a                                   # somewhere inside a block
    last, _ = check_prefixes(s, bytes(z), [4096], every=1)
    assert last.status not in (0, 1)
    z = bytearray(zlib.compress(d, 6)); z[-2] ^= 1           # the checksum: only the last push can tell
    last, p = check_prefixes(s, bytes(z), [50000], every=1)
    assert last.status == spng.E_STREAM_CHECKSUM
    assert (last.aux[0], last.aux[1]) == (int.from_bytes(z[-4:], "big"), zlib.adler32(d))
    z = zlib.compress(d, 6)
    last, p = check_prefixes(s, z[:-3], [30000], every=1)    # trailer incomplete: wants more, everything is there
    assert last.status == 1 and last.written == len(d)
    res = p.push(z[-3:])
    assert res.status == 0 and res.consumed == len(z)
    bad = bytearray(z); bad[0] = 0x79                        # header errors at the first push
    assert check_prefixes(s, bytes(bad), [10], every=1)[0].status == spng.E_COMPRESSION_METHOD


def test_resume_large_stream_in_idat_sized_pieces(gpu):
    """16 MiB of scanlines in 64 KiB pieces (an encoder's IDAT chunks): every complete block goes through the pipeline
    exactly once -- the resume point follows the input closely -- and the whole thing takes seconds, not the minutes
    of decoding the stream from its first byte on every push."""
    import time
    s = gpu.load()
    d = scanlines(8, 16 << 20)
    z = zlib.compress(d, 6)
    p = Pusher(s, cap=len(d) + 4096)
    t0 = time.perf_counter()
    res = None
    for piece in pieces(z, [65536]):
        res = p.push(piece)
        assert res.status in (0, 1)
        if res.status == 1:
            assert p.n * 8 - p.state[0] < 8 * 200000         # at most a block or two behind the input
    dt = time.perf_counter() - t0
    assert res.status == 0 and res.written == len(d) and res.consumed == len(z) and p.out(len(d)) == d
    print(f"resume: {len(z) // 65536 + 1} pushes, {dt:.2f} s")
    assert dt < 30


def test_resume_argument_checks(gpu):
    s = gpu.load()
    z = zlib.compress(b"abc" * 100)
    d_in, d_out = s.to_device(z), s.empty(4096)
    with pytest.raises(spng.SpngError):
        s.inflate_resume(d_in, len(z), d_out, spng.FORMAT_ZLIB, (len(z) * 8 + 1, 0))
    with pytest.raises(spng.SpngError):
        s.inflate_resume(d_in, len(z), d_out, spng.FORMAT_GZIP + 1, (0, 0))
    with pytest.raises(spng.SpngError):
        s.inflate_resume(d_in, len(z), d_out, spng.FORMAT_ZLIB, (0, 4097))           # (more output than the buffer holds)
    res, _ = s.inflate_resume(d_in, len(z), d_out, spng.FORMAT_GZIP, (0, 0))         # a zlib stream is no gzip member
    assert res.status == spng.E_GZIP_SIGIL
    res, _ = s.inflate_resume(d_in, len(z), d_out, spng.FORMAT_ZLIB, (0, 0))
    assert res.status == 0 and bytes(d_out[:300].cpu().numpy()) == b"abc" * 100


@pytest.mark.parametrize("name", ["common/oi9n2c16.png", "common/basi3p04.png", "common/basi6a16.png"])
def test_context_push_is_incremental_on_the_image_side(gpu, name):
    """VERDICT r2 "finish f4": PNG.Context.push keeps PNG.Decoder.row / pass (PNG.Decoder.swift:20-21, 88-94, 121-135): every
    push defilters only the scanlines that became complete with it (spng_unfilter_resume_batch).  oi9n2c16 arrives in 229
    IDAT chunks; the interlaced / sub-byte fixtures go through the scratch copy and the scatter.  After EVERY push the raster
    equals the oracle's for the same prefix (a short stream yields an incomplete image, no error), and over all pushes every
    scanline byte went through the defilter exactly once: device work linear in the bytes, not chunks x image."""
    import pnghelp as ph
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    ctx = gpu.PNG.Context((png.width, png.height), png.depth, png.channels, png.interlaced, png.fmt)
    chunks = png.idat_chunks if len(png.idat_chunks) > 1 else [min(7, len(png.idat) - i) for i in range(0, len(png.idat), 7)]
    U = gpu.inflated_size(png.width, png.height, png.depth, png.channels, png.interlaced)
    pos, calls = 0, 0
    for n in chunks:
        pos += n
        ctx.push(png.idat[pos - n:pos])
        calls += 1
        if calls % 16 == 0 or pos >= len(png.idat):
            st, want, _ = ph.orc_decode(png, png.idat[:pos])
            assert ctx.storage == want.tobytes(), (name, calls)
    ctx.push_ancillary_iend()
    assert hashlib.sha256(ctx.storage).hexdigest() == json.loads((ph.GOLDEN / "pngsuite.json").read_text())[name]["storage_sha256"]
    assert ctx.defiltered_total == U, (ctx.defiltered_total, U)


@pytest.mark.parametrize("name", ["common/basi3p04.png", "common/basi6a16.png", "common/basi2c08.png", "common/basi0g01.png",
                                  "common/s05i3p02.png", "common/s09i3p02.png", "common/s33i3p04.png"])
def test_context_push_with_overdraw_equals_the_reference_procedure(gpu, name):
    """PNG.Context.push(data:overdraw: true) (PNG.Context.swift:88-102; PNG.Image.overdraw, PNG.Image.swift:134-183) on the
    device: an interlaced image pushed a few bytes at a time -- after EVERY push the raster equals what the reference's
    scanline-by-scanline procedure leaves (oracle/pixels.py `overdrawn`, with the reference's `base.y & 7` brush rule), for every
    element size 1, 2, 3, 8 and for images smaller than a cell; at the end it is the image."""
    import sys
    import pnghelp as ph
    sys.path.insert(0, str(ph.ROOT / "oracle"))
    import pixels as orc_pixels
    png = ph.parse_png((ph.GOLDEN / "pngsuite" / name).read_bytes())
    assert png.interlaced
    st, final, _ = ph.orc_decode(png)
    assert st == 0
    w, h = png.width, png.height
    elem = len(final) // (w * h)
    final = final.reshape(h, w, elem)
    # scanlines complete after `n` inflated bytes, in the order of PNG.Decoder.push
    ends = []
    off = 0
    volume = png.depth * png.channels
    for (bx, by), (ex, ey) in orc_pixels.ADAM7:
        sw, sh = (w + (1 << ex) - bx - 1) >> ex, (h + (1 << ey) - by - 1) >> ey
        if sw <= 0 or sh <= 0:
            continue
        pitch = (sw * volume + 7) >> 3
        for _ in range(sh):
            off += pitch + 1
            ends.append(off)
    ctx = gpu.PNG.Context((w, h), png.depth, png.channels, True, png.fmt)
    step = max(1, len(png.idat) // 40)
    seen = set()
    for i in range(0, len(png.idat), step):
        ctx.push(png.idat[i:i + step], overdraw=True)
        k = sum(1 for e in ends if e <= ctx._defiltered)
        seen.add(k)
        got = np.frombuffer(ctx.storage, dtype=np.uint8).reshape(h, w, elem)
        assert (got == orc_pixels.overdrawn(final, k)).all(), (name, i, k)
    assert len(seen) > 3 or len(ends) < 8
    ctx.push_ancillary_iend()
    assert ctx.storage == final.tobytes()
    # the flag off: untouched pixels stay as they were (zero)
    plain = gpu.PNG.Context((w, h), png.depth, png.channels, True, png.fmt)
    plain.push(png.idat[:len(png.idat) // 2])
    st, want, _ = ph.orc_decode(png, png.idat[:len(png.idat) // 2])
    assert plain.storage == want.tobytes()


def test_gzip_member_arrives_in_pieces(gpu):
    """Gzip.Inflator.push by pieces (LZ77.InflatorBuffers.swift:139-230): the member's DEFLATE payload goes on from the block
    boundary the previous push stopped at; after every push the available bytes are a prefix of the plain data, and the CRC-32
    trailer is checked by the push that completes the member."""
    import gzip
    rng = np.random.default_rng(4)
    data = (rng.integers(0, 5, 700000, dtype=np.uint8) * (rng.random(700000) < 0.3)).astype(np.uint8).tobytes()
    z = gzip.compress(data, 6)
    inf = gpu.Gzip.Inflator()
    got = b""
    for i in range(0, len(z), 9973):
        status = inf.push(z[i:i + 9973])
        got += inf.pull()
        assert data.startswith(got)
        assert status == (() if i + 9973 < len(z) else None)
    assert got == data
    bad = bytearray(z); bad[-6] ^= 1                                # the CRC-32 field
    inf = gpu.Gzip.Inflator()
    with pytest.raises(gpu.DecompressionError) as e:
        for i in range(0, len(bad), 50000):
            inf.push(bytes(bad[i:i + 50000]))
    assert e.value.status == 32


@pytest.mark.parametrize("mode", [spng.INFLATE_AUTO, spng.INFLATE_SERIAL])
def test_resume_single_push_checks_the_adler32(gpu, mode):
    """A whole zlib stream handed to spng_inflate_resume_batch in ONE call with state {0, 0}: whoever finishes it -- the
    pipeline, or the serial kernel alone (SPNG_INFLATE_SERIAL; the token pool could not be allocated) -- the trailer is
    compared (invalidStreamChecksum(declared:computed:), LZ77.InflatorBuffers.swift:112-130).  ADVICE r3: the serial kernel
    defers the comparison of every caller-resumable stream, and the deferred pass used to skip states that read {0, 0}."""
    s = gpu.load()
    data = scanlines(3, 4096 * 50)
    good = zlib.compress(data, 6)
    bad = bytearray(good); bad[-2] ^= 0x10; bad = bytes(bad)
    s.configure(spng.CFG_INFLATE_MODE, mode)
    try:
        for z in (good, bad):
            p = Pusher(s, cap=len(data) + 64)
            res = p.push(z)
            st, out, consumed, aux = ph.orc_inflate(z, 0, cap=len(data) + 64)
            assert res.status == st, (mode, res.status, st)
            assert res.written == len(out) and p.out(res.written) == out
            if st == spng.E_STREAM_CHECKSUM:
                assert (res.aux[0], res.aux[1]) == tuple(aux)
            assert res.reserved == (0 if mode == spng.INFLATE_SERIAL else 1)
    finally:
        s.configure(spng.CFG_INFLATE_MODE, spng.INFLATE_AUTO)


# ---- spng_deflate_resume_batch: LZ77.Deflator.push(_:last:) with the compressor's state on the device ------------------------
def _push_deflate(s, data, sizes, level, fmt=spng.FORMAT_ZLIB, exponent=15):
    """-> (final stream, [stream bytes available after each push])"""
    t = s.torch
    d_in = s.empty(len(data) + 16)
    d_out = s.empty(int(s.lib.spng_deflate_bound(len(data))) + 64)
    state = t.zeros(int(s.lib.spng_deflate_state_bytes()), dtype=t.uint8, device=s.tdev)
    hstate, at, k, avail, prefix = (0, 0), 0, 0, [], b""
    while True:
        n = sizes[k % len(sizes)]; k += 1
        piece = data[at:at + n]
        if piece:
            d_in[at:at + len(piece)] = s.to_device(piece)
        at += len(piece)
        last = at >= len(data)
        res, hstate = s.deflate_resume(d_in, at, d_out, level, state, last, fmt, exponent, hstate)
        assert res.status == (spng.DONE if last else spng.NEED_MORE_INPUT), (res.status, at, last)
        assert res.written >= (avail[-1] if avail else 0)
        now = bytes(d_out[:res.written].cpu().numpy())
        assert now[:len(prefix)] == prefix                               # what was handed out stays what it was
        prefix = now
        avail.append(res.written)
        if last:
            assert res.consumed == len(data)
            return now, avail


@pytest.mark.parametrize("level", [0, 3, 6, 7, 8, 9, 13])
def test_deflate_pushed_in_pieces_equals_one_shot(gpu, level):
    """The concatenated output of LZ77.Deflator does not depend on how the input was pushed: pieces of 1 byte
    to 200 KB, whole streams from 0 bytes to 3 MB -- crossing the 2047-term blocks of levels 0-7 and the 2047 / 4095 / ... vertex
    blocks of the full search, 2^21 included -- give the oracle's stream, and every push but the last leaves a prefix of it."""
    s = gpu.load()
    rng = np.random.default_rng(level)
    text = b"lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor " * 3000
    cases = [(b"", [10]), (b"ab", [1]), (b"abc", [1]), (scanlines(5, 4096 * 3), [1, 2, 3, 700]), (scanlines(6, 4096 * 30), [5000]),
             (text[:150001], [4097, 33, 70000]), (rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(), [2047, 2048, 1]),
             (scanlines(7, 4096 * 200), [200000, 1000])]
    if level in (6, 9):
        cases.append((scanlines(8, 4096 * 768), [1 << 20, 4096]))      # 3 MiB: the first block at the vertex cap arrives in pieces
    for data, sizes in cases:
        if level == 13 and len(data) > 200000:
            continue
        got, avail = _push_deflate(s, data, sizes, level)
        assert got == ph.orc_deflate(data, level), (len(data), sizes)
        if len(data) > 100000 and len(sizes) > 1:
            assert avail[len(avail) // 2] > 0                           # (bytes came out before the last push)


def test_deflate_pushes_gzip_and_small_window(gpu):
    s = gpu.load()
    data = scanlines(9, 4096 * 40)
    got, _ = _push_deflate(s, data, [30000, 7], 7, spng.FORMAT_GZIP)
    assert got == s.deflate(data, 7, spng.FORMAT_GZIP)
    import gzip as pygzip
    assert pygzip.decompress(got) == data
    got, _ = _push_deflate(s, data, [9999], 9, spng.FORMAT_ZLIB, 9)
    assert got == ph.orc_deflate(data, 9, 0, 9)


def test_mirror_deflator_streams_every_push(gpu):
    """LZ77.Deflator in the mirror no longer buffers until last: every push is a device call, pull() hands out bytes as they come."""
    s = gpu.load()
    data = scanlines(11, 4096 * 120)
    for level in (6, 9):
        d = gpu.LZ77.Deflator(level=level, hint=1 << 12)
        out, pushes, early = b"", 0, 0
        for at in range(0, len(data), 50000):
            d.push(data[at:at + 50000], last=at + 50000 >= len(data))
            pushes += 1
            while (chunk := d.pop()) is not None:
                out += chunk
                early += at + 50000 < len(data)
        while (chunk := d.pull()) is not None:
            out += chunk
        assert out == ph.orc_deflate(data, level)
        assert d.device_calls == pushes and early > 0


def test_resume_inside_a_block_costs_every_byte_once(gpu):
    """VERDICT r5 (missing 1, next 7): a stream that is ONE block, pushed in pieces.  Block-granular resumption decodes the block again
    from its header on every push -- O(n k); with the token position in the state the call goes on where the last one stopped.
    Small enough for the oracle after every push: 1.5 MiB of literals in one fixed-Huffman block (assembled by hand: zlib closes a block every 32 K symbols) in 48 pushes; then 32 MiB in 512 pushes against zlib's own streaming inflate, within 2 x the one-shot time."""
    import time
    import torch
    s = gpu.load()
    rng = np.random.default_rng(3)

    def one_block(n):
        # (zlib closes a block every 32 K symbols whatever it is asked: the block is assembled here -- a final fixed-Huffman block of n
        # literals below 144, eight bits each most significant first, and the end-of-block code)
        vals = rng.integers(0, 144, n, dtype=np.uint8)
        bits = np.concatenate([np.array([1, 1, 0], np.uint8), np.unpackbits((vals + 0x30)[:, None], axis=1, bitorder="big").reshape(-1),
                               np.zeros(7, np.uint8)])
        data = vals.tobytes()
        return data, b"\x78\x01" + np.packbits(bits, bitorder="little").tobytes() + zlib.adler32(data).to_bytes(4, "big")

    data, z = one_block(3 << 19)
    last, p = check_prefixes(s, z, [len(z) // 48 + 1])
    assert last.status == 0 and last.written == len(data)
    inside = [st for st in p.states if st[2]]
    assert len(set(inside)) >= 40 and all(b[2] >= a[2] for a, b in zip(inside, inside[1:])), "the token position moves forward inside the block"
    assert len({st[0] for st in inside}) == 1, "one block"
    # 32 MiB, 512 pushes
    data, z = one_block(32 << 20)
    d_z = s.to_device(z)
    d_out = s.empty(len(data) + 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _ = s.inflate_resume(d_z, len(z), d_out, spng.FORMAT_ZLIB, (0, 0, 0, 0))
    torch.cuda.synchronize()
    one_shot = time.perf_counter() - t0
    assert res.status == 0 and res.written == len(data)
    assert hashlib.sha256(bytes(d_out[:len(data)].cpu().numpy())).digest() == hashlib.sha256(data).digest()
    d_out.zero_()
    step = len(z) // 512 + 1
    ref = zlib.decompressobj()
    avail, state = 0, (0, 0, 0, 0)
    torch.cuda.synchronize()
    spent = 0.0
    for k in range(512):
        n = min(len(z), (k + 1) * step)
        t0 = time.perf_counter()
        res, state = s.inflate_resume(d_z, n, d_out, spng.FORMAT_ZLIB, state)
        spent += time.perf_counter() - t0
        avail += len(ref.decompress(z[k * step:n]))
        assert res.status == (0 if n == len(z) else 1), (k, res.status)
        assert res.written == avail, (k, res.written, avail)          # bytes of the complete tokens so far: zlib's streaming inflate says the same
        if n == len(z):
            break
    assert hashlib.sha256(bytes(d_out[:len(data)].cpu().numpy())).digest() == hashlib.sha256(data).digest()
    print(f"one-block 32 MiB stream: one call {one_shot * 1e3:.1f} ms, 512 pushes {spent * 1e3:.1f} ms")
    assert spent <= 2.0 * one_shot + 0.25, (spent, one_shot)           # (+ the 512 calls' own launch overhead: ~0.4 ms each)
