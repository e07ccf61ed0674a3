"""The kernels of csrc/pinflate2.hip (find -> decode -> scan -> resolve), run on the CPU by the wave emulator of tools/emu
(one fiber per GPU thread; wave builtins are rendezvous points) and compared with zlib's output.  The build container has
no GPU: this is how the pipeline's LOGIC -- token chains, merges, page bookkeeping, replay windows, pointer jumping -- is
checked before a GPU minute is spent, and kept checked.  Timing, bank conflicts and memory ordering are not modelled; the
`-m gpu` tests remain the parity tests proper."""
import os
import shutil
import subprocess
from pathlib import Path
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("emu") / "emu_pinflate2"
    subprocess.run(["g++", "-O1", "-std=c++17", "-DSPNG_EMU", "-I" + os.path.join(ROOT, "tools", "emu"), "-x", "c++", "-fpermissive",
                    "-Wno-attributes", "-w", "-o", str(out), os.path.join(ROOT, "tools", "emu", "emu_pinflate2.cpp")],
                   check=True, capture_output=True, timeout=600)
    return out


def scanlines(seed, n):
    rng = np.random.default_rng(seed)
    a = rng.integers(-3, 4, n).astype(np.int16)
    a[rng.random(n) < 0.6] = 0
    rows = a.astype(np.uint8).reshape(-1, 512)
    rows[::13] = rng.integers(0, 256, (len(rows[::13]), 512), dtype=np.uint8)
    rows[::4, 0] = 1
    return rows.tobytes()


def deflate(data, level=6, wbits=15, strategy=0):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return c.compress(data) + c.flush()


def cases():
    rng = np.random.default_rng(3)
    rows = scanlines(1, 96 * 512)
    yield "zlib6", deflate(rows, 6), rows, 0
    yield "zlib1", deflate(rows, 1), rows, 0                        # (long unmerged chains: a lane replays over several windows)
    yield "raw9", deflate(rows, 9, -15), rows, 1
    yield "fixed", deflate(rows[:20000], 6, 15, zlib.Z_FIXED), rows[:20000], 0
    noise = rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    yield "stored", deflate(noise, 0), noise, 0
    yield "huffonly", deflate(noise, 6, 15, zlib.Z_HUFFMAN_ONLY), noise, 0
    t16 = rng.integers(0, 16, 60000, dtype=np.uint8).tobytes()
    yield "text16", deflate(t16, 6, 15, zlib.Z_HUFFMAN_ONLY), t16, 0  # 4-bit codes: every step decodes two literals
    t3 = rng.integers(0, 3, 60000, dtype=np.uint8).tobytes()
    yield "text3", deflate(t3, 6, 15, zlib.Z_HUFFMAN_ONLY), t3, 0     # 1- and 2-bit codes: pairs in short subsequences
    yield "zeros", deflate(bytes(200000), 6), bytes(200000), 0     # one-bit codes: short subsequences
    yield "period4", deflate(bytes([1, 2, 3, 255]) * 20000, 9), bytes([1, 2, 3, 255]) * 20000, 0
    # Fibonacci-like byte frequencies: zlib's length-limited trees reach 15 bits, codes longer than the 9-bit root index share
    # root prefixes (second-level tables of several depths; round 4 builds the root table in code order and transposes it)
    deep = []
    a, b = 1, 1
    for sym in range(24):
        deep.append(bytes([sym * 7 % 256]) * a)
        a, b = b, a + b
    deep = b"".join(deep)[:200000]
    deep = bytes(np.random.default_rng(5).permutation(np.frombuffer(deep, dtype=np.uint8)))
    yield "deeptree", deflate(deep, 6, 15, zlib.Z_HUFFMAN_ONLY), deep, 0
    # the same for the distance code: matches at Fibonacci-weighted distance decades over a literal background
    rng2 = np.random.default_rng(6)
    buf = bytearray(rng2.integers(0, 256, 70000, dtype=np.uint8).tobytes())
    dists = [1, 2, 3, 4, 6, 9, 14, 22, 35, 55, 90, 140, 230, 370, 600, 960, 1500, 2500, 4000, 6500, 10000, 16000, 26000, 32000]
    w = [max(1, int(1.6 ** k)) for k in range(len(dists))][::-1]
    pick = rng2.choice(len(dists), 6000, p=np.array(w) / sum(w))
    at = 33000
    for k in pick:
        d = dists[k]
        n = int(rng2.integers(4, 12))
        if at + n >= len(buf):
            break
        for i in range(n):
            buf[at + i] = buf[at + i - d]
        at += n + int(rng2.integers(0, 3))
    deepd = bytes(buf[:at])
    yield "deepdist", deflate(deepd, 9), deepd, 0
    co = zlib.compressobj(6)
    parts = []
    for i in range(0, len(rows), 9000):
        parts.append(co.compress(rows[i:i + 9000]))
        parts.append(co.flush(zlib.Z_FULL_FLUSH if (i // 9000) % 2 else zlib.Z_SYNC_FLUSH))
    yield "flushes", b"".join(parts) + co.flush(), rows, 0


CASES = list(cases())


@pytest.mark.parametrize("segment", [1 << 20, 4096])
@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_emulated_pipeline_matches_zlib(emu, tmp_path, name, segment):
    _, z, raw, fmt = next(c for c in CASES if c[0] == name)
    assert zlib.decompress(z, -15 if fmt else 15) == raw
    (tmp_path / "z").write_bytes(z)
    (tmp_path / "raw").write_bytes(raw)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), str(fmt), str(segment)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-300:])
    assert r.stdout.startswith("ok:")


def test_emulated_decode_reaches_its_rare_paths(tmp_path):
    """Round 5's marks (a back-reference marks its start and the bit behind it) have corner cases of their own: a reference that
    starts on the last bit of a subsequence (nothing behind it to mark) or of a bitmap word (the second mark goes into the next
    word), a chain of round 1 that lands on a second mark (not a token start), a pair of literals cut at a subsequence's end, and
    blocks with a one-bit code, which keep the kind masks.  A build with counters (-DSPNG_EMU_COV) says how often each ran over
    the cases above: every one of them must have, with exact output."""
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    out = tmp_path / "emu_cov"
    subprocess.run(["g++", "-O1", "-std=c++17", "-DSPNG_EMU", "-DSPNG_EMU_COV", "-I" + os.path.join(ROOT, "tools", "emu"), "-x", "c++", "-fpermissive",
                    "-Wno-attributes", "-w", "-o", str(out), os.path.join(ROOT, "tools", "emu", "emu_pinflate2.cpp")],
                   check=True, capture_output=True, timeout=600)
    total = [0] * 6
    for name in ("zlib6", "zlib1", "text16", "text3", "deepdist"):
        _, z, raw, fmt = next(c for c in CASES if c[0] == name)
        (tmp_path / "z").write_bytes(z)
        (tmp_path / "raw").write_bytes(raw)
        r = subprocess.run([str(out), str(tmp_path / "z"), str(tmp_path / "raw"), str(fmt), "4096"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), (name, r.stdout[-300:], r.stderr[-300:])
        cov = [ln for ln in r.stderr.splitlines() if ln.startswith("COV")]
        assert cov, r.stderr[-300:]
        total = [a + int(b) for a, b in zip(total, cov[-1].split()[1:])]
    names = ("reference on a subsequence's last bit", "reference on a word's last bit", "chunks with reference marks", "chunks with kind masks",
             "pairs cut at a subsequence's end", "landings on a second mark")
    assert all(total), dict(zip(names, total))


def test_emulated_pipeline_swiftpng_shaped_stream(emu, tmp_path):
    """the oracle's own level-6 deflate: a dynamic block every <= 2047 tokens, as PNG.Image.compress makes them"""
    import pnghelp as ph
    rows = scanlines(2, 64 * 512)
    z = ph.orc_deflate(rows, 6)
    assert zlib.decompress(z) == rows
    (tmp_path / "z").write_bytes(z)
    (tmp_path / "raw").write_bytes(rows)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "8192"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])


def test_emulated_pipeline_with_a_small_page_table(emu, tmp_path):
    """stored data hides every later segment start: the first segment decodes the whole stream and borrows the page-table
    entries of the start-less segments behind it"""
    rng = np.random.default_rng(9)
    noise = rng.integers(0, 256, 150000, dtype=np.uint8).tobytes()
    z = deflate(noise, 0)
    (tmp_path / "z").write_bytes(z)
    (tmp_path / "raw").write_bytes(noise)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "8192"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, EMU_PTCAP="2"))
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])


@pytest.mark.parametrize("kind", ["stored", "fixed"])
def test_emulated_resume_point_moves_past_stored_and_fixed_blocks(emu, tmp_path, kind):
    """a stream of nothing but stored (or fixed) blocks, cut inside a block: no dynamic header for the search to find, yet the first
    segment's wave takes every complete block and the serial kernel is told to start at the LAST block boundary (exit 4) -- pushes
    of such a stream are linear too.  Then the push that brings the rest goes on from that point."""
    import re
    rng = np.random.default_rng(21)
    if kind == "stored":
        data = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
        z = zlib.compress(data, 0)
        least = 65535                                        # (one whole stored block at least lies in front of the cut)
    else:
        data = bytes((np.cumsum(rng.integers(-2, 3, 120000)) % 256).astype(np.uint8))
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        parts = []
        for i in range(0, len(data), 20000):
            parts.append(co.compress(data[i:i + 20000]))
            parts.append(co.flush(zlib.Z_FULL_FLUSH))
        z = b"".join(parts) + co.flush()
        least = 60000
    assert zlib.decompress(z) == data
    cut = len(z) * 2 // 3
    (tmp_path / "z").write_bytes(z[:cut])
    (tmp_path / "raw").write_bytes(data)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "65536"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 4, (r.stdout[-300:], r.stderr[-300:])
    m = re.search(r"starts at bit (\d+) with (\d+) bytes in front", r.stdout)
    bit, pos = int(m.group(1)), int(m.group(2))
    assert pos >= least and bit > 8 * 1000, (bit, pos)
    # the next push: the whole stream, resumed at that point with the bytes so far in place
    (tmp_path / "z").write_bytes(z)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "65536", "4096", str(bit), str(pos)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])


def test_emulated_pipeline_keeps_the_prefix_of_a_truncated_stream(emu, tmp_path):
    """the input ends in the middle of a block: everything in front of that block is decoded by the pipeline, and the
    serial kernel is told to start at the block boundary (exit 4), not at byte 0"""
    rows = scanlines(5, 96 * 512)
    co = zlib.compressobj(6)
    z = b"".join(co.compress(rows[i:i + 6000]) + co.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(rows), 6000)) + co.flush()
    cut = z[:len(z) * 3 // 4]
    (tmp_path / "z").write_bytes(cut)
    (tmp_path / "raw").write_bytes(rows)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "4096"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 4, (r.returncode, r.stdout[-300:], r.stderr[-300:])
    bit, pos = [int(x) for x in r.stdout.split() if x.isdigit()][:2]
    assert pos > len(rows) // 2 and bit // 8 > len(cut) // 2 and bit // 8 <= len(cut)


def test_emulated_pipeline_reports_a_bad_checksum_itself(emu, tmp_path):
    """every block decodes, the Adler-32 trailer is wrong: invalidStreamChecksum(declared, computed) comes from the pipeline
    (exit 5), with the reference's payload, instead of a second, serial decode of the whole stream"""
    rows = scanlines(6, 32 * 512)
    z = bytearray(deflate(rows, 6))
    z[-1] ^= 0x01
    (tmp_path / "z").write_bytes(bytes(z))
    (tmp_path / "raw").write_bytes(rows)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "1048576"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 5, (r.returncode, r.stdout[-300:])
    declared, computed = int.from_bytes(bytes(z[-4:]), "big"), zlib.adler32(rows)
    assert f"error 32 aux {declared:x} {computed:x} written {len(rows)} consumed {len(z)}" in r.stdout, r.stdout


def test_emulated_pipeline_stops_in_front_of_a_corrupt_block(emu, tmp_path):
    """a flipped bit in the middle: the pipeline must not report success; what it decoded in front of the damaged block is
    right (exit 4), or it leaves everything to the serial kernel (exit 3)"""
    rows = scanlines(5, 32 * 512)
    z = bytearray(deflate(rows, 6))
    z[len(z) // 2] ^= 0x10
    (tmp_path / "z").write_bytes(bytes(z))
    (tmp_path / "raw").write_bytes(rows)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "1048576"], capture_output=True, text=True, timeout=600)
    assert r.returncode in (3, 4, 5), (r.returncode, r.stdout[-300:])


def part_cases():
    """streams of several blocks and >= 150 KB of output: the chain can be cut into parts"""
    rng = np.random.default_rng(11)
    rows = scanlines(7, 56 * 4096)
    yield "zlib6", deflate(rows, 6), rows, 4096
    t16 = rng.integers(0, 16, 150000, dtype=np.uint8).tobytes()
    yield "text16", deflate(t16, 6, 15, zlib.Z_HUFFMAN_ONLY), t16, 4096
    # runs that reach back across part boundaries: zeros and a 4-byte period, a block every 40 KB, tiny compressed blocks
    def flushed(data, step):
        co = zlib.compressobj(6)
        out = b""
        for i in range(0, len(data), step):
            out += co.compress(data[i:i + step]) + co.flush(zlib.Z_FULL_FLUSH if (i // step) % 2 else zlib.Z_SYNC_FLUSH)
        return out + co.flush()
    z0 = bytes(200000)
    yield "zeros", flushed(z0, 20000), z0, 256
    p4 = bytes([1, 2, 3, 255]) * 40000
    yield "period4", flushed(p4, 20000), p4, 256
    mixed = bytes(60000) + rows[:80000] + bytes([9, 8, 7]) * 20000 + rows[80000:120000]
    yield "mixed", flushed(mixed, 15000), mixed, 512


PART_CASES = list(part_cases())


@pytest.mark.parametrize("tile", [4096, 8192])
@pytest.mark.parametrize("parts", [4])
@pytest.mark.parametrize("name", [c[0] for c in PART_CASES])
def test_emulated_pipeline_resolves_a_stream_in_parts(emu, tmp_path, name, parts, tile):
    """several workgroups per stream (api.hip: batches of few streams): the chain cut into parts, those behind the first resolved
    to symbols with markers for what lies in front of them, the windows handed from part to part, symbols -> bytes, one verdict
    (Adler-32 over all parts) -- the same bytes and the same result as one workgroup gives.  The marker parts have two geometries
    (4 KiB tiles, two workgroups per CU, when a batch has more of them than CUs; 8 KiB tiles otherwise): both."""
    _, z, raw, segment = next(c for c in PART_CASES if c[0] == name)
    assert zlib.decompress(z) == raw
    (tmp_path / "z").write_bytes(z)
    (tmp_path / "raw").write_bytes(raw)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", str(segment)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, EMU_PARTS=str(parts), EMU_MARK_TILE=str(tile)))
    assert r.returncode == 0, (name, r.stdout[-300:], r.stderr[-300:])
    made = int(r.stdout.split("parts:")[1].split()[0])
    assert 2 <= made <= parts, r.stdout
    # the first part with one and a half shares (what batches whose marker parts share their CUs get)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", str(segment)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, EMU_PARTS=str(parts), EMU_MARK_TILE=str(tile), EMU_FIRST_PART_SHARE="1"))
    assert r.returncode == 0 and 2 <= int(r.stdout.split("parts:")[1].split()[0]) <= parts, (name, r.stdout[-300:], r.stderr[-300:])
    # a wrong Adler-32 is the pipeline's own verdict in parts too
    bad = bytearray(z); bad[-1] ^= 0x20
    (tmp_path / "z").write_bytes(bytes(bad))
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", str(segment)], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, EMU_PARTS=str(parts), EMU_MARK_TILE=str(tile)))
    assert r.returncode == 5 and "error" in r.stdout, (r.returncode, r.stdout[-300:])


@pytest.mark.parametrize("seed", [3, 17, 29, 41, 58, 77, 5003, 7011, 9040, 11013, 15007, 17020, 19033, 30010])
def test_emulated_pipeline_random_streams(emu, tmp_path, seed):
    """tools/emu/fuzz_pinflate.py: random data x zlib parameters x flush pattern x segment length x resolve parts (hundreds of
    seeds were run when the pipeline changed; a few stay in the suite)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_pinflate", str(Path(__file__).resolve().parent.parent / "tools" / "emu" / "fuzz_pinflate.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rc, what = fz.run(str(emu), seed, str(tmp_path))
    assert rc in (0, 3, 4), what


def test_emulated_pipeline_retry_pass_after_a_dry_pool(emu, tmp_path):
    """the token pool runs dry under a segment: the stream is marked for the retry pass (PSEG_NOPAGE -> st.pass = 1), which --
    given pages -- decodes and resolves it; the result is the pipeline's (NOPAGE had become unreachable and the
    retry launches did nothing)"""
    rows = scanlines(21, 600 * 512)
    z = deflate(rows, 6)
    (tmp_path / "z").write_bytes(z)
    (tmp_path / "raw").write_bytes(rows)
    r = subprocess.run([str(emu), str(tmp_path / "z"), str(tmp_path / "raw"), "0", "8192", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, EMU_RETRY_PAGES="64", EMU_PTCAP="64"))
    assert "first pass: ok 0 pass 1" in r.stdout, (r.stdout[-400:], r.stderr[-300:])
    assert "retry pass: ok 1 done 1" in r.stdout, (r.stdout[-400:], r.stderr[-300:])
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])
