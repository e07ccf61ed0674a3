"""The deflate kernels of csrc/deflate.hip -- levels >= 8: dfl2_begin -> dfl3_search -> dfl2_advance -> dfl2_parse, round by round;
levels 0-7: dfl3_begin -> dfl3_search_fast -> dfl3_advance -> dfl4_walk / dfl4_block / dfl4_scan / dfl4_place (one-shot streams) or
dfl3_parse (streams that arrive in pieces), and the one-kernel form, deflate_kernel -- run on the CPU by the wave emulator of tools/emu and compared with the oracle's stream bit for bit.  The build container has no GPU:
this is how the LOGIC of the device deflater -- hash chains and candidate records, the skip rule, offer tables, the shortest-path
passes, trees, the bit writer -- is checked before a GPU minute is spent.  The emulator compiles a COPY of the source prepared by
tools/emu/prep_deflate.py (launches blanked, a few meetings of the wave where the source relies on lock-step execution); timing
and memory ordering are not modelled, the `-m gpu` tests remain the parity tests proper."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))

import pnghelp as ph  # noqa: E402


def build(d, round_vertices=0, waves=4):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    import prep_deflate
    inc = d / "deflate_emu.inc"
    inc.write_text(prep_deflate.prepare(open(os.path.join(ROOT, "swift_png_amd", "csrc", "deflate.hip")).read(), round_vertices))
    out = d / "emu_deflate2"
    # (four waves per search workgroup instead of the product's sixteen: an inserter and three searchers -- the same protocol, a
    #  quarter of the fibers; test_emulated_search_workgroup_of_sixteen_waves runs the product's shape)
    subprocess.run(["g++", "-O1", "-std=c++17", "-DSPNG_EMU", f"-DSPNG_D3_WAVES={waves}", f'-DEMU_DEFLATE_SRC="{inc}"', "-I" + os.path.join(ROOT, "tools", "emu"),
                    "-I" + os.path.join(ROOT, "swift_png_amd", "csrc"), "-x", "c++", "-fpermissive", "-Wno-attributes", "-w", "-o", str(out),
                    os.path.join(ROOT, "tools", "emu", "emu_deflate2.cpp")], check=True, capture_output=True, timeout=600)
    return out


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    return build(tmp_path_factory.mktemp("emu_deflate"))


@pytest.fixture(scope="module")
def emu_small_rounds(tmp_path_factory):
    """rounds of 2^14 vertices: blocks of 2047 + 4095 + 8191 terms fill the first round, the block of 16383 the second"""
    return build(tmp_path_factory.mktemp("emu_deflate_rv"), 1 << 14)


def inputs():
    rng = np.random.default_rng(2)
    walk = bytes((np.cumsum(rng.integers(-2, 3, 20000)) % 256).astype(np.uint8))       # four blocks (2047, 4095, 8191, the rest)
    runs = b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400)) for _ in range(60))   # runs > 100: the skip rule; > 66: beyond the offer table
    text = (b"It was the best of times, it was the worst of times, it was the age of wisdom, it was the age of foolishness, " * 60)[:6000]
    noise = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    mixed = noise[:2500] + bytes(3000) + text[:2000] + rng.integers(0, 4, 3000, dtype=np.uint8).tobytes()
    y, x = np.mgrid[0:40, 0:96]
    img = np.stack([(x * 2 + y) % 256, (x + y * 3) % 256, (x * y) % 256, np.full_like(x, 255)], axis=-1).astype(np.uint8)
    rows = ph.orc_filter(img.reshape(-1), 96, 40, 8, 4, False)                          # filtered scanlines of a smooth RGBA image
    return {"walk": walk, "runs": runs, "text": text, "noise": noise, "mixed": mixed, "rows": rows,
            "zeros": bytes(3000), "two": b"ab", "three": b"abc", "empty": b""}


INPUTS = inputs()
CASES = [("walk", 9, 3), ("runs", 8, 3), ("runs", 9, 2), ("text", 9, 1), ("text", 10, 5), ("noise", 9, 3), ("mixed", 9, 3), ("mixed", 13, 3),
         ("rows", 9, 3), ("rows", 10, 4), ("zeros", 9, 3), ("two", 9, 3), ("three", 9, 3), ("empty", 9, 3)]


@pytest.mark.parametrize("name,level,chunks", CASES)
def test_emulated_level8_rounds_match_the_oracle(emu, tmp_path, name, level, chunks):
    """chunks: search workgroups per stream and round (the chunk boundaries re-warm the hash window: any number gives the same records)"""
    data = INPUTS[name]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", str(chunks)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (name, level, r.stdout[-300:], r.stderr[-300:])
    assert r.stdout.startswith("ok:")


@pytest.mark.parametrize("name,level", [("walk30k", 9), ("mixed30k", 9), ("walk30k", 8)])
def test_emulated_rounds_carry_their_state(emu_small_rounds, tmp_path, name, level):
    """two rounds (the product needs > 2 MiB for that): bit writer, depths, block limit and the search's own cursor carried from
    round to round, the second round's candidates in the second set of records"""
    rng = np.random.default_rng(11)
    data = {"walk30k": bytes((np.cumsum(rng.integers(-2, 3, 30000)) % 256).astype(np.uint8)),
            "mixed30k": (INPUTS["mixed"] * 3)[:30000]}[name]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu_small_rounds), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", "2"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (name, level, r.stdout[-300:], r.stderr[-300:])
    assert r.stdout.strip().endswith("in 2 rounds"), r.stdout


@pytest.mark.parametrize("level", [0, 1, 6, 7])
@pytest.mark.parametrize("name", ["walk", "text", "zeros", "noise", "rows", "mixed", "empty", "two"])
def test_emulated_greedy_lazy_kernel_matches_the_oracle(emu, tmp_path, name, level):
    """levels 0-7, the two-kernel form (round 5): the chip-wide search leaves one answer per position, a parse wave walks them
    (level 6 is what the bench's swift-png-made inputs are made with)"""
    data = INPUTS[name]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (name, level, r.stdout[-300:], r.stderr[-300:])
    assert "blocks side by side" in r.stdout


@pytest.mark.parametrize("level,name,fmt", [(6, "rows", "1"), (0, "text", "1"), (7, "mixed", "0"), (4, "two", "1"), (6, "empty", "1")])
def test_emulated_two_wave_parse_and_raw_format(emu, emu_small_rounds, tmp_path, level, name, fmt):
    """the two forms of the levels 0-7 parse on the same input: the blocks side by side (one-shot streams) and the two-wave
    parser + writer (the form of streams that arrive in pieces: EMU_TWO_WAVE), zlib and raw (LZ77.Format.ios) -- the same bytes"""
    data = (INPUTS[name] * 8)[:40000] if len(INPUTS[name]) > 2 else INPUTS[name]
    want = ph.orc_deflate(data, level, int(fmt))
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    for exe in (emu, emu_small_rounds):
        for env in ({}, {"EMU_TWO_WAVE": "1"}):
            r = subprocess.run([str(exe), str(tmp_path / "in"), str(tmp_path / "want"), str(level), fmt, "2"], capture_output=True, text=True,
                               timeout=900, env=dict(os.environ, **env))
            assert r.returncode == 0, (name, level, env, r.stdout[-300:], r.stderr[-300:])
            assert ("blocks side by side" in r.stdout) == (not env)


@pytest.mark.parametrize("level", [1, 6])
@pytest.mark.parametrize("name", ["walk", "rows", "mixed", "two"])
def test_emulated_one_kernel_greedy_lazy_form(emu, tmp_path, name, level):
    """SPNG_DEFLATE_ONE_KERNEL at levels 0-7: `deflate_kernel`, one wave per stream does everything (the fallback when the search
    records find no memory)"""
    data = INPUTS[name]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, EMU_ONE_KERNEL="1"))
    assert r.returncode == 0, (name, level, r.stdout[-300:], r.stderr[-300:])
    assert "greedy / lazy kernel" in r.stdout


@pytest.mark.parametrize("level,name", [(0, "walk"), (4, "rows"), (6, "mixed"), (7, "runs"), (6, "noise")])
def test_emulated_greedy_lazy_rounds_carry_their_state(emu_small_rounds, tmp_path, level, name):
    """rounds of 2^14 positions: parse position, queued terms and bit writer carried from round to round in the D1State, the
    search a round ahead with its own cursor, the answers of a round in the set of its parity, a lazy look at the position
    behind a round's last"""
    data = (INPUTS[name] * 20)[:50000]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu_small_rounds), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", "2"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (name, level, r.stdout[-300:], r.stderr[-300:])
    assert int(r.stdout.split(" in ")[1].split()[0]) >= 2, r.stdout


@pytest.mark.parametrize("level,cuts", [(6, (100, 5000, 5001, 12000)), (1, (1, 2, 3, 300, 19999)), (7, (263, 264, 265, 266, 530)), (4, (7000,))])
def test_emulated_greedy_lazy_pushes_give_the_one_shot_stream(emu_small_rounds, tmp_path, level, cuts):
    """spng_deflate_resume_batch at levels 0-7 through the two kernels: a call per piece with `more` set -- only positions whose
    look-ahead (and that of the position behind them) is complete are parsed --, the search cursor and the Adler sums of the
    search workgroups kept between the calls: the bytes are those of one call over everything"""
    data = INPUTS["walk"]
    want = ph.orc_deflate(data, level)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu_small_rounds), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", "2"] + [str(c) for c in cuts],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (cuts, r.stdout[-300:], r.stderr[-300:])


def test_emulated_search_window_wraps_its_ring(emu, tmp_path):
    """the search's window is a ring of 36864 positions in LDS: 90 KB in one chunk wrap it twice, with matches 32760 bytes back
    (the far end of the window, read through the mirrored bytes behind the ring's end) -- level 9 records and level 6 answers"""
    rng = np.random.default_rng(5)
    blk = rng.integers(0, 256, 32760, dtype=np.uint8).tobytes()
    far = blk + blk[:20000] + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes() + blk[3000:30000] + blk[:100] * 50
    for level in (9, 6):
        want = ph.orc_deflate(far, level)
        (tmp_path / "in").write_bytes(far)
        (tmp_path / "want").write_bytes(want)
        r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", "1"], capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (level, r.stdout[-300:], r.stderr[-300:])
        # (round 6: the inserter takes a bucket's old head and leaves its position there in ONE atomic exchange where the device's LDS
        # orders the lanes of an address -- the default here; the read-back form of round 5, which devices that do not keep: the same)
        r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", "1"], capture_output=True, text=True, timeout=1800,
                           env=dict(os.environ, EMU_D3_READBACK="1"))
        assert r.returncode == 0, (level, "read-back inserter", r.stdout[-300:], r.stderr[-300:])


def test_emulated_search_workgroup_of_sixteen_waves(tmp_path_factory, tmp_path):
    """the product's shape -- an inserter and fifteen searchers per workgroup -- on the filtered rows of a small interlaced image
    (the input on which the first GPU run of the kernel showed a store forwarded over the heads' read-back)"""
    emu16 = build(tmp_path_factory.mktemp("emu_deflate16"), waves=16)
    rng = np.random.default_rng(77)
    w, h = 37, 23
    storage = ((np.arange(w * h * 4) * 7 + rng.integers(0, 2, w * h * 4)) % 256).astype(np.uint8)
    rows = ph.orc_filter(storage, w, h, 8, 4, True)
    for level, chunks in ((9, "64"), (6, "5")):
        want = ph.orc_deflate(rows, level)
        (tmp_path / "in").write_bytes(rows)
        (tmp_path / "want").write_bytes(want)
        r = subprocess.run([str(emu16), str(tmp_path / "in"), str(tmp_path / "want"), str(level), "0", chunks], capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (level, r.stdout[-300:], r.stderr[-300:])


def test_emulated_level6_over_many_blocks(emu, tmp_path):
    """96 KiB of scanline-like bytes: some twenty blocks of 2047 terms, as PNG.Image.compress makes them"""
    rng = np.random.default_rng(12)
    a = rng.integers(-3, 4, 96 * 1024).astype(np.int16)
    a[rng.random(len(a)) < 0.6] = 0
    data = a.astype(np.uint8).tobytes()
    want = ph.orc_deflate(data, 6)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), "6", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-300:], r.stderr[-300:])


@pytest.mark.parametrize("cuts", [(100, 5000, 5001, 12000), (1, 2, 3, 300, 19999), (7000,)])
def test_emulated_pushes_give_the_one_shot_stream(emu, tmp_path, cuts):
    """spng_deflate_resume_batch as LZ77.Deflator.push(_:last:) drives it: a call per piece with `more` set (only blocks whose every
    vertex sees its whole look-ahead are taken), the state kept between calls -- the bytes are those of one call over everything"""
    data = INPUTS["walk"]
    want = ph.orc_deflate(data, 9)
    (tmp_path / "in").write_bytes(data)
    (tmp_path / "want").write_bytes(want)
    r = subprocess.run([str(emu), str(tmp_path / "in"), str(tmp_path / "want"), "9", "0", "3"] + [str(c) for c in cuts], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (cuts, r.stdout[-300:], r.stderr[-300:])


def test_prepared_copy_only_differs_where_it_says(tmp_path):
    """the copy the emulator compiles = the product source but for the documented replacements"""
    import difflib
    import prep_deflate
    src = open(os.path.join(ROOT, "swift_png_amd", "csrc", "deflate.hip")).read()
    out = prep_deflate.prepare(src)
    changed = [l for l in difflib.unified_diff(src.splitlines(), out.splitlines(), lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
    assert 0 < len(changed) < 80, len(changed)
    for l in changed:
        if l[1:].strip() == "}":                        # (a closing brace between two blanked launches: the diff's alignment, not a change)
            continue
        assert any(k in l for k in ("<<<", "(void)0", "s_waitcnt", "__builtin_amdgcn_fence", 'asm volatile("" ::: "memory")', "emu_bb", "g.bbase[q]", "b.nacc", "hipMemsetAsync", "dfl2_", "dfl3_", "dfl4_", "deflate_")), l
