"""Fuzz of the inflate pipeline's kernels in the CPU wave emulator (tools/emu): random data of eight kinds x random zlib
level / strategy / memLevel / flush pattern x random segment length x random number of resolve parts; the emulated pipeline
must give zlib's bytes (exit 0), or decline / stop in front of a block with a correct prefix (3 / 4: the serial kernel's work).

    g++ -O2 -std=c++17 -DSPNG_EMU -Itools/emu -x c++ -fpermissive -Wno-attributes -o /tmp/emu_pinflate2 tools/emu/emu_pinflate2.cpp
    python tools/emu/fuzz_pinflate.py /tmp/emu_pinflate2 <first seed> <count>
"""
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

def gen(rng):
    kind = rng.integers(0, 8)
    n = int(rng.integers(2000, 300000))
    if kind == 0:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 1:
        d = rng.integers(0, int(rng.integers(2, 40)), n, dtype=np.uint8).tobytes()
    elif kind == 2:
        a = rng.integers(-3, 4, n).astype(np.int16); a[rng.random(n) < rng.random()] = 0; d = a.astype(np.uint8).tobytes()
    elif kind == 3:
        per = rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8).tobytes(); d = (per * (n // len(per) + 1))[:n]
    elif kind == 4:
        d = bytes(n)
    elif kind == 5:
        parts = []
        while sum(map(len, parts)) < n:
            m = int(rng.integers(100, 20000)); t = rng.integers(0, 4)
            parts.append(bytes(m) if t == 0 else rng.integers(0, 256, m, dtype=np.uint8).tobytes() if t == 1 else
                         (rng.integers(0, 4, m, dtype=np.uint8)).tobytes() if t == 2 else bytes([int(rng.integers(0,256))]) * m)
        d = b"".join(parts)[:n]
    elif kind == 6:
        base = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(); d = (base * (n // 5000 + 1))[:n]
    else:
        a = np.cumsum(rng.integers(-2, 3, n)).astype(np.uint8); d = a.tobytes()
    level = int(rng.integers(0, 10)); strat = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
    co = zlib.compressobj(level, zlib.DEFLATED, 15, int(rng.integers(1, 10)), strat)
    out = b""
    if rng.random() < 0.5:
        step = int(rng.integers(1000, 60000))
        for i in range(0, len(d), step):
            out += co.compress(d[i:i + step])
            if rng.random() < 0.7: out += co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK if hasattr(zlib,'Z_BLOCK') else zlib.Z_SYNC_FLUSH])))
        out += co.flush()
    else:
        out = co.compress(d) + co.flush()
    return d, out
def run(emu, seed, tmp):
    """-> (exit code, description)"""
    rng = np.random.default_rng(seed)
    d, z = gen(rng)
    assert zlib.decompress(z) == d
    if os.environ.get("FUZZ_DAMAGE"):
        # a flipped bit or a cut: the pipeline may decline (3), keep a correct prefix (4) or report the checksum itself (5)
        b = bytearray(z)
        if rng.random() < 0.3:
            b = b[:int(rng.integers(2, len(b)))]
        else:
            at = int(rng.integers(2, len(b)))
            b[at] ^= 1 << int(rng.integers(0, 8))
        z = bytes(b)
        # what the bytes in front of the first bad block must be: the CPU oracle's (tests/pnghelp.py), zero-padded
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
        import pnghelp as ph
        st, out, used, aux = ph.orc_inflate(z, 0, cap=len(d) + 4096)
        d = bytes(out) if st == 0 else bytes(out) + bytes(max(0, len(d) - len(out)) + 16)
    fz, fraw = os.path.join(tmp, "z"), os.path.join(tmp, "raw")
    open(fz, "wb").write(z); open(fraw, "wb").write(d)
    seg = int(rng.choice([256, 512, 1024, 4096, 16384, 1 << 20]))
    parts = int(rng.choice([0, 0, 2, 3, 5, 16]))
    env = dict(os.environ)
    if parts:
        env["EMU_PARTS"] = str(parts)
    r = subprocess.run([emu, fz, fraw, "0", str(seg)], capture_output=True, text=True, env=env, timeout=900)
    return r.returncode, f"seed {seed}: n {len(d)} z {len(z)} seg {seg} parts {parts}: {r.stdout.strip()[-160:]} {r.stderr.strip()[-200:]}"


if __name__ == "__main__":
    emu, seed0, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for k in range(count):
            rc, what = run(emu, seed0 + k, tmp)
            if rc != 0:
                print(f"rc {rc} ({ {3: 'declined', 4: 'partial'}.get(rc, 'FAIL') }) {what}", flush=True)
            if rc not in ((0, 3, 4, 5) if os.environ.get("FUZZ_DAMAGE") else (0, 3, 4)):
                bad += 1
    print("done", count, "bad", bad)
