"""Fuzz of the deflate kernels in the CPU wave emulator (tools/emu): random data of the kinds fuzz_pinflate.py makes x random level
0-13 x random chunks per stream x (levels >= 8) random push points; the emulated kernels must give the oracle's stream bit for bit.

    python tools/emu/prep_deflate.py swift_png_amd/csrc/deflate.hip /tmp/deflate_emu.inc
    g++ -O1 -std=c++17 -DSPNG_EMU -DEMU_DEFLATE_SRC='"/tmp/deflate_emu.inc"' -Itools/emu -Iswift_png_amd/csrc -x c++ -fpermissive -w -o /tmp/emu_deflate2 tools/emu/emu_deflate2.cpp
    python tools/emu/fuzz_deflate.py /tmp/emu_deflate2 <first seed> <count>
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import fuzz_pinflate  # noqa: E402
import pnghelp as ph  # noqa: E402


def run(emu, seed, tmp):
    rng = np.random.default_rng(seed)
    d, _ = fuzz_pinflate.gen(rng)
    d = d[:int(rng.integers(0, 40000))]
    level = int(rng.choice([0, 1, 3, 5, 6, 7, 8, 8, 9, 9, 9, 10, 11, 13]))
    want = ph.orc_deflate(d, level)
    fi, fw = os.path.join(tmp, "in"), os.path.join(tmp, "want")
    open(fi, "wb").write(d); open(fw, "wb").write(want)
    args = [emu, fi, fw, str(level), "0", str(int(rng.integers(1, 6)))]
    if level >= 8 and rng.random() < 0.4 and len(d) > 4:
        args += [str(c) for c in sorted(set(int(x) for x in rng.integers(1, len(d), int(rng.integers(1, 4)))))]
    r = subprocess.run(args, capture_output=True, text=True, timeout=1800)
    return r.returncode, f"seed {seed}: n {len(d)} level {level} args {args[5:]}: {r.stdout.strip()[-160:]} {r.stderr.strip()[-200:]}"


if __name__ == "__main__":
    emu, seed0, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for k in range(count):
            rc, what = run(emu, seed0 + k, tmp)
            if rc != 0:
                print("FAIL", what, flush=True)
                bad += 1
    print("done", count, "bad", bad)
