"""GPU diagnostic: spng_deflate at levels >= 8 against the oracle on a ladder of inputs (first differing byte, whether zlib accepts the stream)."""
import sys, time, zlib; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import pnghelp as ph
import swift_png_amd as spng

s = spng.load(0)
rng = np.random.default_rng(3)
text = (b"lorem ipsum dolor sit amet, consectetur adipiscing elit " * 4000)
def scan(n, seed=1):
    r = np.random.default_rng(seed); a = r.integers(-3, 4, n).astype(np.int16); a[r.random(n) < 0.6] = 0
    return a.astype(np.uint8).tobytes()
cases = [("empty", b""), ("two", b"ab"), ("three", b"abc"), ("tiny", b"abcabcabcabcabcabcabcabc" * 3), ("noise2k", rng.integers(0, 256, 2000, dtype=np.uint8).tobytes()),
         ("noise2047", rng.integers(0, 256, 2047, dtype=np.uint8).tobytes()), ("noise2048", rng.integers(0, 256, 2048, dtype=np.uint8).tobytes()),
         ("noise9k", rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()), ("text5k", text[:5000]), ("text40k", text[:40001]),
         ("scan100k", scan(100000)), ("zeros50k", bytes(50000)), ("runs", b"".join(bytes([int(rng.integers(0, 4))]) * int(rng.integers(90, 700)) for _ in range(120))),
         ("few120k", rng.integers(0, 4, 120000, dtype=np.uint8).tobytes()), ("scan3M", scan(3 << 20, 2)), ("noise5M", rng.integers(0, 256, 5 << 20, dtype=np.uint8).tobytes())]
bad = 0
for name, data in cases:
    for level in ((8, 9, 13) if len(data) < 200000 else (9,)):
        t = time.time()
        try:
            got = s.deflate(data, level)
        except Exception as e:
            print(f"{name} L{level}: EXCEPTION {e}", flush=True); bad += 1; continue
        dt = time.time() - t
        want = ph.orc_deflate(data, level)
        if got == want:
            print(f"{name} L{level}: ok ({len(data)} -> {len(got)}, {dt*1e3:.0f} ms)", flush=True); continue
        bad += 1
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
        try: valid = zlib.decompress(got) == data
        except zlib.error as e: valid = f"zlib: {e}"
        print(f"{name} L{level}: MISMATCH at byte {k} of {len(want)} (got {len(got)} bytes); stream valid: {valid}", flush=True)
print("bad:", bad)
s.configure(spng.CFG_DEFLATE_MODE, spng.DEFLATE_ONE_KERNEL)
assert s.deflate(cases[10][1], 9) == ph.orc_deflate(cases[10][1], 9)
print("one-kernel mode still exact")
