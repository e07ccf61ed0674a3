"""GPU diagnostic: spng_unfilter against the oracle on a few shapes, with the coordinates of the first mismatches."""
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import pnghelp as ph
import swift_png_amd as spng

s = spng.load(0)
rng = np.random.default_rng(5)
bad = 0
for (w, h, depth, ch, forced) in [(8, 3, 8, 4, None), (40, 33, 8, 4, None), (300, 200, 8, 4, None), (33, 70, 8, 4, 4), (64, 64, 8, 4, 3), (1000, 130, 8, 4, None),
                                  (40, 33, 16, 4, None), (300, 100, 16, 4, None), (31, 40, 16, 2, None), (4096, 96, 8, 4, None), (4096, 200, 16, 4, 4),
                                  (5000, 70, 8, 4, None), (130, 2100, 8, 4, None)]:
    bpp = depth * ch // 8
    pitch = w * bpp
    rows = rng.integers(0, 256, (h, pitch + 1), dtype=np.uint8)
    rows[:, 0] = rng.integers(0, 5, h) if forced is None else forced
    rows[::17, 0] = 0
    st_o, want = ph.orc_unfilter(rows.tobytes(), w, h, depth, ch, False)
    st, got = s.unfilter(rows.tobytes(), w, h, depth, ch, False)
    got = np.frombuffer(got, np.uint8).reshape(h, pitch); want = np.asarray(want, np.uint8).reshape(h, pitch)
    diff = np.argwhere(got != want)
    print(f"{w}x{h} depth {depth} ch {ch} forced {forced}: status {st}/{st_o}, {len(diff)} mismatching bytes", flush=True)
    if len(diff):
        bad += 1
        print("   first mismatches (row, byte):", [tuple(int(v) for v in d) for d in diff[:12]], " rows with mismatches:", sorted(set(int(d[0]) for d in diff))[:20],
              " filters of those rows:", [int(rows[int(r), 0]) for r in sorted(set(int(d[0]) for d in diff))[:20]])
        r, c = (int(v) for v in diff[0])
        print("   got ", got[r, max(0, c - 8):c + 24].tolist()); print("   want", want[r, max(0, c - 8):c + 24].tolist())
print("bad cases:", bad)
