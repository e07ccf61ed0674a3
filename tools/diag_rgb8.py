"""GPU diagnostic: filter-select and defilter of 4096^2 RGB8 rasters against the oracle, in batches of 1 and of 64 (the
scanline_formats leg of the bench said the defiltered rasters differ)."""
import sys, ctypes
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import swift_png_amd as spng
from swift_png_amd import synth
import pnghelp as ph
s = spng.load(0)
W = H = 4096
import os
im = np.ascontiguousarray(synth.image(int(os.environ.get("DIAG_IMAGE", "500")), W, H).reshape(H, W, 4)[..., :3])
U = spng.inflated_size(W, H, 8, 3, False); S = spng.storage_size(W, H, 8, 3)
want_rows = ph.orc_filter(im.reshape(-1), W, H, 8, 3, False)
for m in (1, 8):
    d_sto = s.to_device(im.tobytes() * 1).repeat(m)
    d_rows = torch.empty(m * U, dtype=torch.uint8, device=s.tdev); d_back = torch.zeros(m * S, dtype=torch.uint8, device=s.tdev)
    fd = [s.image_desc(None, d_rows[j * U:(j + 1) * U], d_sto[j * S:(j + 1) * S], W, H, 8, 3, False, rows_cap=U) for j in range(m)]
    res = s.filter_batch(fd)
    rows = bytes(d_rows[(m - 1) * U:m * U].cpu().numpy())
    ok_f = rows == want_rows
    if not ok_f:
        a = np.frombuffer(rows, np.uint8); b = np.frombuffer(want_rows, np.uint8); k = int(np.flatnonzero(a != b)[0])
        print(f"m={m}: filter differs first at byte {k} = row {k // (W * 3 + 1)} col {k % (W * 3 + 1)}; {int((a != b).sum())} bytes differ")
    ud = [s.image_desc(None, d_rows[j * U:(j + 1) * U], d_back[j * S:(j + 1) * S], W, H, 8, 3, False, rows_cap=U) for j in range(m)]
    r2 = s.unfilter_batch(ud)
    back = d_back[(m - 1) * S:m * S]
    ok_u = bool(torch.equal(back, d_sto[:S]))
    if not ok_u:
        d = (back != d_sto[:S]).nonzero()[:, 0]
        print(f"m={m}: unfilter differs first at byte {int(d[0])} = row {int(d[0]) // (W * 3)} col {int(d[0]) % (W * 3)}; {len(d)} bytes; row filter types around: {[rows[(int(d[0]) // (W*3) + q) * (W*3+1)] for q in (-1,0,1)]}")
    print(f"m={m}: filter ok {ok_f}, unfilter ok {ok_u}, statuses {set(r.status for r in res)} {set(r.status for r in r2)}")
    wr = s.to_device(want_rows)
    badf = [j for j in range(m) if not torch.equal(d_rows[j * U:(j + 1) * U], wr)]
    badu = [j for j in range(m) if not torch.equal(d_back[j * S:(j + 1) * S], d_sto[:S])]
    print(f"m={m}: images whose filtered rows differ: {badf[:10]} ({len(badf)}), whose rasters differ: {badu[:10]} ({len(badu)})")
    for j in badu[:2]:
        d = (d_back[j * S:(j + 1) * S] != d_sto[:S]).nonzero()[:, 0]
        print(f"   image {j}: {len(d)} bytes differ, first at row {int(d[0]) // (W * 3)} col {int(d[0]) % (W * 3)}, last at row {int(d[-1]) // (W * 3)}")
    del wr
