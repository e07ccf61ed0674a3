"""oracle/pixels.py `overdrawn` (the sequential restatement of PNG.Context.push(data:overdraw: true)) against the properties the
reference's procedure has by construction; the device's closed form is compared with it on the GPU (tests/test_gpu_resume.py)."""
import sys

import numpy as np
import pytest

import pnghelp as ph

sys.path.insert(0, str(ph.ROOT / "oracle"))
import pixels as orc_pixels  # noqa: E402


def scanline_count(w, h):
    n = 0
    for (bx, by), (ex, ey) in orc_pixels.ADAM7:
        sw, sh = (w + (1 << ex) - bx - 1) >> ex, (h + (1 << ey) - by - 1) >> ey
        n += sh if sw > 0 and sh > 0 else 0
    return n


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (8, 8), (9, 17), (33, 20)])
def test_overdraw_never_touches_assigned_pixels_and_ends_as_the_image(w, h):
    rng = np.random.default_rng(w * 100 + h)
    final = rng.integers(1, 256, (h, w, 3), dtype=np.uint8)
    total = scanline_count(w, h)
    assert (orc_pixels.overdrawn(final, total) == final).all()
    seen = np.zeros((h, w), dtype=bool)
    k = 0
    for (bx, by), (ex, ey) in orc_pixels.ADAM7:
        sx, sy = 1 << ex, 1 << ey
        sw, sh = (w + sx - bx - 1) >> ex, (h + sy - by - 1) >> ey
        if sw <= 0 or sh <= 0:
            continue
        for y in range(sh):
            k += 1
            seen[by + y * sy, bx::sx] = True
            img = orc_pixels.overdrawn(final, k)
            assert (img[seen] == final[seen]).all()                    # an assigned pixel keeps its value
            if k >= (h + 7) // 8:                                      # after pass 0 every pixel shows SOME assigned pixel
                flat = {tuple(p) for p in final[seen]}
                assert all(tuple(p) in flat for p in img.reshape(-1, 3))


def test_overdraw_first_pass_is_8x8_blocks_and_the_quirk_of_pass_3():
    final = np.arange(16 * 16, dtype=np.uint8).reshape(16, 16, 1) + 1
    img = orc_pixels.overdrawn(final, 2)                              # pass 0 whole (two scanlines)
    for Y in range(16):
        for X in range(16):
            assert img[Y, X, 0] == final[Y & ~7, X & ~7, 0]
    # passes 0-2 whole = 2 + 2 + 2 scanlines, then pass 3's first two scanlines: rows 0 (brush 2 x 4) and 4 (brush 2 x 2)
    img = orc_pixels.overdrawn(final, 8)
    assert img[3, 2, 0] == final[0, 2, 0] and img[5, 2, 0] == final[4, 2, 0]
    assert img[6, 2, 0] == final[4, 0, 0]                             # rows 6-7 keep pass 2's 4 x 4 cell: `base.y & 0b111`


def closed_form(final, done, fill=0):
    """the device kernel's rule (csrc/unfilter.hip, overdraw_kernel), restated: an unassigned pixel shows the source of the
    last pass that has an assigned scanline whose cell covers it"""
    H, W, _ = final.shape
    BX, BY = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1]
    EX, EY = [3, 3, 2, 2, 1, 1, 0], [3, 3, 3, 2, 2, 1, 1]
    img = np.full_like(final, fill)
    for Y in range(H):
        for X in range(W):
            own = 6 if Y & 1 else 5 if X & 1 else 4 if Y & 2 else 3 if X & 2 else 2 if Y & 4 else 1 if X & 4 else 0
            if ((Y - BY[own]) >> EY[own]) < done[own]:
                img[Y, X] = final[Y, X]
                continue
            for q in range(6, -1, -1):
                if not done[q] or X < BX[q] or Y < BY[q]:
                    continue
                yq = (Y - BY[q]) >> EY[q]
                if yq >= done[q]:
                    continue
                B = BY[q] + (yq << EY[q])
                bx, by = (1 << EX[q]) >> (1 if BX[q] else 0), (1 << EY[q]) >> (1 if B & 7 else 0)
                if bx * by <= 1 or Y >= B + by:
                    continue
                img[Y, X] = final[B, BX[q] + (X - BX[q]) // bx * bx]
                break
    return img


@pytest.mark.parametrize("w,h", [(1, 1), (2, 3), (5, 5), (8, 8), (9, 17), (21, 12), (33, 20)])
def test_the_closed_form_the_device_uses_equals_the_sequential_procedure(w, h):
    rng = np.random.default_rng(w * 31 + h)
    final = rng.integers(1, 256, (h, w, 2), dtype=np.uint8)
    done = [0] * 7
    k = 0
    assert (closed_form(final, done) == orc_pixels.overdrawn(final, 0)).all()
    for q, ((bx, by), (ex, ey)) in enumerate(orc_pixels.ADAM7):
        sw, sh = (w + (1 << ex) - bx - 1) >> ex, (h + (1 << ey) - by - 1) >> ey
        if sw <= 0 or sh <= 0:
            continue
        for y in range(sh):
            k += 1
            done[q] = y + 1
            assert (closed_form(final, done) == orc_pixels.overdrawn(final, k)).all(), (q, y)
