"""The multi-device entry of the C ABI (spng_decode_batch_multi), the copy ceiling and spng_trim.
One GPU is what the test box has: several contexts on device 0 stand in for the devices of a node -- the sharding, the
per-context enqueue, the gather copies behind each context's decode and the result merge are the code a node runs."""
import zlib

import numpy as np
import pytest

import pnghelp as ph
import swift_png_amd as spng
from swift_png_amd import synth
from swift_png_amd.distributed import shard

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("count,parts", [(0, 3), (1, 8), (7, 2), (8, 8), (1024, 8), (1000, 7), (5, 9)])
def test_shard_matches_the_python_side(count, parts):
    for k in range(parts):
        lo, hi = shard(count, parts, k)
        assert spng.shard_of(count, parts, k) == (lo, hi - lo)


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_decode_batch_multi(gpu, n_ctx):
    s0 = gpu.load()
    sessions = [s0] + [spng.Session(0, use_torch_stream=False) for _ in range(n_ctx - 1)]
    try:
        shapes = [(300, 200, 8, 4, False), (64, 64, 16, 4, True), (129, 77, 8, 3, False), (500, 40, 8, 4, False), (33, 33, 8, 1, True),
                  (256, 256, 8, 4, False), (40, 300, 16, 2, False)]
        imgs, descs, keep, want, gather, gbuf = [], [], [], [], [], []
        for k, (w, h, depth, ch, il) in enumerate(shapes):
            rng = np.random.default_rng(k)
            S = spng.storage_size(w, h, depth, ch)
            raster = rng.integers(0, 256, S, dtype=np.uint8)
            rows = ph.orc_filter(raster, w, h, depth, ch, il)
            z = zlib.compress(rows, 6)
            if k == 3:
                z = z[:len(z) // 2]                                       # a truncated stream: NEED_MORE_INPUT, a partial image
            st_o, storage_o, _ = ph.orc_decode(ph.Png(w, h, depth, {1: 0, 2: 4, 3: 2, 4: 6}[ch], il, False, z))
            want.append((st_o, storage_o))
            d_z = s0.to_device(z)
            d_rows = s0.empty(spng.inflated_size(w, h, depth, ch, il) + 64)
            d_st = s0.torch.zeros(S, dtype=s0.torch.uint8, device=s0.tdev)
            d_g = s0.torch.zeros(S, dtype=s0.torch.uint8, device=s0.tdev)
            keep += [d_z, d_rows, d_st]
            gbuf.append(d_g)
            descs.append(s0.image_desc(d_z, d_rows, d_st, w, h, depth, ch, il, rows_cap=d_rows.numel()))
            gather.append(d_g.data_ptr())
        res = spng.decode_batch_multi(sessions, descs, gather)
        for k, r in enumerate(res):
            st_o, storage_o = want[k]
            assert r.status == st_o, (k, r.status, st_o)
            if st_o == 0:
                assert bytes(keep[3 * k + 2].cpu().numpy()) == storage_o.tobytes(), k
                assert bytes(gbuf[k].cpu().numpy()) == storage_o.tobytes(), ("gathered", k)
        # without a gather list the rasters stay where they are
        res2 = spng.decode_batch_multi(sessions, descs)
        assert [r.status for r in res2] == [r.status for r in res]
        # a shard in one call (SPNG_CFG_MULTI_GROUPS = 1) and in two (the default where rasters leave: a group's copies on the
        # context's copy stream beside the next group's decode): the same rasters arrive, the caller's device stays current
        import torch
        for groups in (1, 2):
            for s in sessions:
                s.configure(spng.CFG_MULTI_GROUPS, groups)
            for g in gbuf:
                g.zero_()
            before = torch.cuda.current_device()
            res3 = spng.decode_batch_multi(sessions, descs, gather)
            assert torch.cuda.current_device() == before
            assert [r.status for r in res3] == [r.status for r in res]
            for k, (st_o, storage_o) in enumerate(want):
                if st_o == 0:
                    assert bytes(gbuf[k].cpu().numpy()) == storage_o.tobytes(), ("gathered", groups, k)
        for s in sessions:
            s.configure(spng.CFG_MULTI_GROUPS, 0)
    finally:
        for s in sessions[1:]:
            s.close()


def test_copy_ceiling_and_trim(gpu):
    s = gpu.load()
    gbps, ms = s.copy_ceiling(1 << 30, 0, 3)
    skew, _ = s.copy_ceiling(1 << 30, 1, 2)
    assert 1000 < gbps < 8000 and 500 < skew < 8000                  # (an MI355X copies at 4.6 - 5.4 TB/s; the skewed rows at ~3.3)
    # scratch given back, and taken again by the next call that needs it
    data = np.random.default_rng(1).integers(0, 8, 300000, dtype=np.uint8).tobytes()
    a = s.deflate(data, 9)
    free0 = s.torch.cuda.mem_get_info()[0]
    s.trim()
    free1 = s.torch.cuda.mem_get_info()[0]
    assert free1 >= free0
    assert s.deflate(data, 9) == a == ph.orc_deflate(data, 9)
    z = zlib.compress(data, 6)
    assert s.inflate(z, 0, len(data) + 16)[1] == data
    s.trim()
    assert s.inflate(z, 0, len(data) + 16)[1] == data


def test_decode_1024_images_with_40_gib_free(gpu):
    """A drop-in does not get the device to itself: 1024 x 4096^2 RGBA8 images (BASELINE configs[1]) decoded while only 40 GiB
    are free.  The token pool (54 GB would hold the batch's tokens at once) is capped at half of what is free, the batch goes
    through in groups, every image is right; the scratch is given back on request (spng_trim)."""
    s = gpu.load()
    t = s.torch
    W = H = 4096
    U, S = spng.inflated_size(W, H, 8, 4, False), spng.storage_size(W, H, 8, 4)
    n, unique = 1024, 4
    imgs = [synth.image(40 + k, W, H) for k in range(unique)]
    zs = [zlib.compress(s.filter(im.tobytes(), W, H, 8, 4, False), 6) for im in imgs]
    d_z = [s.to_device(z) for z in zs]
    want = [s.to_device(im.reshape(-1)) for im in imgs]
    s.trim()
    t.cuda.empty_cache()
    rows = t.empty(n * U, dtype=t.uint8, device=s.tdev)
    out = t.empty(n * S, dtype=t.uint8, device=s.tdev)
    free, _ = t.cuda.mem_get_info()
    ballast = t.empty(max(free - (40 << 30), 1), dtype=t.uint8, device=s.tdev) if free > (41 << 30) else None
    try:
        free_now, _ = t.cuda.mem_get_info()
        assert free_now < (41 << 30)
        descs = [s.image_desc(d_z[i % unique], rows[i * U:(i + 1) * U], out[i * S:(i + 1) * S], W, H, 8, 4, False, rows_cap=U) for i in range(n)]
        res = s.decode_batch(descs)
        assert all(r.status == 0 for r in res)
        assert sum(r.reserved == 1 for r in res) == n                     # (the pipeline's own work, group by group; nothing fell to the serial kernel)
        for i in (0, 1, 2, 3, 511, 1023):
            assert t.equal(out[i * S:(i + 1) * S], want[i % unique]), i
        held = free_now - t.cuda.mem_get_info()[0]
        assert held <= (21 << 30), held                                   # token pool + segment tables: at most half of what was free
        s.trim()
        assert t.cuda.mem_get_info()[0] >= free_now - (1 << 30)
    finally:
        del ballast, rows, out
        t.cuda.empty_cache()
        s.trim()
