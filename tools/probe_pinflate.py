#!/usr/bin/env python3
"""GPU probe of the parallel inflate pipeline: correctness against zlib on a spread of streams, which
path produced each result (spng_result.reserved), and per-stage times."""
import sys
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import swift_png_amd as spng  # noqa: E402
from swift_png_amd import synth  # noqa: E402


def run(s, name, streams, expect, fmt=0, seg=0, slack=16):
    s.configure(spng.CFG_SEGMENT_BYTES, seg)
    d = [s.to_device(z) for z in streams]
    caps = [len(e) + slack for e in expect]
    s.profile(True)
    t0 = time.perf_counter()
    outs, res = s.inflate_batch(d, caps, fmt)
    dt = time.perf_counter() - t0
    par = s.profile_get(spng.K_PINFLATE)[0]
    ser = s.profile_get(spng.K_INFLATE)[0]
    s.profile(False)
    ok = 0
    fast = 0
    for o, r, e in zip(outs, res, expect):
        good = r.status == 0 and r.written == len(e) and bytes(o[:r.written].cpu().numpy()) == e
        ok += good
        fast += r.reserved == 1
        if not good:
            got = bytes(o[:min(r.written, len(e))].cpu().numpy())
            first = next((i for i, (a, b) in enumerate(zip(got, e)) if a != b), None)
            print(f"  !! {name}: status {r.status} path {r.reserved} written {r.written} want {len(e)} first diff {first}")
    total = sum(len(e) for e in expect)
    print(f"{name:34s} n={len(streams):3d} ok={ok:3d} fast={fast:3d}  parallel {par:9.3f} ms  serial {ser:9.3f} ms  "
          f"wall {dt * 1e3:9.1f} ms  out {total / 1e6:8.1f} MB", flush=True)
    return ok == len(streams), fast


def main():
    s = spng.load(0)
    rng = np.random.default_rng(5)
    payloads = {
        "text": (b"the quick brown fox jumps over the lazy dog. " * 3000),
        "random": rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
        "zeros": bytes(300000),
        "ramp": bytes(range(256)) * 1000,
        "short": b"abc",
        "empty": b"",
        "mixed": (rng.integers(0, 4, 100000, dtype=np.uint8).tobytes() + bytes(50000) +
                  rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()),
    }
    allok = True
    for level in (1, 6, 9, 0):
        names = sorted(payloads)
        zs = [zlib.compress(payloads[k], level) for k in names]
        ok, _ = run(s, f"payloads zlib L{level}", zs, [payloads[k] for k in names])
        allok &= ok
        ok, _ = run(s, f"payloads zlib L{level} seg=4096", zs, [payloads[k] for k in names], seg=4096)
        allok &= ok
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    zf = co.compress(payloads["text"]) + co.flush()
    allok &= run(s, "fixed huffman", [zf], [payloads["text"]])[0]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(payloads["mixed"]) + co.flush()
    allok &= run(s, "raw deflate (ios)", [raw], [payloads["mixed"]], fmt=1, seg=4096)[0]

    img = synth.image(3, 1024, 1024)
    rows = s.filter(img.tobytes(), 1024, 1024, 8, 4, False)
    for level in (1, 6, 9):
        z = zlib.compress(rows, level)
        for seg in (0, 65536, 16384):
            allok &= run(s, f"1024^2 rows zlib L{level} seg={seg}", [z], [rows], seg=seg)[0]
    zsw = s.deflate(rows, 6)
    for seg in (0, 65536, 8192):
        allok &= run(s, f"1024^2 rows swiftpng L6 seg={seg}", [zsw], [rows], seg=seg)[0]
    # many streams at once
    zs = [zlib.compress(rows[k * 1000:], 6) for k in range(64)]
    allok &= run(s, "64 x 1024^2 rows zlib L6", zs, [rows[k * 1000:] for k in range(64)], seg=65536)[0]
    # token budget smaller than the batch: several passes
    s.configure(spng.CFG_TOKEN_BYTES, 8 << 20)
    allok &= run(s, "64 x 1024^2, 8 MiB token budget", zs, [rows[k * 1000:] for k in range(64)], seg=65536)[0]
    s.configure(spng.CFG_TOKEN_BYTES, 0)
    # one 4K image
    img = synth.image(1, 4096, 4096)
    rows = s.filter(img.tobytes(), 4096, 4096, 8, 4, False)
    z = zlib.compress(rows, 6)
    for seg in (0, 1 << 20, 1 << 18):
        allok &= run(s, f"4096^2 rows zlib L6 seg={seg}", [z], [rows], seg=seg)[0]
    allok &= run(s, "8 x 4096^2 rows zlib L6", [z] * 8, [rows] * 8)[0]
    print("ALL OK" if allok else "FAILURES")


if __name__ == "__main__":
    main()
