#!/usr/bin/env python3
"""bench.py -- the north-star measurement: batched PNG decode (inflate + unfilter) of 4K RGBA8 images.

    python bench.py --gpus N --steps K --warmup W [--mode decode|encode] [--scaling strong|weak]
    (N > 1: one rank per GPU over RCCL; started by torch.distributed.run, or by this script itself when
     it is run plainly with --gpus N)

decode (default).  A "step" is one pass of the decode hot path (spng_decode_batch: inflate -> unfilter,
results left in HBM) over the whole batch.  Workload = BASELINE.json configs[1]: 1024 synthetic
4096x4096 RGBA8 PNG streams, mixed filters chosen by the reference's own heuristic, DEFLATE level 6.
The headline line is measured on swift-png-made level-6 streams (the batch's scanlines deflated by the
device deflater, bit-exact with LZ77.Deflator: a dynamic block every <= 2047 tokens; `--streams`); the
same scanlines deflated by host zlib (16 K-token blocks) are measured in the same run and reported under
"zlib_streams".  At N > 1 the default is configs[2]: the SAME 1024
images sharded 1024/N per GPU ("strong"), every rank decoding its shard in groups whose rasters travel
to rank 0 over RCCL (grouped point-to-point, xGMI) while the next group decodes; `--scaling weak`
gives every GPU its own 1024 images instead.  Compressed inputs are resident in HBM before the timed
region; nothing is copied to the host inside it.

encode (`--mode encode`).  BASELINE configs[3]: random 4096x4096 RGBA8 rasters -> filter-select ->
DEFLATE level 9 (spng_encode_batch); a step encodes the whole batch.

The CPU oracle is used only for the `cpu_baseline` leg (rank 0, N = 1, bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W = H = 4096
DEPTH, CHANNELS = 8, 4
MPIX = W * H / 1e6
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
STAGES = ("pinf_find", "pinf_decode", "pinf_resolve", "inflate", "unfilter")
PMC_FILE = "r06_pmc_traffic.json"      # rocprofv3 --pmc passes of this very workload (bench.py --traffic / tools/final_run.sh), committed


def build_inputs(session, unique: int, threads: int, encoder: str):
    """-> (rasters, filtered rows, streams).  Product path only: synthetic rasters -> GPU filter-select
    (spng_filter, the reference heuristic) -> level-6 DEFLATE, by zlib on the host cores or by the
    device deflater (swift-png's own bitstream)."""
    from swift_png_amd import synth
    with ThreadPoolExecutor(threads) as pool:
        images = list(pool.map(lambda s: synth.image(s, W, H, CHANNELS, DEPTH), range(unique)))
        rows = [session.filter(img.tobytes(), W, H, DEPTH, CHANNELS, False) for img in images]
        if encoder == "swiftpng":
            outs, res = session.deflate_batch([session.to_device(r) for r in rows], 6)
            assert all(r.status == 0 for r in res)
            streams = [bytes(o[:r.written].cpu().numpy()) for o, r in zip(outs, res)]
        else:
            streams = list(pool.map(lambda r: zlib.compress(r, 6), rows))
        ok = list(pool.map(lambda rz: zlib.decompress(rz[1]) == rz[0], zip(rows, streams)))
    assert all(ok)
    return images, rows, streams


def host_cores() -> int:
    """cores this process may really use: the affinity mask, cut down by a cgroup CPU quota if there is one (a container
    that sees 256 CPUs and may burn 12 of them runs 12 workers, and says so)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                n = max(1, min(n, int(q / p + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(streams, images, rows, cores: int):
    """The CPU oracle (restatement of swift-png's CPU path: inflate + defilter + assign) on a bounded sample of the
    same streams: all host cores -- one worker PROCESS per core, library loaded and buffers touched before the clock
    starts (bench_cpu.py, a fresh interpreter without torch) --, one core, and -- as a sanity anchor -- zlib's inflate
    followed by the oracle's defilter on one core (the reference publishes swift-png = 1.35 x libpng decode time).
    Test infrastructure used as a reported baseline only."""
    import shutil
    import tempfile
    sys.path.insert(0, str(ROOT / "tests"))
    import pnghelp as ph
    tmp = tempfile.mkdtemp(prefix="spng_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for k, z in enumerate(streams[:8]):
            (Path(tmp) / f"z{k}").write_bytes(z)
        tasks = max(64, 4 * cores)
        out = subprocess.run([sys.executable, str(ROOT / "bench_cpu.py"), "decode", tmp, str(cores), str(tasks), str(W), str(H)],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-400:]
        allc = json.loads(out.stdout.strip().splitlines()[-1])
        out = subprocess.run([sys.executable, str(ROOT / "bench_cpu.py"), "decode", tmp, "1", "2", str(W), str(H)],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-400:]
        one = json.loads(out.stdout.strip().splitlines()[-1])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    t0 = time.perf_counter()
    raw = zlib.decompress(streams[0])
    st, storage = ph.orc_unfilter(raw, W, H, DEPTH, CHANNELS, False)
    dtz = time.perf_counter() - t0
    assert st == 0 and raw == rows[0] and bytes(storage[:4096]) == images[0].reshape(-1)[:4096].tobytes()
    v_all, v_one = tasks * MPIX / allc["wall_s"], 2 * MPIX / one["wall_s"]
    return {"value": round(v_all, 1), "unit": "MPixels/s", "cores": cores, "kind": "port",
            "sample": f"{tasks} decodes of the same 4096x4096 RGBA8 level-6 streams on {cores} worker processes (one per core, "
                      f"buffers touched before the clock), oracle inflate+defilter+assign, {allc['wall_s']:.1f} s wall, "
                      f"{allc['task_s']:.2f} s per image inside a worker",
            "scaling_vs_one_core": round(v_all / v_one / cores, 3),
            "one_core": {"value": round(v_one, 2), "unit": "MPixels/s", "cores": 1, "sample": f"2 images, {one['wall_s']:.1f} s"},
            "zlib_anchor": {"value": round(MPIX / dtz, 2), "unit": "MPixels/s", "cores": 1,
                            "sample": f"zlib 1.2.11 inflate + oracle defilter of 1 image, {dtz:.1f} s"}}


def measure_traffic(args):
    """--traffic: HBM bytes per kernel measured for THIS invocation -- two rocprofv3 --pmc sub-runs (FETCH_SIZE, WRITE_SIZE: separate
    passes, as MI355X_MICROARCH.md prescribes) of this script on the headline workload, a warm-up step and a timed one each, before
    this process touches the GPU; the result replaces profiles/PMC_FILE (which carries the digest of the sources it was taken on)."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="spng_pmc_")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        sub = [sys.executable, str(Path(__file__).resolve()), "--steps", "1", "--warmup", "1", "--no-alt", "--no-cpu-baseline", "--no-extras",
               "--images", str(args.images), "--unique", str(args.unique), "--streams", args.streams]
        for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, d), "--"] + sub,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-300:]
        dst = ROOT / "profiles" / PMC_FILE
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "pmc_traffic.py"), os.path.join(tmp, "fetch"), os.path.join(tmp, "write"),
                            args.streams, str(args.images), str(args.unique), str(dst), "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-300:]
        return True
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def traffic_provenance(kind: str):
    """where `traffic` comes from and whether it describes the sources that are running"""
    try:
        import swift_png_amd as spng
        cfg = json.loads((ROOT / "profiles" / PMC_FILE).read_text())["configs"][kind]
        mine, theirs = spng.source_digest(), cfg.get("source_digest")
        return {"file": "profiles/" + PMC_FILE, "source_digest_measured": theirs, "source_digest_running": mine, "same_sources": mine == theirs}
    except (OSError, KeyError, ValueError):
        return None


def pmc_traffic(kind: str, images: int, unique: int):
    """HBM bytes per decode step of each kernel (summed over its launches of the step: the file's `hbm_bytes_per_launch`
    dates from one launch per step) from the committed rocprofv3 PMC passes of this very workload
    (profiles/ + PMC_FILE: FETCH_SIZE / WRITE_SIZE in separate passes, corrected as
    MI355X_MICROARCH.md prescribes); None when the file does not describe this configuration."""
    try:
        pmc = json.loads((ROOT / "profiles" / PMC_FILE).read_text())
        cfg = pmc["configs"][kind]
        if cfg["images"] != images or cfg["unique"] != unique:
            return {}
        return {k: v["hbm_bytes_per_launch"] for k, v in cfg["kernels"].items()}
    except (OSError, KeyError, ValueError):
        return {}


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU ourselves."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class DecodeJob:
    """One rank's share of a decode workload: descriptors over two slabs (scanline scratch, rasters)."""

    def __init__(self, spng, s, torch, d_streams, n, first, unique, groups):
        self.spng, self.s, self.n = spng, s, n
        self.U = spng.inflated_size(W, H, DEPTH, CHANNELS, False)
        self.S = spng.storage_size(W, H, DEPTH, CHANNELS)
        self.rows_cap = (self.U + 4096 + 255) & ~255
        self.d_rows = torch.empty(n * self.rows_cap, dtype=torch.uint8, device=s.tdev)
        self.d_out = torch.empty(n * self.S, dtype=torch.uint8, device=s.tdev)
        self.descs = (spng.ImageDesc * n)()
        self.src = []
        for j in range(n):
            z = d_streams[(first + j) % unique]
            self.src.append((first + j) % unique)
            self.descs[j] = spng.ImageDesc(z.data_ptr(), z.numel(), self.d_rows.data_ptr() + j * self.rows_cap,
                                           self.rows_cap, self.d_out.data_ptr() + j * self.S, W, H, DEPTH, CHANNELS,
                                           0, 0, 0)
        from swift_png_amd.distributed import group_bounds
        self.groups = [group_bounds(n, groups, g) for g in range(groups)]    # (the same count on every rank)
        self.dres = s.empty(n * ctypes.sizeof(spng.Result))

    def decode_group(self, g):
        lo, hi = self.groups[g]
        if hi <= lo:
            return
        arr = (self.spng.ImageDesc * (hi - lo)).from_address(ctypes.addressof(self.descs) + lo * ctypes.sizeof(self.spng.ImageDesc))
        rp = ctypes.c_void_p(self.dres.data_ptr() + lo * ctypes.sizeof(self.spng.Result))
        st = self.s.lib.spng_decode_batch(self.s.ctx, arr, hi - lo, rp, None)
        assert st == 0, st

    def results(self):
        raw = bytes(self.dres.cpu().numpy())
        return list((self.spng.Result * self.n).from_buffer_copy(raw))


def run_decode(args, torch, dist, spng, s, rank, world, kind, unique, with_gather):
    """Builds the inputs of `kind` ("zlib" / "swiftpng"), times args.steps steps, checks every slot.
    -> dict of measurements (rank 0 gets the max-over-ranks time)."""
    from swift_png_amd.distributed import exchange_plan, shard
    cores = os.cpu_count() or 1
    images, rows, streams = build_inputs(s, unique, min(cores, 32), kind)
    d_streams = [s.to_device(z) for z in streams]
    # (the deflater's slabs and torch's cached blocks of the input generation go back to the device before the decode sizes its
    # own scratch: measured without this, the first 128-image call after build_inputs took 292 ms instead of 97)
    s.trim()
    torch.cuda.empty_cache()
    C = [len(z) for z in streams]
    weak = args.scaling == "weak" or world == 1
    lo, hi = shard(args.images, world, rank)                 # this rank's share of a 1024-image result
    n = args.images if weak else hi - lo
    first = rank * args.images if weak else lo
    do_gather = with_gather and world > 1 and not args.no_gather
    groups = max(1, min(args.groups or 2, -(-args.images // world))) if do_gather else 1
    # what this rank is about to hold: its shard's scanline scratch + rasters, the compressed inputs, on rank 0 the gathered
    # rasters of the whole job, and the pipeline's token pool (sized by the library to at most half of what is then free:
    # planned here with its cautious 3.2 token bytes per compressed byte of one group).  Checked before anything is allocated.
    U_, S_ = spng.inflated_size(W, H, DEPTH, CHANNELS, False), spng.storage_size(W, H, DEPTH, CHANNELS)
    plan = {"shard": n * (((U_ + 4096 + 255) & ~255) + S_), "inputs": sum(C),
            "gathered": args.images * S_ if (do_gather and rank == 0) else 0}
    plan["tokens_min"] = int(3.2 * max(C)) + (1 << 20)                   # (one stream's worth: the pool's lower bound)
    free_b, total_b = torch.cuda.mem_get_info(s.tdev)
    need = plan["shard"] + plan["gathered"] + plan["tokens_min"]
    assert need <= free_b, (f"rank {rank}: planned allocations {need / 2**30:.1f} GiB ({plan}) exceed the {free_b / 2**30:.1f} GiB "
                            f"free of {total_b / 2**30:.1f} GiB")
    job = DecodeJob(spng, s, torch, d_streams, n, first, unique, groups)
    S = job.S
    gathered = None
    if do_gather and rank == 0:
        gathered = torch.empty(args.images * S, dtype=torch.uint8, device=s.tdev)

    pending = []

    gather_on = [True]

    def step():
        for g in range(len(job.groups)):
            job.decode_group(g)
            if do_gather and gather_on[0]:
                # the only exchange step of the path: this group's rasters -> rank 0 over xGMI, as one batch of
                # point-to-point transfers that runs (on RCCL's stream) while the next group decodes
                glo, ghi = job.groups[g]
                # in the weak form a rank contributes the slots of its share of the 1024-image result
                off = lo if weak else 0
                ops = exchange_plan(dist, job.d_out, gathered, S, args.images, world, rank,
                                    glo, ghi, len(job.groups), g, weak_offset=off)
                if ops:
                    pending.extend(dist.batch_isend_irecv(ops))
        for w in pending:
            w.wait()
        pending.clear()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    gather_error = None
    if do_gather:
        try:
            step()
            fence()
        except Exception as exc:                               # noqa: BLE001  (the one piece a 1-GPU box cannot exercise)
            gather_error = repr(exc)[:200]
            do_gather = False
            pending.clear()
    if with_gather and world > 1 and not args.no_gather:
        # (every rank takes the same road from here on: one that lost its exchange takes it from all)
        flag = torch.tensor([1 if gather_error else 0], dtype=torch.int32, device=s.tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and do_gather:
            do_gather = False
            gather_error = gather_error or "another rank's exchange failed"
    for _ in range(args.warmup):
        step()
    fence()
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    prof = {k: s.profile_get(getattr(spng, "K_" + k.upper())) for k in STAGES}
    s.profile(False)
    # N > 1: the same steps without the exchange, so that the line shows what the gather costs on top of the decode
    dt_decode_only = None
    if do_gather:
        gather_on[0] = False
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_decode_only = time.perf_counter() - t0
        gather_on[0] = True

    # parity of the whole batch: every slot equals its source raster, every status is DONE
    res = job.results()
    U = job.U
    assert all(r.status == 0 and r.written == U for r in res), [r.status for r in res if r.status][:8]
    fast = sum(r.reserved == 1 for r in res)
    ref = [s.to_device(img.reshape(-1)) for img in images]
    for j in range(n):
        assert torch.equal(job.d_out[j * S:(j + 1) * S], ref[job.src[j]]), f"slot {first + j} differs"
    if gathered is not None and do_gather:
        for g in range(0, args.images, max(1, args.images // 16)):
            assert torch.equal(gathered[g * S:(g + 1) * S], ref[g % unique]), f"gathered image {g} differs"

    t = torch.tensor([dt], dtype=torch.float64, device=s.tdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if dt_decode_only is not None:
        t = torch.tensor([dt_decode_only], dtype=torch.float64, device=s.tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_decode_only = float(t.item())
    total_c = sum(C[k] for k in job.src)
    # what the pipeline moves BY DESIGN besides C and U: the token pages its decode stage writes and its resolve stage reads (figures
    # of the step's last call, scaled to the step when the shard went through in several calls)
    glo, ghi = [g for g in job.groups if g[1] > g[0]][-1]
    tok_pages, tok_blocks, _ = s.token_stats()
    tok_step = int(tok_pages * n / (ghi - glo)) if fast else 0
    del ref, gathered, d_streams
    job.d_rows = job.d_out = job.dres = None                  # (the slabs go back to the allocator before the next workload)
    torch.cuda.empty_cache()
    per_step = {k: prof[k][0] / args.steps for k in STAGES}
    # launches of a stage per step: the library's event spans, less the retry pass of the pipeline (launched behind every batch,
    # returns at once when no stream ran out of token pages; in kernel traces it is the <1u> instantiation)
    launches = {k: max(1, round(prof[k][1] / args.steps) - (1 if k.startswith("pinf_") else 0)) for k in STAGES}
    alg = {"pinf_find": 0, "pinf_decode": total_c, "pinf_resolve": n * U,
           "inflate": 0 if fast == n else total_c + n * U, "unfilter": n * (U + S)}
    design = {"pinf_decode": total_c + tok_step, "pinf_resolve": tok_step + n * U, "unfilter": n * (U + S)} if tok_step else {}
    return {"design": design, "token_bytes": tok_step, "dt": dt, "n": n, "weak": weak, "per_step_ms": per_step, "launches": launches, "alg": alg, "total_c": total_c, "U": U, "S": S,
            "fast": fast, "gather": bool(do_gather), "gather_error": gather_error, "hi_lo": hi - lo, "dt_decode_only": dt_decode_only,
            "groups": len(job.groups),
            "ratio": round(U * unique / sum(C), 3), "streams": streams, "images": images, "rows": rows}


def kernel_report(m, traffic):
    rep = {}
    for k in STAGES:
        ms = m["per_step_ms"][k]
        if ms <= 0.02:
            continue
        e = {"ms_per_step": round(ms, 3), "algorithmic_bytes": m["alg"][k]}
        if m.get("launches", {}).get(k, 1) > 1:
            # (a batch whose tokens do not fit the pool at once goes through in groups of streams: that many launches per step)
            e["launches_per_step"] = m["launches"][k]
            e["ms_per_launch"] = round(ms / m["launches"][k], 3)
        if m["alg"][k]:
            e["gbps"] = round(m["alg"][k] / (ms * 1e-3) / 1e9, 2)
            e["frac_of_hbm_peak"] = round(e["gbps"] / HBM_PEAK_GBPS, 4)
        if k in traffic:
            e["traffic"] = traffic[k]
        if m.get("design", {}).get(k):
            # bytes the stage moves by design: with the token stream (decode: C read + tokens written; resolve: tokens read + U
            # written) -- traffic above THIS is wasted, not traffic above the algorithmic C / U
            e["design_bytes"] = m["design"][k]
            if k in traffic and traffic[k]:
                e["traffic_over_design"] = round(traffic[k] / m["design"][k], 3)
        rep[k] = e
    return rep


# ---- the other BASELINE configs and the neighbours of the path, reported in the same line --------------------------
def copy_ceiling(torch, s):
    """Measured device copy next to the 8 TB/s spec peak: read + write bytes per second of the library's
    own 16-bytes-per-lane grid-stride copy kernel (spng_copy_ceiling) over 8 GiB, a torch tensor copy of the same buffers beside
    it, and the same bytes in the access pattern the scanline kernel had before round 4 (stores that straddle lines)."""
    n = 8 << 30
    gbps, ms = s.copy_ceiling(n, 0, 5)
    skew_gbps, skew_ms = s.copy_ceiling(n, 1, 3)
    a = torch.empty(n // 8, dtype=torch.int64, device=s.tdev)
    b = torch.empty(n // 8, dtype=torch.int64, device=s.tdev)
    a.zero_(); b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    tms = e0.elapsed_time(e1) / 5
    del a, b
    torch.cuda.empty_cache()
    return {"gbps": round(gbps, 1), "ms": round(ms, 3), "bytes": 2 * n,
            "how": "spng_copy_ceiling pattern 0: the library's 16-bytes-per-lane grid-stride copy of 8 GiB, read + write counted, HIP "
                   "events, 5 repeats",
            "torch_copy_gbps": round(2 * n / (tms * 1e-3) / 1e9, 1),
            "skewed_rows_gbps": round(skew_gbps, 1),
            "skewed_rows_how": "pattern 1: 64 rows per wave, 256-byte tiles, row r trailing row r - 1 by 4 bytes (unaligned stores): the "
                               "unfilter kernel's access pattern of rounds 1-3"}


def parallel_zlib(rows: bytes, level: int, threads: int) -> bytes:
    """A zlib stream of `rows` made of independently compressed slices (raw DEFLATE, each but the last ended with a sync
    flush), so that half a gigabyte of scanlines is compressed on all host cores in a few seconds."""
    n = len(rows)
    k = max(1, min(threads, n >> 22))
    cuts = [n * i // k for i in range(k + 1)]

    def one(i):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        return co.compress(rows[cuts[i]:cuts[i + 1]]) + co.flush(zlib.Z_FINISH if i == k - 1 else zlib.Z_SYNC_FLUSH)

    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(one, range(k)))
    return b"\x78\x9c" + b"".join(parts) + zlib.adler32(rows).to_bytes(4, "big")


def run_single_image(torch, spng, s, stream, image, steps=5):
    """What the reference does per call: ONE 4096 x 4096 RGBA8 image decoded alone (spng_decode_batch of one), stream resident
    in HBM.  The pipeline cuts the one stream into segments (decode) and parts (resolve): latency, not throughput."""
    d_z = s.to_device(stream)
    U = spng.inflated_size(W, H, DEPTH, CHANNELS, False)
    S = spng.storage_size(W, H, DEPTH, CHANNELS)
    d_rows, d_out = s.empty(U + 4096), s.empty(S)
    desc = s.image_desc(d_z, d_rows, d_out, W, H, DEPTH, CHANNELS, False, 0, rows_cap=U + 4096)
    for _ in range(2):
        s.decode_batch([desc])
    torch.cuda.synchronize()
    names = ("pinf_find", "pinf_decode", "pinf_resolve", "inflate", "unfilter")
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = s.decode_batch([desc])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / steps, 3) for k in names}
    s.profile(False)
    assert res[0].status == 0 and res[0].reserved == 1
    assert torch.equal(d_out, s.to_device(image.reshape(-1))), "single image: raster differs"
    return {"workload": "1 x 4096x4096 RGBA8 PNG decode (level 6, zlib encoder), one call per image",
            "ms": round(dt * 1e3, 3), "mpixels_per_s": round(MPIX / dt, 1), "kernels_ms": prof, "bit_exact": True}


def run_shard_probe(torch, spng, s, streams, images, steps=3):
    """The per-GPU unit of BASELINE configs[2] at N = 8, measured where it can be measured: 128 of the headline's images in ONE
    spng_decode_batch call (and in two calls of 64, the form whose rasters could travel while the second half decodes) on this one
    GPU -- what a rank of the 8-GPU strong-scaling run computes per step before any gather.  Linear share of the headline step =
    ms_per_step / 8."""
    unique = len(streams)
    d_streams = [s.to_device(z) for z in streams]
    out = {"workload": "128 x 4096x4096 RGBA8 (the headline's streams) per call on ONE GPU: the shard of configs[2] at N = 8"}
    ref = [s.to_device(img.reshape(-1)) for img in images]
    for groups in (1, 2):
        job = DecodeJob(spng, s, torch, d_streams, 128, 0, unique, groups)
        for _ in range(2):
            for g in range(groups):
                job.decode_group(g)
        torch.cuda.synchronize()
        s.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            for g in range(groups):
                job.decode_group(g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / steps, 2) for k in STAGES}
        s.profile(False)
        res = job.results()
        assert all(r.status == 0 and r.written == job.U and r.reserved == 1 for r in res)
        for j in range(0, 128, 9):
            assert torch.equal(job.d_out[j * job.S:(j + 1) * job.S], ref[job.src[j]]), f"shard probe: slot {j} differs"
        out[f"calls_{groups}"] = {"ms": round(dt * 1e3, 2), "mpixels_per_s": round(128 * MPIX / dt, 1), "stages_ms": prof}
        job.d_rows = job.d_out = job.dres = None
        del job
        torch.cuda.empty_cache()
    return out


def run_config5(torch, spng, s, steps):
    """BASELINE configs[4]: one 8192 x 8192 RGBA16 Adam7 image (536,886,272 inflated bytes, seven sub-images, one
    stream), level 6, decoded on one GPU: inflate pipeline + per-pass unfilter + scatter."""
    import numpy as np
    from swift_png_amd import synth
    w = h = 8192
    tile = synth.image(11, 2048, 2048, 4, 16)                       # (rows of 2048 * 8 bytes)
    img = np.tile(tile, (4, 4))
    raw = img.tobytes()
    rows = s.filter(raw, w, h, 16, 4, True)
    z = parallel_zlib(rows, 6, min(os.cpu_count() or 1, 64))
    d_idat, d_rows, d_out = s.to_device(z), s.empty(len(rows) + 4096), s.empty(len(raw))
    desc = s.image_desc(d_idat, d_rows, d_out, w, h, 16, 4, True, 0, rows_cap=len(rows) + 4096)
    s.decode_batch([desc])
    torch.cuda.synchronize()
    names = ("pinf_find", "pinf_decode", "pinf_resolve", "inflate", "unfilter", "scatter")
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = s.decode_batch([desc])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = {k: round(s.profile_get(getattr(spng, "K_" + k.upper()))[0] / steps, 3) for k in names}
    s.profile(False)
    assert res[0].status == 0 and res[0].written == len(rows)
    want = s.to_device(raw)
    assert torch.equal(d_out, want), "config 5: raster differs"
    alg = len(z) + 2 * len(rows) + len(raw)                           # inflate: C + U; unfilter (+ scatter): U + S
    return {"workload": "1 x 8192x8192 RGBA16 Adam7 PNG decode, level 6 (host zlib, 64 slices), one stream; BASELINE configs[4]",
            "ms": round(dt * 1e3, 2), "mpixels_per_s": round(w * h / 1e6 / dt, 1), "compressed_bytes": len(z),
            "inflated_bytes": len(rows), "algorithmic_bytes": alg, "gbps": round(alg / dt / 1e9, 2),
            "frac_of_hbm_peak": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 5), "pipeline": res[0].reserved == 1,
            "kernels_ms": prof, "bit_exact": True}


def run_file_to_pixels(torch, spng, s, streams, images, n, steps):
    """File -> pixels, as the reference times it (Benchmarks/Decompression/Swift/Main.swift:103-109: decompress +
    unpack(as: RGBA<UInt8>)): spng_lex_batch (chunk walk, CRC-32 of every chunk, IDAT assembly) -> spng_decode_batch ->
    spng_unpack_batch over n PNG files resident in HBM (the level-6 streams of the headline, cut into 64 KiB IDATs)."""
    import struct
    U = spng.inflated_size(W, H, DEPTH, CHANNELS, False)
    S = spng.storage_size(W, H, DEPTH, CHANNELS)

    def png_file(z):
        def chunk(t, body):
            return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
        out = [bytes([137, 80, 78, 71, 13, 10, 26, 10]), chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 6, 0, 0, 0))]
        out += [chunk(b"IDAT", z[i:i + 65536]) for i in range(0, len(z), 65536)]
        out.append(chunk(b"IEND", b""))
        return b"".join(out)

    files = [png_file(z) for z in streams]
    d_files = [s.to_device(f) for f in files]
    unique = len(files)
    cap = max(len(f) for f in files)
    d_idat = torch.empty(n * cap, dtype=torch.uint8, device=s.tdev)
    d_rows = torch.empty(n * (U + 4096), dtype=torch.uint8, device=s.tdev)
    d_sto = torch.empty(n * S, dtype=torch.uint8, device=s.tdev)
    d_rgba = torch.empty(n * S, dtype=torch.uint8, device=s.tdev)
    fdescs = (spng.FileDesc * n)()
    for j in range(n):
        f = d_files[j % unique]
        fdescs[j] = spng.FileDesc(f.data_ptr(), f.numel(), d_idat.data_ptr() + j * cap, cap)
    infos = (spng.Lexed * n)()
    idescs = (spng.ImageDesc * n)()
    udescs = (spng.UnpackDesc * n)()
    dres = s.empty(n * ctypes.sizeof(spng.Result))

    def step(first):
        assert s.lib.spng_lex_batch(s.ctx, fdescs, n, None, infos) == 0       # (the IHDR fields come back to the host)
        if first:
            for j in range(n):
                r = infos[j]
                assert r.status == 0 and (r.width, r.height, r.depth, r.color, r.interlace) == (W, H, 8, 6, 0)
                idescs[j] = spng.ImageDesc(d_idat.data_ptr() + j * cap, r.idat_len, d_rows.data_ptr() + j * (U + 4096), U + 4096,
                                           d_sto.data_ptr() + j * S, W, H, 8, 4, 0, 0, 0)
                udescs[j] = spng.UnpackDesc(d_sto.data_ptr() + j * S, d_rgba.data_ptr() + j * S, None, W, H, 0, (ctypes.c_uint16 * 3)(),
                                            8, 4, 0, 0, 0, 8)
        assert s.lib.spng_decode_batch(s.ctx, idescs, n, ctypes.c_void_p(dres.data_ptr()), None) == 0
        assert s.lib.spng_unpack_batch(s.ctx, udescs, n) == 0

    step(True)
    torch.cuda.synchronize()
    s.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lex_ms, unp_ms = s.profile_get(spng.K_LEX)[0] / steps, s.profile_get(spng.K_UNPACK)[0] / steps
    s.profile(False)
    ref = [s.to_device(img.reshape(-1)) for img in images]
    for j in range(0, n, max(1, n // 16)):
        assert torch.equal(d_rgba[j * S:(j + 1) * S], ref[j % unique]), f"file {j}: RGBA8 differs from its source raster"
    total_f = sum(len(files[j % unique]) for j in range(n))
    return {"workload": f"{n} PNG files (4096x4096 RGBA8, level 6, 64 KiB IDATs) in HBM -> lex + CRC-32 -> inflate -> unfilter -> "
                        f"RGBA<UInt8>", "ms_per_step": round(dt * 1e3, 2), "mpixels_per_s": round(n * MPIX / dt, 1),
            "kernels": {"lex": {"ms_per_step": round(lex_ms, 3), "algorithmic_bytes": total_f,
                                "gbps": round(total_f / (lex_ms * 1e-3) / 1e9, 2) if lex_ms else None},
                        "unpack": {"ms_per_step": round(unp_ms, 3), "algorithmic_bytes": 2 * n * S,
                                   "gbps": round(2 * n * S / (unp_ms * 1e-3) / 1e9, 2) if unp_ms else None,
                                   "frac_of_hbm_peak": round(2 * n * S / (unp_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if unp_ms else None}},
            "bit_exact": True}


def run_scanline_formats(torch, spng, s, n=256):
    """The scanline kernels on the pixel sizes the headline does not touch: n x 4096^2 RGB8 (bpp 3: the most common PNG format) and
    RGB16 (bpp 6) -- filter-select (spng_filter_batch) and defilter (spng_unfilter_batch) alone, each against the HBM roofline
    (algorithmic bytes U + S per image and direction), the defiltered rasters equal to the sources."""
    from swift_png_amd import synth
    out = {}
    import numpy as np
    # (indexed8 / gray1: the byte-wise kernel of the 1- and 2-byte pixel formats and of the sub-byte ones -- what palette and bilevel
    # PNGs go through; storage of a sub-byte format is one unscaled sample per byte, its scanlines are packed)
    for name, depth, ch in (("rgb8", 8, 3), ("rgb16", 16, 3), ("indexed8", 8, 1), ("gray1", 1, 1)):
        m = n if depth == 8 and ch == 3 else n // 2
        U = spng.inflated_size(W, H, depth, ch, False)
        S = spng.storage_size(W, H, depth, ch)
        unique = 4
        base = [synth.image(500 + k, W, H).reshape(H, W, 4) for k in range(unique)]
        if name == "rgb8":
            srcs = [s.to_device(np.ascontiguousarray(im[..., :3]).tobytes()) for im in base]
        elif name == "indexed8":
            srcs = [s.to_device(np.ascontiguousarray((im[..., 0] >> 2) + (im[..., 1] >> 6)).astype(np.uint8).tobytes()) for im in base]   # (palette indices with structure)
        elif name == "gray1":
            srcs = [s.to_device(np.ascontiguousarray(im[..., 1] >> 7).astype(np.uint8).tobytes()) for im in base]                           # (one sample, 0 / 1, per byte)
        else:
            srcs = [s.to_device(np.repeat(im[..., :3], 2, axis=-1).tobytes()) for im in base]   # (big-endian samples v << 8 | v)
        d_sto = torch.empty(m * S, dtype=torch.uint8, device=s.tdev)
        for j in range(m):
            d_sto[j * S:(j + 1) * S] = srcs[j % unique]
        d_rows = torch.empty(m * U, dtype=torch.uint8, device=s.tdev)
        d_back = torch.empty(m * S, dtype=torch.uint8, device=s.tdev)
        fd = (spng.ImageDesc * m)()
        ud = (spng.ImageDesc * m)()
        for j in range(m):
            fd[j] = spng.ImageDesc(None, 0, d_rows.data_ptr() + j * U, U, d_sto.data_ptr() + j * S, W, H, depth, ch, 0, 0, 0)
            ud[j] = spng.ImageDesc(None, 0, d_rows.data_ptr() + j * U, U, d_back.data_ptr() + j * S, W, H, depth, ch, 0, 0, 0)
        dres = s.empty(m * ctypes.sizeof(spng.Result))
        torch.cuda.synchronize()                               # (the fills above ran on torch's stream, the kernels run on the context's)
        assert s.lib.spng_filter_batch(s.ctx, fd, m, ctypes.c_void_p(dres.data_ptr()), None) == 0
        assert s.lib.spng_unfilter_batch(s.ctx, ud, m, None, ctypes.c_void_p(dres.data_ptr()), None) == 0
        torch.cuda.synchronize()
        if not torch.equal(d_back, d_sto):
            d = (d_back != d_sto).nonzero()[:, 0]
            rowb = max(1, W * ch * depth // 8 if depth >= 8 else W)
            raise AssertionError(f"{name}: defiltered rasters differ from their sources: {len(d)} bytes, first at image {int(d[0]) // S} row "
                                 f"{int(d[0]) % S // rowb} byte {int(d[0]) % rowb}, last at image {int(d[-1]) // S}")
        s.profile(True)
        for _ in range(3):
            assert s.lib.spng_filter_batch(s.ctx, fd, m, ctypes.c_void_p(dres.data_ptr()), None) == 0
            assert s.lib.spng_unfilter_batch(s.ctx, ud, m, None, ctypes.c_void_p(dres.data_ptr()), None) == 0
        torch.cuda.synchronize()
        f_ms, u_ms = s.profile_get(spng.K_FILTER)[0] / 3, s.profile_get(spng.K_UNFILTER)[0] / 3
        s.profile(False)
        moved = m * (U + S)
        extra = {}
        if name == "indexed8":
            # what libpng writes for palette images: every row filtered with None -- defiltering is a pass through the kernel
            plain = [torch.cat([torch.zeros(H, 1, dtype=torch.uint8, device=s.tdev), sr.view(H, -1)], dim=1).reshape(-1) for sr in srcs]
            for j in range(m):
                d_rows[j * U:(j + 1) * U] = plain[j % unique]
            torch.cuda.synchronize()
            assert s.lib.spng_unfilter_batch(s.ctx, ud, m, None, ctypes.c_void_p(dres.data_ptr()), None) == 0
            torch.cuda.synchronize()
            assert torch.equal(d_back, d_sto), "indexed8, all rows None: rasters differ"
            s.profile(True)
            for _ in range(3):
                assert s.lib.spng_unfilter_batch(s.ctx, ud, m, None, ctypes.c_void_p(dres.data_ptr()), None) == 0
            torch.cuda.synchronize()
            n_ms = s.profile_get(spng.K_UNFILTER)[0] / 3
            s.profile(False)
            extra = {"unfilter_all_rows_none": {"ms": round(n_ms, 3), "gbps": round(moved / (n_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(moved / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}}
            del plain
        out[name] = {"images": m, "algorithmic_bytes": moved, **extra,
                     "unfilter": {"ms": round(u_ms, 3), "gbps": round(moved / (u_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(moved / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                     "filter": {"ms": round(f_ms, 3), "gbps": round(moved / (f_ms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(moved / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
                     "bit_exact": True}
        del d_sto, d_rows, d_back, srcs
        torch.cuda.empty_cache()
    return out


def run_pixels_to_file(torch, spng, s, n, size, level=9):
    """Pixels -> file, the mirror of file_to_pixels (PNG.Image.init(packing:size:layout:) + compress): spng_pack_batch ([RGBA<UInt8>]
    -> storage of an rgb8 image: alpha dropped) -> spng_encode_batch (filter-select + DEFLATE at the reference's default level 9) ->
    spng_write_idat_batch (IDAT chunks with their CRC-32) over n synthetic size x size photographs; the pack kernel alone on the same
    pixels for its roofline; the files' IDAT payloads decoded again (lex -> decode -> unpack) must give the pixels back."""
    import numpy as np
    from swift_png_amd import synth
    w = h = size
    unique = min(8, n)
    px = [s.to_device(synth.image(300 + k, w, h).tobytes()) for k in range(unique)]       # r, g, b, a per pixel
    P = w * h * 4
    S = spng.storage_size(w, h, 8, 3)
    U = spng.inflated_size(w, h, 8, 3, False)
    cap = s.lib.spng_deflate_bound(U)
    fcap = cap + 12 * (cap // 65536 + 2)
    d_px = torch.empty(n * P, dtype=torch.uint8, device=s.tdev)
    for j in range(n):
        d_px[j * P:(j + 1) * P] = px[j % unique]
    d_sto = torch.empty(n * S, dtype=torch.uint8, device=s.tdev)
    d_rows = torch.empty(n * U, dtype=torch.uint8, device=s.tdev)
    d_z = torch.empty(n * cap, dtype=torch.uint8, device=s.tdev)
    d_file = torch.empty(n * fcap, dtype=torch.uint8, device=s.tdev)
    pdescs = (spng.PackDesc * n)()
    idescs = (spng.ImageDesc * n)()
    for j in range(n):
        pdescs[j] = spng.PackDesc(d_px.data_ptr() + j * P, d_sto.data_ptr() + j * S, None, w, h, 0, 8, 3, 0, 0, 8, spng.TARGET_RGBA)
        idescs[j] = spng.ImageDesc(d_z.data_ptr() + j * cap, cap, d_rows.data_ptr() + j * U, U, d_sto.data_ptr() + j * S, w, h, 8, 3, 0, 0, 0)
    dres = s.empty(n * ctypes.sizeof(spng.Result))
    res = (spng.Result * n)()
    cres = (spng.Result * n)()
    cdescs = (spng.ChunkingDesc * n)()

    def step():
        assert s.lib.spng_pack_batch(s.ctx, pdescs, n) == 0
        assert s.lib.spng_encode_batch(s.ctx, idescs, level, n, None, res) == 0      # (the stream lengths come back to the host)
        for j in range(n):
            assert res[j].status == 0
            cdescs[j] = spng.ChunkingDesc(d_z.data_ptr() + j * cap, res[j].written, d_file.data_ptr() + j * fcap, fcap, 65536)
        assert s.lib.spng_write_idat_batch(s.ctx, cdescs, n, None, cres) == 0

    torch.cuda.synchronize()                                   # (the fills ran on torch's stream, the kernels run on the context's)
    step()
    torch.cuda.synchronize()
    s.profile(True)
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {k: s.profile_get(getattr(spng, "K_" + k.upper()))[0] for k in ("pack", "filter", "deflate", "lex")}
    s.profile(False)
    # back again: the IDAT payloads of file 0 .. unique - 1 -> storage -> RGBA<UInt8> = the pixels with alpha 255
    for j in range(unique):
        z = bytes(d_z[j * cap:j * cap + res[j].written].cpu().numpy())
        st, storage, _ = s.decode(z, w, h, 8, 3, False)
        back = np.frombuffer(s.unpack(storage, w, h, 8, 3, target=8), dtype=np.uint8).reshape(-1, 4)
        want = np.frombuffer(bytes(px[j].cpu().numpy()), dtype=np.uint8).reshape(-1, 4)
        assert st == 0 and (back[:, :3] == want[:, :3]).all() and (back[:, 3] == 255).all(), f"file {j} does not decode to its pixels"
        assert cres[j].written == res[j].written + 12 * ((res[j].written + 65535) // 65536)
    # the pack kernel by itself on 4096^2 images, both ways it moves bytes: rgba8 (16 bytes per lane each way) and rgb8 (alpha dropped)
    packs = {}
    m = 64
    big = torch.randint(0, 256, (m * 4096 * 4096 * 4,), dtype=torch.uint8, device=s.tdev)
    out = torch.empty(m * 4096 * 4096 * 4, dtype=torch.uint8, device=s.tdev)
    for name, ch in (("rgba8", 4), ("rgb8", 3)):
        bd = (spng.PackDesc * m)()
        for j in range(m):
            bd[j] = spng.PackDesc(big.data_ptr() + j * (1 << 26), out.data_ptr() + j * 4096 * 4096 * ch, None, 4096, 4096, 0, 8, ch, 0, 0, 8, spng.TARGET_RGBA)
        torch.cuda.synchronize()
        assert s.lib.spng_pack_batch(s.ctx, bd, m) == 0
        torch.cuda.synchronize()
        s.profile(True)
        for _ in range(3):
            assert s.lib.spng_pack_batch(s.ctx, bd, m) == 0
        torch.cuda.synchronize()
        ms = s.profile_get(spng.K_PACK)[0] / 3
        s.profile(False)
        moved = m * 4096 * 4096 * (4 + ch)
        packs[name] = {"ms": round(ms, 3), "algorithmic_bytes": moved, "gbps": round(moved / (ms * 1e-3) / 1e9, 1),
                       "frac_of_hbm_peak": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        v = out[:4096 * ch].cpu().numpy().reshape(-1, ch)
        assert (v == big[:4096 * 4].cpu().numpy().reshape(-1, 4)[:, :ch]).all()
    total_c = sum(r.written for r in res)
    return {"workload": f"{n} x {w}x{h} [RGBA<UInt8>] in HBM -> pack (rgb8) -> filter-select -> DEFLATE level {level} -> IDAT chunks + CRC-32",
            "ms_per_step": round(dt * 1e3, 2), "mpixels_per_s": round(n * w * h / 1e6 / dt, 1), "compressed_ratio": round(n * U / total_c, 3),
            "kernels_ms": {k: round(v, 3) for k, v in prof.items()}, "pack_kernel_64x4096x4096": packs, "decodes_to_its_pixels": True}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=("decode", "encode"), default="decode")
    ap.add_argument("--images", type=int, default=1024, help="batch size (BASELINE: 1024)")
    ap.add_argument("--unique", type=int, default=32, help="distinct images; slot i holds image i mod unique")
    ap.add_argument("--streams", choices=("zlib", "swiftpng"), default="swiftpng",
                    help="level-6 encoder of the headline's input streams: the device deflater = swift-png's own bitstream "
                         "(BASELINE.md section 1: what generates the config-2 inputs), or host zlib")
    ap.add_argument("--no-swiftpng", "--no-alt", dest="no_swiftpng", action="store_true",
                    help="skip the second measurement (the same batch from the other encoder)")
    ap.add_argument("--swiftpng-unique", "--alt-unique", dest="swiftpng_unique", type=int, default=16,
                    help="distinct images of the second measurement (>= 10: the compressed input exceeds the 256 MiB Infinity Cache)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: --images in total, sharded (strong, BASELINE configs[2]); or --images per GPU (weak)")
    ap.add_argument("--groups", type=int, default=0,
                    help="N > 1: calls a rank cuts its shard into, each one's rasters leaving for rank 0 while the next decodes "
                         "(0 = automatic: 2 -- a call pays every stream's serial resolve once, so 128 images cost 99 ms in one call, "
                         "111 ms in two and 134 ms in four (profiles/r03_probe_groups.log); with two, the first half's rasters (4 GiB, "
                         "28 ms on one xGMI link) travel under the second half's decode and only the second half's gather is exposed)")
    ap.add_argument("--level", type=int, default=9, help="encode mode: DEFLATE level (BASELINE configs[3]: 9)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="decode mode: skip the copy ceiling, configs[3] / configs[4] and file -> pixels legs")
    ap.add_argument("--legs", default="", help="decode mode: comma list of the extra legs to run (default: all of them)")
    ap.add_argument("--traffic", action="store_true",
                    help="measure `traffic` inside this invocation: two rocprofv3 --pmc sub-runs of the headline workload first (about 3 min)")
    ap.add_argument("--encode-steps", type=int, default=3, help="decode mode: timed steps of the two encode legs (after one warm-up)")
    ap.add_argument("--encode-images", type=int, default=1024,
                    help="decode mode: images of the configs[3] leg (BASELINE: 1024)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))

    measured_now = False
    if args.traffic and args.gpus == 1 and args.mode == "decode":
        try:
            measured_now = measure_traffic(args)
        except Exception as exc:                               # noqa: BLE001  (the line still carries the committed figures, labelled)
            print(f"--traffic: PMC sub-runs failed ({repr(exc)[:200]}); using profiles/{PMC_FILE}", file=sys.stderr)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import swift_png_amd as spng
    s = spng.load(local)
    cores = host_cores()

    if args.mode == "encode":
        from bench_encode import run_encode
        out = run_encode(args, torch, dist, spng, s, rank, world)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    m = run_decode(args, torch, dist, spng, s, rank, world, args.streams, args.unique, True)
    other = None
    if world == 1 and not args.no_swiftpng:
        alt = "swiftpng" if args.streams == "zlib" else "zlib"
        try:
            other = (alt, run_decode(args, torch, dist, spng, s, rank, world, alt, args.swiftpng_unique, False))
        except Exception as exc:                               # noqa: BLE001  (never lose the headline to the second workload)
            other = (alt, {"error": repr(exc)[:300]})
    if rank == 0:
        weak, n = m["weak"], m["n"]
        ms = m["dt"] / args.steps * 1e3
        kernels = kernel_report(m, pmc_traffic(args.streams, args.images, args.unique) if world == 1 else {})
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        d = kernels[dominant]
        infl_ms = sum(m["per_step_ms"][k] for k in ("pinf_find", "pinf_decode", "pinf_resolve", "inflate"))
        out = {
            "metric": "decoded_mpixels_per_s",
            "value": round(args.images * (world if weak else 1) * MPIX / (m["dt"] / args.steps), 1),
            "unit": "MPixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True,
            # (N = 1 is the first point of the strong-scaling series the driver assembles: the same 1024 images)
            "scaling": "weak" if (weak and world > 1) else "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.images} x 4096x4096 RGBA8 PNG decode (inflate+unfilter), mixed filters "
                                   f"(reference heuristic), level 6 ({'swift-png-made streams: the device deflater, bit-exact with LZ77.Deflator' if args.streams == 'swiftpng' else 'zlib-made streams'}); BASELINE configs[1]"
                                   + ("" if world == 1 else
                                      f" per GPU (global batch {args.images * world}); each rank's {m['hi_lo']}-image share "
                                      f"gathered to rank 0 over RCCL" if weak else
                                      f" sharded {n}/GPU, rasters gathered to rank 0 over RCCL while decoding (configs[2])"),
                       "unique_images": args.unique, "compressed_ratio": m["ratio"],
                       "parallel_inflate_streams": m["fast"], "serial_inflate_streams": n - m["fast"],
                       "gather": m["gather"], **({"gather_error": m["gather_error"]} if m["gather_error"] else {})},
            "inflate_gbps": round(n * m["U"] / (infl_ms * 1e-3) / 1e9, 2),
            "inflate_pipeline": {"ms_per_step": round(infl_ms, 3), "algorithmic_bytes": m["total_c"] + n * m["U"],
                                 "gbps": round((m["total_c"] + n * m["U"]) / (infl_ms * 1e-3) / 1e9, 2)},
            "roofline": {"bound": "hbm", "kernel": dominant + "_kernel", "achieved": d.get("gbps"),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d.get("frac_of_hbm_peak"),
                         "traffic": (d.get("traffic") // d.get("launches_per_step", 1)) if d.get("traffic") else d.get("traffic"),
                         "ms_per_launch": d.get("ms_per_launch", d["ms_per_step"]), "launches_per_step": d.get("launches_per_step", 1),
                         "algorithmic_bytes_per_launch": d["algorithmic_bytes"] // d.get("launches_per_step", 1),
                         **({"traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE sub-runs of this invocation (--traffic)" if measured_now else
                                                "profiles/" + PMC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, not measured in this "
                                                "run; `traffic_provenance` says on which sources)"),
                             "traffic_provenance": traffic_provenance(args.streams)} if d.get("traffic") else {})},
            "kernels": kernels,
        }
        if world > 1:
            # what the exchange costs: the same steps with and without it, and the floor of the gather itself -- every peer's share
            # over its own xGMI link into rank 0 (7 links x ~153 GB/s per GPU both ways: ~76.5 GB/s per link and direction,
            # MI355X_MICROARCH.md); with g groups per shard the last group's rasters cannot hide behind a decode
            per_peer = m["hi_lo"] * m["S"]
            floor_ms = per_peer / 76.5e9 * 1e3
            out["multi"] = {"with_gather_mpixels_per_s": out["value"],
                            "decode_only_mpixels_per_s": round(args.images * (world if weak else 1) * MPIX / (m["dt_decode_only"] / args.steps), 1) if m.get("dt_decode_only") else None,
                            "decode_only_ms_per_step": round(m["dt_decode_only"] / args.steps * 1e3, 3) if m.get("dt_decode_only") else None,
                            "groups_per_shard": m.get("groups"), "gathered_bytes_per_peer": per_peer,
                            "xgmi_floor_ms_whole_share": round(floor_ms, 2), "xgmi_floor_ms_last_group": round(floor_ms / max(1, m.get("groups") or 1), 2),
                            "link_gbps_assumed_per_direction": 76.5}
        if other:
            alt, mo = other
        if other and "error" in other[1]:
            out[alt + "_streams"] = other[1]
        elif other:
            ko = kernel_report(mo, pmc_traffic(alt, args.images, args.swiftpng_unique))
            io = sum(mo["per_step_ms"][k] for k in ("pinf_find", "pinf_decode", "pinf_resolve", "inflate"))
            out[alt + "_streams"] = {
                "value": round(args.images * MPIX / (mo["dt"] / args.steps), 1), "unit": "MPixels/s",
                "ms_per_step": round(mo["dt"] / args.steps * 1e3, 3), "unique_images": args.swiftpng_unique,
                "compressed_ratio": mo["ratio"], "parallel_inflate_streams": mo["fast"],
                "inflate_gbps": round(mo["n"] * mo["U"] / (io * 1e-3) / 1e9, 2), "kernels": ko}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m["streams"], m["images"], m["rows"], cores)
        if world == 1 and not args.no_extras:
            # the other BASELINE configs and the path's neighbours, each on a bounded workload (never lose the headline to one)
            only = set(args.legs.split(",")) if args.legs else None

            def leg(name, fn):
                if only is not None and name not in only:
                    return
                try:
                    s.trim()                                   # (the context's scratch of the leg before: each leg sizes its own)
                    torch.cuda.empty_cache()
                    out[name] = fn()
                except Exception as exc:                       # noqa: BLE001
                    out[name] = {"error": repr(exc)[:300]}
            leg("copy_ceiling", lambda: copy_ceiling(torch, s))
            if isinstance(out.get("copy_ceiling", {}).get("gbps"), float) and "unfilter" in kernels:
                kernels["unfilter"]["frac_of_copy_ceiling"] = round(kernels["unfilter"]["gbps"] / out["copy_ceiling"]["gbps"], 4)
            leg("single_image", lambda: run_single_image(torch, spng, s, m["streams"][0], m["images"][0]))

            def shard_probe():
                r = run_shard_probe(torch, spng, s, m["streams"][:8], m["images"][:8])
                r["linear_share_of_headline_ms"] = round(ms / 8, 2)
                return r
            leg("shard_probe", shard_probe)

            def small():
                from bench_small import run_small_images
                return run_small_images(torch, spng, s, 8192, 3, cpu=not args.no_cpu_baseline, cores=cores)
            leg("small_images", small)
            leg("config5", lambda: run_config5(torch, spng, s, 3))
            leg("file_to_pixels", lambda: run_file_to_pixels(torch, spng, s, m["streams"][:8], m["images"][:8], min(256, args.images), 2))

            def enc():
                from bench_encode import run_encode
                ea = argparse.Namespace(**vars(args))
                ea.images, ea.unique, ea.steps, ea.warmup, ea.no_cpu_baseline = args.encode_images, min(8, args.encode_images), args.encode_steps, 1, args.no_cpu_baseline   # (the first call of a context allocates the deflate slab)
                return run_encode(ea, torch, dist, spng, s, rank, world)
            leg("encode", enc)

            def enc6():
                # the level the decode headline's inputs are made with, on the structured rasters they are made from (VERDICT r4:
                # levels 0-7 through the chip-wide search; whole-stream digest of stream 0 against the oracle's)
                from bench_encode import run_encode
                ea = argparse.Namespace(**vars(args))
                ea.images, ea.unique, ea.steps, ea.warmup, ea.no_cpu_baseline, ea.level = args.encode_images, min(8, args.encode_images), args.encode_steps, 1, args.no_cpu_baseline, 6
                return run_encode(ea, torch, dist, spng, s, rank, world, rasters_kind="synthetic")
            leg("encode_level6", enc6)

            leg("pixels_to_file", lambda: run_pixels_to_file(torch, spng, s, 256, 1024))
            leg("scanline_formats", lambda: run_scanline_formats(torch, spng, s, min(256, args.images)))

            def enc_photo():
                from bench_encode import run_encode_photographic
                return run_encode_photographic(torch, spng, s, args.level, cpu=not args.no_cpu_baseline)
            leg("encode_photographic", enc_photo)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
