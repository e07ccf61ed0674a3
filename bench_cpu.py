"""bench_cpu.py -- the `cpu_baseline` leg of bench.py: the CPU oracle (the C restatement of swift-png's CPU path,
oracle/liboracle.so) on the host cores of the machine the benchmark runs on.  Test infrastructure used as a REPORTED
baseline only; nothing here is part of the product.

Run as a fresh interpreter (no torch, no HIP runtime in this process tree), one worker PROCESS per core with its buffers
allocated and touched before the clock starts:

    python bench_cpu.py decode <dir with stream files z0, z1, ...> <cores> <tasks> <w> <h>
    python bench_cpu.py deflate <file with scanline bytes> <cores> <level> <slice bytes>
    python bench_cpu.py files <dir with *.baseline.png> <cores> <tasks>       (bench_small.py: whole small PNG files)

prints one JSON object: {"wall_s": ..., "tasks": ..., "cores": ..., "one_core_s": [...per-task seconds of worker 0...]}
"""
import ctypes
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "tests"))

_state = {}


def _init_decode(paths, w, h):
    import pnghelp as ph
    lib = ph.oracle()
    streams = [np.frombuffer(Path(p).read_bytes(), dtype=np.uint8) for p in paths]
    storage = np.zeros(w * h * 4, dtype=np.uint8)          # (touched: the clock never sees a page fault)
    _state.update(lib=lib, ph=ph, streams=streams, storage=storage, w=w, h=h)


def _decode(k):
    lib, ph, st = _state["lib"], _state["ph"], _state
    z = st["streams"][k % len(st["streams"])]
    aux = (ctypes.c_uint64 * 2)()
    t0 = time.perf_counter()
    rc = lib.orc_decode(ph._ptr(z), len(z), 0, st["w"], st["h"], 8, 4, 0, ph._ptr(st["storage"]), aux)
    return rc, time.perf_counter() - t0


def _init_deflate(path, level, nslice):
    import pnghelp as ph
    _state.update(ph=ph, data=Path(path).read_bytes(), level=level, nslice=nslice)
    ph.orc_deflate(_state["data"][:4096], level)               # (library loaded, tables built)


def _deflate(k):
    st = _state
    n = len(st["data"]) // st["nslice"] * st["nslice"]
    off = (k * st["nslice"]) % max(n, 1)
    t0 = time.perf_counter()
    out = st["ph"].orc_deflate(st["data"][off:off + st["nslice"]], st["level"])
    return len(out), time.perf_counter() - t0


def _init_files(d):
    import pnghelp as ph
    lib = ph.oracle()
    pngs = [ph.parse_png(p.read_bytes()) for p in sorted(Path(d).glob("*.baseline.png"))]
    items = []
    for p in pngs:
        z = np.frombuffer(p.idat, dtype=np.uint8)
        storage = np.zeros(lib.orc_storage_size(p.width, p.height, p.depth, p.channels), dtype=np.uint8)
        items.append((z, storage, p))
    _state.update(lib=lib, ph=ph, items=items)


def _file(k):
    lib, ph = _state["lib"], _state["ph"]
    z, storage, p = _state["items"][k % len(_state["items"])]
    aux = (ctypes.c_uint64 * 2)()
    t0 = time.perf_counter()
    rc = lib.orc_decode(ph._ptr(z), len(z), 0, p.width, p.height, p.depth, p.channels, int(p.interlaced), ph._ptr(storage), aux)
    return rc, time.perf_counter() - t0


def _noop(_):
    return 0


def main():
    mode = sys.argv[1]
    if mode == "decode":
        d, cores, tasks, w, h = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        paths = sorted(str(p) for p in Path(d).glob("z*"))
        with mp.Pool(cores, initializer=_init_decode, initargs=(paths, w, h)) as pool:
            pool.map(_decode, range(cores), chunksize=1)                 # every worker up, library loaded, pages touched
            t0 = time.perf_counter()
            res = pool.map(_decode, range(tasks), chunksize=max(1, tasks // (cores * 4)))
            wall = time.perf_counter() - t0
        assert all(rc == 0 for rc, _ in res)
        print(json.dumps({"wall_s": wall, "tasks": tasks, "cores": cores, "task_s": sorted(t for _, t in res)[len(res) // 2]}))
    elif mode == "deflate":
        path, cores, level, nslice = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
        tasks = cores
        with mp.Pool(cores, initializer=_init_deflate, initargs=(path, level, nslice)) as pool:
            pool.map(os.getpid if False else _noop, range(cores), chunksize=1)      # every worker up
            t0 = time.perf_counter()
            res = pool.map(_deflate, range(tasks), chunksize=1)
            wall = time.perf_counter() - t0
        print(json.dumps({"wall_s": wall, "tasks": tasks, "cores": cores, "slice": nslice,
                          "task_s": sorted(t for _, t in res)[len(res) // 2]}))
    elif mode == "files":
        d, cores, tasks = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        with mp.Pool(cores, initializer=_init_files, initargs=(d,)) as pool:
            pool.map(_file, range(28 * cores), chunksize=28)                 # every worker up, every file decoded once, pages touched
            reps = 1
            while True:                                                      # (small images: repeat until the clock has something to see)
                t0 = time.perf_counter()
                res = pool.map(_file, range(tasks * reps), chunksize=max(1, tasks * reps // (cores * 4)))
                wall = time.perf_counter() - t0
                if wall >= 2.0 or reps >= 64:
                    break
                reps *= 4
        assert all(rc == 0 for rc, _ in res)
        print(json.dumps({"wall_s": wall, "tasks": tasks * reps, "cores": cores, "task_s": sum(t for _, t in res) / len(res)}))
    else:
        raise SystemExit("mode")


if __name__ == "__main__":
    main()
