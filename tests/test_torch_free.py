"""The C ABI stands on its own: a process that never imports torch (the situation of a Swift host,
INTEGRATION.md sections 2-3) creates a context with its own stream, and decodes / filters / deflates
host buffers through plain ctypes.  Checked against the committed golden digests and, for the encode
side, by inflating the result with zlib.  Also runs the driver's smoke() in a fresh interpreter."""
import subprocess
import sys

import pytest

import pnghelp as ph

pytestmark = pytest.mark.gpu

CHILD = r"""
import ctypes, hashlib, json, sys, zlib
from pathlib import Path
root = Path(sys.argv[1])
sys.path.insert(0, str(root / "tests"))
import pnghelp as ph                       # numpy + zlib only
assert "torch" not in sys.modules

lib = ctypes.CDLL(str(root / "swift_png_amd" / "libspng_mi355.so"))
class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("reserved", ctypes.c_int32), ("written", ctypes.c_uint64),
                ("consumed", ctypes.c_uint64), ("aux", ctypes.c_uint64 * 2)]
vp, u64, u32, i32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int
rp = ctypes.POINTER(Result)
lib.spng_last_error_string.restype = ctypes.c_char_p
lib.spng_create.argtypes = [ci, vp, ctypes.POINTER(vp)]
lib.spng_destroy.argtypes = [vp]; lib.spng_destroy.restype = None
lib.spng_storage_size.restype = u64; lib.spng_storage_size.argtypes = [u32, u32, ci, ci]
lib.spng_inflated_size.restype = u64; lib.spng_inflated_size.argtypes = [u32, u32, ci, ci, ci]
lib.spng_decode.argtypes = [vp, vp, u64, i32, u32, u32, ci, ci, ci, vp, rp]
lib.spng_inflate.argtypes = [vp, vp, u64, i32, vp, u64, rp]
lib.spng_filter.argtypes = [vp, vp, u32, u32, ci, ci, ci, vp, rp]
lib.spng_deflate_bound.restype = u64; lib.spng_deflate_bound.argtypes = [u64]
lib.spng_deflate.argtypes = [vp, vp, u64, i32, i32, vp, u64, rp]

ctx = vp()
st = lib.spng_create(0, None, ctypes.byref(ctx))          # NULL stream: the context owns one
assert st == 0, (st, lib.spng_last_error_string())

def buf(b):
    return (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")

table = json.loads((root / "tests" / "golden" / "pngsuite.json").read_text())
names = [n for n in sorted(table) if n.split("/")[1][:4] in ("basn", "basi", "f00n", "f01n", "f02n", "f03n",
                                                                 "f04n", "oi9n", "z00n", "z09n", "PngS")]
checked = 0
for name in names:
    png = ph.parse_png((root / "tests" / "golden" / "pngsuite" / name).read_bytes())
    s = lib.spng_storage_size(png.width, png.height, png.depth, png.channels)
    out = (ctypes.c_uint8 * max(s, 1))()
    res = Result()
    st = lib.spng_decode(ctx, buf(png.idat), len(png.idat), png.fmt, png.width, png.height, png.depth,
                         png.channels, int(png.interlaced), out, ctypes.byref(res))
    assert st == 0 and res.status == 0, (name, st, res.status, lib.spng_last_error_string())
    assert hashlib.sha256(bytes(out[:s])).hexdigest() == table[name]["storage_sha256"], name
    checked += 1
assert checked >= 40, checked

# encode side: filter + deflate of a decoded golden, verified by zlib (every level the device implements)
png = ph.parse_png((root / "tests" / "golden" / "pngsuite" / "common" / "basn6a08.png").read_bytes())
s = lib.spng_storage_size(png.width, png.height, 8, 4)
u = lib.spng_inflated_size(png.width, png.height, 8, 4, 0)
storage = (ctypes.c_uint8 * s)(); res = Result()
assert lib.spng_decode(ctx, buf(png.idat), len(png.idat), 0, png.width, png.height, 8, 4, 0, storage, ctypes.byref(res)) == 0
rows = (ctypes.c_uint8 * u)()
assert lib.spng_filter(ctx, storage, png.width, png.height, 8, 4, 0, rows, ctypes.byref(res)) == 0 and res.status == 0
for level in LEVELS:
    cap = lib.spng_deflate_bound(u)
    z = (ctypes.c_uint8 * cap)()
    assert lib.spng_deflate(ctx, rows, u, 0, level, z, cap, ctypes.byref(res)) == 0 and res.status == 0, level
    assert zlib.decompress(bytes(z[:res.written])) == bytes(rows), level
    back = (ctypes.c_uint8 * (u + 16))(); r2 = Result()
    assert lib.spng_inflate(ctx, z, res.written, 0, back, u + 16, ctypes.byref(r2)) == 0
    assert r2.status == 0 and bytes(back[:r2.written]) == bytes(rows) and r2.consumed == res.written
lib.spng_destroy(ctx)
assert "torch" not in sys.modules
print("torch-free ok:", checked, "goldens")
"""


def _run(code, *args, timeout=600):
    return subprocess.run([sys.executable, "-c", code, *args], capture_output=True, text=True, timeout=timeout)


def test_c_abi_without_torch(gpu):
    r = _run(CHILD.replace("LEVELS", "(0, 4, 6, 9)"), str(ph.ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "torch-free ok" in r.stdout


def test_smoke_in_fresh_process(gpu):
    """What the driver runs: build() then smoke(), in an interpreter that has imported nothing yet."""
    r = _run("import sys; sys.path.insert(0, sys.argv[1]); import __graft_entry__ as g; g.smoke()", str(ph.ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "smoke ok" in r.stdout
