"""swift_png_amd -- host-side mirror of swift-png's hot-path API over libspng_mi355.so.

The product is the C-ABI library (include/spng_mi355.h, built from swift_png_amd/csrc/*.hip for
gfx950).  swift-png itself is Swift; no Swift toolchain exists in this environment, so the host
code above the C ABI is Python and mirrors the reference's seams for this path:

    LZ77.Inflator              Sources/LZ77/Inflator/LZ77.Inflator.swift:8-62
    PNG.Decoder.defilter       Sources/PNG/Decoding/PNG.Decoder.swift:152-196
    PNG.Encoder.filter         Sources/PNG/Encoding/PNG.Encoder.swift:132-204
    PNG.Context.push(data:)    Sources/PNG/Decoding/PNG.Context.swift:88-147

PyTorch is used only for device memory and streams.  There is no CPU fallback: `load()` raises
when the HIP library or the GPU is missing.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SPNG_LIB", _HERE / "libspng_mi355.so"))   # SPNG_LIB: tuning builds only

# ---- status vocabulary (include/spng_mi355.h) -------------------------------------------------
DONE, NEED_MORE_INPUT = 0, 1
E_COMPRESSION_METHOD, E_WINDOW_SIZE, E_CHECK_BITS, E_DICTIONARY = 16, 17, 18, 19
E_GZIP_SIGIL, E_GZIP_METHOD, E_GZIP_FLAG_BITS, E_GZIP_HEADER_CHECKSUM = 24, 25, 26, 27
(E_STREAM_CHECKSUM, E_BLOCK_TYPE, E_BLOCK_COUNT_PARITY, E_RUNLITERAL_COUNT, E_CODELENGTH_TABLE,
 E_CODELENGTH_SEQUENCE, E_HUFFMAN_TABLE, E_STRING_REFERENCE) = range(32, 40)
E_EXTRANEOUS_IMAGE_DATA, E_EXTRANEOUS_COMPRESSED_DATA, E_INCOMPLETE_DATASTREAM = 48, 49, 50
E_OUTPUT_CAPACITY, E_ARGUMENT, E_DEVICE, E_REFERENCE_UNDEFINED = 64, 65, 66, 67
(E_TRUNCATED_SIGNATURE, E_SIGNATURE, E_TRUNCATED_CHUNK_HEADER, E_TRUNCATED_CHUNK_BODY, E_CHUNK_TYPE,
 E_CHUNK_CHECKSUM) = range(80, 86)
FORMAT_ZLIB, FORMAT_IOS, FORMAT_GZIP = 0, 1, 2
K_INFLATE, K_UNFILTER, K_SCATTER, K_FILTER, K_DEFLATE, K_ADLER, K_PINFLATE = 0, 1, 2, 3, 4, 5, 6
K_UNPACK = 7
K_PACK = 10
IMAGE_OVERDRAW = 1
TARGET_RGBA, TARGET_VA, TARGET_SCALAR = 0, 1, 2
PREMULTIPLY, PREMULTIPLY_AS_U8 = 1, 2
K_LEX = 12
K_PINF_FIND, K_PINF_DECODE, K_PINF_RESOLVE = 8, 9, 11
K_DFL_SEARCH, K_DFL_PARSE = 13, 14
CFG_INFLATE_MODE, CFG_SEGMENT_BYTES, CFG_TOKEN_BYTES, CFG_UNFILTER_PIECE_ROWS, CFG_INFLATE_OVERLAP, CFG_RESOLVE_PARTS = 0, 1, 2, 3, 4, 5
CFG_DEFLATE_MODE, CFG_DEFLATE_BYTES, CFG_MULTI_GROUPS = 6, 7, 8
DEFLATE_AUTO, DEFLATE_ONE_KERNEL = 0, 1
OVERLAP_AUTO, OVERLAP_ALWAYS, OVERLAP_NEVER = 0, 1, 2
INFLATE_AUTO, INFLATE_SERIAL = 0, 1

EXPORTS = [
    "spng_version", "spng_status_string", "spng_last_error_string", "spng_inflated_size",
    "spng_storage_size", "spng_create", "spng_destroy", "spng_stream", "spng_sync", "spng_profile",
    "spng_profile_get", "spng_token_stats", "spng_configure", "spng_inflate_batch", "spng_inflate_resume_batch", "spng_unfilter_batch",
    "spng_unfilter_resume_batch", "spng_decode_batch",
    "spng_inflate", "spng_unfilter", "spng_decode", "spng_adler32", "spng_filter_batch", "spng_filter",
    "spng_lex_batch", "spng_write_idat_batch", "spng_crc32", "spng_unpack_batch", "spng_unpack", "spng_unpack_as", "spng_pack_batch", "spng_pack_as", "spng_deflate_bound", "spng_deflate_batch", "spng_deflate", "spng_deflate_window", "spng_encode_batch",
    "spng_shard", "spng_decode_batch_multi", "spng_copy_ceiling", "spng_trim", "spng_lds_exchange_ordered", "spng_deflate_state_bytes", "spng_deflate_resume_batch",
]


def source_digest() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and the ABI header: what a measurement file (profiles/*_pmc_*.json)
    names as the build it was taken on, and what bench.py compares with the sources it runs"""
    import hashlib
    h = hashlib.sha256()
    root = Path(__file__).resolve().parent
    for f in sorted(list((root / "csrc").glob("*.hip")) + list((root / "csrc").glob("*.hpp")) + list((root.parent / "include").glob("*.h"))):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("reserved", ctypes.c_int32), ("written", ctypes.c_uint64),
                ("consumed", ctypes.c_uint64), ("aux", ctypes.c_uint64 * 2)]


class StreamDesc(ctypes.Structure):
    _fields_ = [("d_src", ctypes.c_void_p), ("src_len", ctypes.c_uint64), ("d_dst", ctypes.c_void_p),
                ("dst_cap", ctypes.c_uint64), ("format", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ImageDesc(ctypes.Structure):
    _fields_ = [("d_idat", ctypes.c_void_p), ("idat_len", ctypes.c_uint64), ("d_rows", ctypes.c_void_p),
                ("rows_cap", ctypes.c_uint64), ("d_storage", ctypes.c_void_p), ("width", ctypes.c_uint32),
                ("height", ctypes.c_uint32), ("depth", ctypes.c_uint8), ("channels", ctypes.c_uint8),
                ("interlaced", ctypes.c_uint8), ("format", ctypes.c_uint8), ("reserved", ctypes.c_uint32)]


class FileDesc(ctypes.Structure):
    _fields_ = [("d_png", ctypes.c_void_p), ("len", ctypes.c_uint64), ("d_idat", ctypes.c_void_p), ("idat_cap", ctypes.c_uint64)]


class Lexed(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("chunks", ctypes.c_uint32), ("aux", ctypes.c_uint64 * 2),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("depth", ctypes.c_uint8), ("color", ctypes.c_uint8),
                ("compression", ctypes.c_uint8), ("filter", ctypes.c_uint8), ("interlace", ctypes.c_uint8),
                ("ios", ctypes.c_uint8), ("pad", ctypes.c_uint8 * 2), ("idat_len", ctypes.c_uint64),
                ("plte_off", ctypes.c_uint64), ("trns_off", ctypes.c_uint64), ("plte_len", ctypes.c_uint32),
                ("trns_len", ctypes.c_uint32), ("consumed", ctypes.c_uint64)]


class UnpackDesc(ctypes.Structure):
    _fields_ = [("d_storage", ctypes.c_void_p), ("d_out", ctypes.c_void_p), ("d_palette", ctypes.c_void_p),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("palette_count", ctypes.c_uint32),
                ("key", ctypes.c_uint16 * 3), ("depth", ctypes.c_uint8), ("channels", ctypes.c_uint8), ("indexed", ctypes.c_uint8),
                ("bgr", ctypes.c_uint8), ("has_key", ctypes.c_uint8), ("target", ctypes.c_uint8), ("layout", ctypes.c_uint8),
                ("premultiply", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 6)]


class PackDesc(ctypes.Structure):
    _fields_ = [("d_pixels", ctypes.c_void_p), ("d_storage", ctypes.c_void_p), ("d_palette", ctypes.c_void_p),
                ("width", ctypes.c_uint32), ("height", ctypes.c_uint32), ("palette_count", ctypes.c_uint32),
                ("depth", ctypes.c_uint8), ("channels", ctypes.c_uint8), ("indexed", ctypes.c_uint8), ("bgr", ctypes.c_uint8),
                ("source", ctypes.c_uint8), ("layout", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 6)]


class ChunkingDesc(ctypes.Structure):
    _fields_ = [("d_stream", ctypes.c_void_p), ("len", ctypes.c_uint64), ("d_out", ctypes.c_void_p),
                ("out_cap", ctypes.c_uint64), ("chunk_bytes", ctypes.c_uint64)]


# ---- error mirror --------------------------------------------------------------------------------
class SpngError(Exception):
    """Base of the mirrored reference errors; `.status` is the C-ABI code, `.aux` its payload."""

    def __init__(self, status, aux=(0, 0)):
        self.status, self.aux = status, tuple(aux)
        super().__init__(f"{_NAMES.get(status, status)}{self.aux if any(self.aux) else ''}")


class StreamHeaderError(SpngError):      # LZ77.StreamHeaderError
    pass


class GzipStreamHeaderError(SpngError):  # Gzip.StreamHeaderError
    pass


class DecompressionError(SpngError):     # LZ77.DecompressionError
    pass


class DecodingError(SpngError):          # PNG.DecodingError
    pass


class LexingError(SpngError):            # PNG.LexingError
    pass


_NAMES = {
    16: "invalidCompressionMethod", 17: "invalidWindowSize", 18: "invalidCheckBits", 19: "unexpectedDictionary",
    24: "invalidSigil", 25: "invalidCompressionMethod", 26: "invalidFlagBits", 27: "_headerChecksumUnsupported",
    32: "invalidStreamChecksum", 33: "invalidBlockTypeCode", 34: "invalidBlockElementCountParity",
    35: "invalidHuffmanRunLiteralSymbolCount", 36: "invalidHuffmanCodelengthHuffmanTable",
    37: "invalidHuffmanCodelengthSequence", 38: "invalidHuffmanTable", 39: "invalidStringReference",
    48: "extraneousImageData", 49: "extraneousImageDataCompressedData",
    50: "incompleteImageDataCompressedDatastream", 64: "outputCapacity", 65: "invalidArgument",
    66: "deviceError", 67: "referenceUndefined",
    80: "truncatedSignature", 81: "invalidSignature", 82: "truncatedChunkHeader", 83: "truncatedChunkBody",
    84: "invalidChunkTypeCode", 85: "invalidChunkChecksum",
}


def raise_for(status, aux=(0, 0)):
    if status in (DONE, NEED_MORE_INPUT):
        return
    if 24 <= status < 28:
        raise GzipStreamHeaderError(status, aux)
    if 16 <= status < 32:
        raise StreamHeaderError(status, aux)
    if 32 <= status < 48:
        raise DecompressionError(status, aux)
    if 48 <= status < 64:
        raise DecodingError(status, aux)
    if 80 <= status < 96:
        raise LexingError(status, aux)
    raise SpngError(status, aux)


# ---- library loading -----------------------------------------------------------------------------
_lib = None


def load_library():
    """dlopens libspng_mi355.so and declares prototypes.  Needs no GPU (used by the symbol tests)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process.  libspng_mi355.so NEEDs libamdhip64.so.7; the torch wheel bundles its
    # own copy with the same SONAME (plus its own HSA runtime).  If the system copy is mapped first and
    # torch is imported afterwards, torch is bound to the system libamdhip64 but still loads its bundled
    # HSA runtime, and device discovery fails ("no ROCm-capable device").  This Python host uses torch for
    # device memory, so torch's runtime goes in first and the library rides it.  A torch-free host (the
    # Swift binding of INTEGRATION.md, tests/test_torch_free.py) simply gets the system runtime.
    import torch  # noqa: F401
    lib = ctypes.CDLL(str(LIB_PATH))
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32
    rp = ctypes.POINTER(Result)
    lib.spng_version.restype = i32
    lib.spng_status_string.restype = ctypes.c_char_p
    lib.spng_status_string.argtypes = [i32]
    lib.spng_last_error_string.restype = ctypes.c_char_p
    lib.spng_inflated_size.restype = u64
    lib.spng_inflated_size.argtypes = [u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.spng_storage_size.restype = u64
    lib.spng_storage_size.argtypes = [u32, u32, ctypes.c_int, ctypes.c_int]
    lib.spng_create.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
    lib.spng_destroy.restype = None
    lib.spng_destroy.argtypes = [vp]
    lib.spng_stream.restype = vp
    lib.spng_stream.argtypes = [vp]
    lib.spng_sync.argtypes = [vp]
    lib.spng_configure.argtypes = [vp, ctypes.c_int, ctypes.c_int64]
    lib.spng_profile.argtypes = [vp, ctypes.c_int]
    lib.spng_profile_get.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(u64)]
    lib.spng_token_stats.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_int32)]
    lib.spng_inflate_batch.argtypes = [vp, ctypes.POINTER(StreamDesc), u32, vp, rp]
    lib.spng_inflate_resume_batch.argtypes = [vp, ctypes.POINTER(StreamDesc), ctypes.POINTER(ctypes.c_uint64), u32, vp, rp]
    lib.spng_unfilter_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), u32, vp, vp, rp]
    lib.spng_unfilter_resume_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(u64),
                                               ctypes.POINTER(u64), u32, vp, rp]
    lib.spng_decode_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), u32, vp, rp]
    lib.spng_filter_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), u32, vp, rp]
    lib.spng_inflate.argtypes = [vp, vp, u64, i32, vp, u64, rp]
    lib.spng_unfilter.argtypes = [vp, vp, u64, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, rp]
    lib.spng_decode.argtypes = [vp, vp, u64, i32, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, rp]
    lib.spng_adler32.argtypes = [vp, vp, u64, ctypes.POINTER(u32)]
    lib.spng_filter.argtypes = [vp, vp, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, rp]
    lib.spng_lex_batch.argtypes = [vp, ctypes.POINTER(FileDesc), u32, vp, ctypes.POINTER(Lexed)]
    lib.spng_write_idat_batch.argtypes = [vp, ctypes.POINTER(ChunkingDesc), u32, vp, rp]
    lib.spng_crc32.argtypes = [vp, vp, u64, ctypes.POINTER(u32)]
    lib.spng_unpack_batch.argtypes = [vp, vp, u32]
    lib.spng_unpack.argtypes = [vp, vp, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                vp, u32, vp, vp]
    lib.spng_unpack_as.argtypes = [vp, vp, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, vp, u32, vp, vp]
    lib.spng_pack_batch.argtypes = [vp, vp, u32]
    lib.spng_pack_as.argtypes = [vp, vp, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, vp, u32, vp]
    lib.spng_deflate_bound.restype = u64
    lib.spng_deflate_bound.argtypes = [u64]
    lib.spng_deflate_batch.argtypes = [vp, ctypes.POINTER(StreamDesc), ctypes.POINTER(i32), u32, vp, rp]
    lib.spng_deflate.argtypes = [vp, vp, u64, i32, i32, vp, u64, rp]
    lib.spng_deflate_window.argtypes = [vp, vp, u64, i32, i32, i32, vp, u64, rp]
    lib.spng_encode_batch.argtypes = [vp, ctypes.POINTER(ImageDesc), i32, u32, vp, rp]
    lib.spng_deflate_state_bytes.restype = u64
    lib.spng_deflate_resume_batch.argtypes = [vp, ctypes.POINTER(StreamDesc), ctypes.POINTER(i32), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint8),
                                              ctypes.POINTER(u64), u32, vp, rp]
    lib.spng_shard.argtypes = [u32, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    lib.spng_decode_batch_multi.argtypes = [ctypes.POINTER(vp), u32, ctypes.POINTER(ImageDesc), u32, ctypes.POINTER(vp), rp]
    lib.spng_copy_ceiling.argtypes = [vp, vp, vp, u64, i32, i32, ctypes.POINTER(ctypes.c_double)]
    lib.spng_trim.argtypes = [vp]
    for name in EXPORTS:
        getattr(lib, name)
    _lib = lib
    return lib


_sessions = {}


def load(device: int = 0) -> "Session":
    """Returns the process-wide Session for `device`; raises if the HIP path cannot run."""
    if device not in _sessions:
        _sessions[device] = Session(device)
    return _sessions[device]


def shard_of(count: int, parts: int, index: int):
    """(first, n) of block `index` when `count` images are cut into `parts` contiguous blocks (spng_shard)"""
    first, n = ctypes.c_uint32(0), ctypes.c_uint32(0)
    lib = load_library()
    _check(lib, lib.spng_shard(count, parts, index, ctypes.byref(first), ctypes.byref(n)))
    return first.value, n.value


def decode_batch_multi(sessions, descs, gather=None):
    """spng_decode_batch_multi: `descs` (ImageDesc, device pointers on the device of each block's session) decoded by the
    sessions' contexts, block k by sessions[k]; gather: per image a device pointer on sessions[0]'s device (or 0 / None)
    where its raster is wanted.  -> list[Result]"""
    lib = load_library()
    n = len(descs)
    arr = descs if isinstance(descs, ctypes.Array) else (ImageDesc * n)(*descs)
    ctxs = (ctypes.c_void_p * len(sessions))(*[s.ctx for s in sessions])
    res = (Result * n)()
    g = None
    if gather is not None:
        g = (ctypes.c_void_p * n)(*[int(x) if x else None for x in gather])
    _check(lib, lib.spng_decode_batch_multi(ctxs, len(sessions), arr, n, g, res))
    return list(res)


def inflated_size(w, h, depth, channels, interlaced) -> int:
    return load_library().spng_inflated_size(w, h, depth, channels, int(bool(interlaced)))


def storage_size(w, h, depth, channels) -> int:
    return load_library().spng_storage_size(w, h, depth, channels)


def _check(lib, st):
    if st != DONE:
        msg = lib.spng_last_error_string().decode() if st == E_DEVICE else lib.spng_status_string(st).decode()
        raise SpngError(st) if st != E_DEVICE else RuntimeError(f"libspng_mi355: {msg}")


class Session:
    """One spng_ctx: a device, a HIP stream and its workspaces (re-entrant per handle)."""

    def __init__(self, device: int = 0, use_torch_stream: bool = True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("swift_png_amd needs an MI355X (gfx950); no HIP device is visible and there is "
                               "no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device = device
        self.tdev = torch.device("cuda", device)
        handle = ctypes.c_void_p()
        stream = None
        if use_torch_stream:
            with torch.cuda.device(device):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _check(self.lib, self.lib.spng_create(device, stream, ctypes.byref(handle)))
        self.ctx = handle

    def close(self):
        if self.ctx:
            self.lib.spng_destroy(self.ctx)
            self.ctx = None

    # -- plumbing -------------------------------------------------------------------------------
    def sync(self):
        _check(self.lib, self.lib.spng_sync(self.ctx))

    def configure(self, key, value):
        _check(self.lib, self.lib.spng_configure(self.ctx, key, int(value)))

    def profile(self, enable=True):
        _check(self.lib, self.lib.spng_profile(self.ctx, int(enable)))

    def trim(self):
        """gives the context's scratch (token pool, deflate slab ...) back to the device"""
        _check(self.lib, self.lib.spng_trim(self.ctx))

    def copy_ceiling(self, nbytes=8 << 30, pattern=0, repeats=5):
        """GB/s (read + write) of the library's own copy kernel between two buffers of nbytes: the measured ceiling next to the
        8 TB/s spec peak (pattern 0), or the bandwidth of the scanline kernel's former access pattern (pattern 1)"""
        t = self.torch
        a = t.zeros(nbytes // 8, dtype=t.int64, device=self.tdev)
        b = t.empty(nbytes // 8, dtype=t.int64, device=self.tdev)
        ms = ctypes.c_double(0)
        _check(self.lib, self.lib.spng_copy_ceiling(self.ctx, b.data_ptr(), a.data_ptr(), nbytes, pattern, repeats, ctypes.byref(ms)))
        del a, b
        return 2 * nbytes / (ms.value * 1e-3) / 1e9, ms.value

    def profile_get(self, kernel):
        ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
        _check(self.lib, self.lib.spng_profile_get(self.ctx, kernel, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def token_stats(self):
        """(bytes of token pages, DEFLATE blocks, ran dry) of the most recent parallel-inflate call read back."""
        b, k, d = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int32(0)
        _check(self.lib, self.lib.spng_token_stats(self.ctx, ctypes.byref(b), ctypes.byref(k), ctypes.byref(d)))
        return b.value, k.value, bool(d.value)

    def to_device(self, data):
        t = self.torch
        if isinstance(data, (bytes, bytearray, memoryview)):
            if len(data) == 0:
                return t.empty(0, dtype=t.uint8, device=self.tdev)
            return t.frombuffer(bytearray(data), dtype=t.uint8).to(self.tdev)
        return t.as_tensor(data, dtype=t.uint8).to(self.tdev)

    def empty(self, n):
        return self.torch.empty(max(int(n), 1), dtype=self.torch.uint8, device=self.tdev)

    @staticmethod
    def _ptr(tensor):
        return ctypes.c_void_p(tensor.data_ptr() if tensor is not None and tensor.numel() else None)

    # -- batch entry points (device tensors in, device tensors out) ------------------------------
    def inflate_batch(self, streams, caps, fmt=FORMAT_ZLIB):
        """streams: list of uint8 device tensors; caps: output capacities.
        -> (list of output tensors, list[Result])"""
        n = len(streams)
        outs = [self.empty(c) for c in caps]
        descs = (StreamDesc * n)()
        for i, (s, o, c) in enumerate(zip(streams, outs, caps)):
            f = fmt[i] if isinstance(fmt, (list, tuple)) else fmt
            descs[i] = StreamDesc(self._ptr(s), s.numel(), self._ptr(o), int(c), f, 0)
        res = (Result * n)()
        _check(self.lib, self.lib.spng_inflate_batch(self.ctx, descs, n, None, res))
        return outs, list(res)

    def inflate_resume(self, src, src_len, dst, fmt=FORMAT_ZLIB, state=(0, 0, 0, 0)):
        """One push of a stream that arrives in pieces (spng_inflate_resume_batch): src / dst are device tensors that
        hold ALL compressed bytes so far (src_len of them) / the output so far; state is what the previous call
        returned: four words -- {header bit of the block the input ended in, bytes in front of it, first bit inside it that is
        still to decode (0: its header), bytes in front of that} (a pair is taken as a state at a block boundary).
        -> (Result, next state)"""
        desc = (StreamDesc * 1)(StreamDesc(self._ptr(src), int(src_len), self._ptr(dst), dst.numel(), fmt, 0))
        state = tuple(state) + (0, 0) * (len(state) == 2)
        st = (ctypes.c_uint64 * 4)(*[int(v) for v in state])
        res = (Result * 1)()
        _check(self.lib, self.lib.spng_inflate_resume_batch(self.ctx, desc, st, 1, None, res))
        r = res[0]
        if r.status == NEED_MORE_INPUT:
            tok = int(r.consumed)
            return r, (int(r.aux[0]), int(r.aux[1]), tok, int(r.written) if tok else 0)
        return r, tuple(state)

    def unfilter_resume(self, desc, work, prev_len, now_len):
        """The scanlines of one image that became complete between prev_len and now_len inflated bytes are defiltered and
        assigned (spng_unfilter_resume_batch); work: device scratch of the size of the scanline buffer (interlaced and
        sub-byte images), or None.  -> Result (written = scanline bytes defiltered by this call)"""
        arr = (ImageDesc * 1)(desc)
        wp = (ctypes.c_void_p * 1)(self._ptr(work) if work is not None else None)
        prev, now = (ctypes.c_uint64 * 1)(int(prev_len)), (ctypes.c_uint64 * 1)(int(now_len))
        res = (Result * 1)()
        _check(self.lib, self.lib.spng_unfilter_resume_batch(self.ctx, arr, wp, prev, now, 1, None, res))
        return res[0]

    def image_desc(self, idat, rows, storage, w, h, depth, channels, interlaced, fmt=FORMAT_ZLIB, rows_cap=None):
        return ImageDesc(self._ptr(idat), idat.numel() if idat is not None else 0, self._ptr(rows),
                         int(rows_cap if rows_cap is not None else rows.numel()), self._ptr(storage),
                         w, h, depth, channels, int(bool(interlaced)), fmt, 0)

    def decode_batch(self, descs, wait=True):
        """PNG.Context.push for a batch.  wait=True: -> list[Result]; wait=False: results stay on the
        device (fetch_results) and the call returns as soon as the kernels are enqueued."""
        n = len(descs)
        arr = descs if isinstance(descs, ctypes.Array) else (ImageDesc * n)(*descs)
        if wait:
            res = (Result * n)()
            _check(self.lib, self.lib.spng_decode_batch(self.ctx, arr, n, None, res))
            return list(res)
        if getattr(self, "_dres", None) is None or self._dres.numel() < n * ctypes.sizeof(Result):
            self._dres = self.empty(n * ctypes.sizeof(Result))
        _check(self.lib, self.lib.spng_decode_batch(self.ctx, arr, n, self._ptr(self._dres), None))
        return None

    def fetch_results(self, n):
        raw = bytes(self._dres[:n * ctypes.sizeof(Result)].cpu().numpy())
        return list((Result * n).from_buffer_copy(raw))

    def unfilter_batch(self, descs, rows_len=None):
        n = len(descs)
        arr = (ImageDesc * n)(*descs)
        res = (Result * n)()
        dl = None
        if rows_len is not None:
            dl = self.torch.tensor(list(rows_len), dtype=self.torch.int64, device=self.tdev)
        _check(self.lib, self.lib.spng_unfilter_batch(self.ctx, arr, n, self._ptr(dl) if dl is not None else None,
                                                      None, res))
        return list(res)

    def filter_batch(self, descs):
        n = len(descs)
        arr = (ImageDesc * n)(*descs)
        res = (Result * n)()
        _check(self.lib, self.lib.spng_filter_batch(self.ctx, arr, n, None, res))
        return list(res)

    # -- host-buffer conveniences ------------------------------------------------------------------
    def inflate(self, data: bytes, fmt=FORMAT_ZLIB, cap=None):
        """Whole-stream LZ77.Inflator: -> (status, bytes, consumed, aux)"""
        cap = int(cap if cap is not None else max(1 << 16, 1100 * len(data)))
        src = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        dst = (ctypes.c_uint8 * max(cap, 1))()
        res = Result()
        _check(self.lib, self.lib.spng_inflate(self.ctx, src, len(data), fmt, dst, cap, ctypes.byref(res)))
        return res.status, bytes(dst[:res.written]), res.consumed, (res.aux[0], res.aux[1])

    def decode(self, idat: bytes, w, h, depth, channels, interlaced, fmt=FORMAT_ZLIB, storage=None):
        """PNG.Context.push over the concatenated IDAT payload: -> (status, storage bytes, aux)"""
        s = storage_size(w, h, depth, channels)
        buf = (ctypes.c_uint8 * max(s, 1))()
        if storage is not None:
            ctypes.memmove(buf, bytes(storage), s)
        src = (ctypes.c_uint8 * max(len(idat), 1)).from_buffer_copy(bytes(idat) or b"\0")
        res = Result()
        _check(self.lib, self.lib.spng_decode(self.ctx, src, len(idat), fmt, w, h, depth, channels,
                                              int(bool(interlaced)), buf, ctypes.byref(res)))
        return res.status, bytes(buf[:s]), (res.aux[0], res.aux[1])

    def unfilter(self, rows: bytes, w, h, depth, channels, interlaced, storage=None):
        s = storage_size(w, h, depth, channels)
        buf = (ctypes.c_uint8 * max(s, 1))()
        if storage is not None:
            ctypes.memmove(buf, bytes(storage), s)
        src = (ctypes.c_uint8 * max(len(rows), 1)).from_buffer_copy(bytes(rows) or b"\0")
        res = Result()
        _check(self.lib, self.lib.spng_unfilter(self.ctx, src, len(rows), w, h, depth, channels,
                                                int(bool(interlaced)), buf, ctypes.byref(res)))
        return res.status, bytes(buf[:s])

    def filter(self, storage: bytes, w, h, depth, channels, interlaced) -> bytes:
        u = inflated_size(w, h, depth, channels, interlaced)
        src = (ctypes.c_uint8 * max(len(storage), 1)).from_buffer_copy(bytes(storage) or b"\0")
        dst = (ctypes.c_uint8 * max(u, 1))()
        res = Result()
        _check(self.lib, self.lib.spng_filter(self.ctx, src, w, h, depth, channels, int(bool(interlaced)), dst,
                                              ctypes.byref(res)))
        return bytes(dst[:u])

    def lex_batch(self, files):
        """files: list of PNG file bytes -> (list[Lexed], list of concatenated IDAT payloads as bytes).
        The lexing half of PNG.Image.decompress(stream:) for files resident in HBM."""
        n = len(files)
        d_png = [self.to_device(f) for f in files]
        d_idat = [self.empty(len(f)) for f in files]
        descs = (FileDesc * n)()
        for i, (f, a, b) in enumerate(zip(files, d_png, d_idat)):
            descs[i] = FileDesc(self._ptr(a), len(f), self._ptr(b), len(f))
        infos = (Lexed * n)()
        _check(self.lib, self.lib.spng_lex_batch(self.ctx, descs, n, None, infos))
        return list(infos), [bytes(b[:r.idat_len].cpu().numpy()) for b, r in zip(d_idat, infos)]

    def write_idat(self, stream: bytes, chunk_bytes: int) -> bytes:
        """The IDAT chunks (length, type, data, CRC-32 each) of a zlib stream cut every chunk_bytes bytes."""
        pieces = -(-len(stream) // chunk_bytes) if stream else 0
        cap = len(stream) + 12 * pieces
        src, dst = self.to_device(stream), self.empty(cap)
        d = (ChunkingDesc * 1)(ChunkingDesc(self._ptr(src), len(stream), self._ptr(dst), cap, chunk_bytes))
        res = (Result * 1)()
        _check(self.lib, self.lib.spng_write_idat_batch(self.ctx, d, 1, None, res))
        raise_for(res[0].status)
        return bytes(dst[:res[0].written].cpu().numpy())

    def crc32(self, data: bytes) -> int:
        src = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        out = ctypes.c_uint32(0)
        _check(self.lib, self.lib.spng_crc32(self.ctx, src, len(data), ctypes.byref(out)))
        return out.value

    def unpack(self, storage: bytes, w, h, depth, channels, indexed=False, bgr=False, target=16, palette=None, key=None,
               layout=0, premultiply=0):
        """PNG.Image.unpack(as: PNG.RGBA<UInt8 / UInt16>.self) (layout = TARGET_VA: PNG.VA<T>): -> bytes of r, g, b, a
        (v, a) per pixel (host order).  palette: bytes of (r, g, b, a) quadruplets (PLTE with the tRNS alphas folded in);
        key: tRNS chroma key; premultiply: 0, PREMULTIPLY (.premultiplied) or PREMULTIPLY_AS_U8 (.premultiplied(as: UInt8.self))."""
        n = w * h * (4, 2, 1)[layout] * (target // 8)
        src = (ctypes.c_uint8 * max(len(storage), 1)).from_buffer_copy(bytes(storage) or b"\0")
        out = (ctypes.c_uint8 * max(n, 1))()
        pal = (ctypes.c_uint8 * max(len(palette or b""), 1)).from_buffer_copy(bytes(palette or b"\0"))
        k = (ctypes.c_uint16 * 3)(*(list(key) + [0, 0, 0])[:3]) if key is not None else None
        _check(self.lib, self.lib.spng_unpack_as(self.ctx, src, w, h, depth, channels, int(bool(indexed)), int(bool(bgr)), target,
                                                 int(layout), int(premultiply), pal if palette else None, len(palette or b"") // 4, k, out))
        return bytes(out[:n])

    def pack(self, pixels: bytes, w, h, depth, channels, indexed=False, bgr=False, source=16, palette=None, layout=0) -> bytes:
        """PNG.Image(packing:size:layout:).storage for [PNG.RGBA<T>] (layout TARGET_RGBA), [PNG.VA<T>] (TARGET_VA) or [T]
        (TARGET_SCALAR), T = UInt8 / UInt16 (`source` bits), with the default indexer: pixels = bytes of r, g, b, a | v, a | v per
        pixel (host order) -> storage bytes.  palette: bytes of (r, g, b, a) quadruplets."""
        px = (ctypes.c_uint8 * max(len(pixels), 1)).from_buffer_copy(bytes(pixels) or b"\0")
        if len(pixels) != w * h * (4, 2, 1)[layout] * (source // 8):
            raise ValueError("pixel array `count` must be equal to `size.x * size.y`")
        n = self.lib.spng_storage_size(w, h, depth, channels)
        out = (ctypes.c_uint8 * max(n, 1))()
        pal = (ctypes.c_uint8 * max(len(palette or b""), 1)).from_buffer_copy(bytes(palette or b"\0"))
        _check(self.lib, self.lib.spng_pack_as(self.ctx, px, w, h, depth, channels, int(bool(indexed)), int(bool(bgr)), source,
                                               int(layout), pal if palette else None, len(palette or b"") // 4, out))
        return bytes(out[:n])

    def deflate(self, data: bytes, level: int, fmt=FORMAT_ZLIB, exponent: int = 15) -> bytes:
        """Whole-stream LZ77.Deflator (push(all, last: true) + concatenated pull()): -> stream bytes"""
        cap = self.lib.spng_deflate_bound(len(data))
        src = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        dst = (ctypes.c_uint8 * cap)()
        res = Result()
        _check(self.lib, self.lib.spng_deflate_window(self.ctx, src, len(data), fmt, level, exponent, dst, cap,
                                                      ctypes.byref(res)))
        if res.status != DONE:
            raise SpngError(res.status)
        return bytes(dst[:res.written])

    def deflate_resume(self, src, src_len, dst, level, state, last, fmt=FORMAT_ZLIB, exponent=15, hstate=(0, 0)):
        """One push of a stream that is compressed as it arrives (spng_deflate_resume_batch): src holds ALL input so far, dst the
        stream so far, state: a zero-initialised device tensor of spng_deflate_state_bytes() bytes kept between the calls, hstate
        what the previous call returned.  -> (Result, next hstate)"""
        desc = (StreamDesc * 1)(StreamDesc(self._ptr(src), int(src_len), self._ptr(dst), dst.numel(), fmt, exponent))
        lv = (ctypes.c_int32 * 1)(level)
        st = (ctypes.c_void_p * 1)(state.data_ptr())
        la = (ctypes.c_uint8 * 1)(1 if last else 0)
        hs = (ctypes.c_uint64 * 2)(int(hstate[0]), int(hstate[1]))
        res = (Result * 1)()
        _check(self.lib, self.lib.spng_deflate_resume_batch(self.ctx, desc, lv, st, la, hs, 1, None, res))
        r = res[0]
        return r, (r.aux[0], r.aux[1])

    def deflate_batch(self, streams, level, fmt=FORMAT_ZLIB):
        """streams: list of uint8 device tensors -> (list of output tensors, list[Result])"""
        n = len(streams)
        caps = [self.lib.spng_deflate_bound(t.numel()) for t in streams]
        outs = [self.empty(c) for c in caps]
        descs = (StreamDesc * n)()
        for i, (t, o, c) in enumerate(zip(streams, outs, caps)):
            descs[i] = StreamDesc(self._ptr(t), t.numel(), self._ptr(o), int(c), fmt, 0)
        levels = (ctypes.c_int32 * n)(*([level] * n))
        res = (Result * n)()
        _check(self.lib, self.lib.spng_deflate_batch(self.ctx, descs, levels, n, None, res))
        return outs, list(res)

    def adler32(self, data: bytes) -> int:
        src = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        out = ctypes.c_uint32(0)
        _check(self.lib, self.lib.spng_adler32(self.ctx, src, len(data), ctypes.byref(out)))
        return out.value


from .mirror import LZ77, PNG, Gzip  # noqa: E402,F401
