"""TEST INFRASTRUCTURE ONLY (like the C files beside it): CPU restatement of swift-png's gzip wrapper around the DEFLATE
oracle.  Never imported by the product path.

Follows
  Gzip.StreamHeader.read / write      Sources/LZ77/Gzip/Gzip.StreamHeader.swift:17-97
  Gzip.StreamHeaderError              Sources/LZ77/Gzip/Gzip.StreamHeaderError.swift:4-11
  InflatorBuffers<Gzip.Format>.advance  Sources/LZ77/Inflator/LZ77.InflatorBuffers.swift:139-230
  DeflatorBuffers<Gzip.Format>        Sources/LZ77/Deflator/LZ77.DeflatorBuffers.swift:96-135
  Gzip.Format.Integral                Sources/LZ77/Gzip/Gzip.Format.Integral.swift:5-31 (CRC-32 + byte count)
The CRC-32 is swift-hash 0.7.1's CRC32 (Package.resolved; not vendored): the standard reflected CRC-32, i.e.
zlib.crc32.  Pinned by tests/test_oracle_gzip.py against Python's own gzip module (an independent implementation
of RFC 1952) in both directions, and on the header rules the reference states.

Status codes as include/spng_mi355.h.
"""
import zlib

DONE, NEED_MORE_INPUT = 0, 1
E_GZIP_SIGIL, E_GZIP_METHOD, E_GZIP_FLAG_BITS, E_GZIP_HEADER_CHECKSUM = 24, 25, 26, 27
E_STREAM_CHECKSUM = 32


def read_header(data: bytes):
    """-> (status, aux0, payload offset)"""
    n = len(data)
    if n < 10:                                               # :22-26 (bit + 80 <= input.count)
        return NEED_MORE_INPUT, 0, 0
    if data[0] != 0x1f or data[1] != 0x8b:                   # :28-33
        return E_GZIP_SIGIL, 0, 0
    if data[2] != 8:                                         # :35-39
        return E_GZIP_METHOD, data[2], 0
    flags = data[3]
    if flags & 0xe0:                                         # :41-45
        return E_GZIP_FLAG_BITS, flags, 0
    if flags & 0x02:                                         # :58-61 (FHCRC)
        return E_GZIP_HEADER_CHECKSUM, 0, 0
    off = 10
    if flags & 0x04:                                         # :63-76: XLEN little-endian; InflatorBuffers :186-190 skips it
        if n < 12:
            return NEED_MORE_INPUT, 0, 0
        off = 12 + (data[10] | data[11] << 8)
        if off > n:
            return NEED_MORE_INPUT, 0, 0
    for bit in (0x08, 0x10):                                 # FNAME, FCOMMENT: zero-terminated (:172-185, readString)
        if flags & bit:
            end = data.find(b"\0", off)
            if end < 0:
                return NEED_MORE_INPUT, 0, 0
            off = end + 1
    return DONE, 0, off


def inflate(data: bytes, inflate_raw, cap=None):
    """inflate_raw(payload, cap) -> (status, out, consumed, aux): the raw-DEFLATE oracle (format 1).
    -> (status, out, consumed, (aux0, aux1))"""
    st, aux0, off = read_header(data)
    if st != DONE:
        return st, b"", 0, (aux0, 0)
    st, out, consumed, aux = inflate_raw(data[off:], cap)
    if st != DONE:
        return st, out, off + consumed, aux
    at = off + consumed
    if at + 4 > len(data):                                   # .checksum needs four bytes (:205-217)
        return NEED_MORE_INPUT, out, at, (0, 0)
    declared = int.from_bytes(data[at:at + 4], "little")
    computed = zlib.crc32(out) & 0xffffffff
    if declared != computed:
        return E_STREAM_CHECKSUM, out, at, (declared, computed)
    if at + 8 > len(data):                                   # .epilogue: ISIZE is read, never compared (:219-223)
        return NEED_MORE_INPUT, out, at + 4, (0, 0)
    return DONE, out, at + 8, (0, 0)


HEADER = bytes([0x1f, 0x8b, 0x08, 0x00, 0, 0, 0, 0, 0x00, 0xff])   # StreamHeader.write (:84-96)


def deflate(data: bytes, deflate_raw):
    """deflate_raw(data) -> the raw-DEFLATE oracle's stream (format 1)"""
    return HEADER + deflate_raw(data) + (zlib.crc32(data) & 0xffffffff).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
