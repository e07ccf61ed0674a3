// chunks.hip -- PNG chunk framing for gfx950: the callers' side of the decode / encode path.
//
// Replaces, for a batch of whole files resident in HBM:
//   signature + chunk lexing + CRC-32 check   Sources/PNG/Lexing/PNG.BytestreamSource.swift:44-108
//   chunk type validation                     Sources/PNG/Lexing/PNG.Chunk.swift:69-88
//   IHDR fields                               Sources/PNG/Parsing/PNG.Header.swift:73-129 (layout only)
//   the IDAT loop of decompress(stream:)      Sources/PNG/PNG.Image.swift:385-389 (payloads concatenated)
//   chunk emission + CRC-32                   Sources/PNG/Lexing/PNG.BytestreamDestination.swift:66-88
// CRC-32 is swift-hash 0.7.1's CRC32 (Package.resolved; source not in the reference checkout): the standard
// reflected CRC-32, polynomial 0xEDB88320, initial value and final xor 0xFFFFFFFF -- pinned by the two
// checksums in Sources/PNGIntegrationTests/ErrorHandling.swift:30,42 and by every fixture lexing cleanly.
//
// One wave per file walks the chunk chain (a chunk's length field gives the next header).  The CRC of a
// chunk is wave-parallel: 64 equal pieces, one per lane (byte-wise, 256-entry table in LDS), folded with
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P (the raw, zero-initialised CRC is linear), the
// per-level shift factor being the previous one squared; the < 64 leftover bytes go through the table.
#include "common.hpp"
#include "crc32.hpp"

namespace spng {

__device__ __forceinline__ uint32_t be32(const gbyte *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// PNG.Chunk.init(validating:) (PNG.Chunk.swift:69-88)
__device__ __forceinline__ bool valid_type(uint32_t name)
{
    switch (name) {
    case 0x43674249: case 0x49484452: case 0x504c5445: case 0x49444154: case 0x49454e44:   // CgBI IHDR PLTE IDAT IEND
    case 0x6348524d: case 0x67414d41: case 0x69434350: case 0x73424954: case 0x73524742:   // cHRM gAMA iCCP sBIT sRGB
    case 0x624b4744: case 0x68495354: case 0x74524e53: case 0x70485973: case 0x73504c54:   // bKGD hIST tRNS pHYs sPLT
    case 0x74494d45: case 0x69545874: case 0x74455874: case 0x7a545874:                     // tIME iTXt tEXt zTXt
        return true;
    default:
        return (name & 0x20002000u) == 0x20000000u;
    }
}

__global__ __launch_bounds__(64) void lex_kernel(const spng_file_desc *__restrict__ files, spng_lexed *__restrict__ out)
{
    __shared__ uint32_t tab[256];
    const int lane = threadIdx.x;
    const spng_file_desc f = files[blockIdx.x];
    const gbyte *p = (const gbyte *)uni64((uint64_t)f.d_png);
    const uint64_t n = uni64(f.len);
    gbyte *idat = (gbyte *)uni64((uint64_t)f.d_idat);
    const uint64_t cap = uni64(f.idat_cap);
    crc_table(tab, lane);
    spng_lexed r;
    memset(&r, 0, sizeof r);
    r.status = SPNG_DONE;
    uint64_t off = 8, idat_len = 0;
    // signature() (:44-56)
    if (n < 8) r.status = SPNG_E_TRUNCATED_SIGNATURE;
    else {
        const uint64_t sig = (uint64_t)be32(p) << 32 | be32(p + 4);
        if (sig != 0x89504e470d0a1a0aull) { r.status = SPNG_E_SIGNATURE; r.aux[0] = sig; }
    }
    while (r.status == SPNG_DONE) {
        // chunk() (:71-108)
        if (off + 8 > n) { r.status = SPNG_E_TRUNCATED_CHUNK_HEADER; break; }
        const uint32_t length = be32(p + off), name = be32(p + off + 4);
        if (!valid_type(name)) { r.status = SPNG_E_CHUNK_TYPE; r.aux[0] = name; break; }
        const uint64_t bytes = (uint64_t)length + 4;
        if (off + 8 + bytes > n) { r.status = SPNG_E_TRUNCATED_CHUNK_BODY; r.aux[0] = bytes; break; }
        const uint32_t declared = be32(p + off + 8 + length);
        const uint32_t computed = wave_crc32(tab, p + off + 4, (uint64_t)length + 4, 0, lane);
        if (declared != computed) { r.status = SPNG_E_CHUNK_CHECKSUM; r.aux[0] = declared; r.aux[1] = computed; break; }
        r.chunks += 1;
        const gbyte *data = p + off + 8;
        if (name == 0x43674249) r.ios = 1;
        else if (name == 0x49484452 && length >= 13) {
            r.width = be32(data); r.height = be32(data + 4);
            r.depth = data[8]; r.color = data[9]; r.compression = data[10]; r.filter = data[11]; r.interlace = data[12];
        } else if (name == 0x504c5445) { r.plte_off = off + 8; r.plte_len = length; }
        else if (name == 0x74524e53) { r.trns_off = off + 8; r.trns_len = length; }
        else if (name == 0x49444154) {
            if (idat_len + length > cap) { r.status = SPNG_E_OUTPUT_CAPACITY; break; }
            for (uint64_t i = lane; i < length; i += 64) idat[idat_len + i] = data[i];
            idat_len += length;
        }
        off += 8 + bytes;
        if (name == 0x49454e44) break;                         // IEND
    }
    r.idat_len = idat_len; r.consumed = off < n ? off : n;
    if (lane == 0) out[blockIdx.x] = r;
}

// PNG.BytestreamDestination.format(type: .IDAT, data:) for every piece of a stream (:66-88)
__global__ __launch_bounds__(64) void write_idat_kernel(const spng_chunking_desc *__restrict__ descs, spng_result *__restrict__ results)
{
    __shared__ uint32_t tab[256];
    const int lane = threadIdx.x;
    const spng_chunking_desc d = descs[blockIdx.y];
    const uint64_t n = uni64(d.len), piece = uni64(d.chunk_bytes);
    const uint64_t pieces = n ? (n + piece - 1) / piece : 0;
    const uint64_t need = n + 12 * pieces;
    if (blockIdx.x == 0 && lane == 0) {
        spng_result &res = results[blockIdx.y];
        res.status = need > d.out_cap ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
        res.written = need; res.consumed = n; res.aux[0] = pieces; res.aux[1] = 0;
    }
    if (need > d.out_cap) return;
    crc_table(tab, lane);
    const gbyte *src = (const gbyte *)uni64((uint64_t)d.d_stream);
    gbyte *out = (gbyte *)uni64((uint64_t)d.d_out);
    for (uint64_t k = blockIdx.x; k < pieces; k += gridDim.x) {
        const uint64_t lo = k * piece, len = n - lo < piece ? n - lo : piece;
        gbyte *o = out + lo + 12 * k;
        if (lane < 4) o[lane] = (uint8_t)(len >> (8 * (3 - lane)));
        if (lane < 4) o[4 + lane] = (uint8_t)(0x49444154u >> (8 * (3 - lane)));
        for (uint64_t i = lane; i < len; i += 64) o[8 + i] = src[lo + i];
        // CRC over the type code and the data
        uint32_t crc = 0;
        for (int b = 0; b < 4; ++b) crc = tab[((crc ^ 0xffffffffu) ^ (0x49444154u >> (8 * (3 - b)))) & 0xff] ^ ((crc ^ 0xffffffffu) >> 8) ^ 0xffffffffu;
        crc = wave_crc32(tab, src + lo, len, crc, lane);
        if (lane < 4) o[8 + len + lane] = (uint8_t)(crc >> (8 * (3 - lane)));
    }
}

// raw (zero-initialised, un-finalised) CRC of 1 MiB pieces: the host folds them (spng_crc32)
__global__ __launch_bounds__(64) void crc_partial_kernel(const uint8_t *__restrict__ data, uint64_t n, uint64_t piece, uint32_t *__restrict__ partial)
{
    __shared__ uint32_t tab[256];
    const int lane = threadIdx.x;
    crc_table(tab, lane);
    const uint64_t lo = (uint64_t)blockIdx.x * piece, len = n - lo < piece ? n - lo : piece;
    // wave_crc32 with crc = 0 returns the standard CRC of the piece; undo its conditioning to get the raw one
    const uint32_t std_crc = wave_crc32(tab, (const gbyte *)data + lo, len, 0, lane);
    if (lane == 0) partial[blockIdx.x] = std_crc ^ 0xffffffffu ^ multmodp(xpow8(len), 0xffffffffu);
}

uint32_t crc32_fold(const uint32_t *partial, uint64_t pieces, uint64_t n, uint64_t piece)
{
    uint32_t c = 0;
    for (uint64_t k = 0; k < pieces; ++k) {
        const uint64_t len = n - k * piece < piece ? n - k * piece : piece;
        c = multmodp(xpow8(len), c) ^ partial[k];
    }
    return c ^ multmodp(xpow8(n), 0xffffffffu) ^ 0xffffffffu;
}

hipError_t launch_lex(const spng_file_desc *d_files, uint32_t count, spng_lexed *d_out, hipStream_t stream)
{
    if (!count) return hipSuccess;
    lex_kernel<<<count, 64, 0, stream>>>(d_files, d_out);
    return hipGetLastError();
}
hipError_t launch_write_idat(const spng_chunking_desc *d_descs, uint32_t count, uint32_t blocks_x, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    write_idat_kernel<<<dim3(blocks_x ? blocks_x : 1, count), 64, 0, stream>>>(d_descs, d_results);
    return hipGetLastError();
}
hipError_t launch_crc_partial(const uint8_t *d, uint64_t n, uint64_t piece, uint32_t *d_partial, uint32_t pieces, hipStream_t stream)
{
    if (!pieces) return hipSuccess;
    crc_partial_kernel<<<pieces, 64, 0, stream>>>(d, n, piece, d_partial);
    return hipGetLastError();
}

}  // namespace spng
