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
// Three kernels.  `lex_walk_kernel`: one wave per file walks the chunk chain (a chunk's length field gives the next
// header: the only serial part), checks signature, types and bounds, notes IHDR / PLTE / tRNS and lists every chunk with
// the place its payload takes in the IDAT stream.  `lex_chunk_kernel`: one wave per listed chunk, all files at once:
// CRC-32 of type + data against the declared one (wave-parallel: 64 pieces of 16-byte loads, slicing-by-4, folded in
// GF(2)[x] / P, crc32.hpp) and the IDAT payload copied into place with 16-byte moves.  `lex_finish_kernel`: the first
// thing that went wrong in FILE ORDER decides the result, as in the reference's loop (a bad checksum in chunk 3 hides a
// bad type in chunk 5).  A file with more chunks than its list holds is walked on by its own wave the serial way.
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

// a listed chunk
struct LexChunk { uint64_t off, idat_off; uint32_t length, name; uint32_t declared, computed; };
// what the walk leaves for the finish
struct LexWalk { uint32_t listed, stop_index; int32_t stop_status; uint32_t pad; uint64_t stop_aux, stop_off, stop_idat; uint32_t bad_crc, ihdr_at, plte_at, trns_at, cgbi_at, pad2; };

__device__ __forceinline__ void copy_bytes(gbyte *dst, const gbyte *src, uint64_t n, int lane)
{
    typedef uint32_t c4 __attribute__((vector_size(16)));
    struct __attribute__((packed)) P16 { c4 v; };
    typedef P16 __attribute__((address_space(1))) gP16;
    const uint64_t units = n / 16;
    for (uint64_t u = lane; u < units; u += 64) ((gP16 *)(dst + u * 16))->v = ((const gP16 *)(src + u * 16))->v;
    for (uint64_t i = units * 16 + lane; i < n; i += 64) dst[i] = src[i];
}

// CRC-32 over a chunk's type code and data (BytestreamSource.chunk, :95-103); idat_to: where the payload goes (IDAT chunks), or null
__device__ __forceinline__ uint32_t chunk_crc32(const uint32_t *tab, const gbyte *type_and_data, uint32_t length, int lane, gbyte *idat_to)
{
    uint32_t c = 0xffffffffu;
    for (int b = 0; b < 4; ++b) c = tab[(c ^ UNI(type_and_data[b])) & 0xff] ^ (c >> 8);
    return wave_crc32(tab, type_and_data + 4, length, c ^ 0xffffffffu, lane, idat_to);
}

__global__ __launch_bounds__(64) void lex_walk_kernel(const spng_file_desc *__restrict__ files, spng_lexed *__restrict__ out,
                                                      LexChunk *__restrict__ table, const uint64_t *__restrict__ table_at,
                                                      LexWalk *__restrict__ walks)
{
    __shared__ uint32_t tab[CRC_TAB];
    const int lane = threadIdx.x;
    const spng_file_desc f = files[blockIdx.x];
    const gbyte *p = (const gbyte *)uni64((uint64_t)f.d_png);
    const uint64_t n = uni64(f.len);
    gbyte *idat = (gbyte *)uni64((uint64_t)f.d_idat);
    const uint64_t cap = uni64(f.idat_cap);
    LexChunk *list = table + uni64(table_at[blockIdx.x]);
    const uint32_t list_cap = (uint32_t)(uni64(table_at[blockIdx.x + 1]) - uni64(table_at[blockIdx.x]));
    spng_lexed r;
    memset(&r, 0, sizeof r);
    LexWalk w;
    memset(&w, 0, sizeof w);
    w.stop_status = SPNG_DONE; w.bad_crc = 0xffffffffu;
    w.ihdr_at = w.plte_at = w.trns_at = w.cgbi_at = 0xffffffffu;
    uint64_t off = 8, idat_len = 0;
    uint32_t index = 0;
    bool tabled = false;
    // signature() (:44-56)
    if (n < 8) w.stop_status = SPNG_E_TRUNCATED_SIGNATURE;
    else {
        const uint64_t sig = (uint64_t)be32(p) << 32 | be32(p + 4);
        if (sig != 0x89504e470d0a1a0aull) { w.stop_status = SPNG_E_SIGNATURE; w.stop_aux = sig; }
    }
    while (w.stop_status == SPNG_DONE) {
        // chunk() (:71-108)
        if (off + 8 > n) { w.stop_status = SPNG_E_TRUNCATED_CHUNK_HEADER; break; }
        const uint32_t length = be32(p + off), name = be32(p + off + 4);
        if (!valid_type(name)) { w.stop_status = SPNG_E_CHUNK_TYPE; w.stop_aux = name; break; }
        const uint64_t bytes = (uint64_t)length + 4;
        if (off + 8 + bytes > n) { w.stop_status = SPNG_E_TRUNCATED_CHUNK_BODY; w.stop_aux = bytes; break; }
        const gbyte *data = p + off + 8;
        if (name == 0x49444154 && idat_len + length > cap) {
            // (the reference's order: the checksum first)
            if (!tabled) { crc_table(tab, lane); tabled = true; }
            const uint32_t declared = be32(data + length), computed = wave_crc32(tab, p + off + 4, (uint64_t)length + 4, 0, lane);
            if (declared != computed) { w.stop_status = SPNG_E_CHUNK_CHECKSUM; w.stop_aux = (uint64_t)declared << 32 | computed; }
            else w.stop_status = SPNG_E_OUTPUT_CAPACITY;
            break;
        }
        if (index < list_cap) {
            if (lane == 0) {
                LexChunk c;
                c.off = off; c.idat_off = idat_len; c.length = length; c.name = name; c.declared = be32(data + length); c.computed = 0;
                list[index] = c;
            }
        } else {
            // the list is full: this wave checks and copies the rest itself
            if (!tabled) { crc_table(tab, lane); tabled = true; }
            const uint32_t declared = be32(data + length);
            const uint32_t computed = wave_crc32(tab, p + off + 4, (uint64_t)length + 4, 0, lane);
            if (declared != computed) { w.stop_status = SPNG_E_CHUNK_CHECKSUM; w.stop_aux = (uint64_t)declared << 32 | computed; break; }
            if (name == 0x49444154) copy_bytes(idat + idat_len, data, length, lane);
        }
        if (name == 0x43674249) w.cgbi_at = w.cgbi_at == 0xffffffffu ? index : w.cgbi_at;
        else if (name == 0x49484452 && length >= 13) {
            r.width = be32(data); r.height = be32(data + 4);
            r.depth = data[8]; r.color = data[9]; r.compression = data[10]; r.filter = data[11]; r.interlace = data[12];
            w.ihdr_at = index;
        } else if (name == 0x504c5445) { r.plte_off = off + 8; r.plte_len = length; w.plte_at = index; }
        else if (name == 0x74524e53) { r.trns_off = off + 8; r.trns_len = length; w.trns_at = index; }
        else if (name == 0x49444154) idat_len += length;
        index += 1;
        off += 8 + bytes;
        if (name == 0x49454e44) break;                         // IEND
    }
    w.listed = index < list_cap ? index : list_cap;
    w.stop_index = index; w.stop_off = off; w.stop_idat = idat_len;
    if (lane == 0) { out[blockIdx.x] = r; walks[blockIdx.x] = w; }
}

__global__ __launch_bounds__(64) void lex_chunk_kernel(const spng_file_desc *__restrict__ files, LexChunk *__restrict__ table,
                                                       const uint64_t *__restrict__ table_at, LexWalk *__restrict__ walks)
{
    __shared__ uint32_t tab[CRC_TAB];
    const int lane = threadIdx.x;
    const uint32_t file = blockIdx.y;
    const uint32_t listed = UNI(walks[file].listed);
    if (blockIdx.x >= listed) return;
    const spng_file_desc f = files[file];
    const gbyte *p = (const gbyte *)uni64((uint64_t)f.d_png);
    gbyte *idat = (gbyte *)uni64((uint64_t)f.d_idat);
    LexChunk *list = table + uni64(table_at[file]);
    crc_table(tab, lane);
    for (uint32_t k = blockIdx.x; k < listed; k += gridDim.x) {
        const uint64_t off = uni64(list[k].off);
        const uint32_t length = UNI(list[k].length), name = UNI(list[k].name), declared = UNI(list[k].declared);
        // (the copy stays a pass of its own: consecutive lanes, consecutive 16-byte units.  Copied by the lanes that sum them -- a
        // piece per lane, 2^k bytes apart -- the payloads of 64 KiB chunks cost a line per lane and store: 256 4K files lexed in 10.0
        // instead of 6.9 ms)
        const uint32_t computed = chunk_crc32(tab, p + off + 4, length, lane, nullptr);
        if (declared != computed) {
            if (lane == 0) { list[k].computed = computed; atomicMin(&walks[file].bad_crc, k); }
        } else if (name == 0x49444154) copy_bytes(idat + uni64(list[k].idat_off), p + off + 8, length, lane);
    }
}

__global__ void lex_finish_kernel(spng_lexed *__restrict__ out, const LexChunk *__restrict__ table, const uint64_t *__restrict__ table_at,
                                  const LexWalk *__restrict__ walks, const spng_file_desc *__restrict__ files, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    spng_lexed r = out[i];
    const LexWalk w = walks[i];
    const LexChunk *list = table + table_at[i];
    uint32_t upto = w.stop_index;                              // chunks lexed in full
    r.status = w.stop_status;
    if (w.stop_status == SPNG_E_CHUNK_CHECKSUM) { r.aux[0] = w.stop_aux >> 32; r.aux[1] = (uint32_t)w.stop_aux; }
    else r.aux[0] = w.stop_aux;
    r.consumed = w.stop_off; r.idat_len = w.stop_idat;
    if (w.bad_crc < upto) {
        // a checksum failed in front of whatever stopped the walk
        const LexChunk c = list[w.bad_crc];
        upto = w.bad_crc;
        r.status = SPNG_E_CHUNK_CHECKSUM; r.aux[0] = c.declared; r.aux[1] = c.computed;
        r.consumed = c.off; r.idat_len = c.idat_off;
    }
    r.chunks = upto;
    // what lies behind the first failure was never seen by the reference's loop
    if (w.ihdr_at >= upto) { r.width = r.height = 0; r.depth = r.color = r.compression = r.filter = r.interlace = 0; }
    if (w.plte_at >= upto) { r.plte_off = 0; r.plte_len = 0; }
    if (w.trns_at >= upto) { r.trns_off = 0; r.trns_len = 0; }
    r.ios = w.cgbi_at < upto ? 1 : 0;
    if (r.consumed > files[i].len) r.consumed = files[i].len;
    out[i] = r;
}

// PNG.BytestreamDestination.format(type: .IDAT, data:) for every piece of a stream (:66-88)
__global__ __launch_bounds__(64) void write_idat_kernel(const spng_chunking_desc *__restrict__ descs, spng_result *__restrict__ results)
{
    __shared__ uint32_t tab[CRC_TAB];
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
    __shared__ uint32_t tab[CRC_TAB];
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

size_t lex_chunk_bytes() { return sizeof(LexChunk); }
size_t lex_walk_bytes() { return sizeof(LexWalk); }
hipError_t launch_lex(const spng_file_desc *d_files, uint32_t count, spng_lexed *d_out, void *d_table, const uint64_t *d_table_at,
                      void *d_walks, uint32_t max_listed, hipStream_t stream)
{
    if (!count) return hipSuccess;
    lex_walk_kernel<<<count, 64, 0, stream>>>(d_files, d_out, (LexChunk *)d_table, d_table_at, (LexWalk *)d_walks);
    uint32_t bx = max_listed < 1 ? 1 : max_listed;
    // (enough waves to fill the chip; a wave strides over its file's chunks -- four waves per file at least: a batch of small
    // files is as slow as its file with the most chunks)
    uint32_t want = (8192 + count - 1) / count;
    if (want < 4) want = 4;
    if (bx > want) bx = want < 1 ? 1 : want;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)             // (grid y stops at 65535)
        lex_chunk_kernel<<<dim3(bx, count - y0 < 65535u ? count - y0 : 65535u), 64, 0, stream>>>(d_files + y0, (LexChunk *)d_table, d_table_at + y0,
                                                                                                 (LexWalk *)d_walks + y0);
    lex_finish_kernel<<<(count + 63) / 64, 64, 0, stream>>>(d_out, (const LexChunk *)d_table, d_table_at, (const LexWalk *)d_walks, d_files, count);
    return hipGetLastError();
}
hipError_t launch_write_idat(const spng_chunking_desc *d_descs, uint32_t count, uint32_t blocks_x, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)
        write_idat_kernel<<<dim3(blocks_x ? blocks_x : 1, count - y0 < 65535u ? count - y0 : 65535u), 64, 0, stream>>>(d_descs + y0, d_results + y0);
    return hipGetLastError();
}
hipError_t launch_crc_partial(const uint8_t *d, uint64_t n, uint64_t piece, uint32_t *d_partial, uint32_t pieces, hipStream_t stream)
{
    if (!pieces) return hipSuccess;
    crc_partial_kernel<<<pieces, 64, 0, stream>>>(d, n, piece, d_partial);
    return hipGetLastError();
}

}  // namespace spng
