// gzip.hip -- the gzip wrapper around the DEFLATE kernels: SPNG_FORMAT_GZIP in
// spng_inflate_batch / spng_deflate_batch.
//
// Replaces:
//   header            Sources/LZ77/Gzip/Gzip.StreamHeader.swift:17-97 (read: sigil, method, flag bits, FEXTRA length;
//                     write: the fixed ten bytes), errors Gzip.StreamHeaderError.swift:4-11
//   member layout     Sources/LZ77/Inflator/LZ77.InflatorBuffers.swift:139-230 (.initial -> .strings -> .block ->
//                     .checksum -> .epilogue): FEXTRA bytes skipped, FNAME / FCOMMENT zero-terminated strings skipped,
//                     raw DEFLATE blocks, CRC-32 (little-endian, CHECKED), ISIZE (little-endian, read, NOT checked)
//   trailer           Sources/LZ77/Deflator/LZ77.DeflatorBuffers.swift:96-135: CRC-32 and byte count (mod 2^32) of the input
//   integral          Sources/LZ77/Gzip/Gzip.Format.Integral.swift:5-31 (CRC-32 + byte count)
// One member per stream, as the reference ("this currently only supports one member").
//
// Nothing here inflates or deflates: a header kernel moves each gzip stream's source window onto its raw DEFLATE
// payload before the inflate kernels run (they then treat it like SPNG_FORMAT_IOS: no zlib header, no Adler-32),
// and the CRC-32 of the inflated bytes -- 256 pieces per stream, one wave each, folded with
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B) -- is checked against the trailer afterwards.  Deflate: the kernels write
// the ten header bytes themselves; the trailer is appended here.
#include "common.hpp"
#include "crc32.hpp"

namespace spng {

static constexpr uint32_t GZ_PIECES = 256;
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}

__device__ __forceinline__ uint32_t le32(const gbyte *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// One thread per stream.  gz[i]: payload offset of a gzip stream whose header was accepted; GZ_NONE for any other
// stream (not gzip, or already answered here).  A header that is not complete yet answers SPNG_NEED_MORE_INPUT
// (StreamHeader.read returns nil; readString returns nil), as the reference's push would return ().
__global__ void gzip_pre_kernel(InflateJob *__restrict__ jobs, PStream *__restrict__ streams, spng_result *__restrict__ results,
                                uint64_t *__restrict__ gz, int32_t *__restrict__ done, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    InflateJob &j = jobs[i];
    gz[i] = GZ_NONE;
    if (j.format != SPNG_FORMAT_GZIP) return;
    const gbyte *p = (const gbyte *)j.src;
    const uint64_t n = j.src_len;
    int32_t status = SPNG_DONE;
    uint64_t aux = 0, off = 10;
    if (n < 10) status = SPNG_NEED_MORE_INPUT;
    else if (p[0] != 0x1f || p[1] != 0x8b) status = SPNG_E_GZIP_SIGIL;
    else if (p[2] != 0x08) { status = SPNG_E_GZIP_METHOD; aux = p[2]; }
    else if (p[3] & 0xe0) { status = SPNG_E_GZIP_FLAG_BITS; aux = p[3]; }
    else if (p[3] & 0x02) status = SPNG_E_GZIP_HEADER_CHECKSUM;
    else {
        const uint32_t flags = p[3];
        if (flags & 0x04) {                                    // FEXTRA: XLEN (little-endian) bytes follow
            if (n < 12) status = SPNG_NEED_MORE_INPUT;
            else {
                off = 12 + ((uint64_t)p[10] | (uint64_t)p[11] << 8);
                if (off > n) status = SPNG_NEED_MORE_INPUT;
            }
        }
        for (int k = 0; k < 2 && status == SPNG_DONE; ++k) {   // FNAME, FCOMMENT
            if (!(flags & (k ? 0x10 : 0x08))) continue;
            while (off < n && p[off] != 0) ++off;
            if (off >= n) status = SPNG_NEED_MORE_INPUT;
            else ++off;
        }
    }
    if (status == SPNG_DONE) {
        gz[i] = off;
        j.src += off; j.src_len -= off; j.format = SPNG_FORMAT_IOS;
        if (streams) { PStream &st = streams[i]; st.src += off; st.src_len -= off; st.format = SPNG_FORMAT_IOS; }
    } else {
        spng_result &r = results[j.image];
        r.status = status; r.reserved = 0; r.written = 0; r.consumed = 0; r.aux[0] = aux; r.aux[1] = 0;
        done[i] = 1;                                           // the serial kernel leaves it alone
        if (streams) { PStream &st = streams[i]; st.src_len = 0; st.format = SPNG_FORMAT_IOS; }   // (nor can the pipeline take it)
    }
}

// raw (zero-initialised, unfinalised) CRC of one piece of a buffer; grid = (GZ_PIECES, streams)
__device__ __forceinline__ uint64_t piece_len(uint64_t n) { return (n + GZ_PIECES - 1) / GZ_PIECES; }
__device__ __forceinline__ void piece_crc(const uint32_t *tab, const gbyte *p, uint64_t n, uint32_t *out, int lane)
{
    const uint64_t len = piece_len(n), from = (uint64_t)blockIdx.x * len;
    const uint64_t m = from >= n ? 0 : (n - from < len ? n - from : len);
    // wave_crc32 of a fresh CRC returns the finished value; the raw one is that with the initial value's
    // contribution and the final xor taken off again
    uint32_t c = 0;
    if (m) c = wave_crc32(tab, p + from, m, 0, lane) ^ multmodp(xpow8(m), 0xffffffffu) ^ 0xffffffffu;
    if (lane == 0) out[blockIdx.x] = c;
}
__device__ __forceinline__ uint32_t fold_pieces(const uint32_t *part, uint64_t n)
{
    const uint64_t len = piece_len(n);
    const uint32_t whole = xpow8(len);                        // (every piece but the last has this length)
    uint32_t acc = 0;
    for (uint32_t k = 0; k < GZ_PIECES; ++k) {
        const uint64_t from = (uint64_t)k * len;
        if (from >= n) break;
        const uint64_t m = n - from < len ? n - from : len;
        acc = multmodp(m == len ? whole : xpow8(m), acc) ^ part[k];
    }
    return acc ^ multmodp(xpow8(n), 0xffffffffu) ^ 0xffffffffu;   // initial value 0xffffffff shifted past n bytes, final xor
}

__global__ __launch_bounds__(64) void gzip_inflate_crc_kernel(const InflateJob *__restrict__ jobs, const spng_result *__restrict__ results,
                                                              const uint64_t *__restrict__ gz, uint32_t *__restrict__ parts)
{
    __shared__ uint32_t tab[CRC_TAB];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.y;
    if (uni64(gz[i]) == GZ_NONE) return;
    const spng_result &r = results[UNI(jobs[i].image)];
    if ((int32_t)UNI(r.status) != SPNG_DONE) return;
    crc_table(tab, lane);
    piece_crc(tab, (const gbyte *)uni64((uint64_t)jobs[i].dst), uni64(r.written), parts + (uint64_t)i * GZ_PIECES, lane);
}

// .checksum / .epilogue (InflatorBuffers.swift:205-222) and the byte accounting of the whole member
__global__ void gzip_inflate_post_kernel(const InflateJob *__restrict__ jobs, spng_result *__restrict__ results,
                                         const uint64_t *__restrict__ gz, const uint32_t *__restrict__ parts, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || gz[i] == GZ_NONE) return;
    const InflateJob &j = jobs[i];
    spng_result &r = results[j.image];
    const uint64_t off = gz[i];
    // (spng_inflate_resume_batch: `consumed` of a stream that wants more input is a BIT position inside the payload -- part of the
    // state the caller hands back --, not a byte count of the member)
    if (r.status == SPNG_NEED_MORE_INPUT && j.state && !j.internal) return;
    if (r.status == SPNG_DONE) {
        const gbyte *p = (const gbyte *)j.src;                 // (the payload: gzip_pre_kernel moved the window)
        const uint64_t n = j.src_len, at = r.consumed;
        if (at + 4 > n) { r.status = SPNG_NEED_MORE_INPUT; }
        else {
            const uint32_t declared = le32(p + at), computed = fold_pieces(parts + (uint64_t)i * GZ_PIECES, r.written);
            if (declared != computed) { r.status = SPNG_E_STREAM_CHECKSUM; r.aux[0] = declared; r.aux[1] = computed; }
            else if (at + 8 > n) { r.status = SPNG_NEED_MORE_INPUT; r.consumed = at + 4; }
            else r.consumed = at + 8;
        }
    }
    r.consumed += off;
}

// deflate: CRC-32 of the input, then the trailer behind the stream the kernel wrote
__global__ __launch_bounds__(64) void gzip_deflate_crc_kernel(const DeflateJob *__restrict__ jobs, uint32_t *__restrict__ parts)
{
    __shared__ uint32_t tab[CRC_TAB];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.y;
    if ((int32_t)UNI(jobs[i].format) != SPNG_FORMAT_GZIP) return;
    crc_table(tab, lane);
    piece_crc(tab, (const gbyte *)uni64((uint64_t)jobs[i].src), uni64(jobs[i].src_len), parts + (uint64_t)i * GZ_PIECES, lane);
}
__global__ void gzip_deflate_post_kernel(const DeflateJob *__restrict__ jobs, spng_result *__restrict__ results,
                                         const uint32_t *__restrict__ parts, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || jobs[i].format != SPNG_FORMAT_GZIP) return;
    const DeflateJob &j = jobs[i];
    spng_result &r = results[j.image];
    if (r.status != SPNG_DONE) return;
    const uint32_t crc = fold_pieces(parts + (uint64_t)i * GZ_PIECES, j.src_len), size = (uint32_t)j.src_len;
    if (r.written + 8 > j.dst_cap) { r.status = SPNG_E_OUTPUT_CAPACITY; r.written += 8; return; }
    gbyte *q = (gbyte *)j.dst + r.written;
    for (int k = 0; k < 4; ++k) { q[k] = (uint8_t)(crc >> (8 * k)); q[4 + k] = (uint8_t)(size >> (8 * k)); }
    r.written += 8;
}

// ---- resumable streams (spng_inflate_resume_batch) ------------------------------------------------------
// A stream inflated push by push ends in a call that saw only its tail: the zlib trailer -- Adler-32 over ALL
// inflated bytes (MRC32.swift:26-50; .checksum, InflatorBuffers.swift:112-130) -- is checked here, once, when a
// call reports SPNG_DONE: S = sum b_i and I = sum i * b_i (mod 65521) over 256 pieces, one wave each (the same
// closed form as the inflate kernels: s1 = 1 + S, s2 = N + N S - I).
// Who compared the Adler-32 of a finished stream?  The pipeline's verdict (reserved == 1) over a stream it held from its
// first bit is final.  The serial kernel compares only when it is not `resumed` (inflate.hip: an internal job from {0, 0});
// every other stream -- all of spng_inflate_resume_batch's, whatever their state -- is compared here.
__device__ __forceinline__ bool checked_by_pipeline(const InflateJob &j, const spng_result &r)
{
    const bool from_start = j.state[0] == 0 && j.state[1] == 0;
    return from_start && (r.reserved == 1 || j.internal != 0);
}
__global__ __launch_bounds__(64) void resume_adler_kernel(const InflateJob *__restrict__ jobs, const spng_result *__restrict__ results,
                                                          uint64_t *__restrict__ parts)
{
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x / GZ_PIECES, piece = blockIdx.x % GZ_PIECES;      // (streams on x: grid y stops at 65535)
    const InflateJob &j = jobs[i];
    if (!uni64((uint64_t)j.state) || (int32_t)UNI(j.format) != SPNG_FORMAT_ZLIB) return;
    const spng_result &r = results[UNI(j.image)];
    if ((int32_t)UNI(r.status) != SPNG_DONE) return;
    // (the pipeline finished a stream it saw from its first bit: it compared the checksum itself; the serial kernel leaves
    // the comparison of every caller-resumable stream to this pass, also when it happened to see all of it in one call)
    if (checked_by_pipeline(j, r)) return;
    const gbyte *p = (const gbyte *)uni64((uint64_t)j.dst);
    const uint64_t n = uni64(r.written), len = piece_len(n), from = (uint64_t)piece * len;
    const uint64_t m = from >= n ? 0 : (n - from < len ? n - from : len);
    uint32_t S = 0, I = 0, g = (uint32_t)((from + (uint64_t)lane) % 65521);      // position mod 65521, kept incrementally
    for (uint64_t k = lane; k < m; k += 64) {
        const uint32_t b = p[from + k];
        S += b;                                                // < 2^32: a piece is far below 2^24 bytes
        I = (I + g * b) % 65521;
        g += 64; g = g >= 65521 ? g - 65521 : g;
    }
    S = wave_sum32(S % 65521) % 65521; I = wave_sum32(I) % 65521;
    if (lane == 0) parts[(uint64_t)i * GZ_PIECES + piece] = (uint64_t)S << 32 | I;
}
__global__ void resume_post_kernel(const InflateJob *__restrict__ jobs, spng_result *__restrict__ results,
                                   const uint64_t *__restrict__ parts, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const InflateJob &j = jobs[i];
    if (!j.state || j.format != SPNG_FORMAT_ZLIB) return;
    spng_result &r = results[j.image];
    if (r.status != SPNG_DONE) return;
    if (checked_by_pipeline(j, r)) return;
    uint64_t S = 0, I = 0;
    for (uint32_t k = 0; k < GZ_PIECES; ++k) { const uint64_t v = parts[(uint64_t)i * GZ_PIECES + k]; S += v >> 32; I += (uint32_t)v; }
    S %= 65521; I %= 65521;
    const uint64_t N = r.written % 65521;
    const uint32_t computed = (uint32_t)((N + N * S % 65521 + 65521 - I) % 65521) << 16 | (uint32_t)((1 + S) % 65521);
    const gbyte *t = (const gbyte *)j.src + r.consumed - 4;
    const uint32_t declared = (uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3];
    if (declared != computed) { r.status = SPNG_E_STREAM_CHECKSUM; r.aux[0] = declared; r.aux[1] = computed; }
}
hipError_t launch_resume_post(const InflateJob *d_jobs, spng_result *d_results, uint64_t *d_parts, uint32_t count, hipStream_t stream)
{
    resume_adler_kernel<<<count * GZ_PIECES, 64, 0, stream>>>(d_jobs, d_results, d_parts);
    resume_post_kernel<<<(count + 63) / 64, 64, 0, stream>>>(d_jobs, d_results, d_parts, count);
    return hipGetLastError();
}

hipError_t launch_gzip_pre(InflateJob *d_jobs, PStream *d_streams, spng_result *d_results, uint64_t *d_gz, int32_t *d_done,
                           uint32_t count, hipStream_t stream)
{
    gzip_pre_kernel<<<(count + 63) / 64, 64, 0, stream>>>(d_jobs, d_streams, d_results, d_gz, d_done, count);
    return hipGetLastError();
}
hipError_t launch_gzip_inflate_post(const InflateJob *d_jobs, spng_result *d_results, const uint64_t *d_gz, uint32_t *d_parts,
                                    uint32_t count, hipStream_t stream)
{
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)             // (grid y stops at 65535)
        gzip_inflate_crc_kernel<<<dim3(GZ_PIECES, count - y0 < 65535u ? count - y0 : 65535u), 64, 0, stream>>>(d_jobs + y0, d_results, d_gz + y0,
                                                                                                            d_parts + (uint64_t)y0 * GZ_PIECES);
    gzip_inflate_post_kernel<<<(count + 63) / 64, 64, 0, stream>>>(d_jobs, d_results, d_gz, d_parts, count);
    return hipGetLastError();
}
hipError_t launch_gzip_deflate_post(const DeflateJob *d_jobs, spng_result *d_results, uint32_t *d_parts, uint32_t count,
                                    hipStream_t stream)
{
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)
        gzip_deflate_crc_kernel<<<dim3(GZ_PIECES, count - y0 < 65535u ? count - y0 : 65535u), 64, 0, stream>>>(d_jobs + y0, d_parts + (uint64_t)y0 * GZ_PIECES);
    gzip_deflate_post_kernel<<<(count + 63) / 64, 64, 0, stream>>>(d_jobs, d_results, d_parts, count);
    return hipGetLastError();
}
uint32_t gzip_pieces() { return GZ_PIECES; }

}  // namespace spng
