// inflate.hip -- batched zlib / raw-DEFLATE inflate for gfx950: one wavefront per stream.
//
// Replaces LZ77.Inflator over whole streams:
//   state machine      Sources/LZ77/Inflator/LZ77.InflatorBuffers.swift:25-137
//   block readers      Sources/LZ77/Inflator/LZ77.InflatorBuffers.Stream.swift:59-429
//   zlib header        Sources/LZ77/Inflator/LZ77.StreamHeader.swift:16-54
//   tree validation    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:80-174
//   length/distance    Sources/LZ77/LZ77.Composites.swift:19-111
//   Adler-32           Sources/LZ77/Wrappers/LZ77.MRC32.swift:26-50
// with the same accept/reject behaviour and error vocabulary (status codes in spng_mi355.h).
//
// Design (v1).  A DEFLATE stream is one serial dependency chain (bit position of token k+1
// depends on token k), so the unit of parallelism is the stream: one 64-lane wave per stream, 4
// waves per CU -> 1024 streams in flight on the chip.  Inside a wave:
//   * the symbol decode is wave-uniform (every lane computes the same scalar state; the compiler
//     keeps most of it on the scalar unit) with LDS-resident tables: a 2^9-entry lit/len LUT and a
//     2^8-entry distance LUT whose 32-bit entries already carry base value + extra-bit count, and
//     a canonical first-code/count fallback for the rare longer codes.  Tables are rebuilt
//     cooperatively (ballot/popcount ranking, lanes fill LUT replicas in parallel) per block --
//     swift-png's own encoder emits a dynamic block every <= 2047 tokens, so this is hot;
//   * compressed input is staged through a 2 KiB LDS ring with coalesced 16 B/lane loads;
//   * output goes to a 32 KiB LDS ring (the whole DEFLATE window), so LZ77 back-references are
//     LDS->LDS copies done by all 64 lanes (overlapping runs replicate via i mod distance), and
//     is flushed to HBM in aligned 4 KiB pieces with 16 B/lane coalesced stores; only the few
//     references that reach beyond 32 KiB - run go through HBM (already flushed);
//   * Adler-32 is folded into the flush (udot4 weighted sums + wave reduction), so the inflated
//     bytes are never re-read.
#include "common.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };

static constexpr int RING = 32768;           // output window in LDS (power of two)
static constexpr int INR = 2048;             // input ring (two 1 KiB halves)
static constexpr int FLUSH = 4096;           // flush granularity
static constexpr int LBITS = 9, DBITS = 8, MBITS = 7;

// LUT entry: [3:0] code length (0 = longer than the LUT index), [7:4] extra bits,
// [9:8] kind, [31:16] literal / base run / base distance.
enum { K_LIT = 0, K_EOB = 1, K_MATCH = 2, K_UNDEF = 3 };
__device__ __forceinline__ uint32_t entry(uint32_t len, uint32_t extra, uint32_t kind, uint32_t value)
{
    return len | extra << 4 | kind << 8 | value << 16;
}

// LZ77.Composites.swift:25-66 (run decades; symbols 286/287 are zero padding rows) in closed form
__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len)
{
    if (sym < 256) return entry(len, 0, K_LIT, sym);
    if (sym == 256) return entry(len, 0, K_EOB, 0);
    if (sym < 265) return entry(len, 0, K_MATCH, sym - 254);
    if (sym < 285) {
        const uint32_t e = (sym - 261) >> 2;
        return entry(len, e, K_MATCH, ((4 + ((sym - 265) & 3)) << e) + 3);
    }
    if (sym == 285) return entry(len, 0, K_MATCH, 258);
    return entry(len, 0, K_MATCH, 0);                    // 286, 287: (extra 0, base 0)
}
// LZ77.Composites.swift:68-110 (distance decades; 30/31 are zero padding rows)
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len)
{
    if (sym < 4) return entry(len, 0, K_MATCH, sym + 1);
    if (sym < 30) {
        const uint32_t e = (sym >> 1) - 1;
        return entry(len, e, K_MATCH, ((2 + (sym & 1)) << e) + 1);
    }
    return entry(len, 0, K_MATCH, 0);
}
__device__ __forceinline__ uint32_t meta_entry(uint32_t sym, uint32_t len) { return entry(len, 0, K_LIT, sym); }

struct Tree {                    // canonical description for codes longer than the LUT index
    uint16_t first[16], count[16], offset[16];
};

struct Lds {
    uint8_t  ring[RING];
    uint8_t  in[INR];
    uint32_t lit[1 << LBITS];
    uint32_t dist[1 << DBITS];
    uint32_t sorted_lit[288];
    uint32_t sorted_dist[32];
    uint32_t meta[1 << MBITS];
    uint8_t  lens[464];            // 286 + 32 code lengths + worst-case RLE overshoot (138)
    Tree     tlit, tdist;
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Builds LUT + canonical fallback for `n` code lengths (lens[], LDS).  KIND: 0 lit/len, 1 distance,
// 2 code-length code.  Returns false when the code is not complete (HuffmanTree.size, :80-108).
// `normalizing` restates validate(symbols:normalizing:) (:112-135): 0 or 1 used symbol of length 1
// gives a stub whose unused half the reference leaves uninitialised (K_UNDEF here).
template <int KIND>
__device__ bool build(const uint8_t *lens, int n, uint32_t *lut, int lbits, uint32_t *sorted, Tree *tree,
                      bool normalizing, int lane)
{
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t my = s < n ? lens[s] : 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) cnt[l] += __popcll(__ballot(my == (uint32_t)l));
    }
    uint32_t used = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) used += cnt[l];
    const int size = 1 << lbits;
    if (normalizing && (used == 0 || (used == 1 && cnt[1] == 1))) {
        // stub tree (HuffmanTree.swift:52-65)
        uint32_t sym = 0;
        for (int base = 0; base < n; base += 64) {
            const int s = base + lane;
            const unsigned long long m = __ballot(s < n && lens[s] == 1);
            if (m) sym = base + __ffsll((long long)m) - 1;
        }
        for (int j = lane; j < size; j += 64)
            lut[j] = (used && !(j & 1)) ? (KIND == 1 ? dist_entry(sym, 1) : litlen_entry(sym, 1))
                                        : entry(1, 0, K_UNDEF, 0);
        __syncthreads();
        return true;
    }
    int interior = 1;
#pragma unroll
    for (int l = 1; l < 16; ++l) interior = 2 * interior - (int)cnt[l];
    if (interior != 0) return false;

    uint32_t first[16], off[16];
    {
        uint32_t code = 0, o = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) { first[l] = code; off[l] = o; code = (code + cnt[l]) << 1; o += cnt[l]; }
    }
    if (lane < 16 && lane > 0) {
        uint32_t f = 0, c = 0, o = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) if (lane == l) { f = first[l]; c = cnt[l]; o = off[l]; }
        tree->first[lane] = (uint16_t)f; tree->count[lane] = (uint16_t)c; tree->offset[lane] = (uint16_t)o;
    }
    for (int j = lane; j < size; j += 64) lut[j] = 0;      // 0 = "longer than lbits"
    __syncthreads();

    uint32_t run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) run[l] = 0;
    const unsigned long long below = (1ull << lane) - 1;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const uint32_t my = s < n ? lens[s] : 0;
        uint32_t rank = 0, f = 0, o = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) {
            const unsigned long long m = __ballot(my == (uint32_t)l);
            if (my == (uint32_t)l) { rank = run[l] + __popcll(m & below); f = first[l]; o = off[l]; }
            run[l] += __popcll(m);
        }
        if (my) {
            const uint32_t e = KIND == 0 ? litlen_entry(s, my) : KIND == 1 ? dist_entry(s, my) : meta_entry(s, my);
            sorted[o + rank] = e;
            if ((int)my <= lbits) {
                const uint32_t rev = __brev(f + rank) >> (32 - my);
                for (int j = rev; j < size; j += 1 << my) lut[j] = e;
            }
        }
    }
    __syncthreads();
    return true;
}

struct Reader {
    uint64_t buf;        // unread bits, LSB first
    uint32_t cnt;        // number of valid bits in buf
    uint64_t next;       // byte offset of the next dword to fetch (multiple of 4)
    uint32_t ahead;      // dword at `next`, already fetched from the LDS ring
};

// stage 1 KiB of the stream starting at `from` (multiple of 1024) into the input ring; bytes past
// the end read as zero (the reference pads 48 zero bits, LZ77.InflatorIn.swift:130-133)
__device__ void stage(Lds &s, const uint8_t *src, uint64_t n, uint64_t from, int lane)
{
    const uint64_t off = from + (uint64_t)lane * 16;
    u32x4 v = {0, 0, 0, 0};
    if (off + 16 <= n) v = ((const U128u *)(src + off))->v;
    else if (off < n) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; ++k) if (off + k < n) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    }
    *(u32x4 *)(s.in + ((from + lane * 16) & (INR - 1))) = v;
    __syncthreads();
}

__device__ __forceinline__ void seek(Lds &s, Reader &r, const uint8_t *src, uint64_t n, uint64_t byte, int lane)
{
    const uint64_t a = byte & ~(uint64_t)3;
    stage(s, src, n, a & ~(uint64_t)1023, lane);
    stage(s, src, n, (a & ~(uint64_t)1023) + 1024, lane);
    const uint32_t w = *(const uint32_t *)(s.in + (a & (INR - 1)));
    const uint32_t sh = 8 * (uint32_t)(byte & 3);
    r.buf = w >> sh; r.cnt = 32 - sh; r.next = a + 4;
    if ((r.next & 1023) == 0) stage(s, src, n, r.next + 1024, lane);
    r.ahead = *(const uint32_t *)(s.in + (r.next & (INR - 1)));
}

// after this, cnt >= 33
__device__ __forceinline__ void refill(Lds &s, Reader &r, const uint8_t *src, uint64_t n, int lane)
{
    if (r.cnt <= 32) {
        r.buf |= (uint64_t)r.ahead << r.cnt;
        r.cnt += 32;
        r.next += 4;
        if ((r.next & 1023) == 0) stage(s, src, n, r.next + 1024, lane);
        r.ahead = *(const uint32_t *)(s.in + (r.next & (INR - 1)));
    }
}
__device__ __forceinline__ uint32_t take(Reader &r, uint32_t k)
{
    const uint32_t v = (uint32_t)r.buf & ((1u << k) - 1);
    r.buf >>= k; r.cnt -= k;
    return v;
}
__device__ __forceinline__ uint64_t bitpos(const Reader &r) { return r.next * 8 - r.cnt; }

// canonical decode of a code longer than the LUT index (uniform)
__device__ __forceinline__ uint32_t decode_long(const Reader &r, const Tree &t, const uint32_t *sorted, int lbits)
{
    const uint32_t v = __brev((uint32_t)r.buf) >> 17;          // next 15 bits, MSB first
    for (int l = lbits + 1; l < 16; ++l) {
        const uint32_t d = (v >> (15 - l)) - t.first[l];
        if (d < t.count[l]) return sorted[t.offset[l] + d];
    }
    return entry(15, 0, K_UNDEF, 0);                           // unreachable for complete codes
}

struct Out {
    uint8_t *dst; uint64_t cap;
    uint64_t pos, flushed;
    uint32_t s1, s2;                                           // Adler-32 state (MRC32.swift:14-24)
};

// flush ring bytes [flushed, upto) to HBM and fold them into the Adler-32 state
__device__ void flush(Lds &s, Out &o, uint64_t upto, int lane)
{
    __syncthreads();
    while (o.flushed < upto) {
        const uint64_t rem = upto - o.flushed;
        const uint32_t n = rem > FLUSH ? FLUSH : (uint32_t)rem;
        uint32_t l1 = 0, l2 = 0;
        for (uint32_t off = lane * 16; off < n; off += 1024) {
            const u32x4 v = *(const u32x4 *)(s.ring + ((o.flushed + off) & (RING - 1)));
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const uint32_t valid = n - off >= 16 ? 16 : n - off;
            if (valid == 16) {
                ((U128u *)(o.dst + o.flushed + off))->v = v;
            } else {
                for (uint32_t k = 0; k < valid; ++k) o.dst[o.flushed + off + k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                for (uint32_t k = valid; k < 16; ++k) w[k >> 2] &= ~(0xffu << (8 * (k & 3)));
            }
            // A = sum b_j, B = sum (16 - j) b_j over the 16-byte piece
            uint32_t A = 0, B = 0;
            A = __builtin_amdgcn_sad_u8(w[0], 0, A); A = __builtin_amdgcn_sad_u8(w[1], 0, A);
            A = __builtin_amdgcn_sad_u8(w[2], 0, A); A = __builtin_amdgcn_sad_u8(w[3], 0, A);
            B = __builtin_amdgcn_udot4(w[0], 0x0d0e0f10u, B, false);
            B = __builtin_amdgcn_udot4(w[1], 0x090a0b0cu, B, false);
            B = __builtin_amdgcn_udot4(w[2], 0x05060708u, B, false);
            B = __builtin_amdgcn_udot4(w[3], 0x01020304u, B, false);
            l1 += A;
            l2 += (n - off - 16 + 0u) * A + B;                 // bytes of this flush after the piece: n-off-16
        }
        // (a final piece shorter than 16 bytes is handled by the zeroed bytes: its weight
        //  n-off-16 is negative mod 2^32 and cancels against B's 16-j weights exactly)
        l2 %= 65521;
        const uint32_t t1 = wave_sum(l1), t2 = wave_sum(l2);
        o.s2 = (o.s2 + (n % 65521) * o.s1 % 65521 + t2) % 65521;
        o.s1 = (o.s1 + t1) % 65521;
        o.flushed += n;
    }
    __syncthreads();
}

#define FAIL(code, a0, a1) do { status = (code); aux0 = (a0); aux1 = (a1); goto done; } while (0)

__global__ __launch_bounds__(64) void inflate_kernel(const InflateJob *__restrict__ jobs,
                                                     spng_result *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) Lds s;
    const InflateJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint8_t *src = job.src;
    const uint64_t n = job.src_len, total = n * 8;

    int32_t status = SPNG_NEED_MORE_INPUT;
    uint64_t aux0 = 0, aux1 = 0;
    Out o = { job.dst, job.dst_cap, 0, 0, 1, 0 };
    Reader r;
    seek(s, r, src, n, 0, lane);

    // .initial (InflatorBuffers.swift:92-104, StreamHeader.swift:16-54)
    if (job.format != SPNG_FORMAT_IOS) {
        if (16 > total) goto done;
        const uint32_t cm = take(r, 4);
        if (cm != 8) FAIL(SPNG_E_COMPRESSION_METHOD, cm, 0);
        const uint32_t e = take(r, 4);
        if (e >= 8) FAIL(SPNG_E_WINDOW_SIZE, e + 8, 0);
        const uint32_t flags = take(r, 8);
        if (((e << 12 | 8 << 8) + flags) % 31 != 0) FAIL(SPNG_E_CHECK_BITS, 0, 0);
        if (flags & 0x20) FAIL(SPNG_E_DICTIONARY, 0, 0);
    }

    for (;;) {
        // .metadata: readBlockMetadata (InflatorBuffers.Stream.swift:59-141)
        refill(s, r, src, n, lane);
        if (bitpos(r) + 3 > total) goto done;
        const uint32_t bfinal = take(r, 1);
        const uint32_t type = take(r, 2);
        if (type == 0) {
            const uint64_t boundary = (bitpos(r) + 7) & ~(uint64_t)7;
            if (boundary + 32 > total) goto done;
            take(r, (uint32_t)(boundary - bitpos(r)));
            refill(s, r, src, n, lane);
            const uint32_t l = take(r, 16);
            refill(s, r, src, n, lane);
            const uint32_t m = take(r, 16);
            if (l != (~m & 0xffffu)) FAIL(SPNG_E_BLOCK_COUNT_PARITY, l, m);
            // readBlock(upTo:) (:384-399): copies as many of the LEN bytes as the input holds
            const uint64_t from = boundary / 8 + 4;
            const uint64_t have = n - from < l ? n - from : l;
            if (o.pos + have > o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
            for (uint64_t done_ = 0; done_ < have;) {
                const uint64_t piece = have - done_ < 1024 ? have - done_ : 1024;
                for (uint64_t i = lane; i < piece; i += 64) s.ring[(o.pos + i) & (RING - 1)] = src[from + done_ + i];
                __syncthreads();
                o.pos += piece; done_ += piece;
                if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
            }
            if (have < l) { r.next = (n + 3) & ~(uint64_t)3; r.cnt = 0; r.buf = 0; goto done; }
            seek(s, r, src, n, from + l, lane);
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                // fixed trees, HuffmanTree.swift:24-47
                for (int i = lane; i < 288; i += 64) s.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                __syncthreads();
                build<0>(s.lens, 288, s.lit, LBITS, s.sorted_lit, &s.tlit, false, lane);
                for (int i = lane; i < 32; i += 64) s.lens[i] = 5;
                __syncthreads();
                build<1>(s.lens, 32, s.dist, DBITS, s.sorted_dist, &s.tdist, false, lane);
            } else {
                if (bitpos(r) - 3 + 17 > total) goto done;
                const uint32_t literals = 257 + take(r, 5);
                const uint32_t distances = 1 + take(r, 5);
                const uint32_t codelengths = 4 + take(r, 4);
                if (bitpos(r) + 3 * (uint64_t)codelengths > total) goto done;
                if (literals > 286) FAIL(SPNG_E_RUNLITERAL_COUNT, literals, 0);
                // 19 code-length-code lengths in zig-zag order (:120-125)
                uint64_t packed = 0;                           // 19 x 3 bits = 57 bits
                for (uint32_t i = 0; i < codelengths; ++i) {
                    refill(s, r, src, n, lane);
                    packed |= (uint64_t)take(r, 3) << (3 * i);
                }
                if (lane < 19) {
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    s.lens[order[lane]] = (uint32_t)lane < codelengths ? (uint8_t)((packed >> (3 * lane)) & 7) : 0;
                }
                __syncthreads();
                if (!build<2>(s.lens, 19, s.meta, MBITS, s.sorted_lit, &s.tlit, false, lane))
                    FAIL(SPNG_E_CODELENGTH_TABLE, 0, 0);

                // .tables: readBlockTables (:144-263), sequential RLE decode of the code lengths
                const uint32_t want = literals + distances;
                uint32_t have = 0, last = 0;
                while (have < want) {
                    refill(s, r, src, n, lane);
                    if (bitpos(r) >= total) goto done;
                    const uint32_t e = s.meta[(uint32_t)r.buf & ((1 << MBITS) - 1)];
                    const uint32_t len = e & 15, sym = e >> 16;
                    if (bitpos(r) + len > total) goto done;
                    if (sym < 16) {
                        take(r, len);
                        if (lane == 0) s.lens[have] = (uint8_t)sym;
                        last = sym; have += 1;
                        continue;
                    }
                    uint32_t element, extra, base;
                    if (sym == 16) {
                        if (!have) FAIL(SPNG_E_CODELENGTH_SEQUENCE, 0, 0);
                        element = last; extra = 2; base = 3;
                    } else if (sym == 17) { element = 0; extra = 3; base = 3; }
                    else                  { element = 0; extra = 7; base = 11; }
                    if (bitpos(r) + len + extra > total) goto done;
                    take(r, len);
                    const uint32_t reps = base + take(r, extra);
                    for (uint32_t i = lane; i < reps; i += 64) s.lens[have + i] = (uint8_t)element;
                    last = element; have += reps;
                }
                __syncthreads();
                if (have != want) FAIL(SPNG_E_CODELENGTH_SEQUENCE, 0, 0);
                                const bool okd = build<1>(s.lens + literals, (int)distances, s.dist, DBITS, s.sorted_dist,
                                          &s.tdist, true, lane);
                const bool okl = build<0>(s.lens, (int)literals, s.lit, LBITS, s.sorted_lit, &s.tlit,
                                          false, lane);
                if (!okl || !okd) FAIL(SPNG_E_HUFFMAN_TABLE, 0, 0);
            }

            // .compressed: readBlock(with:) (:266-381)
            for (;;) {
                refill(s, r, src, n, lane);
                const uint64_t b0 = bitpos(r);
                if (b0 >= total) goto done;
                uint32_t e = s.lit[(uint32_t)r.buf & ((1 << LBITS) - 1)];
                if ((e & 15) == 0) e = decode_long(r, s.tlit, s.sorted_lit, LBITS);
                const uint32_t len = e & 15, kind = (e >> 8) & 3;
                if (kind == K_LIT) {
                    if (b0 + len > total) goto done;
                    if (o.pos >= o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
                    take(r, len);
                    if (lane == 0) s.ring[o.pos & (RING - 1)] = (uint8_t)(e >> 16);
                    o.pos += 1;
                } else if (kind == K_EOB) {
                    if (b0 + len > total) goto done;
                    take(r, len);
                    break;
                } else {
                    Reader t = r;                              // commit only if the whole token fits
                    take(t, len);
                    const uint32_t cx = (e >> 4) & 15;
                    const uint32_t count = (e >> 16) + take(t, cx);
                    refill(s, t, src, n, lane);
                    uint32_t d = s.dist[(uint32_t)t.buf & ((1 << DBITS) - 1)];
                    if ((d & 15) == 0) d = decode_long(t, s.tdist, s.sorted_dist, DBITS);
                    if (((d >> 8) & 3) == K_UNDEF) FAIL(SPNG_E_REFERENCE_UNDEFINED, 0, 0);
                    take(t, d & 15);
                    const uint32_t ox = (d >> 4) & 15;
                    const uint32_t offset = (d >> 16) + take(t, ox);
                    if (bitpos(t) > total) goto done;
                    if (offset > o.pos) FAIL(SPNG_E_STRING_REFERENCE, 0, 0);
                    if (count && !offset) FAIL(SPNG_E_REFERENCE_UNDEFINED, 0, 0);
                    if (o.pos + count > o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
                    r = t;
                    // InflatorOut.expand (InflatorOut.swift:124-139): forward copy, overlap replicates
                    __syncthreads();
                    if (offset <= RING - 258) {
                        for (uint32_t i = lane; i < count; i += 64) {
                            const uint32_t k = offset >= count ? i : i % offset;
                            s.ring[(o.pos + i) & (RING - 1)] = s.ring[(o.pos - offset + k) & (RING - 1)];
                        }
                    } else {
                        // source may already be overwritten in the ring; it was flushed long ago
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        for (uint32_t i = lane; i < count; i += 64)
                            s.ring[(o.pos + i) & (RING - 1)] = o.dst[o.pos - offset + i];
                    }
                    __syncthreads();
                    o.pos += count;
                }
                if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
            }
        } else {
            FAIL(SPNG_E_BLOCK_TYPE, type, 0);
        }
        if (bfinal) break;
    }

    // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429)
    if (job.format != SPNG_FORMAT_IOS) {
        refill(s, r, src, n, lane);
        const uint64_t boundary = (bitpos(r) + 7) & ~(uint64_t)7;
        if (boundary + 32 > total) goto done;
        take(r, (uint32_t)(boundary - bitpos(r)));
        uint32_t declared = 0;
        for (int k = 0; k < 4; ++k) { refill(s, r, src, n, lane); declared = declared << 8 | take(r, 8); }
        flush(s, o, o.pos, lane);
        const uint32_t computed = o.s2 << 16 | o.s1;
        if (declared != computed) FAIL(SPNG_E_STREAM_CHECKSUM, declared, computed);
    }
    status = SPNG_DONE;
done:
    flush(s, o, o.pos, lane);
    if (lane == 0) {
        spng_result &res = results[job.image];
        res.status = status; res.reserved = 0;
        res.written = o.pos;
        const uint64_t bp = bitpos(r);
        res.consumed = (bp + 7) / 8 > n ? n : (bp + 7) / 8;
        res.aux[0] = aux0; res.aux[1] = aux1;
    }
}

hipError_t launch_inflate(const InflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    inflate_kernel<<<count, 64, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

}  // namespace spng
