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
// Design.  A DEFLATE stream is one serial dependency chain (the bit position of token k+1 depends
// on token k), so the unit of parallelism is the stream: one 64-lane wave per stream, 4 waves per
// CU -> 1024 streams in flight on the chip.  Inside the wave the chain is attacked two ways:
//   * speculative literal runs: every lane looks up the lit/len LUT (LDS) at bit offset
//     position + lane, all 64 offsets at once; the true chain of symbol boundaries through those
//     64 bits is then resolved on the scalar unit (v_readlane hop per symbol, ~10 cycles instead
//     of an LDS round trip per symbol), the lanes on the chain form a ballot-style mask, and each
//     of them stores its literal at (output position + popcount(mask below me)).  One LDS latency
//     therefore buys every literal that fits in 64 bits (6-8 for PNG residuals);
//   * the token on which a run stops (match, end of block, long code) is decoded wave-uniformly
//     from the same LUT entry; its LZ77 copy is done by all 64 lanes as a second, parallel pass
//     (overlapping runs replicate via i mod distance).
// Tables are LDS resident: a 2^9-entry lit/len LUT and a 2^8-entry distance LUT whose 32-bit
// entries already carry base value + extra-bit count, and a canonical first-code/count fallback
// for the rare longer codes.  They are rebuilt cooperatively per block (ballot/popcount ranking,
// lanes fill LUT replicas in parallel) -- swift-png's own encoder emits a dynamic block every
// <= 2047 tokens, so this is hot.  Compressed input is staged through a 2 KiB LDS ring with
// coalesced 16 B/lane loads and read through a 6-dword register window; output goes to a 32 KiB
// LDS ring (the whole DEFLATE window), so back-references are LDS->LDS copies, and is flushed to
// HBM in aligned 4 KiB pieces with 16 B/lane coalesced stores; only references that reach beyond
// 32 KiB - run go through HBM (already flushed).  Adler-32 is folded into the flush (v_sad_u8 /
// v_dot4 weighted sums + wave reduction), so the inflated bytes are never re-read.
#include "common.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };

static constexpr int RING = 32768;           // output window in LDS (power of two)
static constexpr int INR = 1024;             // input ring (two 512-byte halves)
static constexpr int HALF = INR / 2;
static constexpr int FLUSH = 4096;           // flush granularity
static constexpr int WINOUT = 1024;          // most bytes one speculative window may produce
// Literals of a window are stored before its back-references are resolved, up to WINOUT bytes
// ahead of a reference; a source is still intact in the ring if it is not further back than this.
static constexpr uint32_t LDS_REACH = RING - WINOUT - 258 - 16;
static constexpr int LBITS = 10, DBITS = 8, MBITS = 7;

// Wave-uniform values loaded through the vector path (LDS) are pinned to scalar registers so that
// the whole bit reader and the symbol-boundary chain run on the scalar unit.
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)UNI(v >> 32) << 32 | UNI((uint32_t)v); }

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

struct Lds {                       // 40,208 bytes: four streams per CU
    uint8_t  ring[RING];
    uint8_t  in[INR];
    uint32_t lit[1 << LBITS];      // the code-length-code LUT (2^MBITS entries) lives here while a
                                   // dynamic header is parsed, i.e. before this table is built
    uint32_t dist[1 << DBITS];
    uint16_t sorted_lit[288];      // symbols in canonical order, for codes longer than the LUT index
    uint16_t sorted_dist[32];
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
__device__ bool build(const uint8_t *lens, int n, uint32_t *lut, int lbits, uint16_t *sorted, Tree *tree,
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
            sorted[o + rank] = (uint16_t)s;
            if ((int)my <= lbits) {
                const uint32_t rev = __brev(f + rank) >> (32 - my);
                for (int j = rev; j < size; j += 1 << my) lut[j] = e;
            }
        }
    }
    __syncthreads();
    return true;
}

// Bit reader: a window of six consecutive stream dwords held in (wave-uniform) registers.
struct Reader {
    uint32_t w0, w1, w2, w3, n0;         // dwords wd .. wd+4, pinned to scalar registers
    uint32_t n1;                         // dword wd+5 as it came back from LDS (pinned one shift later,
                                         // so that its latency never sits on the critical path)
    uint64_t wd;                         // dword index of w0
    uint32_t bit;                        // position inside w0, 0..31
};

// stage HALF bytes of the stream starting at `from` (multiple of HALF) into the input ring; bytes
// past the end read as zero (the reference pads 48 zero bits, LZ77.InflatorIn.swift:130-133)
__device__ void stage(Lds &s, const uint8_t *src, uint64_t n, uint64_t from, int lane)
{
    if (lane < HALF / 16) {
        const uint64_t off = from + (uint64_t)lane * 16;
        u32x4 v = {0, 0, 0, 0};
        if (off + 16 <= n) v = ((const U128u *)(src + off))->v;
        else if (off < n) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; ++k) if (off + k < n) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        *(u32x4 *)(s.in + ((from + lane * 16) & (INR - 1))) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}
__device__ __forceinline__ uint32_t dword_at(const Lds &s, uint64_t d)
{
    return *(const uint32_t *)(s.in + ((d * 4) & (INR - 1)));
}

__device__ __forceinline__ void seek(Lds &s, Reader &r, const uint8_t *src, uint64_t n, uint64_t byte, int lane)
{
    const uint64_t h = byte & ~(uint64_t)(HALF - 1);
    stage(s, src, n, h, lane);
    stage(s, src, n, h + HALF, lane);
    r.wd = uni64(byte >> 2); r.bit = UNI(8 * (uint32_t)(byte & 3));
    r.w0 = UNI(dword_at(s, r.wd));     r.w1 = UNI(dword_at(s, r.wd + 1)); r.w2 = UNI(dword_at(s, r.wd + 2));
    r.w3 = UNI(dword_at(s, r.wd + 3)); r.n0 = UNI(dword_at(s, r.wd + 4)); r.n1 = dword_at(s, r.wd + 5);
}
__device__ __forceinline__ void shift(Lds &s, Reader &r, const uint8_t *src, uint64_t n, int lane)
{
    r.w0 = r.w1; r.w1 = r.w2; r.w2 = r.w3; r.w3 = r.n0; r.n0 = UNI(r.n1);
    r.wd = uni64(r.wd + 1);
    const uint64_t d = r.wd + 5;
    if (((d * 4) & (HALF - 1)) == 0) stage(s, src, n, d * 4, lane);
    r.n1 = dword_at(s, d);
}
__device__ __forceinline__ void advance(Lds &s, Reader &r, const uint8_t *src, uint64_t n, int lane, uint32_t k)
{
    r.bit = UNI(r.bit + k);
    while (r.bit >= 32) { shift(s, r, src, n, lane); r.bit = UNI(r.bit - 32); }
}
__device__ __forceinline__ uint32_t peek32(const Reader &r) { return __builtin_amdgcn_alignbit(r.w1, r.w0, r.bit); }
__device__ __forceinline__ uint64_t peek64(const Reader &r)
{
    return (uint64_t)__builtin_amdgcn_alignbit(r.w2, r.w1, r.bit) << 32 | __builtin_amdgcn_alignbit(r.w1, r.w0, r.bit);
}
// 32 stream bits starting `rel` bits after the current position (bit + rel <= 127)
__device__ __forceinline__ uint32_t peek32_at(const Reader &r, uint32_t rel)
{
    const uint32_t a = r.w0, b = r.w1, c = r.w2, d = r.w3, e = r.n0;   // by value: stays in registers
    const uint32_t off = r.bit + rel, sel = off >> 5;
    const uint32_t lo = sel == 0 ? a : (sel == 1 ? b : (sel == 2 ? c : d));
    const uint32_t hi = sel == 0 ? b : (sel == 1 ? c : (sel == 2 ? d : e));
    return __builtin_amdgcn_alignbit(hi, lo, off & 31);
}
#define TAKE(k) take(s, r, src, n, lane, (k))
__device__ __forceinline__ uint32_t take(Lds &s, Reader &r, const uint8_t *src, uint64_t n, int lane, uint32_t k)
{
    const uint32_t v = peek32(r) & ((1u << k) - 1);          // k <= 16
    advance(s, r, src, n, lane, k);
    return v;
}
__device__ __forceinline__ uint64_t bitpos(const Reader &r) { return r.wd * 32 + r.bit; }

// canonical decode of a code longer than the LUT index (uniform); `bits` = next >= 15 stream bits
template <int KIND>
__device__ __forceinline__ uint32_t decode_long(uint32_t bits, const Tree &t, const uint16_t *sorted, int lbits)
{
    const uint32_t v = __brev(bits) >> 17;                     // next 15 bits, MSB first
    for (int l = lbits + 1; l < 16; ++l) {
        const uint32_t d = (v >> (15 - l)) - UNI(t.first[l]);
        if (d < UNI(t.count[l])) {
            const uint32_t sym = UNI(sorted[UNI(t.offset[l]) + d]);
            return KIND == 0 ? litlen_entry(sym, l) : dist_entry(sym, l);
        }
    }
    return entry(15, 0, K_UNDEF, 0);                           // unreachable for complete codes
}

struct Out {
    uint8_t *dst; uint64_t cap;
    uint64_t pos, flushed;
    // Adler-32 (MRC32.swift:26-50) in a form that needs no cross-lane traffic until the very end:
    // with S = sum b_i and I = sum i*b_i over the whole stream of N bytes,
    //     s1 = 1 + S,   s2 = N + N*S - I      (mod 65521)
    // so every lane just accumulates its share of S and I (mod 65521) while flushing.
    uint32_t accS, accI;
    uint32_t base;                                             // flushed mod 65521 (uniform)
};

#define LDS_ORDER() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")

// flush ring bytes [flushed, upto) to HBM and fold them into the lane's Adler-32 accumulators
__device__ __attribute__((always_inline)) void flush(Lds &s, Out &o, uint64_t upto, int lane)
{
    LDS_ORDER();
    while (o.flushed < upto) {
        const uint64_t rem = upto - o.flushed;
        const uint32_t n = rem > FLUSH ? FLUSH : (uint32_t)rem;
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
            // A = sum b_j, J = sum j*b_j over the 16-byte piece at stream offset g = flushed + off
            uint32_t A = 0, J = 0;
            A = __builtin_amdgcn_sad_u8(w[0], 0, A); A = __builtin_amdgcn_sad_u8(w[1], 0, A);
            A = __builtin_amdgcn_sad_u8(w[2], 0, A); A = __builtin_amdgcn_sad_u8(w[3], 0, A);
            J = __builtin_amdgcn_udot4(w[0], 0x03020100u, J, false);
            J = __builtin_amdgcn_udot4(w[1], 0x07060504u, J, false);
            J = __builtin_amdgcn_udot4(w[2], 0x0b0a0908u, J, false);
            J = __builtin_amdgcn_udot4(w[3], 0x0f0e0d0cu, J, false);
            uint32_t g = o.base + off;                         // < 65521 + 4096
            g = g >= 65521 ? g - 65521 : g;
            o.accS += A;                                       // <= 4080 per piece
            o.accI = (o.accI + g * A + J) % 65521;             // 65520*4080 + 30600 + 65520 < 2^32
        }
        o.accS %= 65521;
        o.flushed = uni64(o.flushed + n);
        o.base = UNI((o.base + n) % 65521);
    }
    LDS_ORDER();
}

// cooperative LZ77 copy of `count` (<= 258) bytes from `offset` back (InflatorOut.expand,
// InflatorOut.swift:124-139: forward byte copy, an overlapping run replicates).  All loads are
// issued before the first store; sources are always below `pos`, so they never alias the stores.
__device__ __forceinline__ void copy_match(Lds &s, const Out &o, uint64_t pos, uint32_t count, uint32_t offset, int lane)
{
    if (count <= 64 && offset >= count && offset <= LDS_REACH) {      // the common case: one pass, no overlap
        if ((uint32_t)lane < count)
            s.ring[(pos + lane) & (RING - 1)] = s.ring[(pos - offset + lane) & (RING - 1)];
        return;
    }
    uint32_t v[5];
    if (offset <= LDS_REACH) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t i = lane + 64 * j;
            if (i < count) {
                const uint32_t k = offset >= count ? i : i % offset;
                v[j] = s.ring[(pos - offset + k) & (RING - 1)];
            }
        }
    } else {
        // the source may already be overwritten in the ring; it was flushed to HBM long ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t i = lane + 64 * j;
            if (i < count) v[j] = o.dst[pos - offset + i];
        }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t i = lane + 64 * j;
        if (i < count) s.ring[(pos + i) & (RING - 1)] = (uint8_t)v[j];
    }
}

#ifdef SPNG_INFLATE_PROF
#define PROF_DECL uint64_t pt[8] = {0,0,0,0,0,0,0,0}, pc[8] = {0,0,0,0,0,0,0,0}, p_t0 = 0;
#define PROF_BEGIN() p_t0 = __builtin_readcyclecounter()
#define PROF_END(k) do { pt[k] += __builtin_readcyclecounter() - p_t0; pc[k] += 1; } while (0)
#else
#define PROF_DECL
#define PROF_BEGIN()
#define PROF_END(k)
#endif
#define FAIL(code, a0, a1) do { status = (code); aux0 = (a0); aux1 = (a1); goto done; } while (0)

__global__ __launch_bounds__(64) void inflate_kernel(const InflateJob *__restrict__ jobs,
                                                     spng_result *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) Lds s;
    const InflateJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const uint8_t *src = job.src;
    const uint64_t n = job.src_len, total = n * 8;
    const unsigned long long below = (1ull << lane) - 1;

    int32_t status = SPNG_NEED_MORE_INPUT;
    uint64_t aux0 = 0, aux1 = 0;
    PROF_DECL
    Out o = { job.dst, job.dst_cap, 0, 0, 0, 0, 0 };
    Reader r;
    seek(s, r, src, n, 0, lane);

    // .initial (InflatorBuffers.swift:92-104, StreamHeader.swift:16-54)
    if (job.format != SPNG_FORMAT_IOS) {
        if (16 > total) goto done;
        const uint32_t cm = TAKE(4);
        if (cm != 8) FAIL(SPNG_E_COMPRESSION_METHOD, cm, 0);
        const uint32_t e = TAKE(4);
        if (e >= 8) FAIL(SPNG_E_WINDOW_SIZE, e + 8, 0);
        const uint32_t flags = TAKE(8);
        if (((e << 12 | 8 << 8) + flags) % 31 != 0) FAIL(SPNG_E_CHECK_BITS, 0, 0);
        if (flags & 0x20) FAIL(SPNG_E_DICTIONARY, 0, 0);
    }

    for (;;) {
        // .metadata: readBlockMetadata (InflatorBuffers.Stream.swift:59-141)
        PROF_BEGIN();
        if (bitpos(r) + 3 > total) goto done;
        const uint32_t bfinal = TAKE(1);
        const uint32_t type = TAKE(2);
        if (type == 0) {
            const uint64_t boundary = (bitpos(r) + 7) & ~(uint64_t)7;
            if (boundary + 32 > total) goto done;
            TAKE((uint32_t)(boundary - bitpos(r)));
            const uint32_t l = TAKE(16);
            const uint32_t m = TAKE(16);
            if (l != (~m & 0xffffu)) FAIL(SPNG_E_BLOCK_COUNT_PARITY, l, m);
            // readBlock(upTo:) (:384-399): copies as many of the LEN bytes as the input holds
            const uint64_t from = boundary / 8 + 4;
            const uint64_t have = n - from < l ? n - from : l;
            if (o.pos + have > o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
            for (uint64_t done_ = 0; done_ < have;) {
                const uint64_t piece = have - done_ < 1024 ? have - done_ : 1024;
                for (uint64_t i = lane; i < piece; i += 64) s.ring[(o.pos + i) & (RING - 1)] = src[from + done_ + i];
                LDS_ORDER();
                o.pos = uni64(o.pos + piece); done_ += piece;
                if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
            }
            if (have < l) { r.wd = (n + 3) >> 2; r.bit = 0; goto done; }
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
                const uint32_t literals = 257 + TAKE(5);
                const uint32_t distances = 1 + TAKE(5);
                const uint32_t codelengths = 4 + TAKE(4);
                if (bitpos(r) + 3 * (uint64_t)codelengths > total) goto done;
                if (literals > 286) FAIL(SPNG_E_RUNLITERAL_COUNT, literals, 0);
                // 19 code-length-code lengths in zig-zag order (:120-125)
                uint64_t packed = 0;                           // 19 x 3 bits = 57 bits
                for (uint32_t i = 0; i < codelengths; ++i) packed |= (uint64_t)TAKE(3) << (3 * i);
                if (lane < 19) {
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    s.lens[order[lane]] = (uint32_t)lane < codelengths ? (uint8_t)((packed >> (3 * lane)) & 7) : 0;
                }
                __syncthreads();
                if (!build<2>(s.lens, 19, s.lit, MBITS, s.sorted_lit, &s.tlit, false, lane))
                    FAIL(SPNG_E_CODELENGTH_TABLE, 0, 0);

                // .tables: readBlockTables (:144-263), sequential RLE decode of the code lengths
                const uint32_t want = literals + distances;
                uint32_t have = 0, last = 0;
                while (have < want) {
                    if (bitpos(r) >= total) goto done;
                    const uint32_t bits = peek32(r);
                    const uint32_t e = UNI(s.lit[bits & ((1 << MBITS) - 1)]);
                    const uint32_t len = e & 15, sym = e >> 16;
                    if (bitpos(r) + len > total) goto done;
                    if (sym < 16) {
                        advance(s, r, src, n, lane, len);
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
                    const uint32_t reps = base + ((bits >> len) & ((1u << extra) - 1));
                    advance(s, r, src, n, lane, len + extra);
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

            PROF_END(6);
            // .compressed: readBlock(with:) (:266-381)
            for (;;) {
                const uint64_t b0 = bitpos(r);
                if (b0 >= total) goto done;
                // ---- speculative window: lane i decodes the whole token that would start at bit
                //      b0 + i (lit/len LUT, extra bits, distance LUT, extra bits)
                PROF_BEGIN();
                const uint32_t lo = peek32_at(r, (uint32_t)lane), hi = peek32_at(r, (uint32_t)lane + 32);
                uint32_t e = s.lit[lo & ((1 << LBITS) - 1)];
                const uint32_t len1 = e & 15, kind1 = (e >> 8) & 3, cx = (e >> 4) & 15;
                const uint32_t run = (e >> 16) + ((lo >> len1) & ((1u << cx) - 1));
                const uint32_t p2 = len1 + cx;                                   // <= 14
                const uint32_t dbits = (uint32_t)(((uint64_t)hi << 32 | lo) >> p2);
                const uint32_t d = s.dist[dbits & ((1 << DBITS) - 1)];
                const uint32_t dl = d & 15, ox = (d >> 4) & 15;
                const uint32_t dist = (d >> 16) + ((dbits >> dl) & ((1u << ox) - 1));
                const bool is_lit = kind1 == K_LIT && len1 != 0;
                // anything unusual (long codes, undefined codes, zero runs/offsets, end of block,
                // tokens running past the input) ends the chain and is decoded the slow way
                const bool is_match = kind1 == K_MATCH && len1 != 0 && dl != 0 && ((d >> 8) & 3) == K_MATCH &&
                                      run != 0 && dist != 0;
                const uint32_t tlen = is_lit ? len1 : p2 + dl + ox;              // <= 48
                const bool ok = (is_lit || is_match) && b0 + lane + tlen <= total;
                const uint32_t step = ok ? tlen : 0;
                const uint32_t outlen = is_lit ? 1u : run;

                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                PROF_END(0); PROF_BEGIN();
                // ---- resolve the true chain of token boundaries through the window (scalar unit)
                uint32_t p = 0, acc = 0, offs = 0;
                unsigned long long chain = 0;
                const uint64_t room = o.cap - o.pos;
                const uint32_t maxout = UNI(room < WINOUT ? (uint32_t)room : (uint32_t)WINOUT);
                // Taken scalar branches cost far more than the arithmetic here, so the walk is
                // unrolled and branch-free.  A lane whose step is 0 is absorbing (p stops moving),
                // the output budget is only checked afterwards (it almost never binds).
#pragma unroll
                for (int hop = 0; hop < 8; ++hop) {
                    const uint32_t q = p < 63 ? p : 63;
                    uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)step, (int)q);
                    const uint32_t ol = (uint32_t)__builtin_amdgcn_readlane((int)outlen, (int)q);
                    st = p < 64 ? st : 0u;
                    offs = ((uint32_t)lane == q) & (st != 0) ? acc : offs;
                    chain |= st ? 1ull << q : 0ull;
                    acc += st ? ol : 0u;
                    p += st;
                }
                bool good = p < 64 && __builtin_amdgcn_readlane((int)step, (int)(p < 63 ? p : 63)) != 0;
                if (acc > maxout) {                            // out of room (or > WINOUT bytes): walk again, carefully
                    p = 0; acc = 0; chain = 0; good = true;
                }
                while (good && p < 64) {                       // more than eight tokens in 64 bits: rare
                    const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)step, (int)p);
                    if (!st) break;
                    const uint32_t ol = (uint32_t)__builtin_amdgcn_readlane((int)outlen, (int)p);
                    if (acc + ol > maxout) break;
                    offs = (uint32_t)lane == p ? acc : offs;
                    chain |= 1ull << p;
                    acc += ol; p += st;
                }
                PROF_END(1);
                if (chain) {
                    PROF_BEGIN();
                    const bool mine = (chain >> lane) & 1;
                    // literals first (no token ever reads a later token's bytes) ...
                    if (mine && is_lit) s.ring[(o.pos + offs) & (RING - 1)] = (uint8_t)(e >> 16);
                    LDS_ORDER();
                    // ... then the back-references, in stream order
                    unsigned long long mm = __ballot(mine && !is_lit);
                    while (mm) {
                        const int l = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)run, l);
                        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)dist, l);
                        const uint64_t at = o.pos + (uint32_t)__builtin_amdgcn_readlane((int)offs, l);
                        if (off > at) { o.pos = uni64(at); FAIL(SPNG_E_STRING_REFERENCE, 0, 0); }
                        copy_match(s, o, at, cnt, off, lane);
                        LDS_ORDER();
                    }
                    o.pos = uni64(o.pos + acc);
                    PROF_END(2); PROF_BEGIN();
                    advance(s, r, src, n, lane, p);
                    PROF_END(3); PROF_BEGIN();
                    if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
                    PROF_END(4);
                    if (p >= 64) continue;                     // the chain used the whole window
                    if (bitpos(r) >= total) goto done;
                }
                e = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)p);   // LUT entry of the token the chain stopped on

                // ---- that token, decoded wave-uniformly with every check of the reference
                PROF_BEGIN();
                const uint64_t b1 = bitpos(r);
                uint64_t slug = peek64(r);
                if ((e & 15) == 0) e = decode_long<0>((uint32_t)slug, s.tlit, s.sorted_lit, LBITS);
                const uint32_t len = e & 15, kind = (e >> 8) & 3;
                if (kind == K_LIT) {
                    if (b1 + len > total) goto done;
                    if (o.pos >= o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
                    advance(s, r, src, n, lane, len);
                    if (lane == 0) s.ring[o.pos & (RING - 1)] = (uint8_t)(e >> 16);
                    o.pos = uni64(o.pos + 1);
                } else if (kind == K_EOB) {
                    if (b1 + len > total) goto done;
                    advance(s, r, src, n, lane, len);
                    break;
                } else {
                    slug >>= len;
                    const uint32_t ex = (e >> 4) & 15;
                    const uint32_t count = (e >> 16) + ((uint32_t)slug & ((1u << ex) - 1));
                    slug >>= ex;
                    uint32_t dd = UNI(s.dist[(uint32_t)slug & ((1 << DBITS) - 1)]);
                    if ((dd & 15) == 0) dd = decode_long<1>((uint32_t)slug, s.tdist, s.sorted_dist, DBITS);
                    if (((dd >> 8) & 3) == K_UNDEF) FAIL(SPNG_E_REFERENCE_UNDEFINED, 0, 0);
                    slug >>= dd & 15;
                    const uint32_t dx = (dd >> 4) & 15;
                    const uint32_t offset = (dd >> 16) + ((uint32_t)slug & ((1u << dx) - 1));
                    const uint32_t bits = len + ex + (dd & 15) + dx;     // <= 48
                    if (b1 + bits > total) goto done;
                    if (offset > o.pos) FAIL(SPNG_E_STRING_REFERENCE, 0, 0);
                    if (count && !offset) FAIL(SPNG_E_REFERENCE_UNDEFINED, 0, 0);
                    if (o.pos + count > o.cap) FAIL(SPNG_E_OUTPUT_CAPACITY, 0, 0);
                    advance(s, r, src, n, lane, bits);
                    LDS_ORDER();
                    copy_match(s, o, o.pos, count, offset, lane);
                    LDS_ORDER();
                    o.pos = uni64(o.pos + count);
                }
                PROF_END(5);
                if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
            }
        } else {
            FAIL(SPNG_E_BLOCK_TYPE, type, 0);
        }
        if (bfinal) break;
    }

    // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429)
    if (job.format != SPNG_FORMAT_IOS) {
        const uint64_t boundary = (bitpos(r) + 7) & ~(uint64_t)7;
        if (boundary + 32 > total) goto done;
        TAKE((uint32_t)(boundary - bitpos(r)));
        uint32_t declared = 0;
        for (int k = 0; k < 4; ++k) declared = declared << 8 | TAKE(8);
        flush(s, o, o.pos, lane);
        const uint32_t S = UNI(wave_sum(o.accS % 65521)) % 65521, I = UNI(wave_sum(o.accI)) % 65521;
        const uint32_t N = (uint32_t)(o.pos % 65521);
        const uint32_t computed = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
        if (declared != computed) FAIL(SPNG_E_STREAM_CHECKSUM, declared, computed);
    }
    status = SPNG_DONE;
done:
    flush(s, o, o.pos, lane);
#ifdef SPNG_INFLATE_PROF
    if (lane == 0 && blockIdx.x == 0)
        printf("prof cycles/count: spec %llu/%llu chain %llu/%llu emit %llu/%llu advance %llu/%llu flush %llu/%llu slow %llu/%llu header %llu/%llu\n",
               pt[0], pc[0], pt[1], pc[1], pt[2], pc[2], pt[3], pc[3], pt[4], pc[4], pt[5], pc[5], pt[6], pc[6]);
#endif
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
