// inflate.hip -- batched zlib / raw-DEFLATE inflate for gfx950: one workgroup (three working
// wavefronts) per stream.
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
// on token k), so the unit of parallelism is the stream: one workgroup per stream, four workgroups
// per CU (the 32 KiB window of each lives in LDS) -> 1024 streams in flight on the chip.  A lone
// wavefront issues roughly one dependent instruction every 7-8 cycles, so what bounds a stream is
// the length of the instruction sequence per token on its critical path; the work is therefore cut
// into three stages that run as three wavefronts on three SIMDs, coupled by small LDS rings:
//   * wave 2, the SCOUT, cuts the compressed data of a block into 64-bit windows aligned to its
//     first bit and decodes, for every lane of every window, the whole token that would start at
//     that bit (lit/len LUT, extra bits, distance LUT, extra bits) -- independent of where the real
//     token boundaries fall, hence ahead of everything else and in a software pipeline in which
//     every LDS result is used one iteration after its load was issued.  One 32-bit token per bit
//     position goes into a ring of WD window records.
//   * wave 0, the WALKER, finds the true chain of token boundaries through each window: the lanes'
//     token lengths define a successor map, and the walk over it runs on v_readlane, two tokens per
//     hop over the squared map (a hop is ~50 cycles of pure latency).  The lanes on the chain
//     append their tokens (stream order = popcount of the chain mask below the lane) to a token
//     queue.  The token on which a chain stops (end of block, long code, anything unusual) is
//     decoded wave-uniformly with every check of the reference.  Block headers, table construction
//     and stored blocks are the walker's too; the scout is started per block once the tables stand.
//   * wave 1, the RESOLVER, owns the output window (LDS ring = the whole DEFLATE window).  It takes
//     up to 64 tokens at a time, prefix-sums their lengths across the wave, stores all literals in
//     one LDS write, then replays the back-references in stream order as LDS->LDS copies
//     (overlapping runs replicate via i mod distance), flushes the ring to HBM in aligned 4 KiB
//     pieces with 16 B/lane coalesced stores and folds Adler-32 into the flush (v_sad_u8 / v_dot4
//     weighted sums), so the inflated bytes are never re-read.  Position-dependent checks
//     (reference before the start of the stream, output capacity) live here.
//   * wave 3 does nothing (see the kernel: 256-thread workgroups are placed evenly, 192 are not).
// Both rings are single-producer single-consumer: a counter publishes, a counter releases, LDS
// executes one wave's operations in issue order so neither side ever waits for a store to land.
// Errors keep stream order because the resolver drains everything that was queued before it looks
// at the walker's final status.  All control flow on the hot paths is wave-uniform and the spin
// loops have no trap exits: either makes the compiler thread exec masks and guard flags through
// every loop, which costs more than the work itself.
// Tables are LDS resident: a 2^10-entry lit/len LUT and a 2^8-entry distance LUT whose 32-bit
// entries already carry base value + extra-bit count, and a canonical first-code/count fallback
// for the rare longer codes.  They are rebuilt per block by the whole wave (LDS histogram, DPP
// prefix sums, radix-match ranking; nothing per code length lives in scalar registers) --
// swift-png's own encoder emits a dynamic block every <= 2047 tokens, so this is hot.  Compressed
// input is staged through a 512-byte LDS ring with coalesced 16 B/lane loads; a position is a bit
// index and every fetch reads three dwords straight from the ring.
#include "common.hpp"
#include "huffman.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };
// The stream and output pointers are rebuilt from scalar registers; typed as global memory they
// compile to global_load/global_store (a generic pointer would cost flat instructions, which also
// tie up the LDS counter).
typedef U128u __attribute__((address_space(1))) gU128u;

static constexpr int RING = 32768;           // output window in LDS (power of two)
static constexpr int INR = 512;              // input ring (two 256-byte halves)
static constexpr int HALF = INR / 2;
static constexpr int FLUSH = 4096;           // flush granularity
static constexpr int WINOUT = 1024;          // most bytes one batch of tokens may produce
static constexpr int QN = 128;               // token queue entries (power of two; a window yields <= 64 tokens)
static constexpr int WD = 4;                 // decoded windows in flight between scout and walker (power of two)
#ifndef SPNG_BATCH_MIN
#define SPNG_BATCH_MIN 32
#endif
static constexpr uint32_t BATCH_MIN = SPNG_BATCH_MIN;    // the resolver waits for this many tokens (or the end)
// Literals of a batch are stored before its back-references are resolved, up to WINOUT bytes
// ahead of a reference; a source is still intact in the ring if it is not further back than this.
static constexpr uint32_t LDS_REACH = RING - WINOUT - 258 - 16;
static constexpr int LBITS = 10, DBITS = 8, MBITS = 7;

// Tokens, as the scout writes them (one per bit position) and as they travel through the queue:
//   literal          byte << 8
//   back-reference   1 << 7 | (run - 3) << 8 | (distance - 1) << 16                (15-bit distance)
// with the token's length in bits (scout -> walker only; 0 = not for the fast path) in bits [5:0].
// Tokens the walker decodes itself may be degenerate (run 0, checks the reference leaves to the
// output side) and use the wide forms:
//   back-reference   1 << 31 | run << 16 | distance
//   capacity check   3 << 30 | bytes        ("the next `bytes` bytes must fit the output": stored
//                                             blocks check their whole length up front)
static constexpr uint32_t T_MATCH = 0x80000000u, T_CHECK = 0xc0000000u, T_REF = 0x80u;

struct Ctrl {
    uint32_t tail, head;           // tokens published by the walker / released by the resolver
    uint32_t a_done, b_fail;
    uint32_t a_wait, pad1;         // the walker is blocked on a full queue: take whatever is there
    int32_t  status;               // walker's final status; SPNG_DONE + check => compare Adler-32
    uint32_t check, declared, pad;
    uint64_t aux0, aux1, bits;
    uint64_t blk_bit, tok_bit;     // the header of the block the decoder stopped in; the first token (or stored byte) that was not complete (0: none)
    uint64_t r_tok, r_done;        // a call that goes on inside a block: the bit to go on at, the bytes of the block already produced
    // (all four live here, not in the decoder's registers: its scalar registers are the walker loop's)
    // scout <-> walker
    uint32_t w_gen, w_stop, w_idle, w_quit;   // start order (generation), stop order, acknowledgement, exit
    uint32_t w_prod, w_cons;                  // windows of this generation produced / consumed
    uint64_t w_org;                           // first bit of window 0 of this generation
};

// What the scout hands to the walker for one 64-bit window: per lane the token that would start at
// that bit.
struct WinRec { uint32_t tok[64]; };

struct Lds {                       // 40,928 bytes: four streams per CU
    uint8_t  ring[RING];
    uint32_t q[QN + 4];            // + a slot nobody reads, so that stores need no branch
    Ctrl     c;
    uint8_t  in[INR + 16];         // + mirror of the first 16 bytes
    uint32_t lit[1 << LBITS];      // the code-length-code LUT (2^MBITS entries) lives here while a
                                   // dynamic header is parsed, i.e. before this table is built
    uint32_t dist[1 << DBITS];
    uint16_t sorted_lit[288];      // symbols in canonical order, for codes longer than the LUT index
    uint16_t sorted_dist[32];
    Tree     tlit, tdist;
    union {
        WinRec win[WD];            // compressed data: scout -> walker
        struct {                   // block header (the scout is idle): table construction scratch
            uint8_t  lens[464];    // 286 + 32 code lengths + worst-case RLE overshoot (138)
            uint32_t hist[16], run[16];    // symbols per code length, ranked so far
        };
    };
};

// Bit reader.  The compressed stream is staged through a small LDS ring; a position is just a bit
// index, every fetch reads three consecutive dwords at that position straight from the ring (the
// ring carries a 16-byte mirror of its head so that a fetch never has to wrap), and advancing is an
// addition plus a countdown to the next staging point.
struct Reader {
    uint64_t pos;                        // absolute bit position (wave-uniform)
    int32_t  left;                       // bits until the position enters the half staged last
};

// stage HALF bytes of the stream starting at `from` (multiple of HALF) into the input ring; bytes
// past the end read as zero (the reference pads 48 zero bits, LZ77.InflatorIn.swift:130-133)
__device__ void stage(Lds &s, const gbyte *src, uint64_t n, uint64_t from, int lane)
{
    if (lane < HALF / 16) {
        const uint64_t off = from + (uint64_t)lane * 16;
        u32x4 v = {0, 0, 0, 0};
        if (off + 16 <= n) v = ((const gU128u *)(src + off))->v;
        else if (off < n) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; ++k) if (off + k < n) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        const uint32_t at = (uint32_t)(from + lane * 16) & (INR - 1);
        *(u32x4 *)(s.in + at) = v;
        if (at == 0) *(u32x4 *)(s.in + INR) = v;               // the mirror
    }
    WSYNC();
}
__device__ __forceinline__ void seek(Lds &s, Reader &r, const gbyte *src, uint64_t n, uint64_t byte, int lane)
{
    const uint64_t h = byte & ~(uint64_t)(HALF - 1);
    stage(s, src, n, h, lane);
    stage(s, src, n, h + HALF, lane);
    r.pos = uni64(byte * 8);
    r.left = (int32_t)UNI((uint32_t)(h + HALF - byte) * 8);
}
// reposition at an arbitrary bit (the walker taking the input ring back from the scout)
__device__ __forceinline__ void seek_bits(Lds &s, Reader &r, const gbyte *src, uint64_t n, uint64_t bit, int lane)
{
    const uint64_t h = (bit >> 3) & ~(uint64_t)(HALF - 1);
    stage(s, src, n, h, lane);
    stage(s, src, n, h + HALF, lane);
    r.pos = uni64(bit);
    r.left = (int32_t)UNI((uint32_t)((h + HALF) * 8 - bit));
}
__device__ __forceinline__ void advance(Lds &s, Reader &r, const gbyte *src, uint64_t n, int lane, uint32_t k)
{
    r.pos = uni64(r.pos + k);                                  // k < 8 * HALF
    r.left = (int32_t)UNI((uint32_t)r.left - k);
    if (r.left <= 0) {
        stage(s, src, n, ((r.pos >> 3) & ~(uint64_t)(HALF - 1)) + HALF, lane);
        r.left = (int32_t)UNI((uint32_t)r.left + 8 * HALF);
    }
}
// 64 stream bits starting at bit position `bit` (low 32 bits of the absolute position suffice)
__device__ __forceinline__ void fetch64(const Lds &s, uint32_t bit, uint32_t &lo, uint32_t &hi)
{
    const uint32_t *w = (const uint32_t *)(s.in + ((bit >> 3) & (INR - 4)));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    lo = __builtin_amdgcn_alignbit(d1, d0, bit);
    hi = __builtin_amdgcn_alignbit(d2, d1, bit);
}
__device__ __forceinline__ uint32_t peek32(const Lds &s, const Reader &r)
{
    uint32_t lo, hi;
    fetch64(s, (uint32_t)r.pos, lo, hi);
    return UNI(lo);
}
__device__ __forceinline__ uint64_t peek64(const Lds &s, const Reader &r)
{
    uint32_t lo, hi;
    fetch64(s, (uint32_t)r.pos, lo, hi);
    return (uint64_t)UNI(hi) << 32 | UNI(lo);
}
#define TAKE(k) take(s, r, src, n, lane, (k))
__device__ __forceinline__ uint32_t take(Lds &s, Reader &r, const gbyte *src, uint64_t n, int lane, uint32_t k)
{
    const uint32_t v = peek32(s, r) & ((1u << k) - 1);       // k <= 16
    advance(s, r, src, n, lane, k);
    return v;
}
__device__ __forceinline__ uint64_t bitpos(const Reader &r) { return r.pos; }

template <int KIND>
__device__ __forceinline__ uint32_t decode_long(uint32_t bits, const Tree &t, const uint16_t *sorted, int lbits, int lane)
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
    gbyte *dst; uint64_t cap;
    uint64_t pos, flushed;
    // Adler-32 (MRC32.swift:26-50) in a form that needs no cross-lane traffic until the very end:
    // with S = sum b_i and I = sum i*b_i over the whole stream of N bytes,
    //     s1 = 1 + S,   s2 = N + N*S - I      (mod 65521)
    // so every lane just accumulates its share of S and I (mod 65521) while flushing.
    uint32_t accS, accI;
    uint32_t base;                                             // flushed mod 65521 (uniform)
};

#ifdef SPNG_INFLATE_PROF
#define PROFC(x) ((x) += 1)
#define PROF_DECL uint64_t pt[8] = {0,0,0,0,0,0,0,0}, pc[8] = {0,0,0,0,0,0,0,0}, p_t1 = 0;
#define PROF_BEGIN() p_t1 = __builtin_readcyclecounter()
#define PROF_END(k) do { pt[k] += __builtin_readcyclecounter() - p_t1; pc[k] += 1; } while (0)
#else
#define PROFC(x)
#define PROF_DECL
#define PROF_BEGIN()
#define PROF_END(k)
#endif
#define LDS_ORDER() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define LDS_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LDS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
static constexpr uint32_t SPIN_LIMIT = 1u << 24;               // a lost partner traps instead of hanging the GPU
// (ending the wave with an opaque instruction: a trap would be a block terminator, i.e. one more
//  loop exit for the compiler to thread guard flags around.  The partner waves run into their own
//  limits, the kernel ends, and the caller finds a result that was never written.)
#define SPIN_ABORT() asm volatile("s_endpgm")

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
                ((gU128u *)(o.dst + o.flushed + off))->v = v;
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

// ------------------------------------------------------------------------------------------------
// wave 1: the resolver
// ------------------------------------------------------------------------------------------------
// out_pos: where this call's first byte goes (a call resumed inside a block: behind what earlier calls took of it); blk_out0: the
// output position of the header the call starts from.  The decoder marks every block start in the token stream (a capacity check
// of zero bytes), so that the resolver knows the output position of the LAST block started: what a caller needs to resume there.
__device__ __attribute__((always_inline)) void resolver(Lds &s, gbyte *dst, uint64_t dst_cap, uint64_t src_len, bool resumed, bool internal,
                                                        uint64_t blk_out0, uint64_t out_pos, spng_result *__restrict__ result, int lane)
{
#ifndef SPNG_B_PRIO
#define SPNG_B_PRIO 2
#endif
    __builtin_amdgcn_s_setprio(SPNG_B_PRIO);               // walker 3 > resolver 2 > scout 0 (the scout has slack)
    // resumed: the window is what earlier calls left in the output.  (Flushing starts on the 16-byte piece the
    // position lies in -- flush reads the ring in aligned pieces -- so up to 15 bytes are written once more.)
    Out o = { dst, dst_cap, out_pos, out_pos & ~(uint64_t)15, 0, 0, (uint32_t)((out_pos & ~(uint64_t)15) % 65521) };
    for (uint64_t p = (out_pos > (uint64_t)RING ? out_pos - RING : 0) + lane; p < out_pos; p += 64) s.ring[p & (RING - 1)] = dst[p];
    uint32_t head = 0, spins = 0;
#ifdef SPNG_INFLATE_PROF
    uint64_t p_empty = 0, p_batches = 0, p_t0 = __builtin_readcyclecounter();
#endif
    int32_t status = SPNG_NEED_MORE_INPUT;
    uint64_t aux0 = 0, aux1 = 0, bits = 0;
    uint64_t blk_out = blk_out0;
    bool failed = false;

    for (;;) {
        // Take tokens in large batches: the fixed cost of a batch is paid once, and a resolver that
        // polls an almost empty queue only steals issue slots and LDS cycles from the decoder.
        uint32_t tail = UNI(LDS_LOAD(&s.c.tail));
        if (tail - head < BATCH_MIN && !(tail != head && UNI(LDS_LOAD(&s.c.a_wait)))) {
            if (UNI(LDS_LOAD(&s.c.a_done))) {
                LDS_ACQUIRE();
                tail = UNI(LDS_LOAD(&s.c.tail));
                if (tail == head) break;
            } else {
#ifndef SPNG_B_SLEEP
#define SPNG_B_SLEEP 4
#endif
                __builtin_amdgcn_s_sleep(SPNG_B_SLEEP);
                if (++spins > SPIN_LIMIT) SPIN_ABORT();
                PROFC(p_empty);
                continue;
            }
        }
        spins = 0;
        PROFC(p_batches);
        COMPILER_ORDER();
        uint32_t m = tail - head < 64 ? tail - head : 64;
        const uint32_t t = s.q[(head + lane) & (QN - 1)];
        const bool wide = (t & T_MATCH) != 0, is_check = (t & T_CHECK) == T_CHECK;
        const bool is_match = wide ? !is_check : (t & T_REF) != 0;
        const uint32_t run = wide ? (t >> 16) & 0x1ff : ((t >> 8) & 0xff) + 3;
        const uint32_t dist = wide || is_check ? t & 0xffff : (t >> 16) + 1;
        const uint32_t len = (uint32_t)lane >= m ? 0u : is_match ? run : is_check ? 0u : 1u;
        // inclusive prefix sum of the token lengths: row scan on DPP, rows stitched on the scalar unit
        uint32_t incl = len;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, false);
        {
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
            const uint32_t r1 = r0 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
            const uint32_t r2 = r1 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 47);
            incl += lane < 16 ? 0u : lane < 32 ? r0 : lane < 48 ? r1 : r2;
        }
        // a batch may run at most WINOUT bytes ahead of the references it still has to resolve
        const unsigned long long over = __ballot(incl > WINOUT);
        if (over) m = (uint32_t)__ffsll((long long)over) - 1;  // >= 1: a single token is <= 258 bytes
        const uint32_t offs = incl - len;
        const uint64_t at = o.pos + offs;
        const uint32_t need = is_match ? run : is_check ? dist : 1u;
        const bool bad_ref = is_match && (uint64_t)dist > at;
        const unsigned long long bad = __ballot((uint32_t)lane < m && (bad_ref || at + need > o.cap));
        if (bad) {
            // InflatorBuffers.Stream.swift:352-366: the reference is validated before the capacity
            m = (uint32_t)__ffsll((long long)bad) - 1;
            status = __builtin_amdgcn_readlane(bad_ref ? SPNG_E_STRING_REFERENCE : SPNG_E_OUTPUT_CAPACITY, (int)m);
            failed = true;
        }
        {   // block starts among the tokens taken: the last one's output position
            const unsigned long long marks = __ballot((uint32_t)lane < m && is_check && dist == 0);
            if (marks) blk_out = uni64(o.pos + (uint32_t)__builtin_amdgcn_readlane((int)offs, 63 - __clzll((long long)marks)));
        }
        head = UNI(head + m + (failed ? 1u : 0u));
        COMPILER_ORDER();                                      // the token loads were issued: release their slots
        LDS_STORE(&s.c.head, head);
        const bool mine = (uint32_t)lane < m;
        // ring offsets need only the low bits of the position
        const uint32_t dsto = ((uint32_t)o.pos + offs) & (RING - 1);
        // literals first (no token ever reads a later token's bytes) ...
        // (LDS executes one wave's operations in issue order: a later read sees an earlier write
        //  without any wait, so the compiler only has to keep the order)
        if (mine && !is_match && !is_check) s.ring[dsto] = (uint8_t)(t >> 8);
        COMPILER_ORDER();
        // ... then the back-references, in stream order.  The common kind (one pass of the wave, no
        // overlap, source still in the ring) is a byte read + byte write per lane; everything it
        // needs was computed per lane above, so a reference costs two v_readlane.
        const bool refs = mine && is_match && run != 0;
        const uint32_t srco = (dsto - dist) & (RING - 1);
        const uint32_t both = dsto | run << 16;
        unsigned long long mm = __ballot(refs);
        const unsigned long long plain = __ballot(refs && run <= 64 && dist >= run && dist <= LDS_REACH);
        while (mm) {
            const int l = __ffsll((long long)mm) - 1;
            const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)both, l);
            const uint32_t from = (uint32_t)__builtin_amdgcn_readlane((int)srco, l);
            const uint32_t to = pk & 0xffff, cnt = pk >> 16;
            if ((plain >> l) & 1) {
                if ((uint32_t)lane < cnt) s.ring[(to + lane) & (RING - 1)] = s.ring[(from + lane) & (RING - 1)];
            } else {
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)dist, l);
                const uint64_t dstpos = o.pos + (uint32_t)__builtin_amdgcn_readlane((int)offs, l);
                copy_match(s, o, dstpos, cnt, off, lane);
            }
            COMPILER_ORDER();
            mm &= mm - 1;
        }
        const uint32_t produced = (uint32_t)__builtin_amdgcn_readlane((int)offs, (int)(m < 63 ? m : 63));
        o.pos = uni64(o.pos + (m < 64 ? produced : (uint32_t)__builtin_amdgcn_readlane((int)incl, 63)));
        if (failed) break;
        if (o.pos - o.flushed >= FLUSH) flush(s, o, o.pos & ~(uint64_t)15, lane);
    }

    if (failed) {
        if (lane == 0) LDS_STORE(&s.c.b_fail, 1u);             // the decoder stops at its next full queue
    } else {
        status = (int32_t)UNI(s.c.status);
        aux0 = uni64(s.c.aux0); aux1 = uni64(s.c.aux1); bits = uni64(s.c.bits);
        if (status == SPNG_DONE && UNI(s.c.check)) {
            // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429)
            flush(s, o, o.pos, lane);
            const uint32_t declared = UNI(s.c.declared);
            const uint32_t S = UNI(wave_sum(o.accS % 65521)) % 65521, I = UNI(wave_sum(o.accI)) % 65521;
            const uint32_t N = (uint32_t)(o.pos % 65521);
            const uint32_t computed = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
            if (declared != computed) { status = SPNG_E_STREAM_CHECKSUM; aux0 = declared; aux1 = computed; }
        }
    }
    flush(s, o, o.pos, lane);
#ifdef SPNG_INFLATE_PROF
    if (lane == 0 && blockIdx.x == 0)
        printf("resolver: %llu batches (%.1f tokens each), %llu empty polls, %llu cycles\n", p_batches,
               (double)head / (double)p_batches, p_empty, __builtin_readcyclecounter() - p_t0);
#endif
    if (lane == 0) {
        spng_result &res = *result;
        res.status = status; res.reserved = 0;
        res.written = o.pos;
        res.consumed = (bits + 7) / 8 > src_len ? src_len : (bits + 7) / 8;
        res.aux[0] = aux0; res.aux[1] = aux1;
        if (resumed && !internal && status == SPNG_NEED_MORE_INPUT) {
            // where the next call starts: the header of the block the input ends in and the bytes in front of it, and -- inside
            // that block -- the first token that was not complete (in BITS, in `consumed`; 0: at the header) with the bytes in
            // front of it (= written): spng_inflate_resume_batch's four words
            res.aux[0] = uni64(s.c.blk_bit); res.aux[1] = blk_out;
            res.consumed = uni64(s.c.tok_bit);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wave 2: the scout
// ------------------------------------------------------------------------------------------------
// The scout's software pipeline: three 64-bit windows are in flight, one between each pair of
// stages.  A stage reads the record its predecessor wrote in the previous iteration and overwrites
// the record its successor has just finished with -- the records stay put, so no value is ever
// copied while the load that produces it is still in flight.
struct WinF { uint32_t d0, d1, d2; };                    // fetch ->:  three stream dwords at the lane's bit
struct WinL { uint32_t lo, hi, e; };                     // litlen ->: the lane's next 64 bits, lit/len LUT entry
struct WinD { uint32_t e, run, p2, dbits, d; };          // dist ->:   + run length, bits so far, distance LUT entry

__device__ __forceinline__ void win_fetch(const Lds &s, WinF &f, uint32_t bit)
{
    const uint32_t *in = (const uint32_t *)(s.in + ((bit >> 3) & (INR - 4)));
    f.d0 = in[0]; f.d1 = in[1]; f.d2 = in[2];
}
__device__ __forceinline__ void win_litlen(const Lds &s, const WinF &f, WinL &l, uint32_t bit)
{
    l.lo = __builtin_amdgcn_alignbit(f.d1, f.d0, bit);
    l.hi = __builtin_amdgcn_alignbit(f.d2, f.d1, bit);
    l.e = s.lit[l.lo & ((1 << LBITS) - 1)];
}
__device__ __forceinline__ void win_dist(const Lds &s, const WinL &l, WinD &d)
{
    const uint32_t len1 = l.e & 15, cx = (l.e >> 4) & 15;
    d.e = l.e;
    d.run = (l.e >> 16) + ((l.lo >> len1) & ((1u << cx) - 1));
    d.p2 = len1 + cx;                                                // <= 14
    d.dbits = (uint32_t)(((uint64_t)l.hi << 32 | l.lo) >> d.p2);
    d.d = s.dist[d.dbits & ((1 << DBITS) - 1)];
}
// `rem` = stream bits from the first bit of the window to the end of the input (clamped)
__device__ __forceinline__ uint32_t win_finish(const WinD &d, uint32_t rem, int lane)
{
    const uint32_t dl = d.d & 15, ox = (d.d >> 4) & 15;
    const uint32_t dist = (d.d >> 16) + ((d.dbits >> dl) & ((1u << ox) - 1));
    // anything unusual (long codes, undefined codes, zero runs/offsets, end of block, tokens
    // running past the input) ends the chain and is decoded the slow way by the walker
    const bool is_lit = (d.e & F_LIT) != 0;
    const bool is_match = (d.e & d.d & F_REF) != 0;
    const uint32_t tlen = is_lit ? d.e & 15 : d.p2 + dl + ox;         // <= 48
    const bool ok = (is_lit || is_match) && (uint32_t)lane + tlen <= rem;
    const uint32_t step = ok ? tlen : 0;
    return step | (is_lit ? d.e >> 16 << 8 : T_REF | (d.run - 3) << 8 | (dist - 1) << 16);
}
__device__ __forceinline__ uint32_t window_rem(uint64_t total, uint64_t at)
{
    const uint64_t rem = total > at ? total - at : 0;
    return UNI(rem > 0xffffffffull ? 0xffffffffu : (uint32_t)rem);
}

static constexpr int QUARTER = INR / 4;      // the scout stages the input ring in quarters

// stage QUARTER bytes of the stream starting at `from` (multiple of QUARTER); see stage()
__device__ void stage_quarter(Lds &s, const gbyte *src, uint64_t n, uint64_t from, int lane)
{
    if (lane < QUARTER / 16) {
        const uint64_t off = from + (uint64_t)lane * 16;
        u32x4 v = {0, 0, 0, 0};
        if (off + 16 <= n) v = ((const gU128u *)(src + off))->v;
        else if (off < n) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; ++k) if (off + k < n) w[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
            v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        }
        const uint32_t at = (uint32_t)(from + lane * 16) & (INR - 1);
        *(u32x4 *)(s.in + at) = v;
        if (at == 0) *(u32x4 *)(s.in + INR) = v;               // the mirror
    }
    WSYNC();
}

// Decodes, for every bit position of the compressed data of the current block, the token that would
// start there -- window after window, ahead of the walker and without knowing where the real token
// boundaries fall.  Started by the walker once a block's tables stand (generation counter + first
// bit), stopped by it at the end of the block; what it decodes past the end of a block is never
// looked at.  While it runs it owns the input ring: the quarter behind the newest window stays, so
// the walker's slow path still finds the bytes around its own (older) position.
__device__ __attribute__((always_inline)) void scout(Lds &s, const gbyte *src, uint64_t n, int lane)
{
    const uint64_t total = n * 8;
    uint32_t gen = 0, quit = 0, idle_spins = 0;
#ifdef SPNG_INFLATE_PROF
    uint64_t sp_wait = 0, sp_stage = 0, sp_work = 0, sp_n = 0, sp_t = 0, sp_polls = 0;
#define SPROF_T() (sp_t = __builtin_readcyclecounter())
#define SPROF_ADD(x) ((x) += __builtin_readcyclecounter() - sp_t)
#else
#define SPROF_T()
#define SPROF_ADD(x)
#endif
    while (!quit) {
        const uint32_t g = UNI(LDS_LOAD(&s.c.w_gen));
        quit = UNI(LDS_LOAD(&s.c.w_quit));
        if (g == gen || quit) {
            // long naps: the scout may legitimately idle for as long as the walker copies stored
            // blocks (a whole level-0 stream), which must stay far away from the spin limit
            __builtin_amdgcn_s_sleep(32);
            if (++idle_spins > SPIN_LIMIT) SPIN_ABORT();
            continue;
        }
        idle_spins = 0;
        gen = g;
        COMPILER_ORDER();
        uint64_t wpos = uni64(s.c.w_org);                       // first bit of the window to finish next
        // the ring around the first window: the quarter before, its own, two ahead
        {
            const uint64_t q0 = (wpos >> 3) & ~(uint64_t)(QUARTER - 1);
            if (q0 >= QUARTER) stage_quarter(s, src, n, q0 - QUARTER, lane);
            stage_quarter(s, src, n, q0, lane);
            stage_quarter(s, src, n, q0 + QUARTER, lane);
            stage_quarter(s, src, n, q0 + 2 * QUARTER, lane);
        }
        // bits until the newest window (three ahead of wpos) starts in the next quarter
        int32_t left = (int32_t)UNI((uint32_t)(8 * QUARTER) - (uint32_t)((wpos + 192) & (8 * QUARTER - 1)));
        if (((wpos + 192) >> 3 & ~(uint64_t)(QUARTER - 1)) != ((wpos >> 3) & ~(uint64_t)(QUARTER - 1)))
            stage_quarter(s, src, n, (((wpos + 192) >> 3) & ~(uint64_t)(QUARTER - 1)) + 2 * QUARTER, lane);
#define SBIT(k) ((uint32_t)wpos + 64u * (k) + (uint32_t)lane)
        WinF F; WinL L; WinD D;                                // windows k+2, k+1, k
        {
            WinF f0, f1; WinL l0;
            win_fetch(s, f0, SBIT(0)); win_fetch(s, f1, SBIT(1)); win_fetch(s, F, SBIT(2));
            win_litlen(s, f0, l0, SBIT(0)); win_litlen(s, f1, L, SBIT(1));
            win_dist(s, l0, D);
        }
        uint32_t k = 0, cons_seen = 0, stop = 0;
        while (!stop) {
            // a free slot (and the stop order, which can only matter when the walker has stopped taking)
            if (__builtin_expect(k - cons_seen >= WD, 0)) {
                SPROF_T();
                for (uint32_t spins = 0; !stop; ++spins) {
                    PROFC(sp_polls);
                    cons_seen = UNI(LDS_LOAD(&s.c.w_cons));
                    if (k - cons_seen < WD) break;
                    stop = (UNI(LDS_LOAD(&s.c.w_stop)) == gen) | UNI(LDS_LOAD(&s.c.w_quit));
#ifndef SPNG_S_SLEEP
#define SPNG_S_SLEEP 1
#endif
                    __builtin_amdgcn_s_sleep(SPNG_S_SLEEP);
                    if (spins > SPIN_LIMIT) SPIN_ABORT();
                }
                SPROF_ADD(sp_wait);
            }
            if (!stop) {
                SPROF_T();
                PROFC(sp_n);
                s.win[k & (WD - 1)].tok[lane] = win_finish(D, window_rem(total, wpos), lane);
                k += 1;
                COMPILER_ORDER();                              // LDS executes in issue order: no wait needed
                LDS_STORE(&s.c.w_prod, k);
                // every window behind moves one stage forward
                wpos = uni64(wpos + 64);
                left = (int32_t)UNI((uint32_t)left - 64u);
                SPROF_ADD(sp_work);
                if (left <= 0) {
                    SPROF_T();
                    stage_quarter(s, src, n, (((wpos + 192) >> 3) & ~(uint64_t)(QUARTER - 1)) + 2 * QUARTER, lane);
                    left = (int32_t)UNI((uint32_t)left + 8u * QUARTER);
                    SPROF_ADD(sp_stage);
                }
                SPROF_T();
                win_dist(s, L, D);
                win_litlen(s, F, L, SBIT(1));
                win_fetch(s, F, SBIT(2));
                SPROF_ADD(sp_work);
            }
        }
#undef SBIT
        if (!UNI(LDS_LOAD(&s.c.w_quit))) LDS_STORE(&s.c.w_idle, gen);
    }
#ifdef SPNG_INFLATE_PROF
    if (lane == 0 && blockIdx.x == 0)
        printf("scout: %llu windows, work %llu, slot wait %llu (%llu polls), staging %llu cycles\n", sp_n, sp_work, sp_wait, sp_polls, sp_stage);
#endif
}

// ------------------------------------------------------------------------------------------------
// wave 0: the walker
// ------------------------------------------------------------------------------------------------
struct Queue {
    uint32_t tail, head_seen;
#ifdef SPNG_INFLATE_PROF
    uint64_t p_full, p_push;
#endif
};

// appends the tokens of the lanes in `who` (stream order = lane order).  Single exit on purpose: a
// return from inside the wait loop makes the compiler guard everything after it.  If the resolver
// has given up (it has already written the result) the queue is treated as empty and the decoder
// simply runs to the end of the stream; nobody reads what it queues.
__device__ __forceinline__ void push(Lds &s, Queue &q, unsigned long long who, uint32_t tok, int lane)
{
    const uint32_t k = (uint32_t)__popcll(who);
    if (__builtin_expect(q.tail + k - q.head_seen > QN, 0)) {
        q.head_seen = UNI(LDS_LOAD(&s.c.head));
        if (q.tail + k - q.head_seen > QN) {
            LDS_STORE(&s.c.a_wait, 1u);                        // (the resolver may be waiting for a fuller batch)
            for (uint32_t spins = 0;; ++spins) {
                q.head_seen = UNI(LDS_LOAD(&s.c.head));
                if (q.tail + k - q.head_seen <= QN) break;
                if (UNI(LDS_LOAD(&s.c.b_fail))) { q.head_seen = q.tail; break; }
                PROFC(q.p_full);
                __builtin_amdgcn_s_sleep(1);
                if (spins > SPIN_LIMIT) SPIN_ABORT();
            }
            LDS_STORE(&s.c.a_wait, 0u);
        }
    }
    // LDS executes one wave's operations in issue order, so publishing needs no wait: the compiler
    // only has to keep head load -> token store -> tail store in this order
    COMPILER_ORDER();
    PROFC(q.p_push);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(who >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)who, 0));
    s.q[(who >> lane) & 1 ? (q.tail + rank) & (QN - 1) : QN] = tok;      // q[QN]: write-only slot for idle lanes
    q.tail = UNI(q.tail + k);
    COMPILER_ORDER();
    // every lane stores the same value to the same address: one LDS operation, and no lane-dependent
    // branch for the compiler to fold into wave-uniform control flow
    LDS_STORE(&s.c.tail, q.tail);
}

#define FAIL(code, a0, a1) do { status = (code); aux0 = (a0); aux1 = (a1); goto done; } while (0)
#define PUSH(who, tok) push(s, q, (who), (tok), lane)

// start_bit: the block header to start on (0: the stream's first bit).  inside_: the call goes on INSIDE that block -- its
// header is read again for the tables (stored blocks: for LEN), then decoding continues at bit s.c.r_tok; s.c.r_done = bytes of
// the block earlier calls produced (stored blocks: how far into the LEN bytes).  The reference's inflator stops and resumes at
// any byte (LZ77.InflatorBuffers.Stream.swift:61-65, 284-288, 352-356).
__device__ __attribute__((always_inline)) void decoder(Lds &s, const gbyte *src, uint64_t n, int32_t format, bool resumed,
                                                       uint64_t start_bit, bool inside_, int lane)
{
    const uint64_t total = n * 8;
#ifndef SPNG_NO_PRIO
    __builtin_amdgcn_s_setprio(3);           // the decoder is the critical path of its workgroup
#endif

    int32_t status = SPNG_NEED_MORE_INPUT;
    uint64_t aux0 = 0, aux1 = 0;
    uint32_t check = 0, declared = 0, wgen = 0;
    bool inside = inside_;                                     // the first block: continue at s.c.r_tok
    Queue q = {};
    PROF_DECL
#ifdef SPNG_INFLATE_PROF
    const uint64_t p_t0 = __builtin_readcyclecounter();
#endif
    Reader r;
    seek(s, r, src, n, 0, lane);

    // a resumed stream (spng_inflate_resume_batch) starts on the block header an earlier call stopped in front of
    if (start_bit) seek_bits(s, r, src, n, start_bit, lane);
    // .initial (InflatorBuffers.swift:92-104, StreamHeader.swift:16-54)
    else if (format != SPNG_FORMAT_IOS) {
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
        s.c.blk_bit = bitpos(r);                               // (every lane, same value)
        if (!inside) PUSH(1ull, T_CHECK);                      // (a block starts here: the resolver notes its output position)
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
            // readBlock(upTo:) (:384-399): copies as many of the LEN bytes as the input holds, after
            // making sure all of them fit the output
            uint64_t from = boundary / 8 + 4;
            uint32_t l_left = l;
            if (inside) {                                      // (earlier calls copied r_done of the LEN bytes)
                const uint64_t done_in_block = uni64(s.c.r_done);
                const uint32_t dn = done_in_block < l ? (uint32_t)done_in_block : l;
                from += dn; l_left -= dn; inside = false;
            }
            const uint32_t have = n - from < l_left ? (uint32_t)(n - from) : l_left;
            if (have) PUSH(1ull, T_CHECK | have);               // (a check of 0 bytes is a block start)
            for (uint32_t done_ = 0; done_ < have; done_ += 64) {
                const uint32_t piece = have - done_ < 64 ? have - done_ : 64;
                const uint32_t b = (uint32_t)lane < piece ? src[from + done_ + lane] : 0u;
                PUSH(piece == 64 ? ~0ull : (1ull << piece) - 1, b << 8);
            }
            if (have < l_left) { r.pos = n * 8; s.c.tok_bit = n * 8; goto done; }
            seek(s, r, src, n, from + l_left, lane);
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                // fixed trees, HuffmanTree.swift:24-47
                for (int i = lane; i < 288; i += 64) s.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                WSYNC();
                build<0>(s.hist, s.run, s.lens, 288, s.lit, LBITS, s.sorted_lit, &s.tlit, false, lane);
                for (int i = lane; i < 32; i += 64) s.lens[i] = 5;
                WSYNC();
                build<1>(s.hist, s.run, s.lens, 32, s.dist, DBITS, s.sorted_dist, &s.tdist, false, lane);
            } else {
                if (bitpos(r) - 3 + 17 > total) goto done;
                const uint32_t literals = 257 + TAKE(5);
                const uint32_t distances = 1 + TAKE(5);
                const uint32_t codelengths = 4 + TAKE(4);
                if (bitpos(r) + 3 * (uint64_t)codelengths > total) goto done;
                if (literals > 286) FAIL(SPNG_E_RUNLITERAL_COUNT, literals, 0);
                // 19 code-length-code lengths in zig-zag order (:120-125)
                const uint64_t packed = peek64(s, r) & ((1ull << (3 * codelengths)) - 1);   // 19 x 3 bits = 57 bits
                advance(s, r, src, n, lane, 3 * codelengths);
                if (lane < 19) {
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    s.lens[order[lane]] = (uint32_t)lane < codelengths ? (uint8_t)((packed >> (3 * lane)) & 7) : 0;
                }
                WSYNC();
                if (!UB(build<2>(s.hist, s.run, s.lens, 19, s.lit, MBITS, s.sorted_lit, &s.tlit, false, lane)))
                    FAIL(SPNG_E_CODELENGTH_TABLE, 0, 0);

                // .tables: readBlockTables (:144-263), sequential RLE decode of the code lengths
                const uint32_t want = literals + distances;
                uint32_t have = 0, last = 0;
                while (have < want) {
                    if (bitpos(r) >= total) goto done;
                    const uint32_t bits = peek32(s, r);
                    const uint32_t e = UNI(s.lit[bits & ((1 << MBITS) - 1)]);
                    const uint32_t len = e & 15, sym = e >> 16;
                    if (bitpos(r) + len > total) goto done;
                    if (sym < 16) {
                        advance(s, r, src, n, lane, len);
                        s.lens[have] = (uint8_t)sym;            // every lane, same byte: no lane-dependent branch
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
                    // reps <= 138: three unconditional stores; what lands beyond have + reps is either
                    // rewritten by the following symbols or never read (branch-free on purpose)
#pragma unroll
                    for (uint32_t j = 0; j < 3; ++j) {
                        const uint32_t at = have + lane + 64 * j;
                        s.lens[at < sizeof(s.lens) - 1 ? at : sizeof(s.lens) - 1] = (uint8_t)element;
                    }
                    last = element; have += reps;
                }
                WSYNC();
                if (have != want) FAIL(SPNG_E_CODELENGTH_SEQUENCE, 0, 0);
                const bool okd = UB(build<1>(s.hist, s.run, s.lens + literals, (int)distances, s.dist, DBITS, s.sorted_dist,
                                             &s.tdist, true, lane));
                const bool okl = UB(build<0>(s.hist, s.run, s.lens, (int)literals, s.lit, LBITS, s.sorted_lit, &s.tlit,
                                             false, lane));
                if (!okl || !okd) FAIL(SPNG_E_HUFFMAN_TABLE, 0, 0);
            }

            PROF_END(6);
            // .compressed: readBlock(with:) (:266-381)
            //
            // The block is cut into 64-bit windows aligned to the first bit of its compressed data.
            // The scout decodes, for every lane of every window, the token that would start at that
            // bit; the walker only has to find the true chain of token boundaries through each
            // window and queue the tokens on it.
            {
                if (inside) { seek_bits(s, r, src, n, uni64(s.c.r_tok), lane); inside = false; }     // (the tables stand: on with the block's tokens)
                const uint64_t org = r.pos;
                wgen += 1;
                LDS_STORE(&s.c.w_prod, 0u); LDS_STORE(&s.c.w_cons, 0u);   // (every lane, same value: no branch)
                s.c.w_org = org;
                COMPILER_ORDER();
                LDS_STORE(&s.c.w_gen, wgen);                   // the start order
                // One exit (`stop`), every branch wave-uniform, no breaks out of nested loops or gotos:
                // anything fancier makes the compiler thread guard flags through every path.
                // One loop, one exit (`stop`), one walk per iteration; every branch wave-uniform.
                uint32_t k = 0, prod_seen = 0, ent = 0, stop = 0;   // stop: 1 = end of block, 2 = leave for `done`
                uint32_t step = 0, nxt = 0, nxt2 = 0, tok = 0;
                unsigned long long live = 0;
                bool fresh = true;                             // the next walk starts a new window
                for (;;) {
                    if (fresh) {
                        PROF_BEGIN();
                        if (__builtin_expect(prod_seen <= k, 0)) {
                            for (uint32_t spins = 0;; ++spins) {
                                prod_seen = UNI(LDS_LOAD(&s.c.w_prod));
                                if (prod_seen > k) break;
                                PROFC(q.p_full);
                                __builtin_amdgcn_s_sleep(1);
                                if (spins > SPIN_LIMIT) SPIN_ABORT();
                            }
                        }
                        COMPILER_ORDER();
                        tok = s.win[k & (WD - 1)].tok[lane];
                        step = tok & 63;
                        // successor on the chain; a lane that ends the chain (step 0) or leaves the
                        // window points at itself, so the walk needs no conditions at all
                        nxt = (step != 0 && (uint32_t)lane + step < 64) ? (uint32_t)lane + step : (uint32_t)lane;
                        nxt2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(nxt << 2), (int)nxt);
                        live = __ballot(step != 0);            // lanes the fast path can take
                        PROF_END(0);
                    }
                    const uint64_t wpos = org + ((uint64_t)k << 6);
                    {
                        PROF_BEGIN();
                        // ---- resolve the true chain of token boundaries through the window.
                        // A hop (VALU writes an SGPR, the next v_readlane uses it as its lane select)
                        // costs ~50 cycles of pure latency, so the walk goes two tokens at a time over
                        // the squared successor map, picking up the odd positions with independent
                        // reads that overlap the next hop.  Written out in assembly because the
                        // scheduler otherwise interleaves SALU work between the hops, which makes
                        // every one of them wait for the vector pipeline to drain (s_nop 2 + the odd
                        // read = the 4 wait states of the lane-select hazard).
                        uint32_t p, t7;
                        unsigned long long chain;
                        {
                            uint32_t t1, t2, t3, t4, t5, t6;
                            asm volatile(
                                "v_readlane_b32 %3, %10, %11\n\tv_readlane_b32 %2, %9, %11\n\ts_nop 2\n\t"
                                "v_readlane_b32 %5, %10, %3\n\tv_readlane_b32 %4, %9, %3\n\ts_nop 2\n\t"
                                "v_readlane_b32 %7, %10, %5\n\tv_readlane_b32 %6, %9, %5\n\ts_nop 2\n\t"
                                "v_readlane_b32 %1, %10, %7\n\tv_readlane_b32 %8, %9, %7\n\t"
                                "s_mov_b64 %0, 0\n\ts_bitset1_b64 %0, %11\n\t"
                                "s_bitset1_b64 %0, %2\n\ts_bitset1_b64 %0, %3\n\ts_bitset1_b64 %0, %4\n\t"
                                "s_bitset1_b64 %0, %5\n\ts_bitset1_b64 %0, %6\n\ts_bitset1_b64 %0, %7\n\t"
                                "s_bitset1_b64 %0, %8"
                                : "=&s"(chain), "=&s"(p), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5),
                                  "=&s"(t6), "=&s"(t7)
                                : "v"(nxt), "v"(nxt2), "s"(ent));
                        }
                        if (__builtin_expect(p != t7, 0)) {    // not at rest after eight tokens: rare
                            for (;;) {
                                const uint32_t np = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)p);
                                if (np == p) break;
                                asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(p));
                                p = np;
                            }
                        }
                        // p is the lane the walk came to rest on: either the token that crosses into
                        // the next window (on the chain) or the first one the fast path cannot take
                        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)step, (int)p);
                        chain = (chain & ~(1ull << p)) | (live & 1ull << p);
                        PROF_END(1);
                        if (chain) {
                            PROF_BEGIN();
                            PUSH(chain, tok);
                            PROF_END(2);
                        }
                        uint32_t used = st;
                        if (__builtin_expect(st == 0, 0)) {
                            // ---- the token at lane p, decoded wave-uniformly with every check of
                            //      the reference that does not need the output position (those are
                            //      the resolver's)
                            PROF_BEGIN();
                            const uint64_t b1 = wpos + p;
                            uint32_t slo, shi;
                            fetch64(s, (uint32_t)b1, slo, shi);
                            uint64_t slug = (uint64_t)UNI(shi) << 32 | UNI(slo);
                            if (b1 >= total) {
                                stop = 2;
                            } else {
                                uint32_t e = UNI(s.lit[(uint32_t)slug & ((1 << LBITS) - 1)]);
                                if ((e & 15) == 0) e = decode_long<0>((uint32_t)slug, s.tlit, s.sorted_lit, LBITS, lane);
                                const uint32_t len = e & 15, kind = (e >> 8) & 3;
                                if (kind == K_LIT) {
                                    if (b1 + len > total) stop = 2;
                                    else { PUSH(1ull, e >> 16 << 8); used = len; }
                                } else if (kind == K_EOB) {
                                    if (b1 + len > total) stop = 2;
                                    else { used = len; stop = 1; }
                                } else {
                                    slug >>= len;
                                    const uint32_t ex = (e >> 4) & 15;
                                    const uint32_t count = (e >> 16) + ((uint32_t)slug & ((1u << ex) - 1));
                                    slug >>= ex;
                                    uint32_t dd = UNI(s.dist[(uint32_t)slug & ((1 << DBITS) - 1)]);
                                    if ((dd & 15) == 0) dd = decode_long<1>((uint32_t)slug, s.tdist, s.sorted_dist, DBITS, lane);
                                    slug >>= dd & 15;
                                    const uint32_t dx = (dd >> 4) & 15;
                                    const uint32_t offset = (dd >> 16) + ((uint32_t)slug & ((1u << dx) - 1));
                                    const uint32_t bits = len + ex + (dd & 15) + dx;   // <= 48
                                    // offset > position is the resolver's check and comes first in
                                    // the reference; it cannot fire for offset 0, so the undefined-
                                    // reference case may be raised here
                                    if (((dd >> 8) & 3) == K_UNDEF) { status = SPNG_E_REFERENCE_UNDEFINED; stop = 2; }
                                    else if (b1 + bits > total) stop = 2;
                                    else if (count && !offset) { status = SPNG_E_REFERENCE_UNDEFINED; stop = 2; }
                                    else { PUSH(1ull, T_MATCH | count << 16 | offset); used = bits; }
                                }
                            }
                            PROF_END(5);
                        }
                        ent = p + used;
                    }
                    if (stop) break;
                    fresh = ent >= 64;
                    if (fresh) {
                        ent -= 64;
                        k += 1;
                        COMPILER_ORDER();                      // the record's loads were issued: release its slot
                        LDS_STORE(&s.c.w_cons, k);
                    }
                }
                // stop the scout and take the input ring (and the table scratch it overlays) back
                LDS_STORE(&s.c.w_stop, wgen);
                for (uint32_t spins = 0; UNI(LDS_LOAD(&s.c.w_idle)) != wgen; ++spins) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spins > SPIN_LIMIT) SPIN_ABORT();
                }
                seek_bits(s, r, src, n, org + ((uint64_t)k << 6) + ent, lane);
                if (stop == 2) { s.c.tok_bit = bitpos(r); goto done; }
            }
        } else {
            FAIL(SPNG_E_BLOCK_TYPE, type, 0);
        }
        if (bfinal) break;
    }

    // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429): compared by the resolver
    if (format != SPNG_FORMAT_IOS) {
        const uint64_t boundary = (bitpos(r) + 7) & ~(uint64_t)7;
        if (boundary + 32 > total) goto done;
        TAKE((uint32_t)(boundary - bitpos(r)));
        for (int k = 0; k < 4; ++k) declared = declared << 8 | TAKE(8);
        check = resumed ? 0 : 1;                               // (started in mid-stream: the sum over ALL bytes is taken afterwards, gzip.hip)
    }
    status = SPNG_DONE;
done:
#ifdef SPNG_INFLATE_PROF
    if (lane == 0 && blockIdx.x == 0)
        printf("decoder: %llu pushes (%.2f tokens each), %llu full polls, %llu cycles\n"
               "  cycles/count: spec %llu/%llu chain %llu/%llu push %llu/%llu advance %llu/%llu slow %llu/%llu header %llu/%llu\n",
               q.p_push, (double)q.tail / (double)q.p_push, q.p_full, __builtin_readcyclecounter() - p_t0,
               pt[0], pc[0], pt[1], pc[1], pt[2], pc[2], pt[3], pc[3], pt[5], pc[5], pt[6], pc[6]);
#endif
    if (lane == 0) {
        s.c.status = status; s.c.check = check; s.c.declared = declared;
        s.c.aux0 = aux0; s.c.aux1 = aux1; s.c.bits = bitpos(r);
    }
    LDS_ORDER();
    if (lane == 0) LDS_STORE(&s.c.a_done, 1u);
    LDS_STORE(&s.c.w_quit, 1u);                                // the scout is idle by now; let it go
}

__global__ __launch_bounds__(256) void inflate_kernel(const InflateJob *__restrict__ jobs,
                                                      spng_result *__restrict__ results)
{
    __shared__ __attribute__((aligned(16))) Lds s;
    // job fields are wave-uniform: pin them to scalar registers so that everything derived from
    // them (positions, loop conditions) stays on the scalar unit
    const InflateJob *job = jobs + blockIdx.x;
    // streams the parallel pipeline (pinflate2.hip) has already decoded and verified are not touched
    {
        const int32_t *skip = (const int32_t *)uni64((uint64_t)job->skip);
        if (skip && UNI(*skip)) return;
    }
    const gbyte *src = (const gbyte *)uni64((uint64_t)job->src);
    gbyte *dst = (gbyte *)uni64((uint64_t)job->dst);
    const uint64_t src_len = uni64(job->src_len), dst_cap = uni64(job->dst_cap);
    const int32_t format = (int32_t)UNI(job->format);
    const uint32_t image = UNI(job->image);
    typedef uint64_t __attribute__((address_space(1))) gstate;
    const gstate *state = (const gstate *)uni64((uint64_t)job->state);
    // {header of the block to start from, bytes in front of it, first bit inside it that is still to decode (0: its header), bytes in
    // front of that}
    const uint64_t start_bit = state ? uni64(state[0]) : 0, blk_out = state ? uni64(state[1]) : 0;
    const uint64_t tok_bit = state ? uni64(state[2]) : 0, tok_out = state ? uni64(state[3]) : 0;
    const uint64_t out_pos = tok_bit ? tok_out : blk_out;
    const bool internal = UNI(job->internal) != 0;
    // (a slot of the library's own that still reads {0, 0}: the pipeline got nowhere, this is a whole stream)
    const bool resumed = state != nullptr && !(internal && start_bit == 0 && out_pos == 0);
    if (threadIdx.x == 0) {
        s.c.tail = 0; s.c.head = 0; s.c.a_done = 0; s.c.b_fail = 0; s.c.a_wait = 0;
        s.c.w_gen = 0; s.c.w_stop = 0; s.c.w_idle = 0; s.c.w_quit = 0; s.c.w_prod = 0; s.c.w_cons = 0;
        s.c.blk_bit = start_bit; s.c.tok_bit = 0; s.c.r_tok = tok_bit; s.c.r_done = tok_bit ? tok_out - blk_out : 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t role = UNI(threadIdx.x >> 6);
    if (role == 0)      decoder(s, src, src_len, format, resumed, start_bit, tok_bit != 0, lane);
    else if (role == 1) resolver(s, dst, dst_cap, src_len, resumed, internal, blk_out, out_pos, results + image, lane);
    else if (role == 2) scout(s, src, src_len, lane);
    // (wave 3 has nothing to do.  It is there because the dispatcher places 256-thread workgroups
    //  evenly -- exactly four per CU, all 1024 streams of a batch resident at once -- and 192-thread
    //  ones not: with three waves a few dozen workgroups of every launch were left queued behind
    //  full shader engines until other streams had finished, which doubled the time of the batch.)
}

hipError_t launch_inflate(const InflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    inflate_kernel<<<count, 256, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

}  // namespace spng
