// pinflate2.hip -- the parallel inflate pipeline of spng_inflate_batch / spng_decode_batch (gfx950).
//
// Replaces, for streams the reference accepts, the same functions as inflate.hip:
//   block readers      Sources/LZ77/Inflator/LZ77.InflatorBuffers.Stream.swift:59-429
//   zlib header        Sources/LZ77/Inflator/LZ77.StreamHeader.swift:16-54
//   tree validation    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:80-174
//   output window      Sources/LZ77/Inflator/LZ77.InflatorOut.swift:124-139 (expand)
//   Adler-32           Sources/LZ77/Wrappers/LZ77.MRC32.swift:26-50
//
// A DEFLATE stream is a serial chain at three levels; the pipeline breaks it three times:
//
//   find     a stream is cut into segments of seg_bytes; from every segment's nominal start a wave looks
//            for the first bit at which a complete dynamic block header parses.  Segments decode
//            concurrently; a segment must end exactly on the next segment's start (scan), so a false
//            positive can only send the stream to the serial kernel, never produce wrong output.
//   decode   one wave per segment, block after block, a chunk of 64 subsequences at a time: every lane
//            decodes its subsequence from a GUESSED start (Huffman codes self-synchronise) and marks the
//            token starts it visits (round 0), runs its chain on until it lands on a bit a later lane
//            marked (round 1: a link), the true chain is the set of lanes reachable from lane 0 (pointer
//            doubling), and -- tables and compressed bytes still in LDS -- every lane on it decodes
//            exactly the tokens that start in its subsequence once more, now into a compact token
//            stream: 16-bit halfwords (a literal is one, a back-reference two), every lane storing its
//            own run of them straight into HBM at the place a prefix sum gives it (the lines fill up in
//            L2).  Where the root table holds two literals' codes in one index a step decodes both.
//            Round 2's count + emit (two kernels, a log slab, the tables of every block built twice,
//            4-byte tokens written in pieces) are this one kernel.  Token space comes from a page pool
//            (64 KiB pages, one atomic per page), so nothing has to be counted before it is written.
//   resolve  one 512-thread workgroup per stream turns tokens into bytes, a tile (<= 8 KiB) at a time.
//            Thread t owns bytes t, t + 512, ...: consecutive lanes, consecutive bytes, so every LDS
//            access of a wave is a contiguous row or a gather with few distinct addresses.  Literals go
//            straight to their byte's state; back-references leave a record and one bit (their first
//            byte) in a bitmap, and a byte finds the reference that covers it by a popcount over its
//            row's bitmap word.  Sources before the tile come from the 32 KiB LDS ring; pointers inside
//            the tile are halved by pointer jumping, rows that are complete cost nothing.  The tile
//            leaves for HBM in whole 16-byte units of the output position, Adler-32 folded in.
//            In batches of few streams a stream's chain is cut into parts that several workgroups
//            resolve side by side ("Several workgroups per stream" at the scan kernel).
//
// Exactness.  The pipeline reports SPNG_DONE -- or, when every block was taken and only the Adler-32
// differs, the reference's checksum error -- and only when every check of the reference passed on the
// way (header rules, complete trees, references inside the output, capacity, segment chain).  Anything
// else is left to inflate.hip, started at the first block the pipeline did not take (every call carries
// resume state), which decodes with the reference's exact accept/reject behaviour and error payloads.
// spng_result.reserved tells which path produced a result.
//
// Streams that arrive in pieces (spng_inflate_resume_batch): the resume point is the first segment start; a
// segment that meets a block it cannot take as it stands ends PARTIAL in front of it; resolve begins with
// the window read back from the output, and the block boundary reached goes to the serial kernel.
#include "common.hpp"
#include <type_traits>
#include <cstddef>
#include "huffman.hpp"

namespace spng {

#ifdef SPNG_EMU
#define AS_GLOBAL
#else
#define AS_GLOBAL __attribute__((address_space(1)))
#endif
// a product of two values below 2^24: v_mul_u32_u24 is full rate, v_mul_lo_u32 -- what the compiler picks even for operands it
// knows to be small -- a quarter of it
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b)
{
#ifdef SPNG_EMU
    return (a & 0xffffffu) * (b & 0xffffffu);
#else
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
typedef uint32_t v4u __attribute__((vector_size(16)));
struct __attribute__((packed)) PV4 { v4u v; };
typedef uint8_t AS_GLOBAL g8;
typedef uint16_t AS_GLOBAL g16;
typedef uint32_t AS_GLOBAL g32;
typedef PV4 AS_GLOBAL gPV4;
typedef uint32_t v2u __attribute__((vector_size(8)));
struct __attribute__((packed)) PV2 { v2u v; };
typedef PV2 AS_GLOBAL gPV2;

// a condition that is rarely true in a token step (a code behind the root tables, the end of a block): its block goes out of line,
// the common path falls through (a taken branch costs a wave more than the instructions it skips).  -DSPNG_NO_HINTS: A/B builds
#ifdef SPNG_NO_HINTS
#define RARE(c) (c)
#else
#define RARE(c) __builtin_expect((c), 0)
#endif
static constexpr int LB = 9, DB = 8, MB = 7;         // LUT index bits: lit/len, distance, code-length code
#ifndef SPNG_SDW_MAX
#define SPNG_SDW_MAX 9      // (17: 14.3 KB of LDS per wave, 11 waves per CU instead of 16: decode 291 ms instead of 206 -- the loops are latency-bound)
#endif
static constexpr int SDW_MAX = SPNG_SDW_MAX;           // dwords per lane subsequence at most (odd: conflict-free LDS stride)
static constexpr int STAGE2_DW = SDW_MAX * 64 + 16;    // staged compressed data: a chunk + alignment + a token's reach
static constexpr uint32_t PAGE_SHIFT = 16;             // token pages: 64 KiB
static constexpr uint32_t PAGE_UNITS = 1u << (PAGE_SHIFT - 4);
static constexpr uint64_t NONE2 = ~0ull;
static constexpr uint32_t TK_NULL = 0xffffu;           // padding halfword (a "second half" without a first: length 0)

// token halfwords:  literal 0x0000 | byte;  reference 0x8000 | (distance - 1 & 63) << 8 | (run - 3), then
// 0xC000 | (distance - 1) >> 6
__device__ __forceinline__ uint32_t tk_m0(uint32_t run, uint32_t dist) { return 0x8000u | ((dist - 1) & 63) << 8 | (run - 3); }
__device__ __forceinline__ uint32_t tk_m1(uint32_t dist) { return 0xC000u | (dist - 1) >> 6; }

struct DPool { uint8_t *base; uint32_t *next; uint32_t pages, pad; };   // next[0]: pages taken, next[1]: blocks decoded

// phase cycle counters of a tuning build (-DSPNG_D_PROF; SPNG_LIB=... tools/probe_v2.py): one wave / workgroup prints
#ifdef SPNG_D_PROF
#define DP_ARG , uint64_t *dp
#define DP_PASS , dp
#define DP(k) do { const uint64_t now_ = __builtin_readcyclecounter(); dp[k] += now_ - dp[31]; dp[31] = now_; } while (0)
#define DPN(k, v) (dp[k] += (v))
#define HP_ARG , uint64_t *dp = nullptr
#define HP(k) do { if (dp) DP(k); } while (0)
#else
#define DP_ARG
#define DP_PASS
#define DP(k)
#define DPN(k, v)
#define HP_ARG
#define HP(k)
#endif

// Emulator builds can count how often the rare paths of the decode rounds ran (tests/test_emu_pinflate.py asserts that its cases
// reach every one of them): COV(k) is nothing in the product.
#if defined(SPNG_EMU) && defined(SPNG_EMU_COV)
static long g_cov[8];      // 0 reference on a subsequence's last bit, 1 reference on a word's last bit, 2 chunks with marks for references,
                           // 3 chunks with kind masks, 4 pairs cut at a subsequence's end, 5 round-1 landings on a reference's second mark
#define COV(k) __atomic_fetch_add(&g_cov[k], 1, __ATOMIC_RELAXED)
#else
#define COV(k) ((void)0)
#endif

// ---- per-wave LDS of find / decode ----------------------------------------------------------------------
static constexpr uint32_t EXT = 480;                   // second-level table space, both codes together (zlib's bound for 286 symbols
                                                       // behind a 2^9 root is 340; distances behind a 2^8 root rarely need more than 130)

// Decode-table entries.  The loops that only need token LENGTHS (rounds 0 and 1) read two fields and add:
//   lit/len   [4:0] bits of the code and its extra bits together, [7:5] class, [11:8] code length, [15:12] extra-bit
//             count, [31:16] literal / base run
//   distance  [4:0] bits of the code and its extra bits together, [5] not a usable distance (undefined half of a stub
//             tree, symbols 30 / 31), [11:8] code length, [15:12] extra-bit count, [31:16] base distance
//   link      class C_LINK / bit 7: the code is longer than the root index: [11:8] further index bits, [31:16] where its
//             second-level table starts in ext
enum { C_LIT = 0, C_EOB = 1, C_REF = 2, C_BAD = 3, C_LIT2 = 6, C_LINK = 7 };      // (odd: the chain stops here; bit 1: two halfwords)
__device__ __forceinline__ uint32_t lit_entry2(uint32_t sym, uint32_t len)
{
    if (sym < 256) return len | C_LIT << 5 | len << 8 | sym << 16;
    if (sym == 256) return len | C_EOB << 5 | len << 8;
    // LZ77.Composites.swift:25-66 (run decades; symbols 286 / 287 are zero padding rows: (extra 0, base 0))
    uint32_t base, cx;
    if (sym < 265) { base = sym - 254; cx = 0; }
    else if (sym < 285) { cx = (sym - 261) >> 2; base = ((4 + ((sym - 265) & 3)) << cx) + 3; }
    else if (sym == 285) { base = 258; cx = 0; }
    else { base = 0; cx = 0; }
    return (len + cx) | (base ? C_REF : C_BAD) << 5 | len << 8 | cx << 12 | base << 16;
}
__device__ __forceinline__ uint32_t dist_entry2(uint32_t sym, uint32_t len)
{
    // LZ77.Composites.swift:68-110 (distance decades; 30 / 31 are zero padding rows)
    uint32_t base, ox;
    if (sym < 4) { base = sym + 1; ox = 0; }
    else if (sym < 30) { ox = (sym >> 1) - 1; base = ((2 + (sym & 1)) << ox) + 1; }
    else { base = 0; ox = 0; }
    return (len + ox) | (base ? 0u : 32u) | len << 8 | ox << 12 | base << 16;
}
static constexpr uint32_t DIST_UNDEF = 1 | 32;             // the unused half of a stub tree: one bit, not usable

struct DLds {
    uint32_t lit[1 << LB];
    uint32_t dist[1 << DB];
    uint32_t ext[EXT];                 // second-level tables of the codes longer than the root index
    uint32_t stage[STAGE2_DW];
    union {
        struct {                       // while a header is parsed
            uint8_t  lens[512];        //   code lengths (<= 318 + a run's overshoot)
            uint32_t clut[1 << MB];    //   LUT of the code-length code
            uint32_t hb[6][16];        //   per batch of 64 symbols: how many of each length (5 lit/len batches, distances)
            uint32_t cl[32];
            uint16_t first[2][16];     //   canonical first code of each length (lit/len, distance)
            uint16_t codes[6][64];     //   every lane's canonical codes (five lit/len symbols, one distance symbol)
        } h;
        struct {                       // rounds 0 and 1 of a chunk
            uint32_t ent[64];              // where it enters each subsequence | halfwords it decodes there before merging << 16
                                           // (in front of vmap: round 1 reads it as "the word in front of a lane's first" -- its
                                           // bit 31 is never set)
            uint32_t vmap[SDW_MAX * 64];   // visited-token-start bitmaps, word w of lane l at [w * 64 + l]
            uint16_t flag[64], mpos[64];   // lanes on the true chain; where it merges into their chains
                                           // (behind vmap: round 0's OR of nothing into "the word behind a lane's last" lands here)
        } c;
    };
};
// The token loops count their bit positions from the start of DLds, not of stage: position >> 3 & ~3 is then the LDS address of
// the dword a token starts in as it stands (one instruction less per token step), and positions still fit 16 bits.
static constexpr uint32_t QB = (uint32_t)offsetof(DLds, stage) * 8;
static_assert(QB % 32 == 0 && QB + (SDW_MAX * 64 + 3) * 32 + 64 < 65536, "biased positions: dword-aligned, 16 bits");
static_assert(offsetof(DLds, c.vmap) == offsetof(DLds, c.ent) + 256 && offsetof(DLds, c.flag) == offsetof(DLds, c.vmap) + SDW_MAX * 256,
              "round 1 reads ent as the row in front of vmap; round 0's empty OR behind a lane's last word lands in flag / mpos");

// ---- staging ------------------------------------------------------------------------------------------
__device__ __forceinline__ v4u ld16(const g8 *src, uint64_t n, uint64_t off)
{
    v4u v = {0, 0, 0, 0};
    if (off + 16 <= n) v = ((const gPV4 *)(src + off))->v;
    else if (off < n) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; ++b) if (off + b < n) w[b >> 2] |= (uint32_t)src[off + b] << (8 * (b & 3));
        v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3];
    }
    return v;
}
// copies `dwords` dwords (a multiple of 4) of the stream starting at byte `from` into dst; bytes past the end read as zero
__device__ __forceinline__ void stage_bytes2(uint32_t *dst, const g8 *src, uint64_t n, uint64_t from, int dwords, int lane)
{
    for (int k = 0; k * 256 < dwords; ++k)
        if (k * 256 + lane * 4 < dwords) *(v4u *)(dst + k * 256 + lane * 4) = ld16(src, n, from + (uint64_t)k * 1024 + (uint64_t)lane * 16);
    WSYNC();
}
__device__ __forceinline__ void fetch2(const uint32_t *stage, uint32_t q, uint32_t &lo, uint32_t &hi)
{
    const uint32_t w = q >> 5;
    const uint32_t d0 = stage[w], d1 = stage[w + 1], d2 = stage[w + 2];
    lo = __builtin_amdgcn_alignbit(d1, d0, q);
    hi = __builtin_amdgcn_alignbit(d2, d1, q);
}
__device__ __forceinline__ uint32_t upeek32_2(const uint32_t *stage, uint32_t q)
{
    uint32_t lo, hi;
    fetch2(stage, q, lo, hi);
    return UNI(lo);
}
__device__ __forceinline__ uint64_t upeek64_2(const uint32_t *stage, uint32_t q)
{
    uint32_t lo, hi;
    fetch2(stage, q, lo, hi);
    return (uint64_t)UNI(hi) << 32 | UNI(lo);
}

// ---- block headers ------------------------------------------------------------------------------------
struct Hdr2 {
    uint32_t type, bfinal;
    uint64_t payload;                  // first bit of the compressed data / first BYTE of stored data * 8
    uint32_t stored;                   // stored blocks: LEN
    uint32_t minlen;                   // Huffman blocks: the shortest lit/len code
    uint32_t pairs;                    //   the lit/len table holds pairs of literals
};

template <int ROOT, int KIND>
__device__ __forceinline__ void place_symbol(uint32_t *lut, uint32_t sym, uint32_t my, uint32_t code)
{
    // While it is built the root table is indexed by the code's bits MSB first: a code of length L <= ROOT owns the
    // 2^(ROOT - L) entries from code << (ROOT - L) on and writes the first of them; a root prefix of longer codes learns how
    // long the codes behind it get.  (finish_root fills the entries in between and turns the index round.)
    if (my <= (uint32_t)ROOT) lut[code << ((uint32_t)ROOT - my)] = KIND == 0 ? lit_entry2(sym, my) : dist_entry2(sym, my);
    else atomicMax(&lut[code >> (my - (uint32_t)ROOT)], my);
}
template <int ROOT, int KIND>
__device__ __forceinline__ void place_long(const uint32_t *lut, uint32_t *ext, uint32_t sym, uint32_t my, uint32_t code)
{
    if (my > (uint32_t)ROOT) {
        const uint32_t rev = __brev(code) >> (32 - my);
        const uint32_t link = lut[rev & ((1u << ROOT) - 1)];
        const uint32_t sub = (link >> 8) & 15, at = link >> 16;
        const uint32_t e = KIND == 0 ? lit_entry2(sym, my) : dist_entry2(sym, my);
        for (uint32_t j = rev >> ROOT; j < (1u << sub); j += 1u << (my - ROOT)) ext[at + j] = e;
    }
}
// The root table of a COMPLETE code from the first entries place_symbol left in it (zeros elsewhere; entry 0 is always
// written: the first code is all zeros).  Every lane takes 2^ROOT / 64 neighbouring entries: an empty entry repeats the one
// in front of it (the codes are canonical: their ranges follow one another); entries that hold a length (ROOT < v <= 15)
// become links to second-level tables laid out by a prefix sum (`links`: there are such; false: they do not fit);
// then the table is written back indexed by the next ROOT bits of the stream, which come LSB first.
template <int ROOT>
__device__ __forceinline__ bool finish_root(uint32_t *lut, uint32_t &used, uint32_t cap, bool links, int lane)
{
    constexpr int PER = (1 << ROOT) / 64;
    uint32_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = lut[lane * PER + k];
#pragma unroll
    for (int k = 1; k < PER; ++k) v[k] = v[k] ? v[k] : v[k - 1];
    const unsigned long long have = __ballot(v[PER - 1] != 0) & ((1ull << lane) - 1);
    const uint32_t cin = (uint32_t)__shfl((int)v[PER - 1], have ? 63 - __clzll((long long)have) : 0, 64);
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = v[k] ? v[k] : cin;
    if (links) {
        uint32_t need = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) need += (v[k] > (uint32_t)ROOT && v[k] <= 15) ? 1u << (v[k] - ROOT) : 0u;
        uint32_t tot;
        uint32_t at = used + wave_excl_scan(need, tot, lane);
        tot += used;
#ifdef SPNG_EMU_TRACE
        if (tot > cap && lane == 0) fprintf(stderr, "finish_root<%d>: need %u > cap %u\n", ROOT, tot, cap);
#endif
        if (tot > cap) return false;
        used = tot;
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (v[k] > (uint32_t)ROOT && v[k] <= 15) { const uint32_t sub = v[k] - ROOT; v[k] = C_LINK << 5 | sub << 8 | at << 16; at += 1u << sub; }
    }
    WSYNC();                                                  // (in place: every read before the first write)
#pragma unroll
    for (int k = 0; k < PER; ++k) lut[__brev((uint32_t)(lane * PER + k)) >> (32 - ROOT)] = v[k];
    return true;
}

// The code-length code (readBlockTables, InflatorBuffers.Stream.swift:144-190): 19 lengths of <= 7 bits.  Lane
// i < 19 holds the length of symbol i.  Builds the 2^7-entry LUT (entry = code length | extra bits << 4 | repeat base << 8 | symbol << 16).
// false: not a complete code (HuffmanTree.swift:80-108).
__device__ __forceinline__ bool build_clut(DLds &s, uint32_t mylen, int lane)
{
    // lane l < 8 gets the number of symbols of length l
    uint32_t cnt = 0, before = 0;
#pragma unroll
    for (uint32_t l = 1; l <= 7; ++l) {
        const unsigned long long m = __ballot(lane < 19 && mylen == l);
        cnt = (uint32_t)lane == l ? (uint32_t)__popcll(m) : cnt;
        before = mylen == l ? (uint32_t)__popcll(m & ((1ull << lane) - 1)) : before;   // rank among the symbols of my length
    }
    const uint32_t scaled = (lane >= 1 && lane <= 7) ? cnt << (7 - lane) : 0u;
    if (UNI(wave_sum(scaled)) != 128u) return false;
    const uint32_t first = ((row_scan(scaled) - scaled) >> (7 - (lane & 7))) & 127;      // canonical first code of length `lane`
    if (lane < 16) s.h.cl[lane] = first;
    s.h.clut[lane] = 0; s.h.clut[64 + lane] = 0;
    WSYNC();
    // (built indexed by the code MSB first -- a code's entries are neighbours -- and turned round at the end: finish_root)
    // (entry: code length | extra bits << 4 | smallest repeat count << 8 | symbol << 16 -- decode_lengths2 adds, it does not decide)
    if (lane < 19 && mylen)
        s.h.clut[(s.h.cl[mylen] + before) << (MB - mylen)] =
            mylen | (lane < 16 ? 0u : lane == 16 ? 2u : lane == 17 ? 3u : 7u) << 4 | (lane < 16 ? 1u : lane == 18 ? 11u : 3u) << 8 | (uint32_t)lane << 16;
    WSYNC();
    uint32_t none = 0;
    finish_root<MB>(s.h.clut, none, 0, false, lane);
    WSYNC();
    return true;
}

// The run-length coded code lengths (InflatorBuffers.Stream.swift:191-263), in two passes.  Pass 1, 64 bit positions
// at a time: every lane decodes the symbol that WOULD start at its position, the true chain through the 64 answers is
// walked on the scalar unit (v_readlane), and the symbols on it go, packed, to a list (in the lit/len table's space, which
// is not built yet) -- only one position in four or five starts a symbol, so everything else waits for pass 2, which takes
// the list 64 symbols at a time: a prefix sum over the repeat counts gives the write positions, and "the previous length"
// of a repeat is the nearest lower symbol that defines one.  On success lens[0 .. want) are the code lengths and `rel`
// is the first bit behind them.  false: a sequence the reference rejects (repeat without a previous length, a run
// past the declared count), or one that does not end inside the input.
__device__ __attribute__((always_inline)) bool decode_lengths2(DLds &s, uint32_t &rel, uint32_t rel_end, uint32_t want, int lane)
{
    for (int i = lane; i < 128; i += 64) ((uint32_t *)s.h.lens)[i] = 0;
    uint32_t *list = s.lit;                                  // symbol | repeat count << 8 | the bit behind it << 16
    uint32_t ntok = 0, have = 0, p = rel;
    for (int window = 0; window < 80 && have < want; ++window) {     // (a sequence is at most 318 symbols of >= 1 bit)
        if (p >= rel_end || p + 64 + 16 > 256u * 32) return false;     // (a header parse stages 1 KiB)
        const uint32_t q = p + (uint32_t)lane;
        const uint32_t w = q >> 5;
        const uint32_t bits = __builtin_amdgcn_alignbit(s.stage[w + 1], s.stage[w], q);
        const uint32_t e = s.h.clut[bits & ((1u << MB) - 1)];
        const uint32_t len = e & 15, sym = e >> 16, extra = (e >> 4) & 15;
        const uint32_t rep = ((e >> 8) & 0xff) + ((bits >> len) & ((1u << extra) - 1));
        const uint32_t nb = len + extra;                      // 1 .. 14
        // the chain through this window, and how many lengths it stands for.  A dependent scalar step per symbol is what
        // this loop would cost (~14 per window); three rounds of pointer doubling first, and a step covers eight symbols:
        // jn = bits from my position to the eighth symbol on (or to the first one past the window), jr = the lengths those
        // symbols stand for, jm = where they start.
        uint32_t jn = nb, jr = rep;
        uint32_t jlo = lane < 32 ? 1u << lane : 0u, jhi = lane < 32 ? 0u : 1u << (lane - 32);
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            const uint32_t tgt = (uint32_t)lane + jn;
            const uint32_t on = (uint32_t)__shfl((int)(jn | jr << 8), (int)(tgt & 63), 64);
            const uint32_t olo = (uint32_t)__shfl((int)jlo, (int)(tgt & 63), 64), ohi = (uint32_t)__shfl((int)jhi, (int)(tgt & 63), 64);
            if (tgt < 64) { jn += on & 0xff; jr += on >> 8; jlo |= olo; jhi |= ohi; }
        }
        unsigned long long mask = 0;
        uint32_t pp = 0;
        while (pp < 64) {
            mask |= (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)jhi, (int)pp) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)jlo, (int)pp);
            have += (uint32_t)__builtin_amdgcn_readlane((int)jr, (int)pp);
            pp += (uint32_t)__builtin_amdgcn_readlane((int)jn, (int)pp);
        }
        if ((mask >> lane) & 1) list[ntok + (uint32_t)__popcll(mask & ((1ull << lane) - 1))] = sym | rep << 8 | (q + nb) << 16;
        ntok += (uint32_t)__popcll(mask);
        p += pp;
    }
    if (have < want) return false;
    WSYNC();
    uint32_t prev = 0; bool prev_ok = false;
    have = 0;
    for (uint32_t b = 0; b < ntok; b += 64) {
        const bool is = b + (uint32_t)lane < ntok;
        const uint32_t t = is ? list[b + (uint32_t)lane] : 0u;
        const uint32_t sym = t & 0xff, r = (t >> 8) & 0xff;
        uint32_t tot;
        const uint32_t idx = have + wave_excl_scan(r, tot, lane);
        // the previous length as seen by a repeat (symbol 16): the nearest lower symbol that is not one
        const uint32_t val = sym < 16 ? sym : 0u;
        const unsigned long long defm = __ballot(is && sym != 16);
        const unsigned long long below = defm & ((1ull << lane) - 1);
        const int pl = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t pv = (uint32_t)__shfl((int)val, pl, 64);
        const uint32_t lastcur = below ? pv : prev;
        const bool ok = below ? true : prev_ok;
        const bool active = is && idx < want;
        const bool bad = active && (idx + r > want || (sym == 16 && !ok));
        if (active && !bad) {
            if (sym < 16) s.h.lens[idx] = (uint8_t)sym;
            else if (sym == 16) for (uint32_t k = 0; k < r; ++k) s.h.lens[idx + k] = (uint8_t)lastcur;
        }
        const unsigned long long endm = __ballot(active && idx + r == want);
        if (__ballot(bad)) return false;
        if (endm) {
            rel = (uint32_t)__builtin_amdgcn_readlane((int)(t >> 16), __ffsll((long long)endm) - 1);
            WSYNC();
            return rel <= rel_end;
        }
        have += tot;
        if (defm) { prev = (uint32_t)__shfl((int)val, 63 - __clzll((long long)defm), 64); prev_ok = true; }
    }
    return false;
}

// Both decode tables of a Huffman block from lens[0 .. literals + distances), in LDS phases: per-batch histograms of
// the code lengths; completeness and canonical first codes (lit/len in lanes 0-15, distances in lanes 16-31); every lane
// ranks its own symbols, scatters the root-table copies of the short codes and notes, per root index, the longest code
// behind it; the second-level tables are laid out by a prefix sum over the root table; the long codes fill them.
// Restates HuffmanTree.swift:80-174 (validate / size) and the decade tables of LZ77.Composites.swift.  false: a code
// the reference rejects (or second-level tables that outgrow their space: not this path's case).  minlen = the
// shortest lit/len code.
__device__ __attribute__((always_inline)) bool build_tables2(DLds &s, uint32_t literals, uint32_t distances, uint32_t &minlen, uint32_t &pairs, int lane)
{
    // (the six code lengths of a lane -- five lit/len symbols, one distance symbol -- travel in one register)
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const uint32_t sym = (uint32_t)lane + 64u * k; packed |= (sym < literals ? (uint32_t)s.h.lens[sym] : 0u) << (4 * k); }
    packed |= ((uint32_t)lane < distances ? (uint32_t)s.h.lens[literals + lane] : 0u) << 20;
#define LL(k) ((packed >> (4 * (k))) & 15)
#define DL ((packed >> 20) & 15)
    for (int i = lane; i < 96; i += 64) (&s.h.hb[0][0])[i] = 0;
    WSYNC();
#pragma unroll
    for (int k = 0; k < 5; ++k) __hip_atomic_fetch_add(&s.h.hb[k][LL(k)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&s.h.hb[5][DL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t dl = DL;
    const uint32_t dused = (uint32_t)__popcll(__ballot(dl != 0));
    const unsigned long long d1m = __ballot(dl == 1);
    WSYNC();
    // lanes 0-15: lit/len lengths, lanes 16-31: distance lengths
    const uint32_t L = (uint32_t)lane & 15;
    const bool isd = lane >= 16 && lane < 32;
    uint32_t c = 0;
    if (lane < 16) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) { const uint32_t v = s.h.hb[k][L]; s.h.hb[k][L] = run; run += v; }   // -> symbols of this length in earlier batches
        c = L ? run : 0u;
    } else if (isd) {
        c = L ? s.h.hb[5][L] : 0u;
    }
    const uint32_t scaled = c << (15 - L);                  // (c is 0 in lanes >= 32)
    const uint32_t kraft = row_scan(scaled);                 // inclusive, per row of 16 lanes
    const uint32_t klit = (uint32_t)__builtin_amdgcn_readlane((int)kraft, 15);
    const uint32_t kdist = (uint32_t)__builtin_amdgcn_readlane((int)kraft, 31);
    if (klit != 32768u) return false;
    // distances: 0 or 1 used symbol of length 1 gives a stub whose unused half the reference leaves uninitialised
    // (HuffmanTree.swift:52-65, validate(symbols:normalizing:) :112-135)
    const bool stub = dused == 0 || (dused == 1 && d1m != 0);
    if (!stub && kdist != 32768u) return false;
    if (lane < 32) s.h.first[isd ? 1 : 0][L] = (uint16_t)((kraft - scaled) >> (15 - L));
    {
        const unsigned long long nz = __ballot(lane < 16 && c != 0);
        minlen = (uint32_t)__ffsll((long long)nz) - 1;
    }
#pragma unroll
    for (int j = 0; j < (1 << LB) / 64; ++j) s.lit[j * 64 + lane] = 0;
#pragma unroll
    for (int j = 0; j < (1 << DB) / 64; ++j)
        s.dist[j * 64 + lane] = !stub ? 0u : (dused && !((j * 64 + lane) & 1)) ? dist_entry2((uint32_t)(__ffsll((long long)d1m) - 1), 1) : DIST_UNDEF;
    WSYNC();
    // ---- every lane: its five lit/len symbols and its distance symbol: canonical code = first code of the length + rank
    // (the codes wait in LDS for the second-level pass)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t my = LL(k);
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned long long bk = __ballot((my >> b) & 1);
            same &= (my >> b) & 1 ? bk : ~bk;
        }
        if (my) {
            const uint32_t code = s.h.first[0][my] + s.h.hb[k][my] + (uint32_t)__popcll(same & ((1ull << lane) - 1));
            s.h.codes[k][lane] = (uint16_t)code;
            place_symbol<LB, 0>(s.lit, (uint32_t)lane + 64u * k, my, code);
        }
    }
    if (!stub) {
        const uint32_t my = dl;
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const unsigned long long bk = __ballot((my >> b) & 1);
            same &= (my >> b) & 1 ? bk : ~bk;
        }
        if (my) {
            const uint32_t code = s.h.first[1][my] + (uint32_t)__popcll(same & ((1ull << lane) - 1));
            s.h.codes[5][lane] = (uint16_t)code;
            place_symbol<DB, 1>(s.dist, (uint32_t)lane, my, code);
        }
    }
    WSYNC();
    // ---- the root tables in their final form, and the second-level tables (only when some code is longer than its
    // root index)
    const bool longl = __ballot(LL(0) > (uint32_t)LB || LL(1) > (uint32_t)LB || LL(2) > (uint32_t)LB || LL(3) > (uint32_t)LB || LL(4) > (uint32_t)LB) != 0;
    const bool longd = !stub && __ballot(dl > (uint32_t)DB) != 0;
    uint32_t used = 0;
    if (!UB(finish_root<LB>(s.lit, used, EXT, longl, lane))) return false;
    if (!stub) { if (!UB(finish_root<DB>(s.dist, used, EXT, longd, lane))) return false; }
    WSYNC();
    if (longl || longd) {
#pragma unroll
        for (int k = 0; k < 5; ++k) if (LL(k) > (uint32_t)LB) place_long<LB, 0>(s.lit, s.ext, (uint32_t)lane + 64u * k, LL(k), s.h.codes[k][lane]);
        if (!stub && dl > (uint32_t)DB) place_long<DB, 1>(s.dist, s.ext, (uint32_t)lane, dl, s.h.codes[5][lane]);
        WSYNC();
    }
#undef LL
#undef DL
    // ---- pairs of literals: a root index whose bits hold a literal's code and then another one's whole code decodes
    // both in one step.  (An entry rewritten under a reader still shows the same first literal in the same fields.)
    bool made = false;
    if (2 * minlen <= (uint32_t)LB)                               // (two codes share the root index only when the shortest does twice)
#pragma unroll
    for (int k = 0; k < (1 << LB) / 64; ++k) {
        const uint32_t idx = (uint32_t)k * 64 + (uint32_t)lane;
        const uint32_t e = s.lit[idx];
        const uint32_t len0 = (e >> 8) & 15;
        if (((e >> 5) & 7) == C_LIT && len0 < (uint32_t)LB) {
            const uint32_t e2 = s.lit[idx >> len0];
            const uint32_t c2 = (e2 >> 5) & 7, len1 = (e2 >> 8) & 15;
            if ((c2 == C_LIT || c2 == C_LIT2) && len0 + len1 <= (uint32_t)LB) {
                s.lit[idx] = (len0 + len1) | C_LIT2 << 5 | len0 << 8 | (e & 0x00ff0000u) | (e2 & 0x00ff0000u) << 8;
                made = true;
            }
        }
    }
    pairs = __ballot(made) != 0 ? 1u : 0u;
    WSYNC();
    return true;
}

// Parses the block header at absolute bit `pos` with the reference's rules (readBlockMetadata /
// readBlockTables, InflatorBuffers.Stream.swift:59-263) and, for Huffman blocks, builds the decode
// tables.  false = anything the reference would not accept as is (errors, truncation): the caller
// gives the block up.  Wave-uniform.
__device__ __attribute__((always_inline)) bool parse_header2(DLds &s, const g8 *src, uint64_t n, uint64_t pos, Hdr2 &h, int lane HP_ARG)
{
    const uint64_t total = n * 8;
    if (pos + 3 > total) return false;
    const uint64_t wbyte = (pos >> 5) << 2;
    stage_bytes2(s.stage, src, n, wbyte, 256, lane);
    HP(24);
    uint32_t rel = (uint32_t)(pos - wbyte * 8);
    const uint32_t first = upeek32_2(s.stage, rel);
    h.bfinal = first & 1; h.type = (first >> 1) & 3;
    h.stored = 0; h.minlen = 7; h.payload = 0; h.pairs = 0;
    if (h.type == 0) {
        const uint64_t boundary = (pos + 3 + 7) & ~(uint64_t)7;
        if (boundary + 32 > total) return false;
        const uint32_t v = upeek32_2(s.stage, (uint32_t)(boundary - wbyte * 8));
        const uint32_t l = v & 0xffff, m = v >> 16;
        if (l != (~m & 0xffffu)) return false;
        const uint64_t from = boundary / 8 + 4;
        if (from + l > n) return false;
        h.stored = l; h.payload = from * 8;
        return true;
    }
    if (h.type == 3) return false;
    uint32_t literals, distances;
    if (h.type == 1) {
        for (int i = lane; i < 320; i += 64) s.h.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
        WSYNC();
        literals = 288; distances = 32;
        h.payload = pos + 3;
    } else {
        if (pos + 17 > total) return false;
        literals = 257 + ((first >> 3) & 31);
        distances = 1 + ((first >> 8) & 31);
        const uint32_t codelengths = 4 + ((first >> 13) & 15);
        rel += 17;
        if (pos + 17 + 3 * (uint64_t)codelengths > total) return false;
        if (literals > 286) return false;
        const uint64_t packed = upeek64_2(s.stage, rel) & ((1ull << (3 * codelengths)) - 1);
        rel += 3 * codelengths;
        // lane i <- the length of code-length symbol i (transmitted in the order 16, 17, 18, 0, 8, 7, ...)
        // (position of symbol i in that order, in closed form: 16-18 first, 0, then 8 7 9 6 10 5 ... alternating)
        const uint32_t sym = (uint32_t)lane;
        const uint32_t at = sym >= 16 ? sym - 16 : sym == 0 ? 3u : sym <= 7 ? 19 - 2 * sym : 2 * sym - 12;
        const uint32_t mylen = (lane < 19 && at < codelengths) ? (uint32_t)((packed >> (3 * (at < 19 ? at : 0))) & 7) : 0u;
        if (!UB(build_clut(s, mylen, lane))) return false;
        HP(25);
        const uint32_t rel_end = (uint32_t)((total - wbyte * 8) > 0xffffffffull ? 0xffffffffu : (total - wbyte * 8));
        if (!UB(decode_lengths2(s, rel, rel_end, literals + distances, lane))) return false;
        h.payload = wbyte * 8 + rel;
        HP(26);
    }
    uint32_t minlen = 7, pairs = 0;
    if (!UB(build_tables2(s, literals, distances, minlen, pairs, lane))) return false;
    h.minlen = UNI(minlen); h.pairs = UNI(pairs);
    HP(27);
    return true;
}

// ---- per-lane token decoding ----------------------------------------------------------------------------
// decodes the token that starts at bit q of the staged data (counted from the start of DLds: QB): kind 0 literal, 2 back-reference, 6 two literals, 1 end of block, 3 not a
// token the fast path takes (undefined code, zero run or distance).  FULL also produces the token's halfwords (h0, and h1
// for a reference or a second literal).
static constexpr uint32_t D2_EOB = C_EOB, D2_REF = C_REF, D2_BAD = C_BAD;   // (the entry's class as it is; 0: a literal)
// -> the token's bits (a pair's: both codes'); k = its class, len0 = the code length of the (first) literal.  Round 0 cuts a
// pair whose second literal starts in the next subsequence back to its first one (marks are kept per subsequence); round 1
// and the replay take pairs whole (they count halfwords).  Bits behind the end of the input read as zeros: the callers
// compare where a chain ends with the end.
template <bool FULL, bool PAIRS>
__device__ __forceinline__ uint32_t decode_at2(const DLds &s, uint32_t q, uint32_t &k, uint32_t &len0, uint32_t &h0, uint32_t &h1)
{
    uint32_t lo, hi;
    {   // (q counts from the start of s: QB)
        const uint32_t *w = (const uint32_t *)((const uint8_t *)&s + ((q >> 3) & ~3u));
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
        lo = __builtin_amdgcn_alignbit(d1, d0, q);
        hi = __builtin_amdgcn_alignbit(d2, d1, q);
    }
    uint32_t e = s.lit[lo & ((1 << LB) - 1)];
    if (RARE((e & 0xe0) == 0xe0)) e = s.ext[(e >> 16) + ((lo >> LB) & ((1u << ((e >> 8) & 15)) - 1))];      // a code longer than the root index (rare: out of line)
    uint32_t p2 = e & 31, cls = (e >> 5) & 7;
    len0 = (e >> 8) & 15;
    uint32_t nbits = p2;
    k = cls;
    if (cls == C_REF) {
        const uint32_t b2 = __builtin_amdgcn_alignbit(hi, lo, p2);       // (p2 <= 31: one full-rate instruction, not a 64-bit shift)
        uint32_t d = s.dist[b2 & ((1 << DB) - 1)];
        if (RARE((d & 0x80) != 0)) d = s.ext[(d >> 16) + ((b2 >> DB) & ((1u << ((d >> 8) & 15)) - 1))];
        nbits = p2 + (d & 31);
        k |= (d >> 5) & 1;                                       // (not a usable distance: C_BAD)
        if (FULL) {
            const uint32_t len1 = (e >> 8) & 15, cx = (e >> 12) & 15, dl = (d >> 8) & 15, ox = (d >> 12) & 15;
            const uint32_t run = (e >> 16) + ((lo >> len1) & ((1u << cx) - 1));
            const uint32_t dd = (d >> 16) + ((b2 >> dl) & ((1u << ox) - 1));
            h0 = tk_m0(run, dd); h1 = tk_m1(dd);
        }
    } else if (FULL) {
        h0 = PAIRS ? (e >> 16) & 0xff : e >> 16; h1 = e >> 24;   // one literal, or two
    }
    return nbits;
}

// ---- segment search ---------------------------------------------------------------------------------------
// (retry: only the streams whose first pass ran out of token pages -- PStream.pass == 1 -- are looked at again)
// (RETRY: the pass for the streams that found the pool empty is launched behind every batch and mostly has nothing to do; as
// an instantiation of its own it has a name of its own in kernel traces)
// Round 6: the screen.  PMC counters of the kernel on zlib-made streams (profiles/archive/r06j_pmc_find_*.json) say that a wave retires
// ~120 vector instructions per step of 64 bit positions and that five sixths of them are the screen, not the header parses of
// its false positives (one position in 1150 passes it; a parse of a look-alike ends early) -- a version that validated 64
// candidates at a time, a lane each, saved nothing.  So the screen itself: the completeness of the code-length code -- 19 fields of
// three bits, sum of 2^(7 - l) over the used ones == 128 -- comes from a table of three fields per look-up (512 bytes of LDS, seven
// look-ups) instead of 19 shift-compare-select-add steps, the fields from two 32-bit words instead of a 64-bit one, and the two bounds
// (inside the input, inside the segment) are 32-bit compares against the window's own limits.
template <uint32_t RETRY>
__global__ __launch_bounds__(64) void pinf2_find_kernel(const PStream *__restrict__ streams, PSeg *__restrict__ segs, uint32_t seg0)
{
    constexpr uint32_t retry = RETRY;
    __shared__ __attribute__((aligned(16))) DLds s;
    __shared__ __attribute__((aligned(16))) uint32_t win[512 + 16];
    __shared__ __attribute__((aligned(16))) uint8_t k3[512];       // three code-length-code lengths -> their 2^(7 - l) summed (l = 0: unused, 0)
    const int lane = threadIdx.x;
    PSeg &sg = segs[seg0 + blockIdx.x];
    const PStream &st = streams[UNI(sg.stream)];
    if (retry && UNI(st.pass) != 1) return;
    const g8 *src = (const g8 *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len), total = n * 8;
    const uint32_t j = UNI(sg.index);
    uint64_t found = NONE2;
    const uint64_t resume_bit = uni64(st.start_bit);       // resumable streams: nothing in front of it is looked at again
    if (UNI(st.serial_only)) {
        // (no segment has a start: the chain is empty, the stream is the serial kernel's -- which resumes INSIDE the block at hand)
    } else if (j == 0) {
        // .initial (InflatorBuffers.swift:92-104, StreamHeader.swift:16-54)
        if (resume_bit) found = resume_bit;
        else if (UNI(st.format) == SPNG_FORMAT_IOS) found = 0;
        else if (n >= 2) {
            const uint32_t cmf = src[0], flg = src[1];
            if ((cmf & 15) == 8 && (cmf >> 4) < 8 && ((cmf << 8) + flg) % 31 == 0 && !(flg & 0x20)) found = 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t x = (uint32_t)lane * 8 + i;
            const uint32_t a = x & 7, b = (x >> 3) & 7, c = x >> 6;
            k3[x] = (uint8_t)(((128u >> a) & 127) + ((128u >> b) & 127) + ((128u >> c) & 127));      // (<= 192)
        }
        const uint64_t sb = uni64(st.seg_bytes) * 8;
        const uint64_t lo_nom = (uint64_t)j * sb;
        const uint64_t lo_bit = lo_nom > resume_bit ? lo_nom : ((resume_bit + 1 + 63) & ~(uint64_t)63);   // (window loads want whole bytes)
        const uint64_t hi_bit = lo_nom + sb < total ? lo_nom + sb : total;
        for (uint64_t wb = lo_bit; wb < hi_bit && found == NONE2; wb += 16384) {
            stage_bytes2(win, src, n, wb >> 3, 512, lane);
            {   // the 64 bytes behind the window (a header straddling its end)
                const uint64_t off = (wb >> 3) + 2048 + (uint64_t)lane * 4;
                uint32_t v = 0;
                if (lane < 16) { for (int b = 0; b < 4; ++b) if (off + b < n) v |= (uint32_t)src[off + b] << (8 * b); win[512 + lane] = v; }
                WSYNC();
            }
            // the window's limits, counted from its first bit: where the segment ends, where the input ends
            const uint32_t hi_rel = hi_bit - wb < 16384 ? (uint32_t)(hi_bit - wb) : 16384u;
            const uint32_t tot_rel = total - wb < 0x40000000ull ? (uint32_t)(total - wb) : 0x40000000u;
            // 2048 bit positions per step: a lane takes the 32 positions that start in ITS dword.  What a position's first 13 bits
            // must look like -- BTYPE = 2; HLIT, HDIST <= 29 (30 and 31: bits 1-4 of the field all set) -- is decided for all 32
            // at once on the shifted words (a fifth of them pass); the lane then walks its survivors, a code-length code each
            for (uint32_t ss = 0; ss * 2048 < hi_rel && found == NONE2; ++ss) {
                const uint32_t w = ss * 64 + (uint32_t)lane, base = w * 32;
                const uint32_t d0 = win[w], d1 = win[w + 1], d2 = win[w + 2], d3 = win[w + 3];
                auto sh = [&](uint32_t k) -> uint32_t { return __builtin_amdgcn_alignbit(d1, d0, k); };      // bits k .. k + 31 of my positions' stream
                uint32_t M = ~sh(1) & sh(2);
                M &= ~(sh(4) & sh(5) & sh(6) & sh(7));
                M &= ~(sh(9) & sh(10) & sh(11) & sh(12));
                if (base + 32 > hi_rel) M &= base >= hi_rel ? 0u : ~(~0u << (hi_rel - base));              // (inside the segment)
                uint32_t R = 0;                                  // my positions whose code-length code is complete
                while (M) {
                    const uint32_t i = (uint32_t)__ffs((int)M) - 1;
                    M &= M - 1;
                    const uint32_t v0 = __builtin_amdgcn_alignbit(d1, d0, i), v1 = __builtin_amdgcn_alignbit(d2, d1, i), v2 = __builtin_amdgcn_alignbit(d3, d2, i);
                    const uint32_t ncl = ((v0 >> 13) & 15) + 4, nb = 3 * ncl;       // its 3 ncl bits (12 .. 57): stream bits 17 .. in two words
                    if (base + i + 17 + nb > tot_rel) continue;                      // (inside the input)
                    // sum of 2^(7-len) over the used lengths == 128: the fields behind the ncl-th cleared, three fields per look-up
                    uint32_t lo = __builtin_amdgcn_alignbit(v1, v0, 17), hi = __builtin_amdgcn_alignbit(v2, v1, 17);
                    lo &= nb >= 32 ? ~0u : ~(~0u << nb);
                    hi &= nb <= 32 ? 0u : ~(~0u << (nb - 32));
                    const uint32_t kraft = (uint32_t)k3[lo & 511] + k3[(lo >> 9) & 511] + k3[(lo >> 18) & 511] +
                                           k3[__builtin_amdgcn_alignbit(hi, lo, 27) & 511] + k3[(hi >> 4) & 511] + k3[(hi >> 13) & 511] + k3[(hi >> 22) & 511];
                    R |= kraft == 128 ? 1u << i : 0u;
                }
                // the survivors in stream order: lane by lane, bit by bit
                unsigned long long lm = __ballot(R != 0);
                while (lm && found == NONE2) {
                    const int l = __ffsll((long long)lm) - 1;
                    uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)R, l);
                    while (r && found == NONE2) {
                        const uint64_t at = wb + (ss * 64 + (uint32_t)l) * 32 + ((uint32_t)__ffs((int)r) - 1);
                        Hdr2 h;
                        if (UB(parse_header2(s, src, n, at, h, lane))) found = at;
                        r &= r - 1;
                    }
                    lm &= lm - 1;
                }
            }
        }
    }
    if (lane == 0) { sg.start_bit = found; sg.end_bit = 0; sg.ntok = 0; sg.tok_base = 0; sg.status = PSEG_FAIL; sg.used = 0; sg.next = 0; }
}

// ---- decode: tokens of a segment ----------------------------------------------------------------------------
// Where a segment's tokens go: halfword x of the segment lives in page x >> 15 of its page table, at byte 2 (x & 32767).
// Every lane stores its own tokens straight there (neighbouring lanes write neighbouring runs of ~30 halfwords; the
// lines fill up in the L2 long before they leave it).
static constexpr uint32_t PAGE_HW = 1u << (PAGE_SHIFT - 1);      // halfwords per page
struct Cursor {
    uint64_t nhw;                      // halfwords stored
    uint32_t npages;                   // pages taken
    g8      *pa;                       // page of halfword nhw (when npages > nhw >> 15)
    bool     dry;                      // a page was wanted and none was left
};

// takes a page.  null: the pool or the segment's page table is exhausted
__device__ __forceinline__ g8 *take_page(const DPool &pool, g32 *pt, uint32_t pt_cap, Cursor &c, int lane)
{
    if (c.npages >= pt_cap) { c.dry = true; return nullptr; }
    uint32_t id = 0;
    if (lane == 0) id = atomicAdd(pool.next, 1u);
    id = UNI(id);
    if (id >= pool.pages) { c.dry = true; return nullptr; }
    if (lane == 0) pt[c.npages] = id;
    c.npages += 1;
    return (g8 *)(pool.base + ((uint64_t)id << PAGE_SHIFT));
}

// Room for halfwords [c.nhw, c.nhw + count), count <= PAGE_HW: afterwards c.pa is the page of halfword c.nhw and pb the
// page behind it where the range reaches into one (else pa).  false: no page.
__device__ __forceinline__ bool reserve_tokens(const DPool &pool, g32 *pt, uint32_t pt_cap, Cursor &c, uint32_t count, g8 *&pb, int lane)
{
    pb = c.pa;
    if (!count) return true;
    const uint64_t first_pg = c.nhw >> (PAGE_SHIFT - 1), last_pg = (c.nhw + count - 1) >> (PAGE_SHIFT - 1);
    if (first_pg == c.npages) { c.pa = take_page(pool, pt, pt_cap, c, lane); if (!c.pa) return false; }
    pb = c.pa;
    if (last_pg != first_pg) { pb = take_page(pool, pt, pt_cap, c, lane); if (!pb) return false; }
    return true;
}
// the address of halfword r counted from the start of page pa (r < 2 PAGE_HW)
__device__ __forceinline__ g16 *token_at(g8 *pa, g8 *pb, uint32_t r)
{
    return (g16 *)((r < PAGE_HW ? pa : pb) + ((uint64_t)(r & (PAGE_HW - 1)) << 1));
}
__device__ __forceinline__ void advance_tokens(Cursor &c, uint32_t count, g8 *pb)
{
    const uint64_t pg = c.nhw >> (PAGE_SHIFT - 1);
    c.nhw += count;
    if ((c.nhw >> (PAGE_SHIFT - 1)) != pg) c.pa = pb;            // (on a page boundary exactly: replaced by the next reserve)
}

// One chunk (64 subsequences of sdw dwords) of a Huffman block.  `cb` = absolute first bit of the chunk, `entry`
// = absolute bit at which the first token of the chunk starts (>= cb).  Appends the chunk's tokens and returns:
// state 0 = the block goes on (next = entry of the next chunk), 1 = end of block (next = bit after the
// end-of-block code), 2 = give up.
//
//   round 0   every lane decodes its own subsequence from a guessed start (lane 0: the true start), marks the
//             token starts it visits in its bitmap and notes which of its tokens are back-references (two
//             halfwords); its chain leaves the subsequence at q.
//   round 1   every lane follows its chain on through the subsequences behind it until it lands on a bit that
//             the owner of that subsequence has marked (from there on the two chains are one), or leaves the
//             chunk, or stops (end of block / not a token): a link (lane it merged into, position).  At every
//             subsequence it enters it notes position and halfwords so far (tokens are shorter than a
//             subsequence, so none is skipped).
//   path      lane 0 starts on a true token boundary, so the true chain is lane 0's chain up to its link, then
//             that lane's chain up to its link, ...: the lanes reachable from lane 0 (pointer doubling).
//   replay    per subsequence the true chain enters it at `e` and `mine` halfwords of tokens start in it: the
//             crossing chain's, then the owner's marks behind the merge point.  Every lane decodes exactly
//             those, straight to its place in the token pages (a prefix sum over the lanes' counts).
//
// What the replay has to know of round 0 is HOW MANY halfwords the owner's tokens behind the merge point make.  RM (no
// lit/len code of one bit: every token is at least two bits long, a back-reference three): a back-reference marks its
// start AND the bit behind it, so the halfwords are the marks' popcount, and a token start is a mark without a mark in
// front of it (two neighbouring marks can only be a back-reference).  A back-reference on the last bit of a subsequence has
// no bit behind it to mark: `edge`.  !RM (a code of one bit exists: neighbouring marks may be two tokens): the kinds of a
// lane's tokens by ordinal in two 64-bit masks, as rounds 1-4 did it.
template <bool PAIRS, bool RM>
__device__ __forceinline__ uint32_t decode_chunk(DLds &s, const g8 *src, uint64_t n, uint64_t cb,
                                                 uint64_t entry, uint32_t sdw, const DPool &pool, g32 *pt, uint32_t pt_cap, Cursor &cur,
                                                 uint64_t &next, uint64_t &nbytes, int lane DP_ARG)
{
    DP(0);
    const uint32_t sb = sdw * 32, chb = sb * 64;
    const uint64_t sbyte = (cb >> 5) << 2;
    stage_bytes2(s.stage, src, n, sbyte, (int)((sdw * 64 + 3 + 3) & ~3u), lane);   // the chunk, the dword it may start in the middle of, a token's reach
    const uint64_t sbit = sbyte * 8;
    const uint64_t left = n * 8 - sbit;
    // (positions inside the chunk: bits from sbit, + QB)
    const uint32_t lim = left > 0xffffffffull - QB ? 0xffffffffu : (uint32_t)left + QB;
    const uint32_t off0 = (uint32_t)(cb - sbit) + QB, cend = off0 + chb;
    const uint32_t sub0 = off0 + (uint32_t)lane * sb, sub1 = sub0 + sb;
#pragma unroll
    for (int w = 0; w < SDW_MAX; ++w) s.c.vmap[w * 64 + lane] = 0;
    s.c.flag[lane] = 0;
    s.c.ent[lane] = 0;
    WSYNC();
    uint32_t d0, d1;
    if (lane == 0) COV(RM ? 2 : 3);
    DP(1);
    const uint32_t q0 = lane == 0 ? (uint32_t)(entry - sbit) + QB : sub0;
    uint32_t q = q0, st = 0;                                    // st: 0 running, 1 end of block, 2 not a token
    uint64_t mb0 = 0, mb1 = 0;                                  // !RM: which of my tokens (by ordinal) are back-references
    uint32_t ntk = 0, edge = 0;
    uint32_t qx = 0;
    while (q < sub1) {
        DPN(16, 1);
        uint32_t k, len0;
        uint32_t nb = decode_at2<false, PAIRS>(s, q, k, len0, d0, d1);
        // (ONE way out of the loop, its condition: a lane that stops parks q behind everything and keeps where it stopped in qx --
        // a `break` costs every step the exec-mask bookkeeping of a second exit: decode 275.3 -> 268.4 ms, r05_tuning.md section 5)
        if (RARE((k & 1) != 0)) { st = k == D2_EOB ? 1u : 2u; qx = k == D2_EOB ? q + nb : q; q = 0xffffffffu; continue; }
        const uint32_t b = q - sub0;
        if (RM) {
            // (the mark and, for a back-reference, the one behind it -- in the next word when the token starts on a word's last
            // bit: an OR of zero elsewhere, into the word behind the lane's last one at most)
            uint32_t m = 1;
            if (k == D2_REF) { const bool last = b + 1 == sb; m = last ? 1u : 3u; edge = last ? 1u : edge; if (last) COV(0); else if ((b & 31) == 31) COV(1); }
            const uint64_t mm = (uint64_t)m << (b & 31);
            atomicOr(&s.c.vmap[(b >> 5) * 64 + lane], (uint32_t)mm);
            atomicOr(&s.c.vmap[(b >> 5) * 64 + 64 + lane], (uint32_t)(mm >> 32));
        } else {
            atomicOr(&s.c.vmap[(b >> 5) * 64 + lane], 1u << (b & 31));
            const uint64_t bit = (uint64_t)(PAIRS ? (k == D2_REF ? 1u : 0u) : k >> 1) << (ntk & 63);
            mb0 |= ntk < 64 ? bit : 0ull; mb1 |= ntk < 64 ? 0ull : bit;
            ntk += 1;
        }
        if (PAIRS && k == C_LIT2) {                             // two literals in one step: two tokens, two marks
            const uint32_t b2 = b + len0;                       // (the second one may belong to the next subsequence: not mine)
            if (b2 < sb) {
                atomicOr(&s.c.vmap[(b2 >> 5) * 64 + lane], 1u << (b2 & 31));
                if (!RM) ntk += 1;
            } else { nb = len0; COV(4); }
        }
        q += nb;
    }
    if (st) q = qx;
    if (q > lim) st = 2;                                        // (a chain that runs off the input: zeros from there on)
    WSYNC();
    DP(2);
    uint32_t link = 64, cnt2 = 0, nh = 0;
    uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;                    // crossings: position | halfwords before it << 16
    if (st == 0) {
        uint32_t j = (uint32_t)lane, jb = sub1;                 // the subsequence q is in, and where it ends
        while (q < cend) {
            DPN(17, 1);
            if (q >= jb) {                                      // (a token is shorter than a subsequence: one step at most)
                if (nh < 4) { x3 = x2; x2 = x1; x1 = x0; x0 = q | cnt2 << 16; }      // (the newest in x0)
                nh += 1; j += 1; jb += sb;
            }
            // (the mark's word travels with the token's bits: its test comes behind the decode it may make useless)
            const uint32_t b = q + sb - jb;
            // RM: a token start of the owner's is a mark with no mark in front of it; the word in front travels along (in front
            // of a lane's first word: its ent, bit 31 clear)
            const uint32_t *wp = s.c.ent + (b >> 5) * 64 + j;
            const uint32_t pword = RM ? wp[0] : 0u, mword = wp[64];
            uint32_t k, len0;
            // (a pair of literals is taken whole even when its second one starts in the next subsequence: halfwords are counted,
            // not attributed by position -- the replay of the subsequence the pair starts in decodes both, the next one's entry
            // lies behind them; every literal's start is marked by its owner, so the merge is found one token later at worst)
            const uint32_t nb = decode_at2<false, PAIRS>(s, q, k, len0, d0, d1);
            // (bit i of `front`: the mark in front of bit i)
            const uint32_t front = RM ? __builtin_amdgcn_alignbit(mword, pword, 31) : 0u;
            if (RM && ((mword & front) >> (b & 31)) & 1) COV(5);
            if (((mword & ~front) >> (b & 31)) & 1) { link = j; qx = q; q = 0xffffffffu; continue; }
            if (RARE((k & 1) != 0)) { st = k == D2_EOB ? 1u : 2u; qx = k == D2_EOB ? q + nb : q; q = 0xffffffffu; continue; }
            cnt2 += 1 + (PAIRS ? (k >> 1) & 1 : k >> 1);
            q += nb;
        }
        if (q == 0xffffffffu) q = qx;
        if (q > lim) st = 2;
    }
    // q: where my chain merged / left the chunk / stopped
    DP(3);
    bool onpath = lane == 0;
    uint32_t jump = link;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (onpath && jump < 64) s.c.flag[jump] = 1;
        WSYNC();
        onpath = onpath || s.c.flag[lane] != 0;
        const uint32_t jj = (uint32_t)__shfl((int)jump, (int)(jump & 63), 64);
        jump = jump < 64 ? jj : 64;
    }
    if (onpath) {
        // the subsequences my chain crossed (beyond the fourth crossing they all count for the fourth: rare.  Fewer records
        // cost less per step of round 1 but leave the owner of the last one a replay of several subsequences: slower)
        const uint32_t nrec = nh < 4 ? nh : 4;
        const uint32_t xs[5] = {nrec == 4 ? x3 : nrec == 3 ? x2 : nrec == 2 ? x1 : x0,           // (the first crossing first)
                                nrec == 4 ? x2 : nrec == 3 ? x1 : x0, nrec == 4 ? x1 : x0, x0, 0};
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if ((uint32_t)h < nrec) {
                const uint32_t upto = (uint32_t)h + 1 < nrec ? xs[h + 1] >> 16 : cnt2;
                s.c.ent[lane + 1 + h] = (xs[h] & 0xffff) | (upto - (xs[h] >> 16)) << 16;
            }
        if (link < 64) {
            s.c.mpos[link] = (uint16_t)q;
            if (link != (uint32_t)lane + nrec) s.c.ent[link] = q;  // (merged behind the recorded crossings: no prefix)
        }
    }
    WSYNC();
    const uint32_t e = lane == 0 ? q0 : s.c.ent[lane];
    const uint32_t m = lane == 0 ? q0 : s.c.mpos[lane];         // where the true chain merges into mine
    uint32_t mine = e >> 16;                                    // halfwords
    if (onpath) {
        const uint32_t mb = m - sub0;                            // 0 .. sb - 1
        if (RM) {
            uint32_t hw = edge;                                  // my marks from the merge point on: a halfword each
#pragma unroll
            for (int w = 0; w < SDW_MAX; ++w) {
                const uint32_t word = s.c.vmap[w * 64 + lane];
                const uint32_t lo = w * 32;
                const uint32_t from = mb >= lo + 32 ? 0u : mb > lo ? ~0u << (mb - lo) : ~0u;
                hw += (uint32_t)__popc(word & from);
            }
            mine += hw;
        } else {
            uint32_t k0 = 0;                                     // my tokens in front of the merge point
#pragma unroll
            for (int w = 0; w < SDW_MAX; ++w) {
                const uint32_t word = s.c.vmap[w * 64 + lane];
                const uint32_t lo = w * 32;
                const uint32_t below = mb >= lo + 32 ? ~0u : mb > lo ? ~(~0u << (mb - lo)) : 0u;
                k0 += (uint32_t)__popc(word & below);
            }
            uint64_t r0m = mb0, r1m = mb1;                      // back-references among my tokens k0 ...
            if (k0 >= 64) { r0m = k0 >= 128 ? 0 : r1m >> (k0 - 64); r1m = 0; }
            else if (k0) { r0m = r0m >> k0 | r1m << (64 - k0); r1m >>= k0; }
            mine += (ntk - k0) + (uint32_t)__popcll(r0m) + (uint32_t)__popcll(r1m);
        }
    }   // (a lane off the path whose subsequence the true chain crossed without merging replays the crossing: ent)
    uint32_t tot;
    const uint32_t off = wave_excl_scan(mine, tot, lane);
#ifdef SPNG_EMU_TRACE
    if (getenv("EMU_TRACE")) fprintf(stderr, "chunk cb %llu lane %2d: q0 %5u ntk %3u st %u link %2u q %5u nh %u cnt2 %u onpath %d e %5u|%u m %5u mine %3u off %4u\n",
                                     (unsigned long long)cb, lane, q0, ntk, st, link, q, nh, cnt2, (int)onpath, e & 0xffff, e >> 16, m, mine, off);
#endif
    // the lane on the path whose chain left the chunk or stopped
    const unsigned long long endm = __ballot(onpath && link == 64);
    const int el = endm ? __ffsll((long long)endm) - 1 : 0;
    const uint32_t qe = (uint32_t)__shfl((int)q, el, 64), ste = endm ? (uint32_t)__shfl((int)st, el, 64) : 2u;
    next = sbit + (qe - QB);
#ifdef SPNG_EMU_TRACE
    if (ste == 2 && lane == 0) fprintf(stderr, "chunk cb %llu: path ends in a bad token (end lane %d, q %u)\n", (unsigned long long)cb, el, qe);
#endif
    if (ste == 2) return 2;
    DP(4);
    DPN(20, 1); DPN(21, tot);
    // ---- replay: every lane its own tokens, to where the prefix sum puts them
    g8 *pb;
    if (!UB(reserve_tokens(pool, pt, pt_cap, cur, tot, pb, lane))) return 2;
    {
        g8 *pa = (g8 *)uni64((uint64_t)cur.pa);
        const uint32_t r0 = (uint32_t)(cur.nhw & (PAGE_HW - 1)) + off;
        uint32_t nout = 0;
        // (two instances: a chunk whose halfwords lie in one page -- fifteen of sixteen -- addresses them by a byte offset from
        // the page, one addition per token; the other one picks the page per store)
        auto replay = [&](auto one_page) {
            constexpr bool ONE = decltype(one_page)::value;
            uint32_t done = 0, qq = e & 0xffff;
            while (done < mine) {
                DPN(18, 1);
                uint32_t h0 = 0, h1 = 0;
                uint32_t k, len0;
                const uint32_t nb = decode_at2<true, PAIRS>(s, qq, k, len0, h0, h1);
                // (a pair's second literal is mine only when my count says so: round 0 cuts a pair at its subsequence's end, the
                // crossing chain of round 1 does not)
                const bool two = PAIRS ? (k & 2) != 0 && done + 2 <= mine : k == D2_REF;
                if (ONE) {
                    g16 *t = (g16 *)(pa + ((r0 + done) << 1));
                    t[0] = (uint16_t)h0;
                    if (two) t[1] = (uint16_t)h1;
                } else {
                    *token_at(pa, pb, r0 + done) = (uint16_t)h0;
                    if (two) *token_at(pa, pb, r0 + done + 1) = (uint16_t)h1;
                }
                done += two ? 2u : 1u;
                nout += !two ? 1u : k == D2_REF ? (h0 & 0xff) + 3 : 2u;      // the bytes the token stands for
                qq += nb;
            }
        };
        if (pb == pa) replay(std::true_type{}); else replay(std::false_type{});
        nbytes += wave_sum(nout);
    }
    advance_tokens(cur, tot, pb);
    DP(5);
    DPN(22, 1);
    return ste;
}

#ifndef SPNG_D_WAVES
#define SPNG_D_WAVES 4
#endif
template <uint32_t RETRY>
__global__ __launch_bounds__(64, SPNG_D_WAVES) void pinf2_decode_kernel(const PStream *__restrict__ streams, PSeg *__restrict__ segs,
                                                          uint32_t *__restrict__ pt_slab, DPool pool, uint32_t seg0)
{
    constexpr uint32_t retry = RETRY;
    // (segs = the whole table -- a stream's seg_first counts from its beginning; this launch's segments start at seg0)
    __shared__ __attribute__((aligned(16))) DLds s;
    const int lane = threadIdx.x;
    PSeg &sg = segs[seg0 + blockIdx.x];
    const PStream &st = streams[UNI(sg.stream)];
    if (retry && UNI(st.pass) != 1) return;
    const uint64_t start = uni64(sg.start_bit);
    if (start == NONE2) return;
    const g8 *src = (const g8 *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len);
    // this segment ends where a later one begins: at the first found start it stops ON.  One it runs past
    // was no block start (a look-alike inside stored data or inside a block); whatever its wave decodes
    // from there stays off the chain (scan).
    uint64_t limit = NONE2;
    uint32_t nk = UNI(sg.index) + 1;
    const uint32_t seg_first = UNI(st.seg_first), seg_count = UNI(st.seg_count);
    auto advance = [&](uint64_t from) {
        limit = NONE2;
        for (; nk < seg_count; ++nk) {
            const uint64_t v = uni64(segs[seg_first + nk].start_bit);
            if (v != NONE2 && v >= from) { limit = v; break; }
        }
    };
    advance(start + 1);
    // the page table: my own entries and those of the segments behind me in which no start was found (nobody else
    // writes there; a stream whose stored or fixed blocks hide every later start needs them)
    g32 *pt = (g32 *)(pt_slab + uni64(sg.log_off));
    uint32_t pt_cap;
    {
        const PSeg &upto = segs[seg_first + (nk < seg_count ? nk : seg_count - 1)];
        const uint64_t end = uni64(upto.log_off) + (nk < seg_count ? 0 : uni64(upto.log_cap));
        const uint64_t room = end - uni64(sg.log_off);
        pt_cap = room > 0xffffff00ull ? 0xffffff00u : (uint32_t)room;
    }
    Cursor cur;
    cur.nhw = 0; cur.npages = 0; cur.pa = nullptr; cur.dry = false;
    uint64_t pos = start;
    int32_t status = PSEG_FAIL;
    // A resumable stream stops in front of the first block that cannot be taken as it stands -- cut off by the end
    // of the input so far, or not acceptable: the serial kernel, started there, tells which -- and keeps the
    // blocks before it.
    const bool resumable = uni64((uint64_t)st.state) != 0;
    uint64_t hw_block = 0, nbytes = 0, bytes_block = 0;
    uint32_t nblocks = 0;                                      // (what the planner sizes the next batch's segments by)
#ifdef SPNG_D_PROF
    uint64_t dp[32];
    for (int i = 0; i < 32; ++i) dp[i] = 0;
    dp[31] = __builtin_readcyclecounter();
    const uint64_t dp_t0 = dp[31];
#endif
    for (;;) {
        if (pos >= limit) {
            if (pos == limit) { status = PSEG_CONT; break; }
            nk += 1; advance(pos);
            continue;
        }
        Hdr2 h;
        hw_block = cur.nhw; bytes_block = nbytes;
        DP(7);
        const bool hok_ = UB(parse_header2(s, src, n, pos, h, lane DP_PASS));
        DP(8);
        DPN(23, 1);
#ifdef SPNG_EMU_TRACE
        if (!hok_ && lane == 0) fprintf(stderr, "header at bit %llu rejected\n", (unsigned long long)pos);
#endif
        if (!hok_) break;
        if (h.type == 0) {
            // stored bytes are literal tokens
            const uint64_t from = h.payload / 8;
            bool ok = true;
            for (uint32_t k = 0; k < h.stored && ok; k += PAGE_HW / 2) {
                const uint32_t m = h.stored - k < PAGE_HW / 2 ? h.stored - k : PAGE_HW / 2;
                g8 *pb;
                ok = UB(reserve_tokens(pool, pt, pt_cap, cur, m, pb, lane));
                if (ok) {
                    const uint32_t r0 = (uint32_t)(cur.nhw & (PAGE_HW - 1));
                    for (uint32_t i = (uint32_t)lane; i < m; i += 64) *token_at(cur.pa, pb, r0 + i) = (uint16_t)src[from + k + i];
                    advance_tokens(cur, m, pb);
                    nbytes += m;
                }
            }
            if (!ok) break;
            pos = h.payload + (uint64_t)h.stored * 8;
        } else {
            // subsequence length: at most 128 tokens may start in one (their kinds are kept in two 64-bit masks), and
            // no token may jump a whole subsequence (it is at most 48 bits long)
            // (sdw <= 4 minlen; odd; at most what the LDS is laid out for).  A block starts with 9 dwords per lane at most: one
            // that ends inside its first chunk leaves the lanes behind its end idle.
            uint32_t big = h.minlen >= 5 ? 17u : h.minlen == 4 ? 15u : h.minlen == 3 ? 11u : h.minlen == 2 ? 7u : 3u;
            if (big > (uint32_t)SDW_MAX) big = (uint32_t)SDW_MAX;
            uint32_t sdw = big < 9u ? big : 9u;
            uint64_t entry = h.payload, cb = h.payload;
            uint32_t state = 0;
            for (;;) {
                uint64_t next;
                // (two instances of the loops: without pairs in the table a step skips their bookkeeping)
                if (h.minlen >= 2)
                    state = h.pairs ? UNI((decode_chunk<true, true>(s, src, n, cb, entry, sdw, pool, pt, pt_cap, cur, next, nbytes, lane DP_PASS)))
                                    : UNI((decode_chunk<false, true>(s, src, n, cb, entry, sdw, pool, pt, pt_cap, cur, next, nbytes, lane DP_PASS)));
                else
                    state = h.pairs ? UNI((decode_chunk<true, false>(s, src, n, cb, entry, sdw, pool, pt, pt_cap, cur, next, nbytes, lane DP_PASS)))
                                    : UNI((decode_chunk<false, false>(s, src, n, cb, entry, sdw, pool, pt, pt_cap, cur, next, nbytes, lane DP_PASS)));
                entry = uni64(next);
                if (state) break;
                cb += (uint64_t)sdw * 32 * 64;
                sdw = big;
            }
            if (state != 1) break;
            pos = entry;
        }
        nblocks += 1;
        if (h.bfinal) { status = PSEG_FINAL; break; }
    }
    if (lane == 0 && nblocks) atomicAdd(pool.next + 1, nblocks);
    uint64_t nhw = cur.nhw;
    // The pool ran dry (a batch unlike the one it was sized by): the stream takes the retry pass, with the pool to itself and
    // its like; dry again there, it stops in front of this block like any other block that cannot be taken.
    if (status == PSEG_FAIL && cur.dry && !retry) status = PSEG_NOPAGE;
    else if (status == PSEG_FAIL && resumable) { status = PSEG_PARTIAL; nhw = hw_block; nbytes = bytes_block; }   // (pos is still the block's first bit)
    // (no padding: resolve reads whole 16-byte units, inside the last page, and masks what lies behind nhw)
    if (lane == 0) { sg.end_bit = pos; sg.ntok = nhw; sg.nbytes = nbytes; sg.status = status; sg.next = nk; }
#ifdef SPNG_D_PROF
    if (blockIdx.x == 1 && lane == 0)
        printf("decode: %lu blocks %lu chunks %lu halfwords %lu windows; lane-0 steps r0 %lu r1 %lu replay %lu; cycles: total %lu stage %lu setup %lu round0 %lu "
               "round1 %lu path %lu replay %lu flush %lu header %lu (stage %lu clut %lu lengths %lu tables %lu) other %lu\n",
               dp[23], dp[20], dp[21], dp[22], dp[16], dp[17], dp[18], __builtin_readcyclecounter() - dp_t0, dp[0], dp[1], dp[2], dp[3], dp[4], dp[5], dp[6], dp[8] + dp[24] + dp[25] + dp[26] + dp[27], dp[24], dp[25], dp[26], dp[27], dp[7]);
#endif
}

// ---- scan: the segment chain of every stream -----------------------------------------------------------------
// One wave per stream walks the chain: segment 0, then the segment decode says it stopped at, ... up to the
// first one that saw the final block.  A found start that no chain member stops at (a bit pattern inside
// stored data or in the middle of a block that happens to parse as a header) is simply not on the chain.
//
// Several workgroups per stream (st.parts_max > 0: a batch of few streams, each of which would otherwise be resolved by one
// workgroup at ~0.9 GB/s): the chain is cut, at segment boundaries, into parts of about equal output; a part other than
// the first does not know the 32 KiB in front of it and resolves to 16-bit SYMBOLS -- a byte, or a marker "byte o of the
// window in front of this part" -- which a second, memory-bound pass turns into bytes once the windows are known
// (pinf2_resolve_kernel<_, true>, pinf2_window_kernel, pinf2_fixup_kernel, pinf2_verdict_kernel).
template <uint32_t RETRY>
__global__ __launch_bounds__(64) void pinf2_scan_kernel(PStream *__restrict__ streams, PSeg *__restrict__ segs, PPart *__restrict__ parts)
{
    constexpr uint32_t retry = RETRY;
    const int lane = threadIdx.x;
    PStream &st = streams[blockIdx.x];
    if (retry && UNI(st.pass) != 1) {
        if (lane == 0) st.ok = 0;                               // (resolved, or given up, by the first pass: not again)
        return;
    }
    const uint32_t first = UNI(st.seg_first), count = UNI(st.seg_count);
    bool ok = false, partial = false, dry = false;
    uint64_t tok = 0, end_bit = 0, out = 0;
    uint32_t k = 0;
    for (uint32_t hops = 0; hops < count; ++hops) {
        PSeg *sg = segs + first + k;
        const uint64_t start = uni64(sg->start_bit), end = uni64(sg->end_bit);
        const int32_t status = (int32_t)UNI(sg->status);
        if (status == PSEG_NOPAGE && start != NONE2) dry = true;
        if (start == NONE2 || status == PSEG_FAIL || status == PSEG_NOPAGE) break;
        if (lane == 0) { sg->tok_base = tok; sg->out_base = out; sg->used = 1; }
        tok += uni64(sg->ntok);
        out += uni64(sg->nbytes);
        if (status == PSEG_FINAL) { ok = true; end_bit = end; break; }
        if (status == PSEG_PARTIAL) { ok = true; partial = true; end_bit = end; break; }
        const uint32_t nx = UNI(sg->next);
        if (nx <= k || nx >= count) break;
        if (uni64(segs[first + nx].start_bit) != end) break;
        k = nx;
    }
    // the parts: a new one begins with the first chain segment whose first byte is at or behind the next multiple of 1 / P of
    // the output (and behind the first 32 KiB, so that every marker names a byte that exists); a stream from its first byte only (no resumed one), and one
    // that fits its buffer.  (The table arrives zeroed.)
    const uint32_t pmax = UNI(st.parts_max);
    uint32_t np = 0;
    if (pmax > 1 && ok && uni64(st.start_bit) == 0 && uni64(st.out_pos) == 0 && out <= uni64(st.dst_cap) && out > 0) {
        PPart *pp = parts + (uint64_t)blockIdx.x * pmax;
        uint64_t w0 = (uint64_t)gridDim.x * (pmax - 1) > 256 ? 1 : 0;     // the first part's extra half share
#ifdef SPNG_EMU
        if (getenv("EMU_FIRST_PART_SHARE")) w0 = 1;            // (the emulator drives one stream: the rule would never fire)
#endif
        uint64_t begun = 0;                                      // first byte of the part being filled
        uint32_t kk = 0, tnext = 1;                              // next boundary: the first segment at or behind tnext / pmax
        np = 1;
        if (lane == 0) { pp[0].seg = 0; pp[0].out_pos = 0; pp[0].present = 1; }
        for (uint32_t hops = 0; hops < count; ++hops) {
            const PSeg *sg = segs + first + kk;
            const uint64_t ob = uni64(sg->out_base);
            // (the first part resolves to bytes, the others to symbols that a second pass turns into bytes: a marker workgroup is ~1.5 x
            // slower per byte -- 128 images: 28.3 ms against 18.9 for equal shares --, so the first part takes 1.5 shares: boundary t
            // lies at (2 t + w0) / (2 pmax + w0) of the output, w0 = 1)
            // (only where the marker parts run on 4 KiB tiles, two to a CU -- launch_pinf2_parts' rule; with a CU each they keep up: one
            // image 2.94 ms with equal shares, 3.30 with the larger first part)
            if (kk != 0 && tnext < pmax && ob >= 32768u && ob > begun && ob * (2 * pmax + w0) >= (uint64_t)(2 * tnext + w0) * out) {
                if (lane == 0) {
                    pp[np].seg = kk; pp[np].out_pos = ob; pp[np].present = 1;
                    pp[np - 1].seg_end = kk; pp[np - 1].out_len = ob - begun;
                }
                begun = ob; np += 1;
                tnext = (uint32_t)((ob * (2 * pmax + w0) / out + 2 - w0) / 2);      // the first boundary behind ob: 2 t + w0 > ob (2 pmax + w0) / out
            }
            if ((int32_t)UNI(sg->status) == PSEG_FINAL || (int32_t)UNI(sg->status) == PSEG_PARTIAL) break;
            kk = UNI(sg->next);
        }
        if (lane == 0) { pp[np - 1].seg_end = ~0u; pp[np - 1].out_len = out - begun; }
    }
    // pass: 1 = the chain broke where the token pool was empty: once more, with the pool to itself and its like (retry)
    if (lane == 0) { st.ok = ok ? (partial ? 2 : 1) : 0; st.ntok = tok; st.end_bit = end_bit; st.pass = (!retry && dry) ? 1 : 0; st.tok_base = 0;
                     st.out_total = out; st.parts = np; }
}

// ---- resolve: tokens -> bytes -------------------------------------------------------------------------------
#ifndef SPNG_R_THREADS
#define SPNG_R_THREADS 512        // (1024 threads, two workgroups per CU: 213 ms instead of 202 per 1024 x 64 MiB)
#define SPNG_R_WAVES 4
#endif
static constexpr uint32_t RT2 = SPNG_R_THREADS;      // threads per stream
static constexpr uint32_t NW2 = RT2 / 64;            // (waves)
// The geometry of a step.  Whole streams and first parts (MARK = false): 8 KiB tiles, 61 KB of LDS, two workgroups per CU.  The
// parts behind the first keep 16-bit SYMBOLS in ring and states -- a 64 KiB ring: with 8 KiB tiles 93 KB, ONE workgroup per CU,
// half the waves -- so a batch with more marker parts than CUs takes 4 KiB tiles and windows of 2048
// halfwords for them (BIG = false): 79 KB, two per CU (128 images: resolve 44.6 -> 33.5 ms).  A batch whose marker parts have a
// CU each keeps the 8 KiB tiles (a tile's fixed costs -- scans, barriers -- are per tile: one 4K image 3.0 against 3.2 ms).
template <bool BIG> struct RGeo {
    static constexpr uint32_t TILE = BIG ? 8192 : 4096;              // output bytes resolved per step
    static constexpr uint32_t BPT = TILE / RT2;                      // of them per thread: byte j of the tile belongs to thread j mod RT2
    static constexpr uint32_t HPT = TILE / 2 / RT2;                  // token halfwords per thread in a window (of half a tile's bytes)
    static constexpr uint32_t EB = BPT < 8 ? BPT : 8;                // bytes of a thread expanded together
    static constexpr uint32_t MAXM = TILE / 8;                       // back-references per tile
};
static constexpr uint32_t WINDOW2 = 32768;           // the DEFLATE window
static constexpr uint32_t R2_DONE = 0x8000;          // state: R2_DONE | byte, or the tile index of an earlier byte
static constexpr uint32_t PTC = 256;                 // page-table entries cached in LDS

// MARK (a part that does not know the window in front of it): the ring and the states hold SYMBOLS -- 0x4000 | byte, or
// 0x8000 | o = byte o of the 32 KiB in front of the part -- and a state below 0x2000 is the tile index of an earlier byte.
template <bool MARK> struct RingOf { typedef uint8_t T; };
template <> struct RingOf<true> { typedef uint16_t T; };
template <bool MARK, bool BIG>
struct RLds2T {
    typename RingOf<MARK>::T ring[WINDOW2];   // the last 32 KiB of output, at position mod 32 KiB
    uint16_t state[RGeo<BIG>::TILE];
    uint32_t rec[RGeo<BIG>::MAXM + 2][2];        // back-references of the tile: first byte | run << 16; distance ([0]: none, run 0; [last]: keeps what follows 16-byte aligned)
    uint32_t bitmap[RGeo<BIG>::TILE / 32];       // their first bytes
    uint16_t h0[RT2 + 8];              // every thread's first halfword (the second half of its neighbour's last reference)
    uint32_t pt[PTC];
    uint32_t part[3 * NW2];
    uint32_t cut[2];                   // where the tile ends when the window holds more than a tile: bytes, halfword
};

// exclusive prefix sum over the workgroup; every thread gets the grand total too.  One barrier: s.part is not touched
// again before the next barrier of the caller.
template <class L>
__device__ __forceinline__ uint32_t block_excl_scan2(L &s, uint32_t v, uint32_t &total, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t wt;
    const uint32_t off = wave_excl_scan(v, wt, lane);
    if (lane == 0) s.part[wave] = wt;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w4 = 0; w4 < (int)NW2; w4 += 4) {
        const v4u p = *(const v4u *)(s.part + w4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { before += w4 + c < wave ? p[c] : 0u; all += p[c]; }
    }
    total = all;
    return off + before;
}

// What a stream whose chain was resolved to its end has to say (one thread).  S = sum b_i, I = sum i * b_i (mod 65521)
// over the bytes this call produced, pos = bytes in the output.
__device__ __forceinline__ void stream_verdict(const PStream &st, uint32_t S, uint32_t I, uint64_t pos, spng_result *__restrict__ results,
                                               int32_t *__restrict__ done)
{
    const g8 *src = (const g8 *)st.src;
    const uint64_t n = st.src_len;
    uint64_t *state = st.state;
    // what was in front of this call (spng_inflate_resume_batch): the sum above then lacks those bytes
    const bool whole = st.start_bit == 0 && st.out_pos == 0;
    if (st.ok == 2) {
        // the chain stopped in front of a block the input does not hold completely (or that is not acceptable): the
        // serial kernel goes on from there
        state[0] = st.end_bit; state[1] = pos; state[2] = 0; state[3] = 0;     // (a block boundary: nothing taken of the block behind it)
    } else if (!whole) {
        // resumed and complete: the trailer must be there; the sum over ALL bytes is compared afterwards (gzip.hip)
        const uint64_t endb = (st.end_bit + 7) / 8, consumed = endb + (st.format == SPNG_FORMAT_ZLIB ? 4 : 0);
        spng_result &res = results[st.image];
        if (consumed <= n) {
            res.status = SPNG_DONE; res.reserved = 1;
            res.written = pos; res.consumed = consumed;
            res.aux[0] = res.aux[1] = 0;
            *done = 1;
        }   // (else: the serial kernel, from where this call started, reports "need more input")
    } else {
        const uint64_t endb = (st.end_bit + 7) / 8;
        spng_result &res = results[st.image];
        if (st.format == SPNG_FORMAT_IOS) {
            res.status = SPNG_DONE; res.reserved = 1;
            res.written = pos; res.consumed = endb;
            res.aux[0] = res.aux[1] = 0;
            *done = 1;
        } else if (endb + 4 <= n) {
            // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429): Adler-32 from S and I:
            // a = 1 + S, b = N + N * S - I  (i counted from 0)
            const uint32_t declared = (uint32_t)src[endb] << 24 | (uint32_t)src[endb + 1] << 16 |
                                      (uint32_t)src[endb + 2] << 8 | (uint32_t)src[endb + 3];
            const uint32_t N = (uint32_t)(pos % 65521);
            const uint32_t computed = (uint32_t)((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
            // every block was taken: a checksum that differs is the stream's own (invalidStreamChecksum(declared:computed:))
            res.status = declared == computed ? SPNG_DONE : SPNG_E_STREAM_CHECKSUM; res.reserved = 1;
            res.written = pos; res.consumed = endb + 4;
            res.aux[0] = declared == computed ? 0 : declared; res.aux[1] = declared == computed ? 0 : computed;
            *done = 1;
        }   // (else the trailer is cut off: the serial kernel says so)
    }
}

// parts / pmax: the part table (scan) when streams may be cut into parts (pmax slots each; 0: they are not).  MARK = false
// resolves a whole stream, or its first part (grid = streams); MARK = true the parts behind the first (grid = streams x
// (pmax - 1)) into symbols at sym[st.sym_off + position].
template <uint32_t RETRY, bool MARK, bool BIG>
__global__ __launch_bounds__(RT2, SPNG_R_WAVES) void pinf2_resolve_kernel(const PStream *__restrict__ streams, const PSeg *__restrict__ segs,
                                                               const uint32_t *__restrict__ pt_slab, DPool pool,
                                                               spng_result *__restrict__ results, int32_t *__restrict__ done,
                                                               PPart *__restrict__ parts, uint32_t pmax, uint16_t *__restrict__ sym)
{
    constexpr uint32_t retry = RETRY;
    constexpr uint32_t TILE2 = RGeo<BIG>::TILE, BPT2 = RGeo<BIG>::BPT, HPT2 = RGeo<BIG>::HPT, EB2 = RGeo<BIG>::EB, MAXM2 = RGeo<BIG>::MAXM;
    static_assert(HPT2 == 8 || HPT2 == 4, "a thread's halfwords of a window are one 16- or 8-byte load");
    constexpr uint32_t DONE = MARK ? 0x4000u : R2_DONE;       // a state that is a byte ...
    constexpr uint32_t KNOWN = MARK ? 0xc000u : R2_DONE;      // ... or, MARK, a marker: nothing left to look up
    __shared__ __attribute__((aligned(16))) RLds2T<MARK, BIG> s;
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)UNI((uint32_t)tid >> 6);
    const uint32_t sidx = MARK ? blockIdx.x / (pmax - 1) : blockIdx.x;
    const uint32_t pidx = MARK ? 1 + blockIdx.x % (pmax - 1) : 0;
    const PStream &st = streams[sidx];
    if (!UNI(st.ok) || (retry && done[sidx])) return;
    const uint32_t nparts = pmax ? UNI(st.parts) : 0;           // > 1: the stream is resolved in parts
    if (MARK && pidx >= nparts) return;
    PPart *part = nparts > 1 ? parts + (uint64_t)sidx * pmax + pidx : nullptr;
    const uint32_t seg_end = part ? UNI(part->seg_end) : ~0u;
    const uint64_t part_pos = part ? uni64(part->out_pos) : 0;  // (MARK: markers count from here)
    g8 *dst = (g8 *)uni64((uint64_t)st.dst);
    const uint64_t cap = uni64(st.dst_cap);
    uint64_t pos = uni64(st.out_pos) + part_pos;       // (resumable streams: the bytes earlier calls produced are in dst)
    uint32_t accS = 0, accI = 0;                     // Adler-32 partial sums (inflate.hip: struct Out)
    uint32_t pmod = (uint32_t)(pos % 65521);          // pos mod 65521, kept along
    bool bad = false, over = false;
    g16 *symo = MARK ? (g16 *)(sym + uni64(st.sym_off)) : nullptr;
    if (!MARK && pos) {
        // the window so far
        for (uint64_t p = (pos > WINDOW2 ? pos - WINDOW2 : 0) + (uint32_t)tid; p < pos; p += RT2) s.ring[p & (WINDOW2 - 1)] = dst[p];
    }
    if (tid < (int)(TILE2 / 32)) s.bitmap[tid] = 0;
    if (tid == 0) { s.rec[0][0] = 0; s.rec[0][1] = 1; }
    __syncthreads();
    const uint32_t seg_first = UNI(st.seg_first), seg_count = UNI(st.seg_count);
    uint32_t sk = part ? UNI(part->seg) : 0;
#ifdef SPNG_D_PROF
    uint64_t rp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, rp_t = __builtin_readcyclecounter();
#define RP2(k) do { const uint64_t now_ = __builtin_readcyclecounter(); rp[k] += now_ - rp_t; rp_t = now_; } while (0)
#define RPN2(k, v) (rp[k] += (v))
#else
#define RP2(k)
#define RPN2(k, v)
#endif
    constexpr uint32_t NULL2 = TK_NULL | TK_NULL << 16;
    for (uint32_t hops = 0; hops < seg_count && !over && sk != seg_end; ++hops) {
        const PSeg &sg = segs[seg_first + sk];
        const uint64_t nhw = uni64(sg.ntok);
        const g32 *ptg = (const g32 *)(pt_slab + uni64(sg.log_off));
        const int32_t sstatus = (int32_t)UNI(sg.status);
        uint64_t cursor = 0;                           // halfwords of this segment consumed
        uint64_t pt_lo = 0;                            // first page in s.pt
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < PTC; i += RT2) s.pt[i] = (pt_lo + i) * PAGE_UNITS * 8 < nhw + 8 ? ptg[pt_lo + i] : 0u;
        __syncthreads();
        // a thread's HPT2 halfwords of the token window that holds halfword `from` (windows start on multiples of HPT2)
        uint32_t tv[HPT2 / 2];
        auto load_window = [&](uint64_t from) {
            const uint64_t h = (from & ~(uint64_t)(HPT2 - 1)) + (uint64_t)tid * HPT2;
#pragma unroll
            for (int k = 0; k < (int)(HPT2 / 2); ++k) tv[k] = NULL2;
            if (h < nhw) {
                const uint64_t u = h >> 3;
                const uint32_t pid = s.pt[(u >> (PAGE_SHIFT - 4)) - pt_lo];
                const g8 *q = (const g8 *)pool.base + ((uint64_t)pid << PAGE_SHIFT) + ((u & (PAGE_UNITS - 1)) << 4) + ((h & 7) << 1);
                if constexpr (HPT2 == 8) { const v4u v = ((const gPV4 *)q)->v; tv[0] = v[0]; tv[1] = v[1]; tv[2] = v[2]; tv[3] = v[3]; }
                else { const v2u v = ((const gPV2 *)q)->v; tv[0] = v[0]; tv[1] = v[1]; }
            }
        };
        load_window(0);
        while (cursor < nhw) {
            RP2(0);
            // ---- the window: HPT2 halfwords per thread (a window starts on a multiple of HPT2: see the cut below)
            const uint64_t wb = cursor;
            uint32_t hh[HPT2];
#pragma unroll
            for (int j = 0; j < (int)HPT2; ++j) {
                const uint32_t idx = (uint32_t)tid * HPT2 + j;
                const uint32_t v = (tv[j >> 1] >> (16 * (j & 1))) & 0xffff;
                hh[j] = wb + idx < nhw ? v : TK_NULL;
            }
            s.h0[tid] = (uint16_t)hh[0];
            if (tid == 0) s.cut[1] = 0xffffffffu;                  // (read behind the take's barrier of the tile before)
            uint32_t bytes = 0, refs = 0;
#pragma unroll
            for (int j = 0; j < (int)HPT2; ++j) {
                const uint32_t v = hh[j];
                const bool lit = !(v & 0x8000), m0 = (v & 0xC000) == 0x8000;
                bytes += lit ? 1u : m0 ? (v & 0xff) + 3 : 0u;
                refs += m0 ? 1u : 0u;
            }
            uint32_t total;
            const uint32_t offp = block_excl_scan2(s, refs << 20 | bytes, total, tid);      // (barrier inside: h0 is visible)
            RP2(1);
            // take tokens while the tile has room, a thread's all or none: the totals in front of a thread never
            // decrease, so the threads that fit are the first F, and thread F -- the one that does not fit although
            // everything in front of it does -- says where the tile ends (nobody: the whole window fits).  The last
            // thread also stays out when its last halfword is the first half of a back-reference.
            uint32_t curb = offp & 0xfffff, curm = offp >> 20;
            const uint32_t hnext = tid + 1 < (int)RT2 ? s.h0[tid + 1] : TK_NULL;
            const bool front = curb <= TILE2 && curm <= MAXM2;
            const bool fits = curb + bytes <= TILE2 && curm + refs <= MAXM2 &&
                              !(tid == (int)RT2 - 1 && (hh[HPT2 - 1] & 0xC000) == 0x8000);
            if (front && !fits) { s.cut[0] = curb; s.cut[1] = (uint32_t)tid * HPT2; }
            if (fits) {
#pragma unroll
                for (int j = 0; j < (int)HPT2; ++j) {
                    const uint32_t v = hh[j];
                    if (!(v & 0x8000)) { s.state[curb] = (uint16_t)(DONE | v); curb += 1; }
                    else if ((v & 0xC000) == 0x8000) {
                        const uint32_t len = (v & 0xff) + 3;
                        const uint32_t h1 = j < (int)HPT2 - 1 ? hh[j < (int)HPT2 - 1 ? j + 1 : (int)HPT2 - 1] : hnext;
                        const uint32_t dd = (((v >> 8) & 63) | (h1 & 0x1ff) << 6) + 1;
                        atomicOr(&s.bitmap[curb >> 5], 1u << (curb & 31));
                        s.rec[curm + 1][0] = curb | len << 16;
                        s.rec[curm + 1][1] = dd;
                        curb += len; curm += 1;
                    }
                }
            }
            __syncthreads();
            const uint32_t cut0 = s.cut[0], cut1 = s.cut[1];
            const uint64_t wend = nhw - wb < (uint64_t)RT2 * HPT2 ? nhw - wb : (uint64_t)RT2 * HPT2;      // window end (halfwords from wb)
            const uint32_t tlen = cut1 != 0xffffffffu ? cut0 : total & 0xfffff;
            const uint32_t wlast = cut1 != 0xffffffffu ? cut1 : (uint32_t)wend;
            if (pos + tlen > cap) { over = true; break; }
            // the next window travels while this tile is resolved
            const uint64_t cursor_next = wb + wlast;
            if ((((cursor_next + 2 * RT2 * HPT2) >> 3) >> (PAGE_SHIFT - 4)) >= pt_lo + PTC) {
                __syncthreads();
                pt_lo = (cursor_next >> 3) >> (PAGE_SHIFT - 4);     // page of the next window's first unit
                for (uint32_t i = (uint32_t)tid; i < PTC; i += RT2) s.pt[i] = (pt_lo + i) * PAGE_UNITS * 8 < nhw + 8 ? ptg[pt_lo + i] : 0u;
                __syncthreads();
            }
            load_window(cursor_next);
            RP2(2);
            // ---- expand: the reference (if any) that covers each of my bytes
            const uint32_t rbase = (uint32_t)pos & (WINDOW2 - 1);
            const bool early = pos - part_pos < WINDOW2;             // (only there can a distance reach in front of the output / the part)
            uint32_t sv[BPT2];
            {
                // back-references that start in front of each row: prefix sum over the bitmap (lane l: rows 2l, 2l + 1)
                v4u bw = {0, 0, 0, 0};                                    // (a 4 KiB tile has 64 rows: lanes 0-31)
                if ((uint32_t)lane < TILE2 / 128) bw = *(const v4u *)(s.bitmap + 4 * lane);
                const uint32_t ca = (uint32_t)__popc(bw[0]) + (uint32_t)__popc(bw[1]), cbb = (uint32_t)__popc(bw[2]) + (uint32_t)__popc(bw[3]);
                uint32_t tt;
                const uint32_t rb = wave_excl_scan(ca + cbb, tt, lane);
                // (row = wave + NW2 k: its parity is the wave's)
                const bool odd = wave & 1;
                const uint32_t srclo = odd ? bw[2] : bw[0], srchi = odd ? bw[3] : bw[1], srcbase = odd ? rb + ca : rb;
                // EB2 rows at a time, so that the record reads and then the ring reads of a batch travel together
                // (two instances: only in the first 32 KiB can a distance reach in front of the output, or of the part --
                // everywhere else the test and its bookkeeping are not there)
                auto expand_rows = [&](auto early_c) {
                constexpr bool EARLY = decltype(early_c)::value;
#pragma unroll
                for (int part = 0; part < (int)(BPT2 / EB2); ++part) {
                    v2u recv[EB2];
#pragma unroll
                    for (int kk = 0; kk < (int)EB2; ++kk) {
                        const int k = part * (int)EB2 + kk;
                        const uint32_t row = (uint32_t)wave + NW2 * k;
                        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)srclo, (int)(row >> 1));
                        const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)srchi, (int)(row >> 1));
                        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)srcbase, (int)(row >> 1));
                        // references that start at or in front of my byte: the row's bits up to my lane's -- v_mbcnt counts the
                        // bits BELOW a lane's, so the row's word goes in shifted down by one and its bit 0 joins the base (scalar)
                        const unsigned long long mw1 = ((unsigned long long)mhi << 32 | mlo) >> 1;
                        const uint32_t id = __builtin_amdgcn_mbcnt_hi((uint32_t)(mw1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw1, base + (mlo & 1)));
                        recv[kk] = *(const v2u *)s.rec[id];            // (no reference in front of this byte: entry 0, run 0)
                    }
                    uint32_t siv[EB2], farv[EB2];
#pragma unroll
                    for (int kk = 0; kk < (int)EB2; ++kk) {
                        const int k = part * (int)EB2 + kk;
                        const uint32_t j = ((uint32_t)wave + NW2 * k) * 64 + (uint32_t)lane;
                        const uint32_t startb = recv[kk][0] & 0xffff, len = recv[kk][0] >> 16, d = recv[kk][1];
                        uint32_t kk2 = j - startb;
                        const bool inside = kk2 < len;             // (the tokens taken cover bytes 0 .. tlen - 1 and nothing else)
                        // A run longer than its distance repeats its first `distance` bytes: a byte beyond the first
                        // period copies the period in front of the run (same value, chain one level deep instead of
                        // run / distance levels).  kk2 mod d: kk2 < 258, so the quotient by v_rcp is exact ((kk2 + 0.5) / d is 0.5 / 257
                        // away from an integer at least) and the product a 24-bit multiply.  (Round 5 also tried the reciprocals from
                        // a table, kept in four-word records -- 5 full-rate instructions per byte here instead of 14 issue slots: the
                        // kernel took as long as before; the expand phase is not bound by its instruction count.  r05_tuning.md)
                        if (inside && kk2 >= d) kk2 -= mul24(d, (uint32_t)(((float)kk2 + 0.5f) * __builtin_amdgcn_rcpf((float)d)));
                        siv[kk] = inside ? startb - d + kk2 : 0x7fffffffu;                     // >= 0x80000000: before the tile
                        farv[kk] = s.ring[(rbase + siv[kk]) & (WINDOW2 - 1)];
                    }
#pragma unroll
                    for (int kk = 0; kk < (int)EB2; ++kk) {
                        const int k = part * (int)EB2 + kk;
                        const uint32_t j = ((uint32_t)wave + NW2 * k) * 64 + (uint32_t)lane;
                        const uint32_t si = siv[kk];
                        sv[k] = DONE;
                        if (si != 0x7fffffffu) {
                            uint32_t v;
                            if (MARK) {
                                // in front of the tile: the symbol in the ring, or -- in front of the part -- a marker
                                const uint32_t rel = (uint32_t)(pos - part_pos) + si;             // (from the part's first byte)
                                v = (int32_t)si >= 0 ? si : (!EARLY || (int32_t)rel >= 0) ? farv[kk] : 0x8000u | (rel + WINDOW2);
                            } else {
                                if (EARLY && (int32_t)si < 0 && (uint32_t)pos < 0u - si) bad = true;
                                v = (int32_t)si < 0 ? DONE | farv[kk] : si;
                            }
                            sv[k] = v;
                            s.state[j] = (uint16_t)v;
                        }
                    }
                }
                };
                if (early) expand_rows(std::true_type{}); else expand_rows(std::false_type{});
            }
            __syncthreads();
            RP2(3);
            if (tid < (int)(TILE2 / 32)) s.bitmap[tid] = 0;          // (read above; next written after the next scan's barriers)
            // ---- pointer jumping: all of a thread's reads of a round travel together.  Round 6: every WAVE jumps until its own bytes are
            // known and the workgroup meets once behind the rounds (it met behind every round and went on while ANY wave had an unknown
            // byte).  A state only ever changes from an index to what stood at that index -- an earlier index or the byte itself --
            // so whatever a lane reads there, before or after its owner's update of this round, stands for the same byte: rounds of
            // different waves may interleave freely (they did within a round before), and every chain ends at a byte that was known
            // when the tile's expansion was complete.
            for (;;) {
                asm volatile("" ::: "memory");                       // (the states are read anew in every round)
                bool more = false;
                uint32_t gv[BPT2];
#pragma unroll
                for (int k = 0; k < (int)BPT2; ++k) { gv[k] = DONE; if (!(sv[k] & KNOWN)) gv[k] = s.state[sv[k]]; }
#pragma unroll
                for (int k = 0; k < (int)BPT2; ++k) {
                    if (!(sv[k] & KNOWN)) {
                        const uint32_t j = ((uint32_t)wave + NW2 * k) * 64 + (uint32_t)lane;
                        sv[k] = gv[k]; s.state[j] = (uint16_t)gv[k]; more = more || !(gv[k] & KNOWN);
                    }
                }
                RPN2(9, 1);
                if (!__ballot(more)) break;
            }
            __syncthreads();
            RP2(4);
            // ---- the bytes: into the ring, then to the output in whole 16-byte units of the position
            {
                uint32_t bv[BPT2];
#pragma unroll
                for (int k = 0; k < (int)BPT2; ++k) bv[k] = s.state[((uint32_t)wave + NW2 * k) * 64 + (uint32_t)lane];
#pragma unroll
                for (int k = 0; k < (int)BPT2; ++k) {
                    const uint32_t j = ((uint32_t)wave + NW2 * k) * 64 + (uint32_t)lane;
                    if (j < tlen) s.ring[(rbase + j) & (WINDOW2 - 1)] = (typename RingOf<MARK>::T)bv[k];
                }
            }
            __syncthreads();
            if constexpr (MARK) {
                // symbols, eight to a 16-byte unit of the position; what lies in front of the part is its neighbour's to write
                const uint64_t u0 = pos >> 3, u1 = (pos + tlen) >> 3;
                for (uint64_t u = u0 + (uint32_t)tid; u < u1; u += RT2) {
                    const uint16_t *r = s.ring + ((u << 3) & (WINDOW2 - 1));
                    if ((u << 3) >= part_pos) ((gPV4 *)(symo + (u << 3)))->v = *(const v4u *)r;
                    else for (int c = 0; c < 8; ++c) if ((u << 3) + c >= part_pos) symo[(u << 3) + c] = r[c];
                }
            } else {
                const uint64_t u0 = pos >> 4, u1 = (pos + tlen) >> 4;
                for (uint64_t u = u0 + (uint32_t)tid; u < u1; u += RT2) {
                    const v4u v = *(const v4u *)(s.ring + ((u << 4) & (WINDOW2 - 1)));
                    ((gPV4 *)(dst + (u << 4)))->v = v;
                    uint32_t A = 0, J = 0;
                    A = __builtin_amdgcn_sad_u8(v[0], 0, A); A = __builtin_amdgcn_sad_u8(v[1], 0, A);
                    A = __builtin_amdgcn_sad_u8(v[2], 0, A); A = __builtin_amdgcn_sad_u8(v[3], 0, A);
                    J = __builtin_amdgcn_udot4(v[0], 0x03020100u, J, false);
                    J = __builtin_amdgcn_udot4(v[1], 0x07060504u, J, false);
                    J = __builtin_amdgcn_udot4(v[2], 0x0b0a0908u, J, false);
                    J = __builtin_amdgcn_udot4(v[3], 0x0f0e0d0cu, J, false);
                    // (position of the unit mod 65521, from the tile's: 32-bit arithmetic; g * A < 2^29)
                    const uint32_t g = (pmod + 65521u - ((uint32_t)pos & 15u) + 16u * (uint32_t)(u - u0)) % 65521u;
                    accS = (accS + A) % 65521u;
                    accI = (accI + g * A + J) % 65521u;
                }
            }
            pos += tlen;
            pmod = (pmod + tlen) % 65521u;
            cursor = cursor_next;
            RP2(5);
            RPN2(8, 1); RPN2(10, tlen);
        }
        if (sstatus == PSEG_FINAL || sstatus == PSEG_PARTIAL) break;
        sk = UNI(sg.next);
    }
#ifdef SPNG_D_PROF
    if (blockIdx.x == 0 && (tid == 0 || tid == 300))
        printf("resolve[t%d]: %lu tiles (%lu bytes), %lu rounds; cycles: loop %lu window+scan %lu take+prefetch %lu expand %lu jump %lu store %lu\n",
               tid, rp[8], rp[10], rp[9], rp[0], rp[1], rp[2], rp[3], rp[4], rp[5]);
#endif
    // the bytes behind the last whole unit
    __syncthreads();
    if constexpr (MARK) {
        const uint64_t u1 = pos >> 3 << 3, at = u1 + (uint32_t)tid;
        if (!over && at < pos && at >= part_pos) symo[at] = s.ring[at & (WINDOW2 - 1)];
    } else {
        const uint64_t u1 = pos >> 4 << 4;
        if (!over && u1 + (uint32_t)tid < pos) {
            const uint32_t b = s.ring[(u1 + (uint32_t)tid) & (WINDOW2 - 1)];
            dst[u1 + (uint32_t)tid] = (uint8_t)b;
            const uint32_t g = (uint32_t)((u1 + (uint32_t)tid) % 65521);
            accS = (accS + b) % 65521;
            accI = (uint32_t)((accI + (uint64_t)g * b) % 65521);
        }
    }
    // ---- verdict
    if (__syncthreads_or(bad || over)) {                        // leave it to the serial kernel
        if (part && tid == 0) part->failed = 1;
        return;
    }
    if constexpr (MARK) return;                                 // (pinf2_fixup_kernel sums the bytes, pinf2_verdict_kernel judges)
    else {
        // S = sum b_i, I = sum i * b_i (mod 65521) over the workgroup
        const uint32_t S1 = wave_sum(accS) % 65521, I1 = wave_sum(accI % 65521) % 65521;
        __syncthreads();
        if (lane == 0) { s.part[wave] = S1; s.part[NW2 + wave] = I1; }
        __syncthreads();
        uint32_t S = 0, I = 0;
        for (int w = 0; w < (int)NW2; ++w) { S += s.part[w]; I += s.part[NW2 + w]; }
        S %= 65521; I %= 65521;
        if (part) {                                             // the first of several parts: its sums, no verdict yet
            if (tid == 0) { part->sumS = S; part->sumI = I; }
            return;
        }
        if (tid == 0) stream_verdict(st, S, I, pos, results, done + sidx);
    }
}

// ---- several workgroups per stream: windows, symbols -> bytes, verdict ------------------------------------------
// The 32 KiB in front of every part behind the first, part after part (one workgroup per stream): the bytes of the first
// part, or the symbols of the part before with ITS window put in for the markers -- 32 KiB per step.  win[(stream * pmax + p)
// * 32768 + o] = byte o of the window of part p.
__global__ __launch_bounds__(512) void pinf2_window_kernel(const PStream *__restrict__ streams, const PPart *__restrict__ parts, uint32_t pmax,
                                                           const uint16_t *__restrict__ sym, uint8_t *__restrict__ win)
{
    const PStream &st = streams[blockIdx.x];
    const uint32_t np = UNI(st.parts);
    if (!UNI(st.ok) || np < 2) return;
    const PPart *pp = parts + (uint64_t)blockIdx.x * pmax;
    const g8 *dst = (const g8 *)uni64((uint64_t)st.dst);
    const g16 *symo = (const g16 *)(sym + uni64(st.sym_off));
    g8 *w = (g8 *)win + (uint64_t)blockIdx.x * pmax * WINDOW2;
    for (uint32_t p = 1; p < np; ++p) {
        const uint64_t from = uni64(pp[p].out_pos) - WINDOW2, before = uni64(pp[p - 1].out_pos);     // (out_pos >= 32768)
        for (uint32_t o = threadIdx.x; o < WINDOW2; o += 512) {
            const uint64_t a = from + o;
            uint32_t b;
            if (p == 1) b = dst[a];
            else if (a >= before) {
                const uint32_t v = symo[a];
                b = v & 0x8000 ? w[(uint64_t)(p - 1) * WINDOW2 + (v & 0x7fff)] : v & 0xff;
            } else b = w[(uint64_t)(p - 1) * WINDOW2 + (uint32_t)(a - (before - WINDOW2))];
            w[(uint64_t)p * WINDOW2 + o] = (uint8_t)b;
        }
        __threadfence();
        __syncthreads();
    }
}

// Symbols -> bytes for the parts behind the first, and their Adler-32 sums (grid = (FIX_WG, streams * (pmax - 1)), 256
// threads, eight bytes per thread and step).
static constexpr uint32_t FIX_WG = 32;
__global__ __launch_bounds__(256) void pinf2_fixup_kernel(const PStream *__restrict__ streams, PPart *__restrict__ parts, uint32_t pmax,
                                                          const uint16_t *__restrict__ sym, const uint8_t *__restrict__ win)
{
    const uint32_t sidx = blockIdx.y / (pmax - 1), pidx = 1 + blockIdx.y % (pmax - 1);
    const PStream &st = streams[sidx];
    if (!UNI(st.ok) || pidx >= UNI(st.parts)) return;
    PPart &part = parts[(uint64_t)sidx * pmax + pidx];
    if (UNI(part.failed)) return;
    const uint64_t p0 = uni64(part.out_pos), p1 = p0 + uni64(part.out_len);
    g8 *dst = (g8 *)uni64((uint64_t)st.dst);
    const g16 *symo = (const g16 *)(sym + uni64(st.sym_off));
    const g8 *w = (const g8 *)win + ((uint64_t)sidx * pmax + pidx) * WINDOW2;
    unsigned long long S = 0, I = 0;
    const uint64_t u0 = p0 >> 3, u1 = (p1 + 7) >> 3;
    for (uint64_t u = u0 + (uint64_t)blockIdx.x * 256 + threadIdx.x; u < u1; u += (uint64_t)FIX_WG * 256) {
        const uint64_t at = u << 3;
        const bool inner = at >= p0 && at + 8 <= p1;
        uint32_t b[8];
        if (inner) {
            const v4u v = ((const gPV4 *)(symo + at))->v;
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = (v[c >> 1] >> (16 * (c & 1))) & 0xffff;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) b[c] = (at + c >= p0 && at + c < p1) ? (uint32_t)symo[at + c] : 0x4000u;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c] = b[c] & 0x8000 ? (uint32_t)w[b[c] & 0x7fff] : b[c] & 0xff;
        if (inner) {
            v2u o;
            o[0] = b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24; o[1] = b[4] | b[5] << 8 | b[6] << 16 | b[7] << 24;
            ((gPV2 *)(dst + at))->v = o;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) if (at + c >= p0 && at + c < p1) dst[at + c] = (uint8_t)b[c];
        }
        const uint32_t g = (uint32_t)(at % 65521);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool mine = inner || (at + c >= p0 && at + c < p1);
            const uint32_t v = mine ? b[c] : 0u;
            S += v; I += (unsigned long long)((g + (uint32_t)c) % 65521u) * v;
        }
    }
    // (64-bit sums without reduction: a part is far below 2^40 bytes)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        S += (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)S, m, 64) | (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(S >> 32), m, 64) << 32;
        I += (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)I, m, 64) | (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(I >> 32), m, 64) << 32;
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&part.sumS, S); atomicAdd(&part.sumI, I); }
}

// The verdict of a stream that was resolved in parts (one thread per stream).
__global__ void pinf2_verdict_kernel(const PStream *__restrict__ streams, const PPart *__restrict__ parts, uint32_t pmax,
                                     spng_result *__restrict__ results, int32_t *__restrict__ done, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const PStream &st = streams[i];
    if (!st.ok || st.parts < 2 || done[i]) return;
    const PPart *pp = parts + (uint64_t)i * pmax;
    unsigned long long S = 0, I = 0;
    for (uint32_t p = 0; p < st.parts; ++p) {
        if (pp[p].failed) return;                               // (the serial kernel takes the stream)
        S = (S + pp[p].sumS % 65521) % 65521; I = (I + pp[p].sumI % 65521) % 65521;
    }
    stream_verdict(st, (uint32_t)S, (uint32_t)I, st.out_total, results, done + i);
}

// pages a pass took: its page counter into the batch's totals = {pages, some pass ran dry}
__global__ void pinf2_account_kernel(uint32_t *ctr, uint32_t *totals, uint32_t pages)
{
    const uint32_t used = ctr[0];
    atomicAdd(&totals[0], used < pages ? used : pages);
    if (used >= pages) atomicOr(&totals[1], 1u);
    atomicAdd(&totals[2], ctr[1]); ctr[1] = 0;                  // blocks decoded (ctr[1]: the pass's, counted by the decode waves;
                                                                // atomic: in overlap mode two groups' accounts run side by side)
}

// ---- host ------------------------------------------------------------------------------------------------
#ifndef SPNG_EMU
hipError_t launch_pinf2_find(PStream *d_streams, PSeg *d_segs, uint32_t seg0, uint32_t nsegs, uint32_t retry, hipStream_t stream)
{
    if (retry) pinf2_find_kernel<1><<<nsegs, 64, 0, stream>>>(d_streams, d_segs, seg0);
    else pinf2_find_kernel<0><<<nsegs, 64, 0, stream>>>(d_streams, d_segs, seg0);
    return hipGetLastError();
}
hipError_t launch_pinf2_decode(PStream *d_streams, PSeg *d_segs, uint32_t seg0, uint32_t nsegs, uint32_t *d_pt, uint8_t *d_pool, uint32_t *d_next,
                               uint32_t pages, uint32_t retry, hipStream_t stream)
{
    DPool pool{d_pool, d_next, pages, 0};
    if (retry) pinf2_decode_kernel<1><<<nsegs, 64, 0, stream>>>(d_streams, d_segs, d_pt, pool, seg0);
    else pinf2_decode_kernel<0><<<nsegs, 64, 0, stream>>>(d_streams, d_segs, d_pt, pool, seg0);
    return hipGetLastError();
}
hipError_t launch_pinf2_scan(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, PPart *d_parts, uint32_t retry, hipStream_t stream)
{
    if (retry) pinf2_scan_kernel<1><<<nstreams, 64, 0, stream>>>(d_streams, d_segs, d_parts);
    else pinf2_scan_kernel<0><<<nstreams, 64, 0, stream>>>(d_streams, d_segs, d_parts);
    return hipGetLastError();
}
hipError_t launch_pinf2_resolve(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint32_t *d_pt, uint8_t *d_pool, uint32_t pages,
                                spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax, uint32_t retry, hipStream_t stream)
{
    DPool pool{d_pool, nullptr, pages, 0};
    if (retry) pinf2_resolve_kernel<1, false, true><<<nstreams, RT2, 0, stream>>>(d_streams, d_segs, d_pt, pool, d_results, d_done, d_parts, pmax, nullptr);
    else pinf2_resolve_kernel<0, false, true><<<nstreams, RT2, 0, stream>>>(d_streams, d_segs, d_pt, pool, d_results, d_done, d_parts, pmax, nullptr);
    return hipGetLastError();
}
// the parts behind the first of every stream (pmax >= 2) ...
hipError_t launch_pinf2_parts(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint32_t *d_pt, uint8_t *d_pool, uint32_t pages,
                              spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax, uint16_t *d_sym, hipStream_t stream)
{
    DPool pool{d_pool, nullptr, pages, 0};
    // (more marker parts than CUs: the geometry two of them share a CU with -- RGeo)
    if (nstreams * (pmax - 1) > 256) pinf2_resolve_kernel<0, true, false><<<nstreams * (pmax - 1), RT2, 0, stream>>>(d_streams, d_segs, d_pt, pool, d_results, d_done, d_parts, pmax, d_sym);
    else pinf2_resolve_kernel<0, true, true><<<nstreams * (pmax - 1), RT2, 0, stream>>>(d_streams, d_segs, d_pt, pool, d_results, d_done, d_parts, pmax, d_sym);
    return hipGetLastError();
}
// ... and, when they and the first parts are done: the windows, symbols -> bytes, the verdicts
hipError_t launch_pinf2_join(PStream *d_streams, uint32_t nstreams, spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax,
                             uint16_t *d_sym, uint8_t *d_win, hipStream_t stream)
{
    pinf2_window_kernel<<<nstreams, 512, 0, stream>>>(d_streams, d_parts, pmax, d_sym, d_win);
    pinf2_fixup_kernel<<<dim3(FIX_WG, nstreams * (pmax - 1)), 256, 0, stream>>>(d_streams, d_parts, pmax, d_sym, d_win);
    pinf2_verdict_kernel<<<(nstreams + 63) / 64, 64, 0, stream>>>(d_streams, d_parts, pmax, d_results, d_done, nstreams);
    return hipGetLastError();
}
hipError_t launch_pinf2_account(uint32_t *d_ctr, uint32_t *d_totals, uint32_t pages, hipStream_t stream)
{
    pinf2_account_kernel<<<1, 1, 0, stream>>>(d_ctr, d_totals, pages);
    return hipGetLastError();
}
#endif

}  // namespace spng
