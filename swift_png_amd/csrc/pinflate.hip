// pinflate.hip -- intra-stream parallel inflate for gfx950: the fast path of spng_inflate_batch /
// spng_decode_batch.
//
// Replaces, for streams the reference accepts, the same functions as inflate.hip:
//   block readers      Sources/LZ77/Inflator/LZ77.InflatorBuffers.Stream.swift:59-429
//   zlib header        Sources/LZ77/Inflator/LZ77.StreamHeader.swift:16-54
//   tree validation    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:80-174
//   output window      Sources/LZ77/Inflator/LZ77.InflatorOut.swift:124-139 (expand)
//   Adler-32           Sources/LZ77/Wrappers/LZ77.MRC32.swift:26-50
//
// Why a second inflate.  inflate.hip decodes a stream as the serial chain it is: one workgroup per
// stream, ~40 MB/s per stream however idle the chip is.  That caps a 1024-stream batch at ~41 GB/s,
// gives 128 streams per GPU no speed-up at all (BASELINE configs[2]) and takes 13 s for one 512 MiB
// stream (configs[4]).  This file breaks the chain three times:
//
//   1. between blocks      -- a stream is cut into segments of seg_bytes; `find` looks, from every
//                             segment's nominal start, for the first bit at which a complete dynamic
//                             block header parses (BTYPE, HLIT/HDIST, a complete code-length code, a
//                             code-length sequence of exactly HLIT+HDIST entries, a complete lit/len
//                             code, a usable distance code).  Segments decode concurrently; a
//                             segment must end exactly on the next segment's start (checked by
//                             `scan`), so a false positive can only send the stream to the serial
//                             kernel, never produce wrong output.
//   2. inside a block      -- the 64 lanes of a wave each take a 288-bit subsequence of the block's
//                             compressed data and decode from a GUESSED start (Huffman codes
//                             self-synchronise: a decoder started at the wrong bit falls into step
//                             with the true token boundaries after a few tokens).  Every lane records
//                             the token starts it visited in an LDS bitmap, then runs its chain on
//                             through the subsequences behind it until it lands on a bit their owner
//                             marked (from there on the two chains are one): a link.  Lane 0 started
//                             on a true boundary, so the true chain is the set of lanes reachable from
//                             it through the links (pointer doubling).  `count` does this and logs, per
//                             subsequence, the bit at which the true chain enters it and the number of
//                             tokens that start in it; `emit` replays exactly those tokens into a
//                             32-bit token stream in HBM (literal byte, or run/distance), every lane
//                             its own subsequence's.
//   3. between Huffman decoding and LZ77 -- `resolve` (one workgroup per stream) turns tokens into
//                             bytes an 8 KiB tile at a time with byte-parallel pointer jumping:
//                             every output byte of the tile is a literal or points at an earlier
//                             byte; bytes before the tile are read back from the output (L2),
//                             pointers inside the tile are halved in log2(depth) rounds.  Adler-32
//                             is folded into the tile store.
//
// Exactness.  The pipeline only ever reports SPNG_DONE, and only when every check of the reference
// passed on the way (header rules, complete trees, references inside the output, capacity, Adler-32,
// segment chain).  Anything else -- every error the reference would throw, truncated input, undefined
// references, log/token space exhausted -- leaves the stream to inflate.hip, which re-decodes it from
// byte 0 with the reference's exact accept/reject behaviour and error payloads.  (A found segment start
// that no chain member stops at -- a look-alike inside stored data -- is merely left off the chain.)
// spng_result.reserved tells which path produced a result (1 = this file).
//
// Streams that arrive in pieces (spng_inflate_resume_batch).  The resume point of a stream is the first
// segment start; a segment that meets a block it cannot take as it stands ends PARTIAL in front of it;
// resolve begins with the window read back from the output, and the block boundary reached goes to the
// serial kernel, which decodes the tail from there and so gives the reference's answer for the prefix.
#include "common.hpp"
#include "huffman.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) PU128 { u32x4 v; };
typedef PU128 __attribute__((address_space(1))) gPU128;

#ifndef SPNG_LBITS
#define SPNG_LBITS 9
#endif
// (a 2^9-entry lit/len LUT: 2 KB less LDS per wave than 2^10 buys three more waves per CU and halves the
//  per-block table fill, which pays for the extra long-code lookups: -2 % on zlib-made streams, -14 % on
//  swift-png's own small blocks; profiles/r02_inflate_tuning.md)
static constexpr int LBITS = SPNG_LBITS, DBITS = 8, MBITS = 7;
#ifndef SPNG_SDW
#define SPNG_SDW 9
#endif
static constexpr int SDW = SPNG_SDW;                // dwords per lane subsequence (odd: conflict-free LDS stride)
static constexpr uint32_t SB = SDW * 32;            // bits per subsequence
static constexpr uint32_t CHB = 64 * SB;            // bits per chunk
static constexpr int STAGE_DW = ((CHB / 8 + 64 + 255) / 256) * 64 < 264 ? 264 : ((CHB / 8 + 64 + 255) / 256) * 64;   // staged compressed data (>= a header window)
static constexpr uint32_t REC_DW = 68;              // chunk record: 4 header dwords + one per lane
static constexpr uint32_t BREC_DW = 88;             // block record: 8 header dwords + 320 bytes of code lengths
static constexpr uint64_t NONE = ~0ull;
static constexpr uint32_t T_MATCH = 0x80000000u;    // token: T_MATCH | (distance - 1) << 16 | run;  else the literal byte

// per-wave LDS of find / count / emit
struct PLds {
    uint32_t lit[1 << LBITS];          // (the code-length-code LUT lives here while a header is parsed)
    uint32_t dist[1 << DBITS];
    uint32_t ext_lit[288];             // LUT entries in canonical code order, for codes longer than the LUT index
    uint32_t ext_dist[32];
    Tree     tlit, tdist;
    uint32_t hist[16], run[16];
    union {
        uint8_t  lens[512];            // a header's code lengths (<= 318 + run-length overshoot)
        struct {                       // count_chunk:
            uint16_t flag[64], mpos[64];   //   lanes on the true chain; where it merges into their chains
            uint32_t ent[64];              //   where it enters each subsequence | tokens it decodes there before merging << 16
        };
    };
    uint32_t stage[STAGE_DW];          // compressed data around the current position
    uint32_t vmap[SDW * 64];           // count: visited-token-start bitmaps; find: the search window
};

// canonical codes longer than the LUT index: left-aligned (15-bit) upper limits of every length, kept
// in scalar registers for the block
struct Lim { uint32_t lit[15 - LBITS], dist[15 - DBITS]; };

// ---- staging ------------------------------------------------------------------------------------
// copies `dwords` dwords (a multiple of 4) of the stream starting at byte `from` into dst; bytes past the end read as zero
__device__ __forceinline__ u32x4 stage_load16(const gbyte *src, uint64_t n, uint64_t off)
{
    u32x4 v = {0, 0, 0, 0};
    if (off + 16 <= n) v = ((const gPU128 *)(src + off))->v;
    else if (off < n) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; ++b) if (off + b < n) w[b >> 2] |= (uint32_t)src[off + b] << (8 * (b & 3));
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    }
    return v;
}
__device__ __forceinline__ void stage_bytes(uint32_t *dst, const gbyte *src, uint64_t n, uint64_t from, int dwords, int lane)
{
    for (int k = 0; k * 256 < dwords; ++k)
        if (k * 256 + lane * 4 < dwords) *(u32x4 *)(dst + k * 256 + lane * 4) = stage_load16(src, n, from + (uint64_t)k * 1024 + (uint64_t)lane * 16);
    WSYNC();
}

// The same in two halves, so that a chunk's bytes travel while the chunk before it is decoded.
struct StageRegs { u32x4 v[(STAGE_DW + 255) / 256]; };
__device__ __forceinline__ void stage_fetch(StageRegs &r, const gbyte *src, uint64_t n, uint64_t from, int lane)
{
#pragma unroll
    for (int k = 0; k < (STAGE_DW + 255) / 256; ++k)
        if (k * 256 + lane * 4 < STAGE_DW) r.v[k] = stage_load16(src, n, from + (uint64_t)k * 1024 + (uint64_t)lane * 16);
}
__device__ __forceinline__ void stage_put(uint32_t *dst, const StageRegs &r, int lane)
{
#pragma unroll
    for (int k = 0; k < (STAGE_DW + 255) / 256; ++k)
        if (k * 256 + lane * 4 < STAGE_DW) *(u32x4 *)(dst + k * 256 + lane * 4) = r.v[k];
    WSYNC();
}

__device__ __forceinline__ void fetch(const uint32_t *stage, uint32_t q, uint32_t &lo, uint32_t &hi)
{
    const uint32_t w = q >> 5;
    const uint32_t d0 = stage[w], d1 = stage[w + 1], d2 = stage[w + 2];
    lo = __builtin_amdgcn_alignbit(d1, d0, q);
    hi = __builtin_amdgcn_alignbit(d2, d1, q);
}
__device__ __forceinline__ uint32_t upeek32(const uint32_t *stage, uint32_t q)
{
    uint32_t lo, hi;
    fetch(stage, q, lo, hi);
    return UNI(lo);
}
__device__ __forceinline__ uint64_t upeek64(const uint32_t *stage, uint32_t q)
{
    uint32_t lo, hi;
    fetch(stage, q, lo, hi);
    return (uint64_t)UNI(hi) << 32 | UNI(lo);
}

// ---- block headers ------------------------------------------------------------------------------
struct Hdr {
    uint32_t type, bfinal;
    uint64_t payload;                  // first bit of the compressed data / first BYTE of stored data * 8
    uint32_t stored;                   // stored blocks: LEN
    uint32_t literals, distances;      // dynamic blocks: HLIT + 257, HDIST + 1 (their code lengths are in s.lens)
};

// The run-length coded code lengths of a dynamic block header (readBlockTables,
// InflatorBuffers.Stream.swift:144-263), decoded by the whole wave instead of symbol after symbol: the
// same self-synchronisation scheme as the block data (see count_chunk), on 32-bit subsequences of a
// 2048-bit window.  s.lit holds the code-length-code LUT.  On success s.lens[0 .. want) are the code
// lengths and `rel` is the first bit behind them.  false: a sequence the reference rejects (repeat
// without a previous length, a run past the declared count), or one that does not end inside the input.
__device__ __forceinline__ uint32_t cl_symbol(const PLds &s, uint32_t q)
{
    uint32_t lo, hi;
    fetch(s.stage, q, lo, hi);
    const uint32_t e = s.lit[lo & ((1 << MBITS) - 1)];
    const uint32_t len = e & 15, sym = e >> 16;
    const uint32_t extra = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
    const uint32_t rep = sym < 16 ? 1u : (sym == 18 ? 11u : 3u) + ((lo >> len) & ((1u << extra) - 1));
    return (len + extra) | sym << 8 | rep << 16;              // bits (1 .. 14) | symbol | lengths it stands for
}

__device__ __attribute__((always_inline)) bool decode_lengths(PLds &s, uint32_t &rel, uint32_t rel_end, uint32_t want, int lane)
{
    uint32_t *vm = s.ext_lit, *pf = s.ext_lit + 64, *mp = s.ext_lit + 128;   // (free until the lit/len table is built)
    for (int i = lane; i < 128; i += 64) ((uint32_t *)s.lens)[i] = 0;
    uint32_t have = 0, w0 = rel;
    uint32_t last_in = 0; bool last_ok = false;
    for (int window = 0; window < 3; ++window) {
        const uint32_t sub0 = w0 + 32u * lane, sub1 = sub0 + 32, wend = w0 + 2048;
        uint32_t q = sub0, V = 0;
        while (q < sub1) { V |= 1u << (q - sub0); q += cl_symbol(s, q) & 255; }
        vm[lane] = V; pf[lane] = 0;
        WSYNC();
        uint32_t link = 64;
        while (q < wend) {
            const uint32_t j = (q - w0) >> 5;
            if ((vm[j] >> ((q - w0) & 31)) & 1) { link = j; break; }
            q += cl_symbol(s, q) & 255;
        }
        bool onpath = lane == 0;
        uint32_t jump = link;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (onpath && jump < 64) pf[jump] = 1;
            WSYNC();
            onpath = onpath || pf[lane] != 0;
            const uint32_t jj = (uint32_t)__shfl((int)jump, (int)(jump & 63), 64);
            jump = jump < 64 ? jj : 64;
        }
        if (onpath && link < 64) mp[link] = q;
        WSYNC();
        const uint32_t m = lane == 0 ? w0 : mp[lane];
        // how many lengths my symbols stand for, and what "the previous length" is behind them
        uint32_t cnt = 0, lastv = 0; bool def = false;
        if (onpath) {
            for (uint32_t p = m; p < q;) {
                const uint32_t t = cl_symbol(s, p), sym = (t >> 8) & 255;
                cnt += t >> 16;
                if (sym != 16) { def = true; lastv = sym < 16 ? sym : 0u; }
                p += t & 255;
            }
        }
        uint32_t tot;
        const uint32_t base = have + wave_excl_scan(cnt, tot, lane);
        const unsigned long long dm = __ballot(onpath && def);
        const unsigned long long below = dm & ((1ull << lane) - 1);
        const int pl = below ? 63 - __clzll((long long)below) : 0;
        const uint32_t pv = (uint32_t)__shfl((int)lastv, pl, 64);
        uint32_t lastcur = below ? pv : last_in;
        bool lastcur_ok = below ? true : last_ok;
        // write them
        bool bad = false;
        uint32_t endpos = 0xffffffffu;
        if (onpath && base < want) {
            uint32_t idx = base;
            for (uint32_t p = m; p < q && idx < want;) {
                const uint32_t t = cl_symbol(s, p), sym = (t >> 8) & 255, rep = t >> 16;
                if (idx + rep > want) { bad = true; break; }
                if (sym < 16) { s.lens[idx] = (uint8_t)sym; lastcur = sym; lastcur_ok = true; }
                else if (sym == 16) {
                    if (!lastcur_ok) { bad = true; break; }
                    for (uint32_t r = 0; r < rep; ++r) s.lens[idx + r] = (uint8_t)lastcur;
                } else { lastcur = 0; lastcur_ok = true; }
                idx += rep;
                p += t & 255;
                if (idx == want) endpos = p;
            }
        }
        if (__ballot(bad)) return false;
        if (have + tot >= want) {
            const unsigned long long em = __ballot(endpos != 0xffffffffu);
            if (!em) return false;
            rel = (uint32_t)__shfl((int)endpos, __ffsll((long long)em) - 1, 64);
            WSYNC();
            return rel <= rel_end;
        }
        // the sequence goes on behind this window
        have += tot;
        if (dm) { const int ll = 63 - __clzll((long long)dm); last_in = (uint32_t)__shfl((int)lastv, ll, 64); last_ok = true; }
        const unsigned long long endm = __ballot(onpath && link == 64);
        if (!endm) return false;
        w0 = (uint32_t)__shfl((int)q, __ffsll((long long)endm) - 1, 64);
        if (w0 >= rel_end) return false;
        WSYNC();
    }
    return false;
}

// Parses the block header at absolute bit `pos` with the reference's rules (readBlockMetadata /
// readBlockTables, InflatorBuffers.Stream.swift:59-263) and, for Huffman blocks, builds the decode
// tables.  false = anything the reference would not accept as is (errors, truncation): the caller
// gives the stream up.  Wave-uniform.
__device__ __forceinline__ void load_limits(const PLds &s, Lim &lc)
{
    // lengths no symbol has: limit = that of the next shorter length (count 0), so the compare chain skips them
#pragma unroll
    for (int k = 0; k < 15 - LBITS; ++k) { const int l = LBITS + 1 + k; lc.lit[k] = UNI((uint32_t)(s.tlit.first[l] + s.tlit.count[l]) << (15 - l)); }
#pragma unroll
    for (int k = 0; k < 7; ++k) { const int l = DBITS + 1 + k; lc.dist[k] = UNI((uint32_t)(s.tdist.first[l] + s.tdist.count[l]) << (15 - l)); }
}

#ifdef SPNG_COUNT_PROF
#define HP(k) do { if (hp) { const uint64_t now_ = __builtin_readcyclecounter(); hp[k] += now_ - hp[7]; hp[7] = now_; } } while (0)
#define HP_ARG , uint64_t *hp = nullptr
#else
#define HP(k)
#define HP_ARG
#endif
__device__ __attribute__((always_inline)) bool parse_header(PLds &s, const gbyte *src, uint64_t n, uint64_t pos, Hdr &h, int lane HP_ARG)
{
    const uint64_t total = n * 8;
    if (pos + 3 > total) return false;
    const uint64_t wbyte = (pos >> 5) << 2;
    HP(6);
    stage_bytes(s.stage, src, n, wbyte, 256, lane);
    HP(0);
    uint32_t rel = (uint32_t)(pos - wbyte * 8);
    const uint32_t first = upeek32(s.stage, rel);
    h.bfinal = first & 1; h.type = (first >> 1) & 3;
    h.stored = 0; h.literals = 0; h.distances = 0;
    if (h.type == 0) {
        const uint64_t boundary = (pos + 3 + 7) & ~(uint64_t)7;
        if (boundary + 32 > total) return false;
        const uint32_t v = upeek32(s.stage, (uint32_t)(boundary - wbyte * 8));
        const uint32_t l = v & 0xffff, m = v >> 16;
        if (l != (~m & 0xffffu)) return false;
        const uint64_t from = boundary / 8 + 4;
        if (from + l > n) return false;
        h.stored = l; h.payload = from * 8;
        return true;
    }
    if (h.type == 3) return false;
    if (h.type == 1) {
        for (int i = lane; i < 288; i += 64) s.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        WSYNC();
        build<0>(s.hist, s.run, s.lens, 288, s.lit, LBITS, (uint16_t *)nullptr, &s.tlit, false, lane, s.ext_lit);
        for (int i = lane; i < 32; i += 64) s.lens[i] = 5;
        WSYNC();
        build<1>(s.hist, s.run, s.lens, 32, s.dist, DBITS, (uint16_t *)nullptr, &s.tdist, false, lane, s.ext_dist);
        h.payload = pos + 3;
        return true;
    }
    if (pos + 17 > total) return false;
    const uint32_t literals = 257 + ((first >> 3) & 31);
    const uint32_t distances = 1 + ((first >> 8) & 31);
    const uint32_t codelengths = 4 + ((first >> 13) & 15);
    rel += 17;
    if (pos + 17 + 3 * (uint64_t)codelengths > total) return false;
    if (literals > 286) return false;
    const uint64_t packed = upeek64(s.stage, rel) & ((1ull << (3 * codelengths)) - 1);
    rel += 3 * codelengths;
    if (lane < 19) {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        s.lens[order[lane]] = (uint32_t)lane < codelengths ? (uint8_t)((packed >> (3 * lane)) & 7) : 0;
    }
    WSYNC();
    if (!UB(build<2>(s.hist, s.run, s.lens, 19, s.lit, MBITS, (uint16_t *)nullptr, &s.tlit, false, lane))) return false;
    HP(1);
    // the code lengths, run-length coded (:144-263); at most 4498 bits: inside the staged KiB
    const uint32_t rel_end = (uint32_t)((total - wbyte * 8) > 0xffffffffull ? 0xffffffffu : (total - wbyte * 8));
    const uint32_t want = literals + distances;
    if (!UB(decode_lengths(s, rel, rel_end, want, lane))) return false;
    HP(2);
    const bool okd = UB(build<1>(s.hist, s.run, s.lens + literals, (int)distances, s.dist, DBITS, (uint16_t *)nullptr, &s.tdist, true, lane, s.ext_dist));
    HP(3);
    const bool okl = UB(build<0>(s.hist, s.run, s.lens, (int)literals, s.lit, LBITS, (uint16_t *)nullptr, &s.tlit, false, lane, s.ext_lit));
    HP(4);
    if (!okl || !okd) return false;
    h.payload = wbyte * 8 + rel;
    h.literals = literals; h.distances = distances;
    return true;
}

// ---- per-lane token decoding ----------------------------------------------------------------------
// A code longer than the LUT index.  The canonical codes of one length are consecutive and lengths
// ascend with the code value, so the length of the code in front of us is the number of (left-aligned)
// per-length upper limits it reaches: a handful of compares against scalar registers instead of a
// search loop; its entry then sits at a computed index of the canonical-order table.
template <int KIND>
__device__ __forceinline__ uint32_t long_code(uint32_t bits, const Lim &lim, const Tree &t, const uint32_t *ext)
{
    const uint32_t v = __brev(bits) >> 17;                     // next 15 bits, MSB first
    uint32_t l;
    if (KIND == 0) {
        l = LBITS + 1;
#pragma unroll
        for (int k = 0; k < 14 - LBITS; ++k) l += v >= lim.lit[k];
        if (v >= lim.lit[14 - LBITS]) return entry(15, 0, K_UNDEF, 0);
    } else {
        l = DBITS + 1 + (v >= lim.dist[0]) + (v >= lim.dist[1]) + (v >= lim.dist[2]) + (v >= lim.dist[3]) +
            (v >= lim.dist[4]) + (v >= lim.dist[5]);
        if (v >= lim.dist[6]) return entry(15, 0, K_UNDEF, 0);
    }
    const uint32_t idx = t.offset[l] + (v >> (15 - l)) - t.first[l];
    return ext[idx < (KIND == 0 ? 288u : 32u) ? idx : 0];
}

// decodes the token that starts at staged bit q.  -> bits | kind << 8 with kind 0 literal / run,
// 1 end of block, 3 not a token the fast path takes (undefined code, zero run or distance, past the
// end of the input `lim`).  FULL also produces the token word.
static constexpr uint32_t D_EOB = 1, D_BAD = 3;
template <bool FULL>
__device__ __forceinline__ uint32_t decode_at(const PLds &s, const Lim &lc, uint32_t q, uint32_t lim, uint32_t &tok)
{
    uint32_t lo, hi;
    fetch(s.stage, q, lo, hi);
    uint32_t e = s.lit[lo & ((1 << LBITS) - 1)];
    if ((e & 15) == 0) e = long_code<0>(lo, lc, s.tlit, s.ext_lit);
    const uint32_t len1 = e & 15, kind = (e >> 8) & 3;
    uint32_t nbits = len1, k = kind == K_LIT ? 0u : kind == K_EOB ? D_EOB : kind == K_MATCH ? 0u : D_BAD;
    if (kind == K_MATCH) {
        const uint32_t cx = (e >> 4) & 15, p2 = len1 + cx;
        const uint32_t b2 = (uint32_t)(((uint64_t)hi << 32 | lo) >> p2);
        uint32_t d = s.dist[b2 & ((1 << DBITS) - 1)];
        if ((d & 15) == 0) d = long_code<1>(b2, lc, s.tdist, s.ext_dist);
        const uint32_t dl = d & 15, ox = (d >> 4) & 15;
        nbits = p2 + dl + ox;
        if (((d >> 8) & 3) == K_UNDEF || (d >> 16) == 0 || (e >> 16) == 0) k = D_BAD;
        if (FULL) {
            const uint32_t run = (e >> 16) + ((lo >> len1) & ((1u << cx) - 1));
            const uint32_t dd = (d >> 16) + ((b2 >> dl) & ((1u << ox) - 1));
            tok = T_MATCH | (dd - 1) << 16 | run;
        }
    } else if (FULL) {
        tok = e >> 16;
    }
    if (q + nbits > lim) k = D_BAD;
    return nbits | k << 8;
}

// ---- segment search --------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pinf_find_kernel(const PStream *__restrict__ streams, PSeg *__restrict__ segs)
{
    __shared__ __attribute__((aligned(16))) PLds s;
    const int lane = threadIdx.x;
    PSeg &sg = segs[blockIdx.x];
    const PStream &st = streams[UNI(sg.stream)];
    const gbyte *src = (const gbyte *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len), total = n * 8;
    const uint32_t j = UNI(sg.index);
    uint64_t found = NONE;
    const uint64_t resume_bit = uni64(st.start_bit);       // resumable streams: nothing in front of it is looked at again
    if (j == 0) {
        // .initial (InflatorBuffers.swift:92-104, StreamHeader.swift:16-54)
        if (resume_bit) found = resume_bit;
        else if (UNI(st.format) == SPNG_FORMAT_IOS) found = 0;
        else if (n >= 2) {
            const uint32_t cmf = src[0], flg = src[1];
            if ((cmf & 15) == 8 && (cmf >> 4) < 8 && ((cmf << 8) + flg) % 31 == 0 && !(flg & 0x20)) found = 16;
        }
    } else {
        const uint64_t sb = uni64(st.seg_bytes) * 8;
        const uint64_t lo_nom = (uint64_t)j * sb;
        const uint64_t lo_bit = lo_nom > resume_bit ? lo_nom : ((resume_bit + 1 + 63) & ~(uint64_t)63);   // (window loads want whole bytes)
        const uint64_t hi_bit = lo_nom + sb < total ? lo_nom + sb : total;
        uint32_t *win = s.vmap;                                 // 2 KiB + slack
        for (uint64_t wb = lo_bit; wb < hi_bit && found == NONE; wb += 16384) {
            stage_bytes(win, src, n, wb >> 3, 512, lane);
            {   // the 64 bytes behind the window (a header straddling its end)
                const uint64_t off = (wb >> 3) + 2048 + (uint64_t)lane * 4;
                uint32_t v = 0;
                if (lane < 16) { for (int b = 0; b < 4; ++b) if (off + b < n) v |= (uint32_t)src[off + b] << (8 * b); win[512 + lane] = v; }
                WSYNC();
            }
            for (uint32_t p = 0; p < 16384 && wb + p < hi_bit && found == NONE; p += 64) {
                const uint32_t q = p + (uint32_t)lane;
                const uint32_t w = q >> 5;
                const uint32_t d0 = win[w], d1 = win[w + 1], d2 = win[w + 2], d3 = win[w + 3];
                const uint32_t v0 = __builtin_amdgcn_alignbit(d1, d0, q);
                const uint32_t v1 = __builtin_amdgcn_alignbit(d2, d1, q);
                const uint32_t v2 = __builtin_amdgcn_alignbit(d3, d2, q);
                const uint32_t ncl = ((v0 >> 13) & 15) + 4;
                bool cand = ((v0 >> 1) & 3) == 2 && ((v0 >> 3) & 31) <= 29 && ((v0 >> 8) & 31) <= 29 &&
                            wb + q + 17 + 3 * ncl <= total && wb + q < hi_bit;
                if (__ballot(cand)) {
                    // the code-length code must be complete: sum of 2^(7-len) over the used lengths == 128
                    const uint64_t W = ((uint64_t)v2 << 47) | ((uint64_t)v1 << 15) | (v0 >> 17);
                    uint32_t kraft = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 19; ++k) {
                        const uint32_t l = (uint32_t)(W >> (3 * k)) & 7;
                        kraft += (k < ncl && l) ? 128u >> l : 0u;
                    }
                    cand = cand && kraft == 128;
                }
                unsigned long long m = __ballot(cand);
                while (m && found == NONE) {
                    const int l = __ffsll((long long)m) - 1;
                    const uint64_t at = wb + p + (uint32_t)l;
                    Hdr h;
                    if (UB(parse_header(s, src, n, at, h, lane))) found = at;
                    m &= m - 1;
                }
            }
        }
    }
    if (lane == 0) { sg.start_bit = found; sg.end_bit = 0; sg.ntok = 0; sg.tok_base = 0; sg.status = PSEG_FAIL; sg.used = 0; sg.next = 0; }
}

// ---- count (pass 1) and emit (pass 2) ----------------------------------------------------------------

// One chunk of a Huffman block in the counting pass.  `cb` = absolute first bit of the chunk, `entry`
// = absolute bit at which the first token of the chunk starts (>= cb).  Writes the chunk record and
// returns: state 0 = the block goes on (next = entry of the next chunk), 1 = end of block (next = bit
// after the end-of-block code), 2 = give up.
//
//   round 0   every lane decodes its own subsequence from a guessed start (lane 0: the true start) and
//             marks the token starts it visits in its bitmap; its chain leaves the subsequence at rV.
//   round 1   every lane follows its chain on through the subsequences behind it until it lands on a
//             bit that the owner of that subsequence has marked (from there on the two chains are
//             one), or leaves the chunk, or stops (end of block / not a token).  That gives every
//             lane a link: (lane it merged into, position).
//   path      lane 0 starts on a true token boundary, so the true chain is lane 0's chain up to its
//             link, then that lane's chain up to its link, ...: the lanes reachable from lane 0
//             (pointer doubling over the links).
//   record    per subsequence: the bit at which the true chain enters it and the number of tokens that
//             start in it.  A lane on the path knows both for every subsequence its chain crossed in
//             round 1 (it notes position and token count at each crossing; tokens are shorter than a
//             subsequence, so none is skipped) and for the prefix of the one it merged into; the rest
//             of that one is the owner's marks behind the merge point.  emit then gives every lane
//             exactly its own subsequence's tokens: no lane replays a long unmerged chain alone.
#ifdef SPNG_COUNT_PROF
#define CP_ARG , uint64_t *cp
#define CP_PASS , cp
#define CP(k) do { const uint64_t now_ = __builtin_readcyclecounter(); cp[k] += now_ - cp[15]; cp[15] = now_; } while (0)
#define CPN(k, v) (cp[k] += (v))
#else
#define CP_ARG
#define CP_PASS
#define CP(k)
#define CPN(k, v)
#endif
__device__ __forceinline__ uint32_t count_chunk(PLds &s, const Lim &lim_codes, StageRegs &sr, const gbyte *src, uint64_t n, uint64_t cb,
                                                uint64_t entry, uint32_t *rec, uint64_t &next, uint32_t &ntok, int lane CP_ARG)
{
    CP(0);
    const uint64_t sbyte = (cb >> 5) << 2;
    stage_put(s.stage, sr, lane);                               // (fetched while the chunk before was decoded)
    stage_fetch(sr, src, n, ((cb + CHB) >> 5) << 2, lane);
    const uint64_t sbit = sbyte * 8;
    const uint64_t left = n * 8 - sbit;
    const uint32_t lim = left > 0xffffffffull ? 0xffffffffu : (uint32_t)left;
    const uint32_t off0 = (uint32_t)(cb - sbit), cend = off0 + CHB;
    const uint32_t sub0 = off0 + (uint32_t)lane * SB, sub1 = sub0 + SB;
#pragma unroll
    for (int w = 0; w < SDW; ++w) s.vmap[w * 64 + lane] = 0;
    s.flag[lane] = 0;
    s.ent[lane] = 0;
    WSYNC();
    uint32_t dummy;
    const uint32_t q0 = lane == 0 ? (uint32_t)(entry - sbit) : sub0;
    uint32_t q = q0, st = 0;                                    // st: 0 running, 1 end of block, 2 not a token
    CP(1);
    while (q < sub1) {
        CPN(8, 1);
        const uint32_t t = decode_at<false>(s, lim_codes, q, lim, dummy);
        const uint32_t k = t >> 8;
        if (k) { st = k == D_EOB ? 1u : 2u; if (k == D_EOB) q += t & 255; break; }
        const uint32_t b = q - sub0;
        atomicOr(&s.vmap[(b >> 5) * 64 + lane], 1u << (b & 31));
        q += t & 255;
    }
    WSYNC();
    CP(2);
    uint32_t link = 64, cnt2 = 0, nh = 0, lastj = (uint32_t)lane;
    uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;                    // crossings: position | tokens before it << 16
    if (st == 0) {
        while (q < cend) {
            CPN(9, 1);
            const uint32_t j = (q - off0) / SB, b = q - off0 - j * SB;
            if (j != lastj) {
                const uint32_t v = q | cnt2 << 16;
                x0 = nh == 0 ? v : x0; x1 = nh == 1 ? v : x1; x2 = nh == 2 ? v : x2; x3 = nh == 3 ? v : x3;
                nh += 1; lastj = j;
            }
            if ((s.vmap[(b >> 5) * 64 + j] >> (b & 31)) & 1) { link = j; break; }
            const uint32_t t = decode_at<false>(s, lim_codes, q, lim, dummy);
            const uint32_t k = t >> 8;
            if (k) { st = k == D_EOB ? 1u : 2u; if (k == D_EOB) q += t & 255; break; }
            cnt2 += 1;
            q += t & 255;
        }
    }
    // q: where my chain merged / left the chunk / stopped
    CP(3);
    bool onpath = lane == 0;
    uint32_t jump = link;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (onpath && jump < 64) s.flag[jump] = 1;
        WSYNC();
        onpath = onpath || s.flag[lane] != 0;
        const uint32_t jj = (uint32_t)__shfl((int)jump, (int)(jump & 63), 64);
        jump = jump < 64 ? jj : 64;
    }
    if (onpath) {
        // the subsequences my chain crossed (beyond the fourth crossing they all count for the fourth: rare)
        const uint32_t nrec = nh < 4 ? nh : 4;
        const uint32_t xs[5] = {x0, x1, x2, x3, 0};
#pragma unroll
        for (int h = 0; h < 4; ++h)
            if ((uint32_t)h < nrec) {
                const uint32_t upto = (uint32_t)h + 1 < nrec ? xs[h + 1] >> 16 : cnt2;
                s.ent[lane + 1 + h] = (xs[h] & 0xffff) | (upto - (xs[h] >> 16)) << 16;
            }
        if (link < 64) {
            s.mpos[link] = (uint16_t)q;
            if (link != (uint32_t)lane + nrec) s.ent[link] = q;  // (merged behind the recorded crossings: no prefix)
        }
    }
    WSYNC();
    const uint32_t e = lane == 0 ? q0 : s.ent[lane];
    const uint32_t m = lane == 0 ? q0 : s.mpos[lane];           // where the true chain merges into mine
    uint32_t mine = e >> 16;
    if (onpath) {
        const uint32_t mb = m - sub0;                            // 0 .. SB - 1
#pragma unroll
        for (int w = 0; w < SDW; ++w) {
            const uint32_t word = s.vmap[w * 64 + lane];
            const uint32_t lo = w * 32;
            const uint32_t mask = mb >= lo + 32 ? 0u : mb > lo ? ~0u << (mb - lo) : ~0u;
            mine += (uint32_t)__popc(word & mask);
        }
    }
    uint32_t tot;
    (void)wave_excl_scan(mine, tot, lane);
    // the lane on the path whose chain left the chunk or stopped
    const unsigned long long endm = __ballot(onpath && link == 64);
    const int el = endm ? __ffsll((long long)endm) - 1 : 0;
    const uint32_t qe = (uint32_t)__shfl((int)q, el, 64), ste = endm ? (uint32_t)__shfl((int)st, el, 64) : 2u;
    next = sbit + qe;
    ntok = tot;
    rec[4 + lane] = (e & 0xffff) | mine << 16;
    if (lane == 0) { rec[0] = tot; rec[1] = ste == 1 ? 1u : 0u; rec[2] = (uint32_t)next; rec[3] = (uint32_t)(next >> 32); }
    CP(4);
    CPN(10, 1);
    CPN(11, tot);
    return ste;
}

__global__ __launch_bounds__(64) void pinf_count_kernel(const PStream *__restrict__ streams, PSeg *__restrict__ segs,
                                                        uint8_t *__restrict__ logs)
{
    __shared__ __attribute__((aligned(16))) PLds s;
    const int lane = threadIdx.x;
    PSeg &sg = segs[blockIdx.x];
    const PStream &st = streams[UNI(sg.stream)];
    const uint64_t start = uni64(sg.start_bit);
    if (start == NONE) return;
    const gbyte *src = (const gbyte *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len);
    // this segment ends where a later one begins: at the first found start it stops ON.  One it runs past
    // was no block start (a look-alike inside stored data or inside a block); whatever its wave decodes
    // from there stays off the chain (scan).
    uint64_t limit = NONE;
    uint32_t nk = UNI(sg.index) + 1;
    const uint32_t seg_first = UNI(st.seg_first), seg_count = UNI(st.seg_count);
    auto advance = [&](uint64_t from) {
        limit = NONE;
        for (; nk < seg_count; ++nk) {
            const uint64_t v = uni64(segs[seg_first + nk].start_bit);
            if (v != NONE && v >= from) { limit = v; break; }
        }
    };
    advance(start + 1);
    // the log: my own space and that of the segments behind me in which no start was found (nobody else
    // writes there; a stream whose stored or fixed blocks hide every later start needs it)
    uint32_t *log = (uint32_t *)(logs + uni64(sg.log_off));
    uint32_t log_cap;
    {
        const PSeg &upto = segs[seg_first + (nk < seg_count ? nk : seg_count - 1)];
        const uint64_t end = uni64(upto.log_off) + (nk < seg_count ? 0 : uni64(upto.log_cap));
        const uint64_t room = (end - uni64(sg.log_off)) / 4;
        log_cap = room > 0xffffff00ull ? 0xffffff00u : (uint32_t)room;
    }
    uint32_t cur = 0;
    uint64_t pos = start, ntok = 0;
    int32_t status = PSEG_FAIL;
    // A resumable stream stops in front of the first block that cannot be taken as it stands -- cut off by the end
    // of the input so far, or not acceptable: the serial kernel, started there, tells which -- and keeps the
    // blocks before it.
    const bool resumable = uni64((uint64_t)st.state) != 0;
    uint64_t ntok_block = 0;
#ifdef SPNG_COUNT_PROF
    uint64_t cp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, hpv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    cp[15] = __builtin_readcyclecounter();
#endif
    for (;;) {
        if (pos >= limit) {
            if (pos == limit) { status = PSEG_CONT; break; }
            nk += 1; advance(pos);
            continue;
        }
        Hdr h;
        CP(5);
        ntok_block = ntok;
#ifdef SPNG_COUNT_PROF
        hpv[7] = __builtin_readcyclecounter();
        const bool hok = UB(parse_header(s, src, n, pos, h, lane, hpv));
#else
        const bool hok = UB(parse_header(s, src, n, pos, h, lane));
#endif
        CP(6);
        CPN(12, 1);
        if (!hok) break;
        // the block record: what emit needs to set the block up again without parsing its header (the code
        // lengths; the payload position; stored length)
        if (cur + BREC_DW > log_cap) break;
        {
            uint32_t *br = log + cur;
            if (lane < 8) {
                const uint32_t hdr[8] = {h.type | h.bfinal << 8, h.literals | h.distances << 16, (uint32_t)h.payload,
                                         (uint32_t)(h.payload >> 32), h.stored, 0, 0, 0};
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) v = lane == k ? hdr[k] : v;
                br[lane] = v;
            }
            if (h.type == 2) for (int i = lane; i < 80; i += 64) br[8 + i] = ((const uint32_t *)s.lens)[i];
            cur += BREC_DW;
        }
        if (h.type == 0) {
            ntok += h.stored;
            pos = h.payload + (uint64_t)h.stored * 8;
        } else {
            uint64_t entry = h.payload, cb = h.payload;
            uint32_t state = 0;
            Lim lc;
            load_limits(s, lc);
            StageRegs sr;
            stage_fetch(sr, src, n, (cb >> 5) << 2, lane);
            for (;;) {
                if (cur + REC_DW > log_cap) { state = 2; break; }
                uint64_t next; uint32_t nt;
                state = UNI(count_chunk(s, lc, sr, src, n, cb, entry, log + cur, next, nt, lane CP_PASS));
                next = uni64(next);
                cur += REC_DW;
                ntok += UNI(nt);
                if (state) { entry = next; break; }
                entry = next; cb += CHB;
            }
            if (state != 1) break;
            pos = entry;
        }
        if (h.bfinal) { status = PSEG_FINAL; break; }
    }
    if (status == PSEG_FAIL && resumable) { status = PSEG_PARTIAL; ntok = ntok_block; }   // (pos is still the block's first bit)
    if (lane == 0) { sg.end_bit = pos; sg.ntok = ntok; sg.status = status; sg.next = nk; }
#ifdef SPNG_COUNT_PROF
    if (blockIdx.x == 1 && lane == 0)
        printf("count: %lu blocks %lu chunks %lu tokens; steps r0 %lu p2 %lu; cycles: stage %lu setup %lu round0 %lu phase2 %lu path+rec %lu header %lu other %lu\n",
               cp[12], cp[10], cp[11], cp[8], cp[9], cp[0], cp[1], cp[2], cp[3], cp[4], cp[6], cp[5]);
    if (blockIdx.x == 1 && lane == 0)
        printf("header: stage %lu precode %lu lengths %lu build-dist %lu build-lit %lu\n", hpv[0], hpv[1], hpv[2], hpv[3], hpv[4]);
#endif
}

__global__ __launch_bounds__(64) void pinf_emit_kernel(const PStream *__restrict__ streams, const PSeg *__restrict__ segs,
                                                       const uint8_t *__restrict__ logs, uint32_t *__restrict__ tokens,
                                                       uint32_t pass)
{
    __shared__ __attribute__((aligned(16))) PLds s;
    const int lane = threadIdx.x;
    const PSeg &sg = segs[blockIdx.x];
    const PStream &st = streams[UNI(sg.stream)];
    if (!UNI(st.ok) || UNI(st.pass) != pass || !UNI(sg.used)) return;
    const gbyte *src = (const gbyte *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len);
    const uint64_t stop = uni64(sg.end_bit);
    const uint32_t *log = (const uint32_t *)(logs + uni64(sg.log_off));
    uint32_t *out = tokens + uni64(st.tok_base) + uni64(sg.tok_base);
    uint32_t cur = 0;
    uint64_t pos = uni64(sg.start_bit), at = 0;
    const bool final_seg = UNI(sg.status) == PSEG_FINAL;
    for (;;) {
        if (!final_seg && pos >= stop) break;
        // the block as count recorded it
        Hdr h;
        {
            const uint32_t *br = log + cur;
            cur += BREC_DW;
            const uint32_t tb = UNI(br[0]), ld = UNI(br[1]);
            h.type = tb & 0xff; h.bfinal = tb >> 8;
            h.literals = ld & 0xffff; h.distances = ld >> 16;
            h.payload = (uint64_t)UNI(br[3]) << 32 | UNI(br[2]);
            h.stored = UNI(br[4]);
            if (h.type == 2) {
                for (int i = lane; i < 80; i += 64) ((uint32_t *)s.lens)[i] = br[8 + i];
                WSYNC();
                build<1>(s.hist, s.run, s.lens + h.literals, (int)h.distances, s.dist, DBITS, (uint16_t *)nullptr, &s.tdist, true, lane, s.ext_dist);
                build<0>(s.hist, s.run, s.lens, (int)h.literals, s.lit, LBITS, (uint16_t *)nullptr, &s.tlit, false, lane, s.ext_lit);
            } else if (h.type == 1) {
                Hdr dummy;
                if (!UB(parse_header(s, src, n, pos, dummy, lane))) break;   // (fixed tables; cannot fail: count parsed the same bits)
            }
        }
        if (h.type == 0) {
            const uint64_t from = h.payload / 8;
            for (uint32_t k = 0; k < h.stored; k += 64)
                if (k + lane < h.stored) out[at + k + lane] = src[from + k + lane];
            at += h.stored;
            pos = h.payload + (uint64_t)h.stored * 8;
        } else {
            uint64_t cb = h.payload;
            Lim lc;
            load_limits(s, lc);
            StageRegs sr;
            stage_fetch(sr, src, n, (cb >> 5) << 2, lane);
            for (;;) {
                const uint32_t *rec = log + cur;
                cur += REC_DW;
                const uint32_t tot = UNI(rec[0]), last = UNI(rec[1]);
                const uint64_t next = (uint64_t)UNI(rec[3]) << 32 | UNI(rec[2]);
                const uint32_t mine = rec[4 + lane];
                stage_put(s.stage, sr, lane);
                if (!last) stage_fetch(sr, src, n, ((cb + CHB) >> 5) << 2, lane);
                uint32_t t2;
                const uint32_t off = wave_excl_scan(mine >> 16, t2, lane);
                uint32_t q = mine & 0xffff;
                uint32_t *o = out + at + off;
                // four tokens per store: a lane's tokens are contiguous, and 4-byte stores from 64 lanes at 64 places
                // leave L2 lines half written for twenty iterations (PMC: 3.7 x the algorithmic write bytes)
                const uint32_t cnt = mine >> 16;
                uint32_t b0 = 0, b1 = 0, b2 = 0;
                for (uint32_t k = 0; k < cnt; ++k) {
                    uint32_t tok;
                    const uint32_t t = decode_at<true>(s, lc, q, 0xffffffffu, tok);
                    const uint32_t r = k & 3;
                    if (r == 3) { const u32x4 v = {b0, b1, b2, tok}; ((gPU128 *)(o + k - 3))->v = v; }
                    b0 = r == 0 ? tok : b0; b1 = r == 1 ? tok : b1; b2 = r == 2 ? tok : b2;
                    q += t & 255;
                }
                {
                    const uint32_t rem = cnt & 3, base = cnt - rem;
                    if (rem > 0) o[base] = b0;
                    if (rem > 1) o[base + 1] = b1;
                    if (rem > 2) o[base + 2] = b2;
                }
                at += tot;
                cb += CHB;
                if (last) { pos = next; break; }
            }
        }
        if (h.bfinal) break;
    }
}

// ---- scan: the segment chain of every stream, token offsets, passes ------------------------------------
// One wave per stream walks the chain: segment 0, then the segment count says it stopped at, ... up to the
// first one that saw the final block.  A found start that no chain member stops at (a bit pattern inside
// stored data or in the middle of a block that happens to parse as a header) is simply not on the chain.
__global__ __launch_bounds__(64) void pinf_scan_kernel(PStream *__restrict__ streams, PSeg *__restrict__ segs)
{
    const int lane = threadIdx.x;
    PStream &st = streams[blockIdx.x];
    const uint32_t first = UNI(st.seg_first), count = UNI(st.seg_count);
    bool ok = false, partial = false;
    uint64_t tok = 0, end_bit = 0;
    uint32_t k = 0;
    for (uint32_t hops = 0; hops < count; ++hops) {
        PSeg *sg = segs + first + k;
        const uint64_t start = uni64(sg->start_bit), end = uni64(sg->end_bit);
        const int32_t status = (int32_t)UNI(sg->status);
        if (start == NONE || status == PSEG_FAIL) break;
        if (lane == 0) { sg->tok_base = tok; sg->used = 1; }
        tok += uni64(sg->ntok);
        if (status == PSEG_FINAL) { ok = true; end_bit = end; break; }
        if (status == PSEG_PARTIAL) { ok = true; partial = true; end_bit = end; break; }
        const uint32_t nx = UNI(sg->next);
        if (nx <= k || nx >= count) break;
        if (uni64(segs[first + nx].start_bit) != end) break;
        k = nx;
    }
    if (lane == 0) { st.ok = ok ? (partial ? 2 : 1) : 0; st.ntok = tok; st.end_bit = end_bit; st.pass = 0; st.tok_base = 0; }
}

// single wave: global token offsets and passes.  capacity = tokens the token buffer holds.
__global__ __launch_bounds__(64) void pinf_assign_kernel(PStream *__restrict__ streams, uint32_t count, uint64_t capacity,
                                                         uint32_t passes)
{
    const int lane = threadIdx.x;
    // the largest stream
    uint64_t big = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t i = base + lane;
        const uint64_t nt = (i < count && streams[i].ok) ? streams[i].ntok : 0;
        big = nt > big ? nt : big;
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const uint64_t o = (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(big >> 32), m, 64) << 32 | (uint32_t)__shfl_xor((int)(uint32_t)big, m, 64);
        big = o > big ? o : big;
    }
    // a pass takes streams while their first token lies below `nominal`; nominal + big <= capacity
    const uint64_t nominal = capacity > big ? capacity - big : 0;
    uint64_t run = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t i = base + lane;
        const bool live = i < count && streams[i].ok;
        const uint64_t nt = live ? streams[i].ntok : 0;
        // inclusive scan of 64-bit values over the wave
        uint64_t incl = nt;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d, 64) << 32 | (uint32_t)__shfl_up((int)(uint32_t)incl, d, 64);
            if (lane >= d) incl += o;
        }
        const uint64_t excl = run + incl - nt;
        if (live) {
            if (nominal == 0) { streams[i].ok = 0; }
            else {
                const uint64_t p = excl / nominal;
                if (p >= passes) streams[i].ok = 0;
                else { streams[i].pass = (uint32_t)p; streams[i].tok_base = excl - p * nominal; }
            }
        }
        run += (uint64_t)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63, 64) << 32 | (uint32_t)__shfl((int)(uint32_t)incl, 63, 64);
    }
}

// ---- resolve: tokens -> bytes -----------------------------------------------------------------------
static constexpr uint32_t RT = 512;                 // threads per stream
static constexpr uint32_t TILE = 8192;              // output bytes resolved per step (16 per thread)
static constexpr uint32_t WINDOW = 32768;           // the DEFLATE window
static constexpr uint32_t R_DONE = 0x8000;          // state: R_DONE | byte, or the tile index of an earlier byte

#ifdef SPNG_RESOLVE_PROF
#define RP_DECL uint64_t rp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rp_t = __builtin_readcyclecounter(), rp_rounds = 0, rp_tiles = 0;
#define RP(k) do { const uint64_t now_ = __builtin_readcyclecounter(); rp[k] += now_ - rp_t; rp_t = now_; } while (0)
#else
#define RP_DECL
#define RP(k)
#endif

template <int TPT>                     // tokens looked at per thread and step
struct RLdsT {
    uint8_t  ring[WINDOW];             // the last 32 KiB of output, at position mod 32 KiB
    uint16_t state[TILE];              // marks while a tile is laid out, then one entry per output byte
    uint32_t tokv[RT * TPT];
    uint16_t toks[RT * TPT];           // first byte of each token on the tile
    uint32_t part[24];
    uint32_t again[3];                 // pointer jumping: somebody still has an unknown byte (flag of round r: r mod 3)
};

// exclusive prefix sum over the workgroup (8 waves); every thread gets the grand total too
template <class RLds>
__device__ __forceinline__ uint32_t block_excl_scan(RLds &s, uint32_t v, uint32_t &total, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t wt;
    const uint32_t off = wave_excl_scan(v, wt, lane);
    __syncthreads();
    if (lane == 0) s.part[wave] = wt;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < (int)(RT / 64); ++w) { const uint32_t p = s.part[w]; before += w < wave ? p : 0u; all += p; }
    total = all;
    return off + before;
}
template <class RLds>
__device__ __forceinline__ uint32_t block_sum(RLds &s, uint32_t v, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t wt = wave_sum(v);
    if (lane == 0) s.part[8 + wave] = wt;
    __syncthreads();
    uint32_t all = 0;
#pragma unroll
    for (int w = 0; w < (int)(RT / 64); ++w) all += s.part[8 + w];
    return all;
}
template <class RLds>
__device__ __forceinline__ uint32_t block_excl_max(RLds &s, uint32_t v, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl = o > incl ? o : incl;
    }
    uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
    if (lane == 0) excl = 0;
    const uint32_t wt = (uint32_t)__shfl((int)incl, 63, 64);
    if (lane == 0) s.part[16 + wave] = wt;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < (int)(RT / 64); ++w) { const uint32_t p = s.part[16 + w]; before = (w < wave && p > before) ? p : before; }
    return excl > before ? excl : before;
}

// One workgroup per stream, one 8 KiB tile of output per step, sixteen bytes per thread.  Every output
// byte of the tile gets a 16-bit state: R_DONE | value once it is known, else the tile index of the
// earlier byte it copies.
//   layout   the next tokens are prefix-summed onto the tile (as many as fit); each token marks its
//            first byte with its number; a thread fills the token numbers forward over its sixteen
//            bytes, fetches the token of every byte and writes a literal's value, the value of a source
//            before the tile (the last 32 KiB of output live in an LDS ring) or a pointer at an earlier
//            byte of the tile (InflatorOut.expand, InflatorOut.swift:124-139: a forward byte copy).
//   jumping  state[j] = state[state[j]] for every unknown byte, until none is left: resolving a byte
//            and halving its pointer chain are the same 16-bit gather, so no ordering between threads
//            is needed (whatever is read is a valid member of the byte's chain).
//   store    sixteen bytes per thread to the output and the ring, Adler-32 folded in (inflate.hip:
//            struct Out).
// Two instantiations: 4 tokens per thread fill a tile when tokens average 4 bytes (zlib-made streams);
// literal-heavy streams (swift-png's own level-6 output: 2 bytes per token) take 8, or their tiles would
// be half empty.  A workgroup leaves at once when its stream belongs to the other instantiation.
template <int TPT>
__global__ __launch_bounds__(RT, 4) void pinf_resolve_kernel(const PStream *__restrict__ streams, const uint32_t *__restrict__ tokens,
                                                             spng_result *__restrict__ results, int32_t *__restrict__ done,
                                                             uint32_t pass)
{
    typedef RLdsT<TPT> RLds;
    constexpr uint32_t TOKS = RT * TPT;
    __shared__ __attribute__((aligned(16))) RLds s;
    const int tid = threadIdx.x;
    const PStream &st = streams[blockIdx.x];
    if (!UNI(st.ok) || UNI(st.pass) != pass) return;
    const uint32_t *tk = tokens + uni64(st.tok_base);
    const uint64_t ntok = uni64(st.ntok);
    if ((uni64(st.dst_cap) < 3 * ntok) != (TPT == 8)) return;   // (capacity ~ output size: under 3 bytes per token)
    gbyte *dst = (gbyte *)uni64((uint64_t)st.dst);
    const uint64_t cap = uni64(st.dst_cap);
    const gbyte *src = (const gbyte *)uni64((uint64_t)st.src);
    const uint64_t n = uni64(st.src_len);
    uint64_t pos = uni64(st.out_pos), ti = 0;      // (resumable streams: the bytes earlier calls produced are in dst)
    uint32_t accS = 0, accI = 0;                     // Adler-32 partial sums (inflate.hip: struct Out)
    bool bad = false;
    const uint32_t j0 = (uint32_t)tid * 16;
    uint64_t *state = (uint64_t *)uni64((uint64_t)st.state);
    if (pos) {
        // the window so far
        for (uint64_t p = (pos > WINDOW ? pos - WINDOW : 0) + (uint32_t)tid; p < pos; p += RT) s.ring[p & (WINDOW - 1)] = dst[p];
        __syncthreads();
    }
    RP_DECL
    uint32_t tokr[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) { const uint64_t i = (uint64_t)tid * TPT + k; tokr[k] = i < ntok ? tk[i] : 0u; }
    while (ti < ntok) {
        RP(0);
        // ---- lay the next tokens out on the tile
        uint32_t len[TPT], sum = 0;
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const uint64_t i = ti + (uint64_t)tid * TPT + k;
            len[k] = i < ntok ? ((tokr[k] & T_MATCH) ? (tokr[k] & 0x1ff) : 1u) : 0u;
            sum += len[k];
        }
        {   // clear my marks
            const u32x4 z = {0, 0, 0, 0};
            u32x4 *p = (u32x4 *)(s.state + j0);
            p[0] = z; p[1] = z;
        }
        uint32_t total;
        uint32_t off = block_excl_scan(s, sum, total, tid);      // (barriers inside: all marks are clear)
        uint32_t pack = 0;                                       // tokens taken << 16 | their bytes
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const uint32_t end = off + len[k];
            if (len[k] && end <= TILE) {
                const uint32_t id = (uint32_t)tid * TPT + k;
                s.tokv[id] = tokr[k];
                s.toks[id] = (uint16_t)off;
                s.state[off] = (uint16_t)(id + 1);
                pack += (1u << 16) + len[k];
            }
            off = end;
        }
        // (everything fits: the usual case with 2048 tokens of ~4 bytes; else count what was taken)
        uint32_t nused = TOKS, tlen = total;
        if (ti + TOKS > ntok) nused = (uint32_t)(ntok - ti);
        if (total > TILE) { const uint32_t ptot = block_sum(s, pack, tid); nused = ptot >> 16; tlen = ptot & 0xffff; }
        else __syncthreads();                                    // (barrier: marks and tokens are visible)
        if (pos + tlen > cap) { bad = true; break; }
        // the next tile's tokens travel while this one is resolved
#pragma unroll
        for (int k = 0; k < TPT; ++k) { const uint64_t i = ti + nused + (uint64_t)tid * TPT + k; tokr[k] = i < ntok ? tk[i] : 0u; }
        RP(1);
        // ---- the token of each of my sixteen bytes
        uint32_t mkw[8];
        {
            const u32x4 *p = (const u32x4 *)(s.state + j0);
            const u32x4 a = p[0], c = p[1];
            mkw[0] = a.x; mkw[1] = a.y; mkw[2] = a.z; mkw[3] = a.w; mkw[4] = c.x; mkw[5] = c.y; mkw[6] = c.z; mkw[7] = c.w;
        }
        uint32_t lastmark = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) { const uint32_t mkb = (b & 1) ? mkw[b >> 1] >> 16 : mkw[b >> 1] & 0xffff; lastmark = mkb ? mkb : lastmark; }
        uint32_t id = block_excl_max(s, lastmark, tid);          // token covering byte j0 - 1 (marks ascend)
        RP(2);
        uint32_t stw[8];                                         // my sixteen states, two per word
        const uint32_t rbase = (uint32_t)pos & (WINDOW - 1);
        uint32_t cstart = id ? s.toks[id - 1] : 0u;              // first byte of the token I am in
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t tokb[8], farb[8], sidx[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int bb = 8 * h + b;
                const uint32_t j = j0 + bb;
                const uint32_t mkb = (bb & 1) ? mkw[bb >> 1] >> 16 : mkw[bb >> 1] & 0xffff;
                id = mkb ? mkb : id;
                cstart = mkb ? j : cstart;
                tokb[b] = s.tokv[(id ? id : 1u) - 1];
                sidx[b] = j - cstart;                            // (for now: my offset inside the token)
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint32_t j = j0 + 8 * h + b;
                const uint32_t d = ((tokb[b] >> 16) & 0x7fff) + 1;
                // A run longer than its distance repeats its first `distance` bytes: a byte beyond the
                // first period copies the period in front of the run (same value, chain one level deep
                // instead of run / distance levels).
                uint32_t k = sidx[b];
                if (k >= d) k -= d * (uint32_t)__fdividef((float)k + 0.5f, (float)d);          // k mod d, k < 258
                sidx[b] = j - sidx[b] + k - d;                   // >= 0x80000000: before the tile
                // the source in the ring (for a byte that needs none: some byte of the ring)
                farb[b] = s.ring[(rbase + sidx[b]) & (WINDOW - 1)];
                if ((tokb[b] & T_MATCH) && (int32_t)sidx[b] < 0 && j < tlen && pos < (uint64_t)(0u - sidx[b])) bad = true;
            }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int bb = 8 * h + b;
                const uint32_t j = j0 + bb;
                uint32_t sv = (tokb[b] & T_MATCH) ? ((int32_t)sidx[b] < 0 ? R_DONE | farb[b] : sidx[b]) : R_DONE | (tokb[b] & 0xff);
                sv = j < tlen ? sv : R_DONE;
                if (bb & 1) stw[bb >> 1] |= sv << 16; else stw[bb >> 1] = sv;
            }
        }
        if (tid < 3) s.again[tid] = 0;
        {
            u32x4 *p = (u32x4 *)(s.state + j0);
            const u32x4 a = {stw[0], stw[1], stw[2], stw[3]}, c = {stw[4], stw[5], stw[6], stw[7]};
            p[0] = a; p[1] = c;
        }
        __syncthreads();
        RP(3);
        // ---- pointer jumping
        for (uint32_t round = 0;; ++round) {
            bool more = false;
            uint32_t unk = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) unk |= ~stw[k] & (R_DONE | R_DONE << 16);
            if (unk) {
                uint32_t g[16];
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t sv = (b & 1) ? stw[b >> 1] >> 16 : stw[b >> 1] & 0xffff;
                    g[b] = s.state[(sv & R_DONE) ? j0 + b : sv];
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t sv = (b & 1) ? stw[b >> 1] >> 16 : stw[b >> 1] & 0xffff;
                    if (!(sv & R_DONE)) {
                        stw[b >> 1] = (b & 1) ? (stw[b >> 1] & 0xffffu) | g[b] << 16 : (stw[b >> 1] & 0xffff0000u) | g[b];
                        more = more || !(g[b] & R_DONE);
                    }
                }
                u32x4 *p = (u32x4 *)(s.state + j0);
                const u32x4 a = {stw[0], stw[1], stw[2], stw[3]}, c = {stw[4], stw[5], stw[6], stw[7]};
                p[0] = a; p[1] = c;
            }
#ifdef SPNG_RESOLVE_PROF
            rp_rounds += 1;
#endif
            // one barrier per round.  Three flags in rotation: the one cleared here was last read before
            // this round's barrier and is next set after the next round's.
            const uint32_t fr = round % 3;
            if (more) s.again[fr] = 1;
            __syncthreads();
            const bool go = s.again[fr] != 0;
            if (tid == 0) s.again[fr == 0 ? 2 : fr - 1] = 0;
            if (!go) break;
        }
        RP(4);
        // ---- store the tile (output + ring), fold it into Adler-32
        if (j0 < tlen) {
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t a = stw[2 * k], c = stw[2 * k + 1];
                w[k] = (a & 0xff) | (a >> 8 & 0xff00) | (c & 0xff) << 16 | (c >> 16 & 0xff) << 24;
            }
            const uint32_t valid = tlen - j0 >= 16 ? 16 : tlen - j0;
            const uint32_t ro = (rbase + j0) & (WINDOW - 1);
            if (valid == 16) {
                const u32x4 v = {w[0], w[1], w[2], w[3]};
                ((gPU128 *)(dst + pos + j0))->v = v;
                if (ro + 16 <= WINDOW) ((PU128 *)(s.ring + ro))->v = v;
                else for (uint32_t k = 0; k < 16; ++k) s.ring[(ro + k) & (WINDOW - 1)] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            } else {
                for (uint32_t k = 0; k < valid; ++k) {
                    const uint8_t by = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                    dst[pos + j0 + k] = by;
                    s.ring[(ro + k) & (WINDOW - 1)] = by;
                }
                for (uint32_t k = valid; k < 16; ++k) w[k >> 2] &= ~(0xffu << (8 * (k & 3)));
            }
            uint32_t A = 0, J = 0;
            A = __builtin_amdgcn_sad_u8(w[0], 0, A); A = __builtin_amdgcn_sad_u8(w[1], 0, A);
            A = __builtin_amdgcn_sad_u8(w[2], 0, A); A = __builtin_amdgcn_sad_u8(w[3], 0, A);
            J = __builtin_amdgcn_udot4(w[0], 0x03020100u, J, false);
            J = __builtin_amdgcn_udot4(w[1], 0x07060504u, J, false);
            J = __builtin_amdgcn_udot4(w[2], 0x0b0a0908u, J, false);
            J = __builtin_amdgcn_udot4(w[3], 0x0f0e0d0cu, J, false);
            uint32_t g = (uint32_t)(pos % 65521) + j0;               // < 65521 + 8192
            g = g >= 65521 ? g - 65521 : g;
            accS = (accS + A) % 65521;
            accI = (accI + g * A + J) % 65521;
        }
        pos += tlen;
        ti += nused;
        RP(5);
#ifdef SPNG_RESOLVE_PROF
        rp_tiles += 1;
#endif
    }
#ifdef SPNG_RESOLVE_PROF
    if (blockIdx.x == 0 && (tid == 0 || tid == 100))
        printf("resolve[t%d]: %lu tiles, %lu rounds; cycles: loop %lu scan+layout %lu marks+maxscan %lu expand %lu jump %lu store %lu\n",
               tid, rp_tiles, rp_rounds, rp[0], rp[1], rp[2], rp[3], rp[4], rp[5]);
#endif
    // ---- verdict
    if (__syncthreads_or(bad)) return;                          // leave it to the serial kernel
    {
        // S = sum b_i, I = sum i * b_i (mod 65521) over the workgroup
        const uint32_t S1 = wave_sum(accS) % 65521, I1 = wave_sum(accI % 65521) % 65521;
        __syncthreads();
        if ((tid & 63) == 0) { s.part[tid >> 6] = S1; s.part[8 + (tid >> 6)] = I1; }
        __syncthreads();
        uint32_t S = 0, I = 0;
        for (int w = 0; w < (int)(RT / 64); ++w) { S += s.part[w]; I += s.part[8 + w]; }
        S %= 65521; I %= 65521;
        if (tid == 0 && st.ok == 2) {
            // resumable, and the chain stopped in front of a block the input does not hold (or that is not
            // acceptable): the serial kernel goes on from there
            state[0] = st.end_bit; state[1] = pos;
        } else if (tid == 0 && state) {
            // resumable and complete: the trailer must be there; the sum over ALL bytes is compared afterwards (gzip.hip)
            const uint64_t endb = (st.end_bit + 7) / 8, consumed = endb + (st.format == SPNG_FORMAT_ZLIB ? 4 : 0);
            spng_result &res = results[st.image];
            if (consumed <= n) {
                res.status = SPNG_DONE; res.reserved = 1;
                res.written = pos; res.consumed = consumed;
                res.aux[0] = res.aux[1] = 0;
                done[blockIdx.x] = 1;
            }   // (else: the serial kernel, from where this call started, reports "need more input")
        } else if (tid == 0) {
            const uint64_t endb = (st.end_bit + 7) / 8;
            bool good = true;
            uint64_t consumed = endb;
            if (st.format != SPNG_FORMAT_IOS) {
                // .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429)
                if (endb + 4 > n) good = false;
                else {
                    const uint32_t declared = (uint32_t)src[endb] << 24 | (uint32_t)src[endb + 1] << 16 |
                                              (uint32_t)src[endb + 2] << 8 | (uint32_t)src[endb + 3];
                    const uint32_t N = (uint32_t)(pos % 65521);
                    const uint32_t computed = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
                    good = declared == computed;
                    consumed = endb + 4;
                }
            }
            if (good) {
                spng_result &res = results[st.image];
                res.status = SPNG_DONE; res.reserved = 1;
                res.written = pos; res.consumed = consumed;
                res.aux[0] = res.aux[1] = 0;
                done[blockIdx.x] = 1;
            }
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------
// The pipeline stage by stage (api.hip times each launch separately): find -> count -> scan -> per pass {emit, resolve}.
hipError_t launch_pinf_find(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint32_t nsegs, int32_t *d_done, hipStream_t stream)
{
    (void)d_done; (void)nstreams;                               // (the flags arrive zeroed with the staged plan; the gzip header kernel may have set some)
    pinf_find_kernel<<<nsegs, 64, 0, stream>>>(d_streams, d_segs);
    return hipGetLastError();
}
hipError_t launch_pinf_count(PStream *d_streams, PSeg *d_segs, uint32_t nsegs, uint8_t *d_logs, hipStream_t stream)
{
    pinf_count_kernel<<<nsegs, 64, 0, stream>>>(d_streams, d_segs, d_logs);
    return hipGetLastError();
}
hipError_t launch_pinf_scan(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint64_t tok_cap, uint32_t passes, hipStream_t stream)
{
    pinf_scan_kernel<<<nstreams, 64, 0, stream>>>(d_streams, d_segs);
    pinf_assign_kernel<<<1, 64, 0, stream>>>(d_streams, nstreams, tok_cap, passes);
    return hipGetLastError();
}
hipError_t launch_pinf_emit(PStream *d_streams, PSeg *d_segs, uint32_t nsegs, uint8_t *d_logs, uint32_t *d_tokens, uint32_t pass,
                            hipStream_t stream)
{
    pinf_emit_kernel<<<nsegs, 64, 0, stream>>>(d_streams, d_segs, d_logs, d_tokens, pass);
    return hipGetLastError();
}
hipError_t launch_pinf_resolve(PStream *d_streams, uint32_t nstreams, uint32_t *d_tokens, spng_result *d_results, int32_t *d_done,
                               uint32_t pass, hipStream_t stream)
{
    pinf_resolve_kernel<4><<<nstreams, RT, 0, stream>>>(d_streams, d_tokens, d_results, d_done, pass);
    pinf_resolve_kernel<8><<<nstreams, RT, 0, stream>>>(d_streams, d_tokens, d_results, d_done, pass);
    return hipGetLastError();
}

}  // namespace spng
