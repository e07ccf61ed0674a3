// deflate.hip -- batched DEFLATE / zlib compression for gfx950, bit-exact with swift-png's
// LZ77.Deflator at every level: greedy (0-3), lazy (4-7) and the shortest-path search (8 and up);
// one wavefront per stream.
//
// Replaces (whole-stream form, i.e. LZ77.Deflator.push(all, last: true)):
//   level table        Sources/LZ77/Deflator/LZ77.DeflatorSearch.swift:13-35
//   compress loops     Sources/LZ77/Deflator/LZ77.DeflatorBuffers.Stream.swift:64-404 (greedy, lazy, full)
//   match graph        Sources/LZ77/Deflator/LZ77.DeflatorMatches.swift:162-379, ...Depths.swift:4-112
//   window / chains    Sources/LZ77/Deflator/LZ77.DeflatorWindow.swift:78-212, F14 exact map
//   terms              Sources/LZ77/Deflator/LZ77.DeflatorTerm.swift:10-56, LZ77.Decades.swift
//   block writer       Sources/LZ77/Deflator/LZ77.DeflatorBuffers.Stream.swift:440-709
//   tree construction  Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:247-404, LZ77.Heap.swift
//   header / trailer   Sources/LZ77/Inflator/LZ77.StreamHeader.swift:56-62, LZ77.MRC32.swift
//
// Design.  The reference's match candidates are a pure function of the input: every position is
// entered into the window, and the candidates of position p are the earlier positions with the
// same 4-byte key, most recent first (first one at distance <= 32767, later ones < 32767), tried
// until `attempts` run out or a run >= `goal` is seen; the first strictly longest run > 5 wins.
// Only the parse (which positions are asked) is sequential.  So the wave works in three layers:
//   * hash insertion, 64 positions per step: every lane hashes its key, reads the bucket head from
//     LDS, an unrolled readlane sweep resolves same-bucket positions inside the batch, and each
//     position's link (distance to the previous same-bucket position + a 16-bit key tag) goes to a
//     64 K-entry ring in HBM.  Insertion runs ahead of the parse -- later positions never appear
//     in an earlier position's chain, which only walks backwards;
//   * match search, 64 positions per step: lane i walks the chain of position w+i (tag filter,
//     then dword-wise comparison straight from the input), all lanes at once, so the HBM/L2
//     latency of a chain hop is paid once per step instead of once per position;
//   * the parse itself walks those 64 results on the scalar unit (readlane per token) with the
//     reference's greedy / lazy rules, packing terms exactly like LZ77.DeflatorTerm.
// When 2047 terms are queued (lazy: 2046/2047) the block is written: histogram with LDS atomics,
// the reference's heap-based length-limited Huffman construction (ranked in parallel, merged on
// one lane because its tie-breaking is order dependent), code-length RLE, and the token bits.
#include "common.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U128u { u32x4 v; };
// Input, output and the link ring are global memory, and say so in their types (a generic pointer
// costs flat instructions, which also tie up the LDS counter).
typedef uint8_t __attribute__((address_space(1))) gbyte;
typedef uint32_t __attribute__((address_space(1))) gword;
typedef U32u __attribute__((address_space(1))) gU32u;
// wave-uniform values, said so to the compiler (see inflate.hip)
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)UNI(v >> 32) << 32 | UNI((uint32_t)v); }

static constexpr int HBITS = 13;                 // bucket heads in LDS
static constexpr uint32_t NONE = 0xffffffffu;
static constexpr int OUTB = 8192;                // output staging ring (bytes, power of two)

// LZ77.Composites.swift:25-110 in closed form (table loads from HBM would sit on the serial path):
// number of extra bits of a run / distance decade, and the extra-bit value of a run / distance.
__device__ __forceinline__ uint32_t run_extra_bits(uint32_t decade) { return decade < 9 || decade == 29 ? 0u : (decade - 5) >> 2; }
__device__ __forceinline__ uint32_t dist_extra_bits(uint32_t decade) { return decade < 4 ? 0u : (decade >> 1) - 1; }
__device__ __forceinline__ uint32_t run_extra_value(uint32_t run, uint32_t decade) { return (run - 3) & ((1u << run_extra_bits(decade)) - 1); }
__device__ __forceinline__ uint32_t dist_extra_value(uint32_t d, uint32_t decade) { return (d - 1) & ((1u << dist_extra_bits(decade)) - 1); }

// LZ77.Decades.swift in closed form
__device__ __forceinline__ uint32_t run_decade(uint32_t run)
{
    if (run < 11) return run - 2;
    if (run == 258) return 29;
    const uint32_t x = run - 3, e = 29 - __builtin_clz(x);       // extra bits: 1 for 8..15, 2 for 16..31, ...
    return 1 + 4 * e + 4 + (x >> e) - 4;                          // 4 codes per extra-bit class
}
__device__ __forceinline__ uint32_t dist_decade(uint32_t d)
{
    if (d < 5) return d - 1;
    const uint32_t x = d - 1, e = 30 - __builtin_clz(x);          // extra bits
    return 2 * e + 2 + ((x >> e) & 1);
}

static constexpr int CHV = 2048;                 // vertices per back-trace chunk
struct DLds {
    uint32_t head[(1 << HBITS) + 1];             // most recent position per bucket (low 32 bits); + a slot nobody reads
    union {
        uint32_t terms[2048];                    // greedy / lazy: the queued terms
        uint32_t cslot[30 * 64];                 // full: per lane, the best run of every distance decade (distance << 16 | run)
    };
    // full search (levels >= 8): forward pass / back-trace scratch
    union {
        struct {
            uint32_t win_depth[512], win_up[512];    // the next vertices' best depth and the edge it came by
            uint32_t batch[64 * 30];                 // edge slots of the 64 vertices being explored
        };
        struct {
            uint32_t upc[CHV];                       // chunk of the per-vertex incoming edges
            uint16_t jump[2][CHV];                   // pointer doubling over the chunk
        };
    };
    uint8_t  onpath[CHV];
    uint8_t  depths[544];                        // LZ77.DeflatorMatches.Depths: cost of every symbol in quarter bits
    uint32_t freq[320];                          // 0..287 lit/len, 288..319 distance
    uint8_t  out[OUTB];
    // Huffman scratch (one tree at a time)
    uint16_t order[288];                         // symbols by descending frequency (stable)
    uint64_t heap[288];                          // heap entries: key << 32 | node id
    uint16_t parent[576];                        // tree nodes: leaves 0..m-1 (in `order`), then merges
    uint16_t depthcnt[300];                      // leaves per depth
    uint8_t  ll[288], dl[32], ml[19];            // code lengths
    uint16_t lcode[288], dcode[32], mcode[19];   // bit-reversed codewords
    uint8_t  msym[320], mbits[320];              // code-length RLE terms
    uint8_t  cl[20];                             // code-length-code lengths in transmission order
};
// One instance per workgroup, at namespace scope so that the (non-inlined) block writer reaches it
// with LDS instructions instead of through a generic pointer.
__shared__ __attribute__((aligned(16))) DLds g_lds;

struct Bits {                                    // LSB-first bit writer (LZ77.DeflatorOut.append)
    uint64_t acc; uint32_t nacc;
    uint64_t total;                              // bytes produced so far
    uint64_t flushed;
    gbyte *dst; uint64_t cap; bool overflow;
};

__device__ __forceinline__ void put(DLds &s, Bits &b, uint32_t bits, uint32_t count, int lane)
{
    b.acc |= (uint64_t)(bits & ((1u << count) - 1)) << b.nacc;
    b.nacc += count;
    while (b.nacc >= 8) {
        s.out[b.total & (OUTB - 1)] = (uint8_t)b.acc;          // (every lane, same byte: no lane-dependent branch)
        b.total++; b.acc >>= 8; b.nacc -= 8;
    }
}
__device__ __forceinline__ void drain(DLds &s, Bits &b, uint64_t upto, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    for (uint64_t i = b.flushed + lane; i < upto; i += 64) {
        if (i < b.cap) b.dst[i] = s.out[i & (OUTB - 1)];
    }
    if (upto > b.cap) b.overflow = true;
    b.flushed = upto;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}
__device__ __forceinline__ void maybe_drain(DLds &s, Bits &b, int lane)
{
    if (b.total - b.flushed >= OUTB / 2) drain(s, b, b.total, lane);
}

// HuffmanTree.init(frequencies:limit:) (HuffmanTree.swift:247-344) for `n` symbols with counts
// in freq[0..n): code length per symbol into len[].  The heap (LZ77.Heap.swift) is replayed
// exactly -- which two nodes merge on equal keys depends on its sift order -- but its values are
// node ids: the reference's per-level leaf-count vectors are the depth histogram of the tree.
__device__ __attribute__((noinline)) void build_tree(const uint32_t *freq, int n, int limit, uint8_t *len, int lane)
{
    DLds &s = g_lds;
    for (int i = lane; i < n; i += 64) len[i] = 0;
    // rank = position in (descending frequency, ascending symbol) order
    int m = 0;
    for (int base = 0; base < n; base += 64) m += __popcll(__ballot(base + lane < n && freq[base + lane] > 0));
    for (int i = lane; i < n; i += 64) {
        const uint32_t f = freq[i];
        if (!f) continue;
        int rank = 0;
        for (int j = 0; j < n; ++j) { const uint32_t g = freq[j]; rank += (g > f) | ((g == f) & (j < i) & (g != 0)); }
        s.order[rank] = (uint16_t)i;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (m <= 1) {                                              // stub tree (HuffmanTree.swift:52-65)
        if (m == 1 && lane == 0) len[s.order[0]] = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        return;
    }
    if (lane == 0) {
        // heap over symbols.reversed(): ascending frequency; leaf k = order[m-1-k].  Entries are
        // (key << 32 | node) so that one LDS access moves an element; the sifts carry the moving
        // element in registers (same comparisons, same outcome as the reference's swap form).
        uint64_t *hp = s.heap;                                 // 1-based: hp[i - 1]
        int count = m, nodes = m;
        for (int k = 0; k < m; ++k) hp[k] = (uint64_t)freq[s.order[m - 1 - k]] << 32 | (uint32_t)k;
        auto sift_down = [&](int i, uint64_t moving) {         // Heap.siftDown / lowest(below:) (:94-135)
            const uint32_t key = (uint32_t)(moving >> 32);
            for (;;) {
                const int l = i << 1, r = l + 1;
                if (l > count) break;
                const uint64_t el = hp[l - 1], er = r <= count ? hp[r - 1] : ~0ull;
                const bool right = r <= count && (uint32_t)(er >> 32) < (uint32_t)(el >> 32);
                const uint64_t ec = right ? er : el;
                if (!((uint32_t)(ec >> 32) < key)) break;
                hp[i - 1] = ec;
                i = right ? r : l;
            }
            hp[i - 1] = moving;
        };
        auto dequeue = [&]() -> uint64_t {                     // Heap.dequeue (:149-164)
            const uint64_t top = hp[0];
            const uint64_t moving = hp[count - 1];
            --count;
            if (count > 0) sift_down(1, moving);
            return top;
        };
        for (int i = count >> 1; i >= 1; --i) sift_down(i, hp[i - 1]);   // heapify (:166-175)
        uint16_t root = 0;
        for (;;) {
            const uint64_t e1 = dequeue();
            if (count == 0) { root = (uint16_t)e1; break; }
            const uint64_t e2 = dequeue();
            const uint16_t id = (uint16_t)nodes++;
            s.parent[(uint16_t)e1] = id; s.parent[(uint16_t)e2] = id;
            const uint32_t key = (uint32_t)(e1 >> 32) + (uint32_t)(e2 >> 32);
            int i = ++count;                                   // enqueue + siftUp (:137-147, :113-123)
            for (;;) {
                const int p = i >> 1;
                if (p < 1) break;
                const uint64_t ep = hp[p - 1];
                if (!(key < (uint32_t)(ep >> 32))) break;
                hp[i - 1] = ep; i = p;
            }
            hp[i - 1] = (uint64_t)key << 32 | id;
        }
        // depth histogram: depthcnt[d-1] = leaves at depth d
        int maxd = 0;
        for (int d = 0; d < 300; ++d) s.depthcnt[d] = 0;
        for (int k = 0; k < m; ++k) {
            int d = 0;
            for (uint16_t v = (uint16_t)k; v != root; v = s.parent[v]) ++d;
            s.depthcnt[d - 1]++;
            if (d > maxd) maxd = d;
        }
        // HuffmanTree.limitHeight (:348-404)
        int nl = maxd;
        if (nl > limit) {
            int unhoused = 0;
            for (int l = nl - 1; l >= limit; --l) {
                const int pairs = s.depthcnt[l] >> 1;
                unhoused += pairs;
                s.depthcnt[l - 1] += (uint16_t)pairs;
            }
            nl = limit;
            int split = limit - 2;
            while (unhoused > 0) {
                if (!(s.depthcnt[split] > 0)) { split--; continue; }
                const int resettled = s.depthcnt[split] < unhoused ? s.depthcnt[split] : unhoused;
                unhoused -= resettled;
                s.depthcnt[split] -= (uint16_t)resettled;
                s.depthcnt[split + 1] += (uint16_t)(2 * resettled);
                if (split < limit - 2) split++;
            }
        }
        // most frequent symbols take the shortest codes (:306-337)
        int at = 0;
        for (int l = 0; l < nl; ++l)
            for (int k = 0; k < s.depthcnt[l]; ++k) len[s.order[at++]] = (uint8_t)(l + 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// canonical codewords, bit-reversed for LSB-first emission (HuffmanTree.codewords :206-230)
__device__ __attribute__((noinline)) void make_codes(const uint8_t *len, int n, uint16_t *code, int lane)
{
    if (lane == 0) {
        uint32_t counter = 0;
        for (int l = 1; l <= 15; ++l) {
            for (int sym = 0; sym < n; ++sym) if (len[sym] == l) {
                code[sym] = (uint16_t)(__brev(counter) >> (32 - l));
                counter++;
            }
            counter <<= 1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// The head of a block once the code lengths stand in s.ll / s.dl: code-length RLE, the code-length code,
// codewords, then writeBlockMetadata + writeBlockTables (DeflatorBuffers.Stream.swift:459-623).
__device__ __attribute__((noinline)) Bits write_tables(Bits b, bool final, int lane)
{
    DLds &s = g_lds;
    if (lane < 2) { s.ll[286 + lane] = 0; s.dl[30 + lane] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    // code-length RLE (:459-552) and the meta tree, on one lane
    int r = 0, dn = 0, nm = 0;
    if (lane == 0) {
        r = 286; while (r > 0 && s.ll[r - 1] == 0) --r;
        if (r < 257) r = 257;
        dn = 30; while (dn > 0 && s.dl[dn - 1] == 0) --dn;
        if (dn < 1) dn = 1;
        auto length_at = [&](int idx) -> uint8_t { return idx < r ? s.ll[idx] : s.dl[idx - r]; };
        int reps = 1; uint8_t last = length_at(0);
        for (int idx = 1; ; ++idx) {
            const bool have = idx < r + dn;
            const uint8_t cur = have ? length_at(idx) : 0;
            if (have && cur == last) { reps++; continue; }
            if (last == 0) {
                while (reps > 138) { s.msym[nm] = 18; s.mbits[nm++] = 138 - 11; reps -= 138; }
                if (reps > 2) { if (reps < 11) { s.msym[nm] = 17; s.mbits[nm++] = (uint8_t)(reps - 3); }
                                else { s.msym[nm] = 18; s.mbits[nm++] = (uint8_t)(reps - 11); } }
                else for (int k = 0; k < reps; ++k) { s.msym[nm] = 0; s.mbits[nm++] = 0; }
            } else {
                s.msym[nm] = last; s.mbits[nm++] = 0; reps -= 1;
                while (reps > 6) { s.msym[nm] = 16; s.mbits[nm++] = 6 - 3; reps -= 6; }
                if (reps > 2) { s.msym[nm] = 16; s.mbits[nm++] = (uint8_t)(reps - 3); }
                else for (int k = 0; k < reps; ++k) { s.msym[nm] = last; s.mbits[nm++] = 0; }
            }
            if (!have) break;
            last = cur; reps = 1;
        }
        for (int k = 0; k < 19; ++k) s.freq[k] = 0;
        for (int k = 0; k < nm; ++k) s.freq[s.msym[k]]++;
    }
    r = __builtin_amdgcn_readfirstlane(r); dn = __builtin_amdgcn_readfirstlane(dn); nm = __builtin_amdgcn_readfirstlane(nm);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    build_tree(s.freq, 19, 7, s.ml, lane);
    make_codes(s.ll, 288, s.lcode, lane);
    make_codes(s.dl, 32, s.dcode, lane);
    make_codes(s.ml, 19, s.mcode, lane);

    // writeBlockMetadata (:577-612)
    // (in LDS: a local array indexed at run time would live in scratch memory)
    if (lane < 19) {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // symbol sent k-th
        uint32_t sym = 0;
#pragma unroll
        for (int k = 0; k < 19; ++k) if (lane == k) sym = order[k];
        s.cl[lane] = s.ml[sym];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    int ncl = 19; while (ncl > 0 && UNI(s.cl[ncl - 1]) == 0) --ncl;
    if (ncl < 4) ncl = 4;
    put(s, b, final ? 5 : 4, 3, lane);
    put(s, b, (uint32_t)(r - 257), 5, lane);
    put(s, b, (uint32_t)(dn - 1), 5, lane);
    put(s, b, (uint32_t)(ncl - 4), 4, lane);
    for (int k = 0; k < ncl; ++k) put(s, b, UNI(s.cl[k]), 3, lane);
    // writeBlockTables (:615-623)
    for (int k = 0; k < nm; ++k) {
        const uint32_t sym = s.msym[k];
        put(s, b, s.mcode[sym], s.ml[sym], lane);
        put(s, b, s.mbits[k], sym == 18 ? 7 : sym == 17 ? 3 : sym == 16 ? 2 : 0, lane);
    }
    maybe_drain(s, b, lane);
    return b;
}

// Stream.writeBlock (DeflatorBuffers.Stream.swift:440-709), greedy / lazy form
__device__ __attribute__((noinline)) Bits write_block(Bits b, int count, bool final, int lane)
{
    DLds &s = g_lds;
    // DeflatorMatches.trees() (:138-159)
    for (int i = lane; i < 320; i += 64) s.freq[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    for (int i = lane; i < count; i += 64) {
        const uint32_t t = s.terms[i];
        atomicAdd(&s.freq[t & 0x1ff], 1u);
        atomicAdd(&s.freq[288 + (t >> 27)], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) s.freq[256] = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    build_tree(s.freq, 286, 15, s.ll, lane);
    build_tree(s.freq + 288, 30, 15, s.dl, lane);
    b = write_tables(b, final, lane);
    // writeBlock(with:) (:626-659)
    for (int i = 0; i < count; ++i) {
        const uint32_t t = s.terms[i];
        const uint32_t sym = t & 0x1ff, dsym = t >> 27;
        put(s, b, s.lcode[sym], s.ll[sym], lane);
        if (sym > 256) {
            put(s, b, (t >> 9) & 0x1f, run_extra_bits(sym & 0xff), lane);
            put(s, b, s.dcode[dsym], s.dl[dsym], lane);
            put(s, b, (t >> 14) & 0x1fff, dist_extra_bits(dsym), lane);
        }
        if ((i & 255) == 255) maybe_drain(s, b, lane);
    }
    put(s, b, s.lcode[256], s.ll[256], lane);
    maybe_drain(s, b, lane);
    return b;
}

__device__ __forceinline__ uint32_t load32(const gbyte *p) { return ((const gU32u *)p)->v; }

// bytes of position q.. and p.. agree for how many bytes (<= limit)?  Dword-wise from the input.
__device__ __forceinline__ uint32_t common_prefix(const gbyte *in, uint64_t q, uint64_t p, uint32_t limit)
{
    uint32_t i = 0;
    while (i + 4 <= limit) {
        const uint32_t x = load32(in + q + i) ^ load32(in + p + i);
        if (x) return i + (__builtin_ctz(x) >> 3);
        i += 4;
    }
    while (i < limit && in[q + i] == in[p + i]) ++i;
    return i;
}

#ifdef SPNG_DEFLATE_PROF
#define DPROF_DECL uint64_t pt[6] = {0,0,0,0,0,0}, p_t0 = 0;
#define DPROF_BEGIN() p_t0 = __builtin_readcyclecounter()
#define DPROF_END(k) pt[k] += __builtin_readcyclecounter() - p_t0
#else
#define DPROF_DECL
#define DPROF_BEGIN()
#define DPROF_END(k)
#endif

__global__ __launch_bounds__(64) void deflate_kernel(const DeflateJob *__restrict__ jobs,
                                                     spng_result *__restrict__ results)
{
    DLds &s = g_lds;
    const DeflateJob *jp = jobs + blockIdx.x;
    const int lane = threadIdx.x;
    // job fields are wave-uniform: pinned to scalar registers, typed as global memory
    const gbyte *in = (const gbyte *)uni64((uint64_t)jp->src);
    const uint64_t n = uni64(jp->src_len);
    gword *ring = (gword *)uni64((uint64_t)jp->ring);          // 65536 links: distance | tag << 16
    struct { gbyte *dst; uint64_t dst_cap; int32_t format, level; uint32_t image; } job = {
        (gbyte *)uni64((uint64_t)jp->dst), uni64(jp->dst_cap), (int32_t)UNI(jp->format), (int32_t)UNI(jp->level), UNI(jp->image) };

    // DeflatorSearch.init(level:) (:13-35), greedy and lazy rows
    const int level = job.level < 0 ? 0 : job.level;
    const bool lazy = level >= 4;
    // (packed constants: run-time indexed local arrays would live in scratch memory)
    const int lv = level & 7;
    const int attempts = lv == 0 ? 1 : lv == 1 ? 2 : lv == 2 ? 4 : lv == 3 ? 40 : lv == 4 ? 20 : lv == 5 ? 40 : lv == 6 ? 64 : 100;
    const int goal = lv == 0 ? 6 : lv == 1 ? 8 : lv == 2 ? 10 : lv == 3 ? 24 : lv == 4 ? 32 : lv == 5 ? 54 : lv == 6 ? 80 : 160;

    Bits b = {0, 0, 0, 0, job.dst, job.dst_cap, false};
    const uint32_t wmask = (1u << UNI(jp->exponent)) - 1;    // window 2^exponent (LZ77.Deflator(exponent:); PNG: 15)
    if (job.format == SPNG_FORMAT_ZLIB) {
        // StreamHeader.write (StreamHeader.swift:56-62)
        const uint32_t unpaired = (UNI(jp->exponent) - 8) << 4 | 0x08;
        const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
        put(s, b, check << 8 | unpaired, 16, lane);
    }
    for (int i = lane; i <= (1 << HBITS); i += 64) s.head[i] = NONE;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    DPROF_DECL
    uint32_t accS = 0, accI = 0;                               // Adler-32 of the input, as in inflate.hip
    int count = 0;                                             // queued terms
    const int limit_terms = 2048;
    auto unfilled = [&]() { return limit_terms - 1 - count; };

    if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434)
        put(s, b, 1, 3, lane);
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, (uint32_t)n, 16, lane); put(s, b, ~(uint32_t)n & 0xffff, 16, lane);
        for (uint64_t k = 0; k < n; ++k) put(s, b, in[k], 8, lane);
        if ((uint64_t)lane < n) { accS = in[lane]; accI = (uint32_t)lane * in[lane]; }
    } else {
        const uint64_t last_main = n - 4 + 1;                  // positions 0 .. n-4 are searched
        uint64_t inserted = 0;                                 // positions < inserted are in the window
        uint64_t w = 0;                                        // parse position
        auto insert_upto = [&](uint64_t target) {
            while (inserted < target && inserted < n) {
                const uint64_t p = inserted + lane;
                const bool live = p + 4 <= n;                  // the last three positions never start a match
                uint32_t key = 0;
                if (live) key = load32(in + p);
                else for (int k = 0; k < 4; ++k) if (p + k < n) key |= (uint32_t)in[p + k] << (8 * k);
                if (p < n) {                                   // Adler-32 accumulators
                    const uint32_t byte = key & 0xff;
                    accS += byte;
                    accI = (accI + (uint32_t)(p % 65521) * byte) % 65521;
                }
                const uint32_t mix = key * 0x9E3779B1u;
                const uint32_t h = live ? mix >> (32 - HBITS) : 0xffffffffu - lane;
                const uint32_t tag = (mix >> 3) & 0xffffu;
                uint32_t prev = live ? s.head[h & ((1 << HBITS) - 1)] : NONE;
                bool later = false;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
                    const bool same = hj == h;
                    prev = (same && j < lane) ? (uint32_t)(inserted + j) : prev;
                    later |= same && j > lane;
                }
                uint32_t dist = 0;
                if (live && prev != NONE) {
                    const uint64_t d = (uint32_t)((uint32_t)p - prev);
                    dist = d <= 32767 ? (uint32_t)d : 0;
                }
                if (p < n) ring[p & 65535] = dist | tag << 16;
                s.head[live && !later ? h & ((1 << HBITS) - 1) : 1 << HBITS] = (uint32_t)p;   // (idle lanes: the spare slot)
                inserted = uni64(inserted + 64);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            }
        };

        while (w < last_main) {
            // keep the window filled well ahead of the 64 positions searched now (+258 of look-ahead
            // is irrelevant for insertion: links only point backwards)
            DPROF_BEGIN();
            insert_upto(w + 128);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // our own ring stores, before we read them back
            DPROF_END(0); DPROF_BEGIN();
            // ---- match search: lane i answers window.match(from: w + i)
            const uint64_t p = w + lane;
            uint32_t best_run = 5, best_dist = 1;
            if (p < last_main) {
                const uint32_t limit = n - p < 258 ? (uint32_t)(n - p) : 258u;
                const uint32_t mine = ring[p & 65535];
                const uint32_t tag = mine >> 16;
                uint32_t d = mine & 0xffff, acc = 0;
                int remaining = attempts;
                bool first = true;
                while (d) {
                    acc += d;
                    if (acc > wmask || (!first && acc >= wmask)) break;
                    const uint32_t e = ring[(p - acc) & 65535];
                    if ((e >> 16) == tag && load32(in + p - acc) == load32(in + p)) {
                        // LZ77.DeflatorWindow.match (:145-208): run, then the stop rules
                        const uint32_t run = common_prefix(in, p - acc, p, limit);
                        if (best_run < run) { best_run = run; best_dist = acc; }
                        first = false;
                        remaining -= 1;
                        if (!(remaining > 0 && goal > (int)run)) break;
                    }
                    d = e & 0xffff;
                }
            }
            const uint32_t mrun = best_run > 5 ? best_run : 0;   // 0: no match (run must exceed 5, :129)
            const uint32_t mylit = p < n ? in[p] : 0u;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DPROF_END(1); DPROF_BEGIN();

            // ---- the parse: Stream.compress greedy (:209-252) / lazy (:268-323) over these 64 answers
            uint32_t t = 0;
            bool stop = false;
            while (t < 64 && w + t < last_main && !stop) {
                if (!(unfilled() > (lazy ? 1 : 0))) { DPROF_END(2); DPROF_BEGIN(); b = write_block(b, count, false, lane); count = 0; DPROF_END(3); DPROF_BEGIN(); }
                const uint32_t run = (uint32_t)__builtin_amdgcn_readlane((int)mrun, (int)t);
                const uint32_t lit = (uint32_t)__builtin_amdgcn_readlane((int)mylit, (int)t);
                if (!run) { s.terms[count] = 0xf8000000u | lit; ++count; t += 1; continue; }
                uint32_t use_run = run, use_dist = (uint32_t)__builtin_amdgcn_readlane((int)best_dist, (int)t);
                uint32_t adv = run;
                if (lazy) {
                    // the answer for position w+t+1 is needed: restart the search there if it is not in this batch
                    if (t + 1 >= 64) { stop = true; break; }
                    // lazy match at a+1 (:293-299); it exists only if that position is still searched
                    const uint32_t lrun = (w + t + 1 < last_main) ? (uint32_t)__builtin_amdgcn_readlane((int)mrun, (int)(t + 1)) : 0u;
                    if (lrun > run) {
                        s.terms[count] = 0xf8000000u | lit;
                        ++count;
                        use_run = lrun; use_dist = (uint32_t)__builtin_amdgcn_readlane((int)best_dist, (int)(t + 1));
                        adv = 1 + lrun;
                    }
                }
                // LZ77.DeflatorTerm.init(run:distance:) (DeflatorTerm.swift:34-56)
                const uint32_t rd = run_decade(use_run), dd = dist_decade(use_dist);
                s.terms[count] = dd << 27 | 0x100u | rd | dist_extra_value(use_dist, dd) << 14 | run_extra_value(use_run, rd) << 9;
                ++count;
                t += adv;
            }
            w = uni64(w + t);
            DPROF_END(2);
        }
        insert_upto(n);                                        // Adler-32 over the tail
        // epilogue: the positions still in the window pipeline become literals (:254-265, :331-342)
        for (uint64_t p = w; p < n; ++p) {
            if (!(unfilled() > 0)) { b = write_block(b, count, false, lane); count = 0; }
            s.terms[count] = 0xf8000000u | UNI(in[p]);
            ++count;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        b = write_block(b, count, true, lane);
    }

    if (job.format == SPNG_FORMAT_ZLIB) {
        // Adler-32 (see inflate.hip): s1 = 1 + S, s2 = N + N*S - I
        uint32_t S = accS % 65521, I = accI % 65521;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
        S %= 65521; I %= 65521;
        const uint32_t N = (uint32_t)(n % 65521);
        const uint32_t sum = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, sum >> 24, 8, lane); put(s, b, (sum >> 16) & 0xff, 8, lane);
        put(s, b, (sum >> 8) & 0xff, 8, lane); put(s, b, sum & 0xff, 8, lane);
    }
    if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);                // DeflatorOut.pull flushes padding bits
    drain(s, b, b.total, lane);
#ifdef SPNG_DEFLATE_PROF
    if (lane == 0 && blockIdx.x == 0) printf("deflate prof cycles: insert %llu search %llu parse %llu write_block %llu\n", pt[0], pt[1], pt[2], pt[3]);
#endif
    if (lane == 0) {
        spng_result &res = results[job.image];
        res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
        res.written = b.total; res.consumed = n; res.aux[0] = res.aux[1] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// levels >= 8: the shortest-path ("full") search
// ------------------------------------------------------------------------------------------------
// The reference makes every input position a vertex that knows, per distance decade, the longest run
// some candidate offers (LZ77.DeflatorMatches.set(edge:), :183-194), closes a block after limit - 1
// vertices (the limit doubles per block, 2048 ... 2^21) and finds the cheapest path through the block
// with per-symbol costs in quarter bits that it re-derives from the trees of the previous pass
// (trees(iterations:), :225-260; minimize / explore, :262-379; Depths, ...Depths.swift:31-98).
// Here, per stream (one wave):
//   * candidates: the 64-positions-at-a-time chain walk of the greedy / lazy kernel, every candidate
//     recorded (per lane, thirty decade slots in LDS); the serial part is only deciding which
//     positions are searched at all (behind a run > 100 the next run - 100 vertices get no edges,
//     DeflatorBuffers.Stream.swift:376-380).  Vertices live in HBM: 30 slots each.
//   * forward pass: vertices in order, 64 at a time through LDS; the best depth / incoming edge of the
//     next 258 vertices sit in an LDS ring; one vertex relaxes its literal edge, then decade after
//     decade (ascending, as the reference: ties go to the first writer) with the lanes spread over the
//     run lengths 3 ... maxlen.
//   * back-trace: 2048 vertices at a time from the end; which vertices lie on the path is found by
//     pointer doubling over the chunk instead of a serial walk; path vertices tally the symbol
//     frequencies and hand their edge to the vertex it starts from.
//   * trees, cost update, repeat (2 x iterations passes for the first block, iterations after);
//     then the block is written walking the path forwards.
struct FullArrays {
    gword *slots;        // [vertex][30]: distance << 16 | longest run of that distance decade
    gword *up;           // [vertex]: incoming edge of the cheapest path, run << 16 | decade << 8 (literal: 1 << 16 | 0xff00)
    gword *step;         // [vertex]: the path's edge that STARTS here (set by the back-trace)
    gbyte *pathb;        // [vertex]: on the path
};

__device__ __forceinline__ uint32_t run_base(uint32_t dec)       // LZ77.Composites.swift:25-63
{
    if (dec < 9) return dec + 2;
    if (dec == 29) return 258;
    const uint32_t e = (dec - 5) >> 2;
    return ((4 + ((dec - 9) & 3)) << e) + 3;
}
__device__ __forceinline__ uint32_t depth_default(uint32_t i)    // Depths.default (Depths.swift:31-44)
{
    return i < 256 ? 33u : i < 512 ? 30u + 4 * run_extra_bits(run_decade(i - 253)) : 19u + 4 * dist_extra_bits(i - 512);
}

// minimize() forwards (:262-280, explore :322-379): best depth and incoming edge of every vertex
__device__ __attribute__((noinline)) void full_forward(const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count, int lane)
{
    DLds &s = g_lds;
    if (lane == 0) { s.win_depth[0] = 0; s.win_up[0] = 0; }
    uint32_t inited = 1;
    for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        const uint32_t need = (b0 + 64 + 258 < count ? b0 + 64 + 258 : count) + 1;
        for (uint32_t j = inited + lane; j < need; j += 64) { s.win_depth[j & 511] = 0xffffffffu; s.win_up[j & 511] = 0; }
        inited = inited > need ? inited : need;
        const uint32_t nv = count - b0 < 64 ? count - b0 : 64;
        for (uint32_t i = lane; i < nv * 30; i += 64) s.batch[i] = g.slots[(uint64_t)b0 * 30 + i];
        const uint32_t lit = (uint32_t)lane < nv ? in[bbase + b0 + lane] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        for (uint32_t k = 0; k < nv; ++k) {
            const uint32_t v = b0 + k;
            const uint32_t cur = UNI(s.win_depth[v & 511]);
            const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lit, (int)k);
            const uint32_t ld = cur + UNI(s.depths[l]);
            const uint32_t t1 = (v + 1) & 511;
            if (ld < UNI(s.win_depth[t1])) { s.win_depth[t1] = ld; s.win_up[t1] = 0x0001ff00u; }   // (every lane, same values)
            const uint32_t rem = count - v;
            if (rem >= 3) {
                const uint32_t mine = lane < 30 ? s.batch[k * 30 + lane] : 0u;
                const uint32_t run = mine & 0xffff;
                unsigned long long m = __ballot(run > 0);
                while (m) {
                    const int dec = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)run, dec);
                    const uint32_t maxlen = r < rem ? r : rem;
                    const uint32_t base = cur + UNI(s.depths[512 + dec]);
                    for (uint32_t L = 3 + lane; L <= maxlen; L += 64) {
                        const uint32_t dd = base + s.depths[253 + L];
                        const uint32_t tt = (v + L) & 511;
                        if (dd < s.win_depth[tt]) { s.win_depth[tt] = dd; s.win_up[tt] = L << 16 | (uint32_t)dec << 8; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                }
            }
        }
        if ((uint32_t)lane < nv) g.up[b0 + 1 + lane] = s.win_up[(b0 + 1 + lane) & 511];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// minimize() backwards (:282-320): the path from the last vertex to the first, symbol frequencies
// into s.freq, every path edge handed to the vertex it starts from
__device__ __attribute__((noinline)) void full_backward(const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count, int lane)
{
    DLds &s = g_lds;
    for (int i = lane; i < 320; i += 64) s.freq[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    uint32_t cur = count;
    while (cur > 0) {
        const uint32_t hi = cur, lo = hi >= CHV ? hi - CHV + 1 : 0, nn = hi - lo + 1;
        for (uint32_t k = lane; k < nn; k += 64) {
            const uint32_t c = lo + k;
            const uint32_t u = c ? g.up[c] : 0u;
            s.upc[k] = u;
            const uint32_t len = u >> 16;
            s.jump[0][k] = (c == 0 || len > k) ? (uint16_t)0xffff : (uint16_t)(k - len);
            s.onpath[k] = k == nn - 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        int cb = 0;
        for (int r = 0; r < 11; ++r) {                         // 2^11 = CHV
            for (uint32_t k = lane; k < nn; k += 64) {
                const uint32_t j = s.jump[cb][k];
                if (s.onpath[k] && j != 0xffff) s.onpath[j] = 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            for (uint32_t k = lane; k < nn; k += 64) {
                const uint32_t j = s.jump[cb][k];
                s.jump[cb ^ 1][k] = j == 0xffff ? (uint16_t)0xffff : s.jump[cb][j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            cb ^= 1;
        }
        uint32_t exit_to = 0;
        for (uint32_t k = lane; k - lane < nn; k += 64) {      // (uniform trip count: ballot inside)
            const bool in_chunk = k < nn;
            const uint32_t c = lo + k;
            const bool on = in_chunk && s.onpath[k];
            if (in_chunk && c < count) g.pathb[c] = on ? 1 : 0;
            const uint32_t u = in_chunk ? s.upc[k] : 0u;
            const uint32_t len = u >> 16;
            const bool hop = on && c > 0;
            if (hop) {
                const uint32_t nxt = c - len;
                g.step[nxt] = u & 0xffffff00u;
                if (len == 1) atomicAdd(&s.freq[in[bbase + nxt]], 1u);
                else { atomicAdd(&s.freq[256 | run_decade(len)], 1u); atomicAdd(&s.freq[288 + ((u >> 8) & 0xff)], 1u); }
            }
            const unsigned long long em = __ballot(hop && len > k);            // the hop that leaves the chunk
            if (em) exit_to = lo + k - lane + (__ffsll((long long)em) - 1) - (uint32_t)__shfl((int)len, __ffsll((long long)em) - 1, 64);
        }
        exit_to = UNI(exit_to);
        // vertices the leaving hop jumped over are not on the path
        if (lo > 0) for (uint32_t c = exit_to + 1 + lane; c < lo; c += 64) g.pathb[c] = 0;
        cur = lo == 0 ? 0 : exit_to;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    }
    if (lane == 0) s.freq[256] = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Depths.update (Depths.swift:53-86): the cost of a symbol = the length of its code (+ extra bits) in
// quarter bits; symbols without a code keep their cost.  The reference writes in (code length, symbol)
// order and run 258 belongs to two symbols (284 with extra bits 31, and 285): the later write wins.
__device__ __forceinline__ void full_depths_update(int lane)
{
    DLds &s = g_lds;
    for (uint32_t sym = lane; sym < 286; sym += 64) {
        const uint32_t len = s.ll[sym];
        if (!len) continue;
        if (sym < 256) s.depths[sym] = (uint8_t)(len << 2);
        else if (sym > 256) {
            const uint32_t dec = sym & 0xff, e = run_extra_bits(dec), base = 253 + run_base(dec);
            for (uint32_t l = base; l < base + (1u << e); ++l) if (l != 253 + 258 || dec == 29) s.depths[l] = (uint8_t)((len + e) << 2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) {
        const uint32_t a = s.ll[284], c = s.ll[285];
        if (a && (!c || a > c)) s.depths[253 + 258] = (uint8_t)((a + 5) << 2);
        else if (c) s.depths[253 + 258] = (uint8_t)(c << 2);
    }
    if (lane < 30 && s.dl[lane]) s.depths[512 + lane] = (uint8_t)((s.dl[lane] + dist_extra_bits(lane)) << 2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// Stream.writeBlock (DeflatorBuffers.Stream.swift:440-709), full form: trees(iterations:), header, the path's tokens
__device__ __attribute__((noinline)) Bits full_block(Bits b, const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count,
                                                     bool final, int iterations, bool generic, int lane)
{
    DLds &s = g_lds;
    for (int i = generic ? -iterations : 0;;) {
        if (count) { full_forward(g, in, bbase, count, lane); full_backward(g, in, bbase, count, lane); }
        else {
            for (int k = lane; k < 320; k += 64) s.freq[k] = k == 256 ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        }
        build_tree(s.freq, 286, 15, s.ll, lane);
        build_tree(s.freq + 288, 30, 15, s.dl, lane);
        ++i;
        if (!(i < iterations)) break;
        full_depths_update(lane);
    }
    b = write_tables(b, final, lane);
    // writeBlock(with:) (:661-707): walk the path
    for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        const uint32_t v = b0 + lane;
        const bool on = v < count && g.pathb[v] != 0;
        const uint32_t st = on ? g.step[v] : 0u;
        const uint32_t cnt = st >> 16, dd = (st >> 8) & 0xff;
        const uint32_t lit = on ? in[bbase + v] : 0u;
        const uint32_t off = (on && cnt > 1) ? g.slots[(uint64_t)v * 30 + dd] >> 16 : 0u;
        unsigned long long m = __ballot(on);
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, k);
            if (c == 1) {
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lit, k);
                put(s, b, s.lcode[l], s.ll[l], lane);
            } else {
                const uint32_t rd = run_decade(c), d2 = (uint32_t)__builtin_amdgcn_readlane((int)dd, k);
                const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, k);
                put(s, b, s.lcode[256 | rd], s.ll[256 | rd], lane);
                put(s, b, run_extra_value(c, rd), run_extra_bits(rd), lane);
                put(s, b, s.dcode[d2], s.dl[d2], lane);
                put(s, b, dist_extra_value(o, d2), dist_extra_bits(d2), lane);
            }
        }
        maybe_drain(s, b, lane);
    }
    put(s, b, s.lcode[256], s.ll[256], lane);
    maybe_drain(s, b, lane);
    // resetGraph -> Depths.generalize (Depths.swift:88-98)
    for (uint32_t i = lane; i < 542; i += 64) {
        const uint32_t x = s.depths[i], d = depth_default(i);
        s.depths[i] = (uint8_t)((x & d) + ((x ^ d) >> 1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    return b;
}

__global__ __launch_bounds__(64) void deflate_full_kernel(const DeflateJob *__restrict__ jobs, spng_result *__restrict__ results)
{
    DLds &s = g_lds;
    const DeflateJob *jp = jobs + blockIdx.x;
    const int lane = threadIdx.x;
    const gbyte *in = (const gbyte *)uni64((uint64_t)jp->src);
    const uint64_t n = uni64(jp->src_len);
    gword *ring = (gword *)uni64((uint64_t)jp->ring);
    struct { gbyte *dst; uint64_t dst_cap; int32_t format, level; uint32_t image; } job = {
        (gbyte *)uni64((uint64_t)jp->dst), uni64(jp->dst_cap), (int32_t)UNI(jp->format), (int32_t)UNI(jp->level), UNI(jp->image) };
    const uint32_t vcap = UNI(jp->graph_vertices);             // vertices the scratch arrays hold
    FullArrays g;
    {
        gword *base = (gword *)uni64((uint64_t)jp->graph);
        g.slots = base; g.up = base + (uint64_t)vcap * 30; g.step = g.up + vcap + 1; g.pathb = (gbyte *)(g.step + vcap + 1);
    }
    // DeflatorSearch.init(level:) (:13-35), full rows
    const int lv = job.level > 13 ? 13 : job.level;
    const int attempts = lv == 8 ? 14 : lv == 9 ? 20 : lv == 10 ? 30 : lv == 11 ? 60 : lv == 12 ? 100 : 0x7fffffff;
    const int goal = lv == 8 ? 20 : lv == 9 ? 32 : lv == 10 ? 50 : lv == 11 ? 80 : lv == 12 ? 133 : 258;
    const int iterations = lv - 7;
    const uint32_t wmask = (1u << UNI(jp->exponent)) - 1;

    Bits b = {0, 0, 0, 0, job.dst, job.dst_cap, false};
    if (job.format == SPNG_FORMAT_ZLIB) {
        // StreamHeader.write (StreamHeader.swift:56-62)
        const uint32_t unpaired = (UNI(jp->exponent) - 8) << 4 | 0x08;
        const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
        put(s, b, check << 8 | unpaired, 16, lane);
    }
    for (int i = lane; i <= (1 << HBITS); i += 64) s.head[i] = NONE;
    for (uint32_t i = lane; i < 542; i += 64) s.depths[i] = (uint8_t)depth_default(i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    uint32_t accS = 0, accI = 0;
    uint32_t count = 0, limit = 2048;
    bool generic = true;
    uint64_t bbase = 0;
    auto unfilled = [&]() { return (int)limit - 1 - (int)count; };
    auto close_block = [&](bool final) {
        const uint32_t doubled = 2 * limit < (1u << 21) ? 2 * limit : 1u << 21;     // trees(iterations:) :229
        b = full_block(b, g, in, bbase, count, final, iterations, generic, lane);
        generic = false; count = 0; limit = doubled < vcap + 1 ? doubled : vcap + 1;
    };

    if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434)
        put(s, b, 1, 3, lane);
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, (uint32_t)n, 16, lane); put(s, b, ~(uint32_t)n & 0xffff, 16, lane);
        for (uint64_t k = 0; k < n; ++k) put(s, b, in[k], 8, lane);
        if ((uint64_t)lane < n) { accS = in[lane]; accI = (uint32_t)lane * in[lane]; }
    } else {
        const uint64_t last_main = n - 4 + 1;
        uint64_t inserted = 0, w = 0;
        auto insert_upto = [&](uint64_t target) {
            while (inserted < target && inserted < n) {
                const uint64_t p = inserted + lane;
                const bool live = p + 4 <= n;
                uint32_t key = 0;
                if (live) key = load32(in + p);
                else for (int k = 0; k < 4; ++k) if (p + k < n) key |= (uint32_t)in[p + k] << (8 * k);
                if (p < n) {
                    const uint32_t byte = key & 0xff;
                    accS += byte;
                    accI = (accI + (uint32_t)(p % 65521) * byte) % 65521;
                }
                const uint32_t mix = key * 0x9E3779B1u;
                const uint32_t h = live ? mix >> (32 - HBITS) : 0xffffffffu - lane;
                const uint32_t tag = (mix >> 3) & 0xffffu;
                uint32_t prev = live ? s.head[h & ((1 << HBITS) - 1)] : NONE;
                bool later = false;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    const uint32_t hj = (uint32_t)__builtin_amdgcn_readlane((int)h, j);
                    const bool same = hj == h;
                    prev = (same && j < lane) ? (uint32_t)(inserted + j) : prev;
                    later |= same && j > lane;
                }
                uint32_t dist = 0;
                if (live && prev != NONE) {
                    const uint64_t d = (uint32_t)((uint32_t)p - prev);
                    dist = d <= 32767 ? (uint32_t)d : 0;
                }
                if (p < n) ring[p & 65535] = dist | tag << 16;
                s.head[live && !later ? h & ((1 << HBITS) - 1) : 1 << HBITS] = (uint32_t)p;
                inserted = uni64(inserted + 64);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            }
        };
        while (w < last_main) {
            insert_upto(w + 128);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- every candidate of position w + lane becomes an edge (DeflatorWindow.match, :132-212)
            const uint64_t p = w + lane;
#pragma unroll
            for (int d = 0; d < 30; ++d) s.cslot[d * 64 + lane] = 0;
            uint32_t extent = 1;
            if (p < last_main) {
                const uint32_t limit_run = n - p < 258 ? (uint32_t)(n - p) : 258u;
                const uint32_t mine = ring[p & 65535];
                const uint32_t tag = mine >> 16;
                uint32_t d = mine & 0xffff, acc = 0;
                int remaining = attempts;
                bool first = true;
                while (d) {
                    acc += d;
                    if (acc > wmask || (!first && acc >= wmask)) break;
                    const uint32_t e = ring[(p - acc) & 65535];
                    if ((e >> 16) == tag && load32(in + p - acc) == load32(in + p)) {
                        const uint32_t run = common_prefix(in, p - acc, p, limit_run);
                        extent = run > extent ? run : extent;
                        const uint32_t at = dist_decade(acc) * 64 + lane;
                        if (run > (s.cslot[at] & 0xffff)) s.cslot[at] = acc << 16 | run;
                        first = false;
                        remaining -= 1;
                        if (!(remaining > 0 && goal > (int)run)) break;
                    }
                    d = e & 0xffff;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            // ---- which positions are vertices with edges (Stream.compress full, :344-400)
            uint32_t t = 0;
            while (t < 64 && w + t < last_main) {
                if (!(unfilled() > 0)) close_block(false);
                if (count == 0) bbase = w + t;
                if (lane < 30) g.slots[(uint64_t)count * 30 + lane] = s.cslot[lane * 64 + t];
                count += 1;
                const int ext = __builtin_amdgcn_readlane((int)extent, (int)t);
                int skip = ext - 100 < unfilled() ? ext - 100 : unfilled();
                if (skip > 0) {
                    for (uint32_t i = lane; i < (uint32_t)skip * 30; i += 64) g.slots[(uint64_t)count * 30 + i] = 0;
                    count += (uint32_t)skip;
                } else skip = 0;
                t += 1 + (uint32_t)skip;
            }
            w = uni64(w + t);
        }
        insert_upto(n);
        // epilogue: the three positions still in the window pipeline (:254-265)
        for (uint64_t p = w; p < n; ++p) {
            if (!(unfilled() > 0)) close_block(false);
            if (count == 0) bbase = p;
            if (lane < 30) g.slots[(uint64_t)count * 30 + lane] = 0;
            count += 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        close_block(true);
    }

    if (job.format == SPNG_FORMAT_ZLIB) {
        uint32_t S = accS % 65521, I = accI % 65521;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
        S %= 65521; I %= 65521;
        const uint32_t N = (uint32_t)(n % 65521);
        const uint32_t sum = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, sum >> 24, 8, lane); put(s, b, (sum >> 16) & 0xff, 8, lane);
        put(s, b, (sum >> 8) & 0xff, 8, lane); put(s, b, sum & 0xff, 8, lane);
    }
    if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
    drain(s, b, b.total, lane);
    if (lane == 0) {
        spng_result &res = results[job.image];
        res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
        res.written = b.total; res.consumed = n; res.aux[0] = res.aux[1] = 0;
    }
}

// bytes of graph scratch a stream of n bytes needs at levels >= 8 (api.hip sizes the slab with it)
uint64_t deflate_graph_vertices(uint64_t n)
{
    const uint64_t cap = (1u << 21) - 1;
    return n + 2 < cap ? n + 2 : cap;
}
uint64_t deflate_graph_bytes(uint64_t vertices)
{
    return ((vertices + 1) * (30 * 4 + 4 + 4 + 1) + 1024 + 255) & ~(uint64_t)255;
}

hipError_t launch_deflate_full(const DeflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    deflate_full_kernel<<<count, 64, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

hipError_t launch_deflate(const DeflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    deflate_kernel<<<count, 64, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

}  // namespace spng
