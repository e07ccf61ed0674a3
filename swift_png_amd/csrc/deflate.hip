// deflate.hip -- batched DEFLATE / zlib compression for gfx950, bit-exact with swift-png's
// LZ77.Deflator at every level: greedy (0-3), lazy (4-7) and the shortest-path search (8 and up);
// one wavefront per stream (plus three helper waves in the forward pass of the search on compressible input).
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
// Only the parse (which positions are asked) is sequential.  So a stream's wave works in layers:
//   * hash insertion, 64 positions per step: every lane hashes its key; its link is the distance to
//     the nearest lower lane with the same bucket (radix match over the hash bits, one ballot per bit),
//     else to the bucket head in LDS; links (+ a 16-bit key tag) go to a 64 K-entry ring in HBM.
//     Insertion runs ahead of the parse -- later positions never appear in an earlier position's
//     chain, which only walks backwards;
//   * match search, 128 positions per step, two per lane (chain_walk2): both chains hop together and the
//     candidate's first four bytes are fetched speculatively with the link, so the latency of a hop is
//     paid once per pair;
//   * the parse itself walks those answers on the scalar unit (readlane per token) with the
//     reference's greedy / lazy rules, packing terms exactly like LZ77.DeflatorTerm.
// When 2047 terms are queued (lazy: 2046/2047) the block is written: histogram with LDS atomics,
// the reference's heap-based length-limited Huffman construction (ranked in parallel, merged on
// one lane because its tie-breaking is order dependent; depths by pointer jumping), canonical codes
// by ballot, code-length RLE, and the bits of 64 terms at a time (prefix sum of their lengths, ds_or
// into a zero-initialised staging ring).  Levels >= 8: see the second half of this file.
#include "common.hpp"
#include "huffman.hpp"      // UNI / uni64, WSYNC, DPP scans

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U32u { uint32_t v; };
struct __attribute__((packed)) U128u { u32x4 v; };
// Input, output and the link ring are global memory, and say so in their types (a generic pointer
// costs flat instructions, which also tie up the LDS counter).
typedef uint32_t __attribute__((address_space(1))) gword;
typedef U32u __attribute__((address_space(1))) gU32u;

static constexpr int HBITS = 13;                 // bucket heads in LDS
static constexpr uint32_t NONE = 0xffffffffu;
static constexpr int OUTB = 2048;                // output staging ring (bytes, power of two; at most half of it + the ~600 bytes of a block's tables stand undrained)

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

struct DLdsFull {                                // the full search (levels >= 8) only
    // forward pass: the best way into each of the next vertices found so far, as one 64-bit key (depth << 32 | writer order:
    // see full_forward)
    uint64_t win[512];
    uint8_t  depths[544];                        // LZ77.DeflatorMatches.Depths: cost of every symbol in quarter bits
};
struct DLds {                                    // what every kernel of this file that writes blocks needs
    uint32_t freq[320];                          // 0..287 lit/len, 288..319 distance
    union { uint8_t out[OUTB]; uint32_t out32[OUTB / 4]; };   // output staging ring; bytes not written yet are zero
    // Huffman scratch (one tree at a time)
    uint16_t order[288];                         // symbols by descending frequency (stable)
    alignas(16) uint64_t heap[292];              // heap entries (1-based): key << 32 | node id; afterwards ancestor / depth of every node
    uint16_t parent[576];                        // tree nodes: leaves 0..m-1 (in `order`), then merges
    uint32_t depthcnt[300];                      // leaves per depth
    uint8_t  ll[288], dl[32], ml[19];            // code lengths
    uint16_t lcode[288], dcode[32], mcode[19];   // bit-reversed codewords
    uint8_t  msym[320], mbits[320];              // code-length RLE terms
    uint8_t  cl[20];                             // code-length-code lengths in transmission order
};
struct DLdsSearch {                              // the match search of the one-kernel forms (greedy / lazy, full)
    uint32_t head[(1 << HBITS) + 1];             // most recent position per bucket (low 32 bits); + a slot nobody reads
    uint32_t cslot[2][30 * 64];                  // full: per lane and half, the best run of every distance decade (distance << 16 | run)
};
struct DLdsTerms {                               // greedy / lazy: the queued terms (of their own: dfl3_parse_kernel has no search in it)
    uint32_t terms[2][2048];                     // two blocks' worth: dfl3_parse_kernel's parser fills one while its writer emits the other
    uint32_t cmd[2];                             // parser -> writer, per buffer: 0 free, else D3_CMD_* | terms
    uint64_t fin_w;                              // the parse position a SAVE / NEED_MORE command carries
};
struct DLdsOld {                                 // the one-kernel full search (deflate_full_kernel)
    uint32_t batch[64 * 30];                     // the edge slots of the 64 vertices at hand
    // the full kernel's helper waves (forward pass): what wave 0 hands them per pass and per batch of 64 vertices
    // (per-batch words twice: a helper may still be reading one batch's while wave 0 sets up the next)
    uint64_t fw_bbase, fw_ems[2][16];            // which vertices of the next sixteen batches have edges
    uint32_t fw_cmd, fw_count, fw_carry[2];
    uint32_t fw_cin[2][64];
};
// One instance per workgroup, at namespace scope so that the (non-inlined) block writer reaches it
// with LDS instructions instead of through a generic pointer.
__shared__ __attribute__((aligned(16))) DLds g_lds;
__shared__ __attribute__((aligned(16))) DLdsFull g_full;
__shared__ __attribute__((aligned(16))) DLdsSearch g_sea;
__shared__ __attribute__((aligned(16))) DLdsTerms g_trm;
__shared__ __attribute__((aligned(16))) DLdsOld g_old;

struct Bits {                                    // LSB-first bit writer (LZ77.DeflatorOut.append)
    uint64_t acc; uint32_t nacc;
    uint64_t total;                              // bytes produced so far
    uint64_t flushed;
    gbyte *dst; uint64_t cap; bool overflow;
};

// Arguments of a function that is not inlined travel in vector registers, and whatever is computed from them -- loop counters,
// masks, branch conditions -- stays there: a single wave then pays a VALU slot and an EXEC detour for what is scalar work
// (and a wave alone on its SIMD issues one instruction every four cycles, scalar or not: the parse is bound by their number).
// These put wave-uniform arguments back into scalar registers at the callee's door.
#define UNIP(T, p) ((T)uni64((uint64_t)(p)))
__device__ __forceinline__ Bits uni_bits(Bits b)
{
    b.acc = uni64(b.acc); b.nacc = UNI(b.nacc); b.total = uni64(b.total); b.flushed = uni64(b.flushed);
    b.dst = UNIP(gbyte *, b.dst); b.cap = uni64(b.cap); b.overflow = UB(b.overflow);
    return b;
}

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
        s.out[i & (OUTB - 1)] = 0;                              // (the ring ahead of the writer is all zero: bulk_put ORs into it)
    }
    if (upto > b.cap) b.overflow = true;
    b.flushed = upto;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}
__device__ __forceinline__ void maybe_drain(DLds &s, Bits &b, int lane)
{
    if (b.total - b.flushed >= OUTB / 2) drain(s, b, b.total, lane);
}

// Up to 64 bit strings at once (lane order = stream order): `n` <= 48 bits of `v` per lane, n = 0 for a lane
// without one.  Prefix sum of the lengths, then every lane ORs its bits into the staging ring (LDS atomics:
// neighbours share dwords).  At most 384 bytes per call; the caller drains in between (maybe_drain).
__device__ __forceinline__ void bulk_put(DLds &s, Bits &b, uint64_t v, uint32_t n, int lane)
{
    uint32_t tot;
    const uint32_t off = wave_excl_scan(n, tot, lane);
    if (b.nacc) s.out[b.total & (OUTB - 1)] = (uint8_t)b.acc;  // the pending bits of the serial writer join the ring
    if (n) {
        const uint64_t pos = b.total * 8 + b.nacc + off;
        const uint32_t w = (uint32_t)(pos >> 5), sh = (uint32_t)pos & 31;
        const uint64_t x = v << sh;
        atomicOr(&s.out32[w & (OUTB / 4 - 1)], (uint32_t)x);
        if (sh + n > 32) atomicOr(&s.out32[(w + 1) & (OUTB / 4 - 1)], (uint32_t)(x >> 32));
        if (sh + n > 64) atomicOr(&s.out32[(w + 2) & (OUTB / 4 - 1)], (uint32_t)(v >> (64 - sh)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    const uint64_t end = b.total * 8 + b.nacc + tot;
    b.total = end >> 3; b.nacc = (uint32_t)end & 7;
    b.acc = b.nacc ? UNI(s.out[b.total & (OUTB - 1)]) : 0;      // (the unfinished byte stays in the ring; put() overwrites it)
}

// HuffmanTree.init(frequencies:limit:) (HuffmanTree.swift:247-344) for `n` symbols with counts
// in freq[0..n): code length per symbol into len[].  The heap (LZ77.Heap.swift) is replayed
// exactly -- which two nodes merge on equal keys depends on its sift order -- but its values are
// node ids: the reference's per-level leaf-count vectors are the depth histogram of the tree.
__device__ __attribute__((noinline)) void build_tree(const uint32_t *freq_, int n_, int limit_, uint8_t *len_, int lane)
{
    const uint32_t *freq = UNIP(const uint32_t *, freq_); uint8_t *len = UNIP(uint8_t *, len_);
    const int n = (int)UNI(n_), limit = (int)UNI(limit_);
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
        // 1-based in place (hp[i], slot 0 unused): the children of i are one aligned 16-byte read at 2 i, its grandchildren
        // two at 4 i -- asked for together, so that a sift-down takes two levels per LDS round trip (this lane is alone with the
        // latency of every dependent read: the replay was ~0.9 M cycles of a 2047-term block's ~1.5 M).  A sift only ever
        // writes the slot it has just left, never one it has read ahead.
        uint64_t *hp = s.heap;
        int count = m, nodes = m;
        for (int k = 0; k < m; ++k) hp[k + 1] = (uint64_t)freq[s.order[m - 1 - k]] << 32 | (uint32_t)k;
        struct alignas(16) u64x2 { uint64_t x, y; };           // (one aligned 16-byte read)
        constexpr int HCAP = 290;                              // (reads ahead of the heap's end are clamped into the array and not looked at)
        auto sift_down = [&](int i, uint64_t moving) {         // Heap.siftDown / lowest(below:) (:94-135)
            const uint32_t key = (uint32_t)(moving >> 32);
            for (;;) {
                const int l = i << 1;
                if (l > count) break;
                const int g = l << 1;                          // first grandchild
                const u64x2 c = *(const u64x2 *)&hp[l < HCAP - 2 ? l : HCAP - 2];
                const u64x2 g0 = *(const u64x2 *)&hp[g < HCAP - 4 ? g : HCAP - 4], g1 = *(const u64x2 *)&hp[g < HCAP - 4 ? g + 2 : HCAP - 2];
                // level 1
                const int r = l + 1;
                const bool right = r <= count && (uint32_t)(c.y >> 32) < (uint32_t)(c.x >> 32);
                const uint64_t ec = right ? c.y : c.x;
                if (!((uint32_t)(ec >> 32) < key)) break;
                hp[i] = ec;
                i = right ? r : l;
                // level 2: the children of the child taken
                const int l2 = i << 1, r2 = l2 + 1;
                if (l2 > count) break;
                const uint64_t el2 = right ? g1.x : g0.x, er2 = right ? g1.y : g0.y;
                const bool right2 = r2 <= count && (uint32_t)(er2 >> 32) < (uint32_t)(el2 >> 32);
                const uint64_t ec2 = right2 ? er2 : el2;
                if (!((uint32_t)(ec2 >> 32) < key)) break;
                hp[i] = ec2;
                i = right2 ? r2 : l2;
            }
            hp[i] = moving;
        };
        auto dequeue = [&]() -> uint64_t {                     // Heap.dequeue (:149-164)
            const uint64_t top = hp[1];
            const uint64_t moving = hp[count];
            --count;
            if (count > 0) sift_down(1, moving);
            return top;
        };
        for (int i = count >> 1; i >= 1; --i) sift_down(i, hp[i]);   // heapify (:166-175)
        uint16_t root = 0;
        for (;;) {
            const uint64_t e1 = dequeue();
            if (count == 0) { root = (uint16_t)e1; break; }
            const uint64_t e2 = dequeue();
            const uint16_t id = (uint16_t)nodes++;
            s.parent[(uint16_t)e1] = id; s.parent[(uint16_t)e2] = id;
            const uint32_t key = (uint32_t)(e1 >> 32) + (uint32_t)(e2 >> 32);
            int i = ++count;                                   // enqueue + siftUp (:137-147, :113-123): parent and grandparent read together
            for (;;) {
                const int p = i >> 1, pp = i >> 2;
                if (p < 1) break;
                const uint64_t ep = hp[p], epp = hp[pp < 1 ? 1 : pp];
                if (!(key < (uint32_t)(ep >> 32))) break;
                hp[i] = ep; i = p;
                if (pp < 1) break;
                if (!(key < (uint32_t)(epp >> 32))) break;
                hp[i] = epp; i = pp;
            }
            hp[i] = (uint64_t)key << 32 | id;
        }
        s.parent[root] = root;                                 // (root == 2 m - 2: the last merge)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    // depth of every node by pointer jumping (the heap's storage is free now): leaves per depth.
    // The reference's per-level leaf-count vectors are exactly this histogram.
    const uint32_t nodes = 2 * (uint32_t)m - 1, root = nodes - 1;
    uint16_t *anc = (uint16_t *)s.heap, *dep = anc + 576;
    for (uint32_t v = lane; v < nodes; v += 64) { anc[v] = s.parent[v]; dep[v] = v == root ? 0 : 1; }
    for (int i = lane; i < 300; i += 64) s.depthcnt[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    for (int round = 0; round < 10; ++round) {
        uint32_t na[9], nd[9];
        bool moving = false;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const uint32_t v = (uint32_t)lane + 64u * j;
            if (v < nodes) {
                const uint32_t a = anc[v];
                na[j] = anc[a]; nd[j] = dep[v] + dep[a];
                moving = moving || a != root;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const uint32_t v = (uint32_t)lane + 64u * j;
            if (v < nodes) { anc[v] = (uint16_t)na[j]; dep[v] = (uint16_t)nd[j]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (!__ballot(moving)) break;
    }
    uint32_t maxd = 0;
    for (int k = lane; k < m; k += 64) {
        const uint32_t d = dep[k];
        atomicAdd(&s.depthcnt[d - 1], 1u);
        maxd = d > maxd ? d : maxd;
    }
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)maxd, w, 64); maxd = o > maxd ? o : maxd; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    // HuffmanTree.limitHeight (:348-404)
    int nl = (int)maxd;
    if (nl > limit) {
        if (lane == 0) {
            int unhoused = 0;
            for (int l = nl - 1; l >= limit; --l) {
                const int pairs = (int)(s.depthcnt[l] >> 1);
                unhoused += pairs;
                s.depthcnt[l - 1] += (uint32_t)pairs;
            }
            int split = limit - 2;
            while (unhoused > 0) {
                if (!(s.depthcnt[split] > 0)) { split--; continue; }
                const int resettled = (int)s.depthcnt[split] < unhoused ? (int)s.depthcnt[split] : unhoused;
                unhoused -= resettled;
                s.depthcnt[split] -= (uint32_t)resettled;
                s.depthcnt[split + 1] += (uint32_t)(2 * resettled);
                if (split < limit - 2) split++;
            }
        }
        nl = limit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    }
    // most frequent symbols take the shortest codes (:306-337): rank r gets the level its slot falls into
    for (int base = 0; base < m; base += 64) {
        const int r = base + lane;
        uint32_t acc = 0, l1 = 1;
        for (int l = 0; l < nl; ++l) { acc += UNI(s.depthcnt[l]); l1 += (uint32_t)r >= acc; }
        if (r < m) len[s.order[r]] = (uint8_t)l1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// canonical codewords, bit-reversed for LSB-first emission (HuffmanTree.codewords :206-230)
__device__ __attribute__((noinline)) void make_codes(const uint8_t *len_, int n_, uint16_t *code_, int lane)
{
    const uint8_t *len = UNIP(const uint8_t *, len_); uint16_t *code = UNIP(uint16_t *, code_);
    const int n = (int)UNI(n_);
    // codes of one length are consecutive in symbol order; the first code of length l is
    // (first of l-1 + count of l-1) << 1
    uint32_t my[5], cw[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) { const int sym = c * 64 + lane; my[c] = sym < n ? len[sym] : 0u; cw[c] = 0; }
    uint32_t next = 0, prev = 0;
    for (uint32_t l = 1; l <= 15; ++l) {
        next = (next + prev) << 1;
        uint32_t cnt = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const unsigned long long mk = __ballot(my[c] == l);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0));
            if (my[c] == l) cw[c] = next + cnt + before;
            cnt += (uint32_t)__popcll(mk);
        }
        prev = cnt;
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const int sym = c * 64 + lane;
        if (sym < n && my[c]) code[sym] = (uint16_t)(__brev(cw[c]) >> (32 - my[c]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// The head of a block once the code lengths stand in s.ll / s.dl: code-length RLE, the code-length code,
// codewords, then writeBlockMetadata + writeBlockTables (DeflatorBuffers.Stream.swift:459-623).
__device__ __attribute__((noinline)) Bits write_tables(Bits b_, bool final_, int lane)
{
    Bits b = uni_bits(b_);
    const bool final = UB(final_);
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
    // writeBlockTables (:615-623), 64 terms at a time
    for (int k0 = 0; k0 < nm; k0 += 64) {
        const int k = k0 + lane;
        uint64_t v = 0; uint32_t nb = 0;
        if (k < nm) {
            const uint32_t sym = s.msym[k], l = s.ml[sym];
            v = (uint64_t)s.mcode[sym] | (uint64_t)s.mbits[k] << l;
            nb = l + (sym == 18 ? 7u : sym == 17 ? 3u : sym == 16 ? 2u : 0u);
        }
        bulk_put(s, b, v, nb, lane);
    }
    maybe_drain(s, b, lane);
    return b;
}

// the bits of one term: a literal, or run code + extra bits + distance code + extra bits (<= 48)
__device__ __forceinline__ uint64_t literal_bits(const DLds &s, uint32_t lit, uint32_t &nb)
{
    nb = s.ll[lit];
    return s.lcode[lit];
}
__device__ __forceinline__ uint64_t match_bits(const DLds &s, uint32_t rd, uint32_t rextra, uint32_t dd, uint32_t dextra, uint32_t &nb)
{
    const uint32_t sym = 256 | rd, l1 = s.ll[sym], e1 = run_extra_bits(rd), l2 = s.dl[dd], e2 = dist_extra_bits(dd);
    nb = l1 + e1 + l2 + e2;
    return (uint64_t)s.lcode[sym] | (uint64_t)rextra << l1 | (uint64_t)s.dcode[dd] << (l1 + e1) | (uint64_t)dextra << (l1 + e1 + l2);
}

// Stream.writeBlock (DeflatorBuffers.Stream.swift:440-709), greedy / lazy form
template <class TP>
__device__ __forceinline__ Bits write_block_from(Bits b_, int count_, bool final_, int lane, TP terms)
{
    Bits b = uni_bits(b_);
    const int count = (int)UNI(count_);
    const bool final = UB(final_);
    DLds &s = g_lds;
    // DeflatorMatches.trees() (:138-159)
    for (int i = lane; i < 320; i += 64) s.freq[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    for (int i = lane; i < count; i += 64) {
        const uint32_t t = terms[i];
        atomicAdd(&s.freq[t & 0x1ff], 1u);
        atomicAdd(&s.freq[288 + (t >> 27)], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) s.freq[256] = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    build_tree(s.freq, 286, 15, s.ll, lane);
    build_tree(s.freq + 288, 30, 15, s.dl, lane);
    b = uni_bits(write_tables(b, final, lane));
    // writeBlock(with:) (:626-659), 64 terms at a time
    for (int i0 = 0; i0 < count; i0 += 64) {
        const int i = i0 + lane;
        uint64_t v = 0; uint32_t nb = 0;
        if (i < count) {
            const uint32_t t = terms[i];
            const uint32_t sym = t & 0x1ff;
            if (sym > 256) v = match_bits(s, sym & 0xff, (t >> 9) & 0x1f, t >> 27, (t >> 14) & 0x1fff, nb);
            else v = literal_bits(s, sym, nb);
        }
        bulk_put(s, b, v, nb, lane);
        maybe_drain(s, b, lane);
    }
    put(s, b, s.lcode[256], s.ll[256], lane);
    maybe_drain(s, b, lane);
    return b;
}
// ... the queued terms of the one-wave / two-wave forms (LDS, buffer tb)
__device__ __attribute__((noinline)) Bits write_block(Bits b_, int count_, bool final_, int lane, int tb_ = 0)
{
    return write_block_from(b_, count_, final_, lane, (const uint32_t *)g_trm.terms[UNI(tb_)]);
}
// ... a block of the block-parallel form: its terms where the walk left them (global memory)
__device__ __attribute__((noinline)) Bits write_block_global(Bits b_, int count_, bool final_, int lane, const gword *terms_)
{
    return write_block_from(b_, count_, final_, lane, UNIP(const gword *, terms_));
}

__device__ __forceinline__ uint32_t load32(const gbyte *p) { return ((const gU32u *)p)->v; }

// bytes of position q.. and p.. agree for how many bytes (<= limit)?  Eight at a time from the input.
struct __attribute__((packed)) U64u { uint64_t v; };
typedef U64u __attribute__((address_space(1))) gU64u;
__device__ __forceinline__ uint64_t load64(const gbyte *p) { return ((const gU64u *)p)->v; }
__device__ __forceinline__ uint32_t common_prefix(const gbyte *in, uint64_t q, uint64_t p, uint32_t limit)
{
    uint32_t i = 0;
    while (i + 8 <= limit) {
        const uint64_t x = load64(in + q + i) ^ load64(in + p + i);
        if (x) return i + ((uint32_t)__builtin_ctzll(x) >> 3);
        i += 8;
    }
    if (i + 4 <= limit) {
        const uint32_t x = load32(in + q + i) ^ load32(in + p + i);
        if (x) return i + (__builtin_ctz(x) >> 3);
        i += 4;
    }
    while (i < limit && in[q + i] == in[p + i]) ++i;
    return i;
}

// LZ77.DeflatorWindow.match (:132-212) for TWO positions per lane at once (pA, pB: the lane's position in
// each half of a 128-position batch): both chains hop together, so the latency of a hop -- the link and,
// speculatively, the candidate's first four bytes -- is paid once for the pair.  hit(which, distance, run)
// sees every candidate whose tag and key match, in chain order, and the reference's stop rules apply per
// chain: `attempts` candidates, a run >= `goal`, the window 2^exponent.
template <class F>
__device__ __forceinline__ void chain_walk2(const gbyte *in, const gword *ring, uint64_t n, uint64_t pA, uint64_t pB,
                                            bool liveA, bool liveB, uint32_t keyA, uint32_t keyB, uint32_t wmask,
                                            int attempts, int goal, F &&hit)
{
    uint32_t tagA = 0, tagB = 0, dA = 0, dB = 0, accA = 0, accB = 0;
    int remA = attempts, remB = attempts;
    bool firstA = true, firstB = true;
    if (liveA) { const uint32_t m = ring[pA & 65535]; tagA = m >> 16; dA = m & 0xffff; }
    if (liveB) { const uint32_t m = ring[pB & 65535]; tagB = m >> 16; dB = m & 0xffff; }
    const uint32_t limA = n - pA < 258 ? (uint32_t)(n - pA) : 258u, limB = n - pB < 258 ? (uint32_t)(n - pB) : 258u;
    while (dA | dB) {
        bool goA = dA != 0, goB = dB != 0;
        if (goA) { accA += dA; if (accA > wmask || (!firstA && accA >= wmask)) goA = false; }
        if (goB) { accB += dB; if (accB > wmask || (!firstB && accB >= wmask)) goB = false; }
        uint32_t eA = 0, eB = 0, kA = 0, kB = 0;
        if (goA) { eA = ring[(pA - accA) & 65535]; kA = load32(in + pA - accA); }
        if (goB) { eB = ring[(pB - accB) & 65535]; kB = load32(in + pB - accB); }
        if (goA && (eA >> 16) == tagA && kA == keyA) {
            const uint32_t run = common_prefix(in, pA - accA, pA, limA);
            hit(0, accA, run);
            firstA = false; remA -= 1;
            if (!(remA > 0 && goal > (int)run)) goA = false;
        }
        if (goB && (eB >> 16) == tagB && kB == keyB) {
            const uint32_t run = common_prefix(in, pB - accB, pB, limB);
            hit(1, accB, run);
            firstB = false; remB -= 1;
            if (!(remB > 0 && goal > (int)run)) goB = false;
        }
        dA = goA ? eA & 0xffff : 0u;
        dB = goB ? eB & 0xffff : 0u;
    }
}

// the 4-byte key of position p (zero-extended at the end of the input)
__device__ __forceinline__ uint32_t load_key(const gbyte *in, uint64_t n, uint64_t p)
{
    if (p + 4 <= n) return load32(in + p);
    uint32_t key = 0;
    for (int k = 0; k < 4; ++k) if (p + k < n) key |= (uint32_t)in[p + k] << (8 * k);
    return key;
}

// Hash insertion of the 64 positions inserted .. inserted + 63 (LZ77.DeflatorWindow.update, :78-128): every
// lane hashes its 4-byte key; a position's link is the distance to the previous position of its bucket --
// the nearest lower lane with the same bucket (radix match over the hash bits: one ballot per bit), else
// the bucket head -- and the last lane of every bucket becomes the new head.  Adler-32 sums ride along.
__device__ __forceinline__ void insert_batch(uint32_t *head, const gbyte *in, uint64_t n, gword *ring, uint64_t inserted, uint32_t key,
                                             uint32_t &accS, uint32_t &accI, int lane, bool sum = true)
{
    const uint64_t p = inserted + lane;
    const bool live = p + 4 <= n;                              // the last three positions never start a match
    if (p < n && sum) {                                        // Adler-32 accumulators
        const uint32_t byte = key & 0xff;
        accS += byte;
        accI = (accI + (uint32_t)(p % 65521) * byte) % 65521;
    }
    const uint32_t mix = key * 0x9E3779B1u;
    const uint32_t h = mix >> (32 - HBITS);
    const uint32_t tag = (mix >> 3) & 0xffffu;
    unsigned long long same = __ballot(live);
#pragma unroll
    for (int k = 0; k < HBITS; ++k) {
        const unsigned long long bk = __ballot((h >> k) & 1);
        same &= (h >> k) & 1 ? bk : ~bk;
    }
    const unsigned long long lower = (1ull << lane) - 1;
    const unsigned long long below = same & lower, above = same & ~lower & ~(1ull << lane);
    uint32_t prev = live ? head[h] : NONE;
    if (below) prev = (uint32_t)(inserted + (63 - __clzll((long long)below)));
    uint32_t dist = 0;
    if (live && prev != NONE) {
        const uint64_t d = (uint32_t)((uint32_t)p - prev);
        dist = d <= 32767 ? (uint32_t)d : 0;
    }
    if (p < n) ring[p & 65535] = dist | tag << 16;
    head[live && !above ? h : 1u << HBITS] = (uint32_t)p;      // (idle lanes: the spare slot)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

#ifdef SPNG_DEFLATE_PROF
// full kernel: cycles per phase, kept in LDS so that the non-inlined passes can add to them
__shared__ uint64_t g_prof[12];
#define FPROF(k) do { const uint64_t now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) { g_prof[k] += now_ - g_prof[11]; g_prof[11] = now_; } } while (0)
#define DPROF_DECL uint64_t pt[6] = {0,0,0,0,0,0}, p_t0 = 0;
#define DPROF_BEGIN() p_t0 = __builtin_readcyclecounter()
#define DPROF_END(k) pt[k] += __builtin_readcyclecounter() - p_t0
#else
#define FPROF(k)
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

    for (int i = lane; i < OUTB / 4; i += 64) s.out32[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    Bits b = {0, 0, 0, 0, job.dst, job.dst_cap, false};
    const uint32_t wmask = (1u << UNI(jp->exponent)) - 1;    // window 2^exponent (LZ77.Deflator(exponent:); PNG: 15)
    // spng_deflate_resume_batch: the stream arrives in pieces.  `more`: src_len is what has arrived so far; st1: where the
    // previous push left off (parse position, queued terms, bit writer, Adler sums)
    const bool more = UNI(jp->more) != 0;
    D1State *st1 = (D1State *)uni64((uint64_t)jp->state);
    const bool resumed = st1 && UNI(st1->started);
    if (resumed) { b.acc = uni64(st1->acc); b.nacc = UNI(st1->nacc); b.total = b.flushed = uni64(st1->total); b.overflow = UNI(st1->overflow) != 0; }
    if (n < 3 && more) {
        // (nothing can be decided yet: not even whether this will be a stored tail)
        if (lane == 0) {
            spng_result &res = results[job.image];
            res.status = SPNG_NEED_MORE_INPUT; res.reserved = 0; res.written = b.total; res.consumed = resumed ? uni64(st1->w) : 0;
            res.aux[0] = res.consumed; res.aux[1] = 0;
        }
        return;
    }
    if (resumed) {}
    else if (job.format == SPNG_FORMAT_ZLIB) {
        // StreamHeader.write (StreamHeader.swift:56-62)
        const uint32_t unpaired = (UNI(jp->exponent) - 8) << 4 | 0x08;
        const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
        put(s, b, check << 8 | unpaired, 16, lane);
    } else if (job.format == SPNG_FORMAT_GZIP) {
        // Gzip.StreamHeader.write (Gzip.StreamHeader.swift:84-96): sigil, method 8, no flags, MTIME 0, XFL 0, OS 255;
        // the trailer (CRC-32, byte count) is appended by gzip.hip
        put(s, b, 0x8b1f, 16, lane); put(s, b, 0x0008, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0xff00, 16, lane);
    }
    for (int i = lane; i <= (1 << HBITS); i += 64) g_sea.head[i] = NONE;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    DPROF_DECL
    uint32_t accS = 0, accI = 0;                               // Adler-32 of the input, as in inflate.hip
    int count = 0;                                             // queued terms
    const int limit_terms = 2048;
    auto unfilled = [&]() { return limit_terms - 1 - count; };
    uint64_t w0 = 0, summed = 0;                               // where the parse goes on; positions below `summed` are in the sums already
    if (resumed) {
        w0 = uni64(st1->w); summed = uni64(st1->inserted); count = (int)UNI(st1->count);
        for (int i = lane; i < count; i += 64) g_trm.terms[0][i] = st1->terms[i];
        if (lane == 0) { accS = st1->adlerS; accI = st1->adlerI; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    }

    if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434)
        put(s, b, 1, 3, lane);
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, (uint32_t)n, 16, lane); put(s, b, ~(uint32_t)n & 0xffff, 16, lane);
        for (uint64_t k = 0; k < n; ++k) put(s, b, in[k], 8, lane);
        if ((uint64_t)lane < n) { accS = in[lane]; accI = (uint32_t)lane * in[lane]; }
    } else {
        const uint64_t last_main = n - 4 + 1;                  // positions 0 .. n-4 are searched
        // (a resumed stream: the 32 KiB in front of the parse position enter the window again)
        uint64_t inserted = (w0 >= 32768 ? w0 - 32768 : 0) & ~(uint64_t)63;    // positions < inserted are in the window
        uint64_t w = w0;                                       // parse position
        uint32_t key_next = load_key(in, n, inserted + (uint64_t)lane);      // (keys travel one batch ahead of their insertion)
        auto insert_upto = [&](uint64_t target) {
            while (inserted < target && inserted < n) {
                const uint32_t key = key_next;
                key_next = load_key(in, n, inserted + 64 + lane);
                insert_batch(g_sea.head, in, n, ring, inserted, key, accS, accI, lane, inserted + lane >= summed);
                inserted = uni64(inserted + 64);
            }
        };

        // (more input to come: a batch of 128 positions is only searched when each of them sees its whole look-ahead)
        while (w < last_main && (!more || w + 128 + 259 <= n)) {
            // keep the window filled well ahead of the 64 positions searched now (+258 of look-ahead
            // is irrelevant for insertion: links only point backwards)
            DPROF_BEGIN();
            insert_upto(w + 192);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // our own ring stores, before we read them back
            DPROF_END(0); DPROF_BEGIN();
            // ---- match search: lane i answers window.match(from: w + i) and (from: w + 64 + i)
            const uint64_t pA = w + lane, pB = pA + 64;
            const bool liveA = pA < last_main, liveB = pB < last_main;
            const uint32_t keyA = liveA ? load32(in + pA) : 0u, keyB = liveB ? load32(in + pB) : 0u;
            uint32_t brA = 5, bdA = 1, brB = 5, bdB = 1;
            chain_walk2(in, ring, n, pA, pB, liveA, liveB, keyA, keyB, wmask, attempts, goal,
                        [&](int which, uint32_t dist, uint32_t run) {
                            // the first strictly longest run wins (:145-208)
                            if (which == 0) { if (brA < run) { brA = run; bdA = dist; } }
                            else            { if (brB < run) { brB = run; bdB = dist; } }
                        });
            const uint32_t mrunA = brA > 5 ? brA : 0, mrunB = brB > 5 ? brB : 0;   // 0: no match (run must exceed 5, :129)
            const uint32_t litA = keyA & 0xff, litB = keyB & 0xff;
            auto at = [&](uint32_t xa, uint32_t xb, uint32_t t) -> uint32_t {
                return t < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)xa, (int)t) : (uint32_t)__builtin_amdgcn_readlane((int)xb, (int)(t - 64));
            };
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DPROF_END(1); DPROF_BEGIN();

            // ---- the parse: Stream.compress greedy (:209-252) / lazy (:268-323) over these 128 answers
            uint32_t t = 0;
            bool stop = false;
            while (t < 128 && w + t < last_main && !stop) {
                if (!(unfilled() > (lazy ? 1 : 0))) { DPROF_END(2); DPROF_BEGIN(); b = uni_bits(write_block(b, count, false, lane)); count = 0; DPROF_END(3); DPROF_BEGIN(); }
                const uint32_t run = at(mrunA, mrunB, t);
                const uint32_t lit = at(litA, litB, t);
                if (!run) { g_trm.terms[0][count] = 0xf8000000u | lit; ++count; t += 1; continue; }
                uint32_t use_run = run, use_dist = at(bdA, bdB, t);
                uint32_t adv = run;
                if (lazy) {
                    // the answer for position w+t+1 is needed: restart the search there if it is not in this batch
                    if (t + 1 >= 128) { stop = true; break; }
                    // lazy match at a+1 (:293-299); it exists only if that position is still searched
                    const uint32_t lrun = (w + t + 1 < last_main) ? at(mrunA, mrunB, t + 1) : 0u;
                    if (lrun > run) {
                        g_trm.terms[0][count] = 0xf8000000u | lit;
                        ++count;
                        use_run = lrun; use_dist = at(bdA, bdB, t + 1);
                        adv = 1 + lrun;
                    }
                }
                // LZ77.DeflatorTerm.init(run:distance:) (DeflatorTerm.swift:34-56)
                const uint32_t rd = run_decade(use_run), dd = dist_decade(use_dist);
                g_trm.terms[0][count] = dd << 27 | 0x100u | rd | dist_extra_value(use_dist, dd) << 14 | run_extra_value(use_run, rd) << 9;
                ++count;
                t += adv;
            }
            w = uni64(w + t);
            DPROF_END(2);
        }
        if (more) {
            // on with the next push: whole bytes out, the rest into the state
            drain(s, b, b.total, lane);
            uint32_t S = accS % 65521, I = accI % 65521;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
            for (int i = lane; i < count; i += 64) st1->terms[i] = g_trm.terms[0][i];
            if (lane == 0) {
                st1->w = w; st1->inserted = inserted > summed ? inserted : summed; st1->acc = b.acc; st1->nacc = b.nacc; st1->total = b.total;
                st1->overflow = b.overflow ? 1u : 0u; st1->count = (uint32_t)count; st1->started = 1;
                st1->adlerS = S % 65521; st1->adlerI = I % 65521;
                spng_result &res = results[job.image];
                res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_NEED_MORE_INPUT; res.reserved = 0;
                res.written = b.total; res.consumed = w; res.aux[0] = w; res.aux[1] = 0;
            }
            return;
        }
        insert_upto(n);                                        // Adler-32 over the tail
        // epilogue: the positions still in the window pipeline become literals (:254-265, :331-342)
        for (uint64_t p = w; p < n; ++p) {
            if (!(unfilled() > 0)) { b = uni_bits(write_block(b, count, false, lane)); count = 0; }
            g_trm.terms[0][count] = 0xf8000000u | UNI(in[p]);
            ++count;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        b = uni_bits(write_block(b, count, true, lane));
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
// Here, per stream:
//   * candidates: the 128-positions-at-a-time chain walk of the greedy / lazy kernel, every candidate
//     recorded (per lane and half, thirty decade slots in LDS); the serial part is only deciding which
//     positions are searched at all (behind a run > 100 the next run - 100 vertices get no edges,
//     DeflatorBuffers.Stream.swift:376-380).  Vertices live in HBM: a flag (has edges) and, for those
//     that have, 30 slots.
//   * forward pass (full_forward / forward_body): the best way into a vertex as the minimum of one 64-bit
//     key over its incoming edges; along 64 vertices the depths are a min-plus prefix scan, match edges
//     go through ds_min_u64 into an LDS ring, three vertices at a time; with helper waves on
//     compressible input.
//   * back-trace (full_backward): 64 vertices at a time from the end, the path hopping through the
//     batch on the scalar unit; path vertices tally the symbol frequencies and hand their edge to the
//     vertex it starts from.
//   * trees, cost update, repeat (2 x iterations passes for the first block, iterations after);
//     then the block is written: the path's terms 64 at a time.
struct FullArrays {      // (two registers' worth: passed to the non-inlined passes in SGPRs, not through scratch memory)
    gword *base; uint32_t vcap;
    // [vertex][30]: distance << 16 | longest run of that distance decade
    __device__ __forceinline__ gword *slots_() const { return base; }
    // [vertex]: incoming edge of the cheapest path, run << 16 | decade << 8 (literal: 1 << 16 | 0xff00)
    __device__ __forceinline__ gword *up_() const { return base + (uint64_t)vcap * 30; }
    // [vertex]: the path's edge that STARTS here (set by the back-trace)
    __device__ __forceinline__ gword *step_() const { return up_() + vcap + 1; }
    // [vertex]: on the path
    __device__ __forceinline__ gbyte *pathb_() const { return (gbyte *)(step_() + vcap + 1); }
    // [vertex]: has edges (its thirty slots are valid); vertices without ones never touch `slots`
    __device__ __forceinline__ gbyte *flag_() const { return pathb_() + vcap + 1; }
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

// minimize() forwards (:262-280, explore :322-379): best depth and incoming edge of every vertex.
//
// The reference visits the vertices in order and lets every edge of a vertex overwrite its target when it is
// strictly cheaper -- so among equally cheap ways into a vertex the FIRST writer stays: the edge from the
// earliest vertex (= the longest edge), then the literal before the matches, then the lowest distance decade.
// That makes the best way into a vertex the minimum of one 64-bit key over all its incoming edges,
//     depth << 32 | (258 - length) << 8 | (literal: 0, match: decade + 1)
// and the order in which edges are tried irrelevant, as long as every edge is tried with the final depth of
// the vertex it leaves.  So, 64 vertices at a time:
//   * the keys the matches have offered so far sit in an LDS ring (ds_min_u64);
//   * depth(v) = min(offered(v), depth(v - 1) + literal cost) along the batch is a prefix scan under
//     (a1, b1) o (a2, b2) = (a1 + a2, min(b1 + a2, b2)), in DPP steps, no memory;
//   * a match is at least 3 long: the depths of three consecutive vertices are final before any of their
//     own edges is relaxed.  Vertices with edges are taken three positions at a time (lanes spread over the
//     run lengths, costs in registers), then the scan is repeated.  Batches without edges (incompressible
//     data) cost one scan.
static constexpr uint32_t DINF = 0x3fffffffu;    // "no way in yet" (real depths stay below 2^28)
__device__ __forceinline__ uint32_t minplus_scan(uint32_t a, uint32_t b, uint32_t x, int lane)
{
    // inclusive composition inside each row of 16 lanes
#define SPNG_MP_STEP(ctrl)                                                                         \
    {                                                                                              \
        const uint32_t as = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, ctrl, 0xf, 0xf, false);          \
        const uint32_t bs = (uint32_t)__builtin_amdgcn_update_dpp((int)DINF, (int)b, ctrl, 0xf, 0xf, false);  \
        const uint32_t nb = bs + a < b ? bs + a : b;                                               \
        a += as; b = nb;                                                                           \
    }
    SPNG_MP_STEP(0x111) SPNG_MP_STEP(0x112) SPNG_MP_STEP(0x114) SPNG_MP_STEP(0x118)
#undef SPNG_MP_STEP
    // the value entering each row
    uint32_t x1, x2, x3;
    {
        const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)a, 15), b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, 15);
        const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)a, 31), b1 = (uint32_t)__builtin_amdgcn_readlane((int)b, 31);
        const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)a, 47), b2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 47);
        x1 = x + a0 < b0 ? x + a0 : b0;
        x2 = x1 + a1 < b1 ? x1 + a1 : b1;
        x3 = x2 + a2 < b2 ? x2 + a2 : b2;
    }
    const uint32_t xin = lane < 16 ? x : lane < 32 ? x1 : lane < 48 ? x2 : x3;
    return xin + a < b ? xin + a : b;
}

// One forward pass, run by all four waves of the workgroup.  Wave 0 owns the pass: it fetches, keeps the ring
// initialised, scans alone through batches without edges and finalises every batch.  In a batch with edges the
// three vertices of a group (their depths are final together, see above) are relaxed by waves 0, 1 and 2 at the
// same time -- each wave scans for itself (same ring, same result), so the only things exchanged are the ring and
// one workgroup barrier per group; wave 3 only keeps the barrier count (256-thread workgroups place evenly).
enum { FW_PASS = 1, FW_EXIT = 2 };
// (WAVES == 1: the same pass on the one wave of a 64-thread workgroup -- no helpers, no barriers)
template <int WAVES> __device__ __forceinline__ void wg_sync()
{
    if (WAVES > 1) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}
template <int WAVES>
__device__ __attribute__((noinline)) void forward_body(const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count, int lane, int wave)
{
    DLds &s = g_lds;
    // costs in registers: distance decade `lane`; run lengths 3 + lane + 64 j
    const uint32_t dcost = lane < 30 ? g_full.depths[512 + lane] : 0u;
    uint32_t rc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t L = 3u + (uint32_t)lane + 64u * j; rc[j] = L <= 258 ? g_full.depths[253 + L] : 0u; }
    if (wave == 0 && lane == 0) g_full.win[0] = 0;                  // vertex 0: depth 0
    uint32_t inited = 1, carry = DINF;                         // carry: depth of the vertex in front of the batch
    // (the literal byte of a batch is fetched while the batch before it is worked on)
    uint32_t lb_next = 0, fl_next = 0;
    if (wave == 0) {
        lb_next = (lane >= 1 && (uint32_t)lane <= count) ? in[bbase + lane - 1] : 0u;
        fl_next = (uint32_t)lane < count ? g.flag_()[lane] : 0u;
    }
    for (uint32_t sb0 = 0; sb0 <= count; sb0 += 1024) {
        // sixteen batches at a time, wave 0 tells the helpers which vertices have edges: through incompressible
        // data they then sleep from one of these barriers to the next, one per 1024 vertices
        const uint32_t spar = (sb0 >> 10) & 1;
        if (WAVES > 1 && wave == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t vq = sb0 + 64u * q + (uint32_t)lane;
                const unsigned long long e = __ballot(vq < count && g.flag_()[vq] != 0);
                if (lane == 0) g_old.fw_ems[spar][q] = e;
            }
        }
        if (WAVES > 1) __syncthreads();
    for (uint32_t b0 = sb0; b0 <= count && b0 < sb0 + 1024; b0 += 64) {
        const uint32_t nv = count + 1 - b0 < 64 ? count + 1 - b0 : 64;      // vertices b0 .. b0 + nv - 1 (the last one: `count`, the end)
        const uint32_t v = b0 + (uint32_t)lane;
        const uint32_t par = (b0 >> 6) & 1;
        // (wave 0 has its own copy of the flags, fetched a batch ahead: no LDS round trip on its way through
        //  edge-less batches)
        unsigned long long em;
        if (wave == 0) {
            em = __ballot(fl_next != 0);
            const uint32_t vn = b0 + 64 + (uint32_t)lane;
            fl_next = vn < count ? g.flag_()[vn] : 0u;
        } else {
            em = uni64(g_old.fw_ems[spar][(b0 - sb0) >> 6]);
            if (!em) continue;                                 // (wave 0 scans through it alone)
        }
        uint32_t cin = 0;
        if (wave == 0) {
            const uint32_t lb = lb_next;
            {
                const uint32_t vn = b0 + 64 + (uint32_t)lane;
                lb_next = vn <= count ? in[bbase + vn - 1] : 0u;
            }
            const uint32_t need = (b0 + 64 + 258 < count ? b0 + 64 + 258 : count) + 1;
            for (uint32_t j = inited + lane; j < need; j += 64) g_full.win[j & 511] = ~0ull;
            inited = inited > need ? inited : need;
            if (em) {
                const uint32_t ns = count - b0 < 64 ? count - b0 : 64;
                for (uint32_t i = lane; i < ns * 30; i += 64) g_old.batch[i] = g.slots_()[(uint64_t)b0 * 30 + i];
            }
            cin = (v >= 1 && v <= count) ? g_full.depths[lb] : 0u;   // the literal edge INTO v
            if (WAVES > 1 && em) {
                g_old.fw_cin[par][lane] = cin;
                if (lane == 0) g_old.fw_carry[par] = carry;
            }
        }
        if (WAVES > 1 && em) {
            __syncthreads();                                   // the batch is set up
            if (wave != 0) { cin = g_old.fw_cin[par][lane]; carry = UNI(g_old.fw_carry[par]); }
        } else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        uint64_t W; uint32_t Wd, D;
        uint32_t k = 0;
        for (;;) {
            W = (uint32_t)lane < nv ? g_full.win[v & 511] : ~0ull;
            Wd = (uint32_t)(W >> 32) < DINF ? (uint32_t)(W >> 32) : DINF;
            D = minplus_scan(cin, Wd, carry, lane);
            const unsigned long long rest = k < 64 ? (em >> k) << k : 0ull;
            if (!rest) break;
            const uint32_t kk = (uint32_t)__ffsll((long long)rest) - 1;
            // (four waves: this wave's vertex of the group; one wave: all three in turn)
            for (uint32_t kq = kk + (WAVES > 1 ? (uint32_t)wave : 0u); kq < kk + (WAVES > 1 ? (uint32_t)wave + 1u : 3u); ++kq) {
                if (!((WAVES == 1 || wave < 3) && kq < 64 && ((em >> kq) & 1) && count - (b0 + kq) >= 3)) continue;
                const uint32_t vv = b0 + kq, rem = count - vv;
                const uint32_t Dk = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)kq);
                const uint32_t run = lane < 30 ? g_old.batch[kq * 30 + lane] & 0xffffu : 0u;
                unsigned long long m = __ballot(run > 0);
                // Of the decades that reach a length, only the cheapest (first among equals: the lowest) can be
                // the way into that target from this vertex: one key per length instead of one per decade.
                uint32_t bc[4] = {~0u, ~0u, ~0u, ~0u}, bd[4] = {0, 0, 0, 0}, reach = 0;
                while (m) {
                    const int dec = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)run, dec);
                    const uint32_t maxlen = r < rem ? r : rem;
                    const uint32_t dc = (uint32_t)__builtin_amdgcn_readlane((int)dcost, dec);
                    reach = maxlen > reach ? maxlen : reach;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (3u + 64u * j > maxlen) break;
                        const uint32_t L = 3u + (uint32_t)lane + 64u * j;
                        if (L <= maxlen && dc < bc[j]) { bc[j] = dc; bd[j] = (uint32_t)dec; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (3u + 64u * j > reach) break;
                    const uint32_t L = 3u + (uint32_t)lane + 64u * j;
                    if (bc[j] != ~0u) {
                        const uint64_t key = (uint64_t)(Dk + bc[j] + rc[j]) << 32 | (258u - L) << 8 | (bd[j] + 1u);
                        __hip_atomic_fetch_min(&g_full.win[(vv + L) & 511], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            k = kk + 3;
            wg_sync<WAVES>();                                  // the group's keys are in the ring
        }
        if (wave != 0) continue;
        // the way in: the literal only when strictly cheaper than what the matches offer
        if ((uint32_t)lane < nv && v >= 1) {
            const uint32_t low = (uint32_t)W;
            g.up_()[v] = D < Wd ? 0x0001ff00u : (258u - (low >> 8)) << 16 | ((low & 0xff) - 1u) << 8;
        }
        carry = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)(nv - 1));
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// (wave 0) one forward pass, with the helper waves if there are any
template <int WAVES>
__device__ __attribute__((noinline)) void full_forward(const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count, int lane)
{
    DLds &s = g_lds;
    if (WAVES > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the vertices' flags and slots are in memory
        if (lane == 0) { g_old.fw_cmd = FW_PASS; g_old.fw_bbase = bbase; g_old.fw_count = count; }
        __syncthreads();
    }
    forward_body<WAVES>(g, in, bbase, count, lane, 0);
    if (WAVES > 1) __syncthreads();                            // (the helpers are back in their loop before anything else changes)
}

// minimize() backwards (:282-320): the path from the last vertex to the first, symbol frequencies into
// s.freq, every path edge handed to the vertex it starts from.  64 vertices at a time, descending from a
// vertex on the path: lane i holds the way into vertex hi - i; the path hops through the batch on the
// scalar unit (v_readlane with the hop's length), and a batch of nothing but literals is all path.
__device__ __attribute__((noinline)) void full_backward(const FullArrays g, const gbyte *in, uint64_t bbase, uint32_t count, int lane)
{
    DLds &s = g_lds;
    for (int i = lane; i < 320; i += 64) s.freq[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    uint32_t hi = count;
    // (the ways into the next 64 vertices are fetched ahead on the guess that the path leaves this batch
    //  exactly at its end, as it does through literals; a longer hop refetches)
    uint32_t u_next = ((uint32_t)lane < hi) ? g.up_()[hi - (uint32_t)lane] : 0u, hi_next = hi;
    for (;;) {
        const bool valid = (uint32_t)lane <= hi;
        const uint32_t c = valid ? hi - (uint32_t)lane : 0u;
        uint32_t u = u_next;
        if (hi_next != hi) u = (valid && c > 0) ? g.up_()[c] : 0u;
        if (hi >= 64) { hi_next = hi - 64; u_next = ((uint32_t)lane < hi_next) ? g.up_()[hi_next - (uint32_t)lane] : 0u; }
        const uint32_t len = u >> 16;                          // 0: vertex 0 (or nothing)
        unsigned long long pm = 0;
        uint32_t pos = 0;
        if (!__ballot(valid && c > 0 && len != 1)) { pm = __ballot(valid); pos = 64; }
        else {
            while (pos < 64) {
                pm |= 1ull << pos;
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)pos);
                if (!l) break;
                pos += l;
            }
        }
        const bool on = ((pm >> lane) & 1) != 0;
        if (valid && c < count) g.pathb_()[c] = on ? 1 : 0;
        if (on && c > 0) {
            const uint32_t nxt = c - len;
            g.step_()[nxt] = u & 0xffffff00u;
            if (len == 1) atomicAdd(&s.freq[in[bbase + nxt]], 1u);
            else { atomicAdd(&s.freq[256 | run_decade(len)], 1u); atomicAdd(&s.freq[288 + ((u >> 8) & 0xff)], 1u); }
        }
        if (hi < 64 || pos > hi) break;                        // vertex 0 was in this batch
        if (pos < 64) break;                                   // (cannot happen: a hop of length 0 above vertex 0)
        const uint32_t nhi = hi - pos;
        // vertices the leaving hop jumped over are not on the path
        for (uint32_t cc = nhi + 1 + (uint32_t)lane; cc + 64 <= hi; cc += 64) g.pathb_()[cc] = 0;
        hi = nhi;
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
        if (sym < 256) g_full.depths[sym] = (uint8_t)(len << 2);
        else if (sym > 256) {
            const uint32_t dec = sym & 0xff, e = run_extra_bits(dec), base = 253 + run_base(dec);
            for (uint32_t l = base; l < base + (1u << e); ++l) if (l != 253 + 258 || dec == 29) g_full.depths[l] = (uint8_t)((len + e) << 2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (lane == 0) {
        const uint32_t a = s.ll[284], c = s.ll[285];
        if (a && (!c || a > c)) g_full.depths[253 + 258] = (uint8_t)((a + 5) << 2);
        else if (c) g_full.depths[253 + 258] = (uint8_t)(c << 2);
    }
    if (lane < 30 && s.dl[lane]) g_full.depths[512 + lane] = (uint8_t)((s.dl[lane] + dist_extra_bits(lane)) << 2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
}

// Stream.writeBlock (DeflatorBuffers.Stream.swift:440-709), full form: trees(iterations:), header, the path's tokens
template <int WAVES>
__device__ __attribute__((noinline)) Bits full_block(Bits b_, const FullArrays g_, const gbyte *in_, uint64_t bbase_, uint32_t count_,
                                                     bool final_, int iterations_, bool generic_, int lane)
{
    Bits b = uni_bits(b_);
    FullArrays g; g.base = UNIP(gword *, g_.base); g.vcap = UNI(g_.vcap);
    const gbyte *in = UNIP(const gbyte *, in_);
    const uint64_t bbase = uni64(bbase_);
    const uint32_t count = UNI(count_);
    const bool final = UB(final_), generic = UB(generic_);
    const int iterations = (int)UNI(iterations_);
    DLds &s = g_lds;
    for (int i = generic ? -iterations : 0;;) {
        if (count) { FPROF(2); full_forward<WAVES>(g, in, bbase, count, lane); FPROF(4); full_backward(g, in, bbase, count, lane); FPROF(5); }
        else {
            for (int k = lane; k < 320; k += 64) s.freq[k] = k == 256 ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        }
        build_tree(s.freq, 286, 15, s.ll, lane);
        build_tree(s.freq + 288, 30, 15, s.dl, lane);
        ++i;
        FPROF(6);
        if (!(i < iterations)) break;
        full_depths_update(lane);
    }
    b = uni_bits(write_tables(b, final, lane));
    FPROF(7);
    // writeBlock(with:) (:661-707): the path's terms, 64 vertices at a time
    uint32_t pb_next = (uint32_t)lane < count ? g.pathb_()[lane] : 0u, st_next = (uint32_t)lane < count ? g.step_()[lane] : 0u,
             lt_next = (uint32_t)lane < count ? in[bbase + lane] : 0u;
    for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        const uint32_t v = b0 + lane;
        const bool on = v < count && pb_next != 0;
        const uint32_t st = st_next, lt = lt_next;
        {
            const uint32_t vn = v + 64;
            const bool inn = vn < count;
            pb_next = inn ? g.pathb_()[vn] : 0u; st_next = inn ? g.step_()[vn] : 0u; lt_next = inn ? in[bbase + vn] : 0u;
        }
        uint64_t bits = 0; uint32_t nb = 0;
        if (on) {
            const uint32_t cnt = st >> 16, dd = (st >> 8) & 0xff;
            if (cnt == 1) bits = literal_bits(s, lt, nb);
            else {
                const uint32_t off = g.slots_()[(uint64_t)v * 30 + dd] >> 16, rd = run_decade(cnt);
                bits = match_bits(s, rd, run_extra_value(cnt, rd), dd, dist_extra_value(off, dd), nb);
            }
        }
        bulk_put(s, b, bits, nb, lane);
        maybe_drain(s, b, lane);
    }
    put(s, b, s.lcode[256], s.ll[256], lane);
    maybe_drain(s, b, lane);
    FPROF(8);
    // resetGraph -> Depths.generalize (Depths.swift:88-98)
    for (uint32_t i = lane; i < 542; i += 64) {
        const uint32_t x = g_full.depths[i], d = depth_default(i);
        g_full.depths[i] = (uint8_t)((x & d) + ((x ^ d) >> 1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    return b;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void deflate_full_kernel(const DeflateJob *__restrict__ jobs, spng_result *__restrict__ results)
{
    DLds &s = g_lds;
    const DeflateJob *jp = jobs + blockIdx.x;
    // Which of the four waves is the stream's main wave rotates with the workgroup: the waves of a workgroup go one to
    // each SIMD, and the two workgroups a CU holds (placed 256 apart) should not keep their busy waves on the same one.
    const int lane = threadIdx.x & 63;
    const int wave = WAVES > 1 ? (int)UNI(((threadIdx.x >> 6) - (blockIdx.x + (blockIdx.x >> 8))) & 3) : 0;
    if (WAVES > 1 && wave != 0) {
        // helper waves: they serve the forward passes wave 0 announces (forward_body) until it says it is done
        FullArrays gh;
        gh.base = (gword *)uni64((uint64_t)jp->graph); gh.vcap = UNI(jp->graph_vertices);
        const gbyte *inh = (const gbyte *)uni64((uint64_t)jp->src);
        for (;;) {
            __syncthreads();
            if (UNI(g_old.fw_cmd) != FW_PASS) return;
            forward_body<WAVES>(gh, inh, uni64(g_old.fw_bbase), UNI(g_old.fw_count), lane, wave);
            __syncthreads();
        }
    }
    const gbyte *in = (const gbyte *)uni64((uint64_t)jp->src);
    const uint64_t n = uni64(jp->src_len);
    gword *ring = (gword *)uni64((uint64_t)jp->ring);
    struct { gbyte *dst; uint64_t dst_cap; int32_t format, level; uint32_t image; } job = {
        (gbyte *)uni64((uint64_t)jp->dst), uni64(jp->dst_cap), (int32_t)UNI(jp->format), (int32_t)UNI(jp->level), UNI(jp->image) };
#ifdef SPNG_DEFLATE_PROF
    if (lane < 12) g_prof[lane] = lane == 11 ? __builtin_readcyclecounter() : 0;
#endif
    const uint32_t vcap = UNI(jp->graph_vertices);             // vertices the scratch arrays hold
    FullArrays g;
    {
        g.base = (gword *)uni64((uint64_t)jp->graph); g.vcap = vcap;
    }
    // DeflatorSearch.init(level:) (:13-35), full rows
    const int lv = job.level > 13 ? 13 : job.level;
    const int attempts = lv == 8 ? 14 : lv == 9 ? 20 : lv == 10 ? 30 : lv == 11 ? 60 : lv == 12 ? 100 : 0x7fffffff;
    const int goal = lv == 8 ? 20 : lv == 9 ? 32 : lv == 10 ? 50 : lv == 11 ? 80 : lv == 12 ? 133 : 258;
    const int iterations = lv - 7;
    const uint32_t wmask = (1u << UNI(jp->exponent)) - 1;

    for (int i = lane; i < OUTB / 4; i += 64) s.out32[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    Bits b = {0, 0, 0, 0, job.dst, job.dst_cap, false};
    if (job.format == SPNG_FORMAT_ZLIB) {
        // StreamHeader.write (StreamHeader.swift:56-62)
        const uint32_t unpaired = (UNI(jp->exponent) - 8) << 4 | 0x08;
        const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
        put(s, b, check << 8 | unpaired, 16, lane);
    } else if (job.format == SPNG_FORMAT_GZIP) {
        // Gzip.StreamHeader.write (Gzip.StreamHeader.swift:84-96): sigil, method 8, no flags, MTIME 0, XFL 0, OS 255;
        // the trailer (CRC-32, byte count) is appended by gzip.hip
        put(s, b, 0x8b1f, 16, lane); put(s, b, 0x0008, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0xff00, 16, lane);
    }
    for (int i = lane; i <= (1 << HBITS); i += 64) g_sea.head[i] = NONE;
    for (uint32_t i = lane; i < 542; i += 64) g_full.depths[i] = (uint8_t)depth_default(i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    uint32_t accS = 0, accI = 0;
    uint32_t count = 0, limit = 2048;
    bool generic = true;
    uint64_t bbase = 0;
    auto unfilled = [&]() { return (int)limit - 1 - (int)count; };
    auto close_block = [&](bool final) {
        const uint32_t doubled = 2 * limit < (1u << 21) ? 2 * limit : 1u << 21;     // trees(iterations:) :229
        b = uni_bits(full_block<WAVES>(b, g, in, bbase, count, final, iterations, generic, lane));
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
        uint32_t key_next = load_key(in, n, (uint64_t)lane);      // (keys travel one batch ahead of their insertion)
        auto insert_upto = [&](uint64_t target) {
            while (inserted < target && inserted < n) {
                const uint32_t key = key_next;
                key_next = load_key(in, n, inserted + 64 + lane);
                insert_batch(g_sea.head, in, n, ring, inserted, key, accS, accI, lane);
                inserted = uni64(inserted + 64);
            }
        };
        while (w < last_main) {
            FPROF(2);
            insert_upto(w + 192);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FPROF(0);
            // ---- every candidate of positions w + lane and w + 64 + lane becomes an edge (DeflatorWindow.match, :132-212)
            const uint64_t pA = w + lane, pB = pA + 64;
            const bool liveA = pA < last_main, liveB = pB < last_main;
            const uint32_t keyA = liveA ? load32(in + pA) : 0u, keyB = liveB ? load32(in + pB) : 0u;
#pragma unroll
            for (int d = 0; d < 30; ++d) { g_sea.cslot[0][d * 64 + lane] = 0; g_sea.cslot[1][d * 64 + lane] = 0; }
            uint32_t extA = 1, extB = 1;
            chain_walk2(in, ring, n, pA, pB, liveA, liveB, keyA, keyB, wmask, attempts, goal,
                        [&](int which, uint32_t dist, uint32_t run) {
                            uint32_t &ext = which ? extB : extA;
                            ext = run > ext ? run : ext;
                            uint32_t *slot = &g_sea.cslot[which][dist_decade(dist) * 64 + lane];
                            if (run > (*slot & 0xffff)) *slot = dist << 16 | run;
                        });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            FPROF(1);
            // ---- which positions are vertices with edges (Stream.compress full, :344-400)
            uint32_t t = 0;
            while (t < 128 && w + t < last_main) {
                const uint32_t hf = t >> 6, tl = t & 63;
                const uint32_t extent = hf ? extB : extA;
                if (tl == 0 && !__ballot(extent > 100) && w + t + 64 <= last_main && unfilled() >= 64) {
                    // a whole half with nothing to skip and no block to close: 64 vertices at once
                    if (count == 0) bbase = w + t;
                    const bool has = extent > 1;               // (a candidate matched: at least the four key bytes)
                    g.flag_()[count + lane] = has ? 1 : 0;
                    if (__ballot(has))
                        for (uint32_t i = lane; i < 64 * 30; i += 64) { const uint32_t vtx = i / 30, d = i - vtx * 30; g.slots_()[(uint64_t)count * 30 + i] = g_sea.cslot[hf][d * 64 + vtx]; }
                    count += 64;
                    t += 64;
                    continue;
                }
                if (!(unfilled() > 0)) close_block(false);
                if (count == 0) bbase = w + t;
                const int ext = __builtin_amdgcn_readlane((int)extent, (int)tl);
                if (ext > 1) { if (lane < 30) g.slots_()[(uint64_t)count * 30 + lane] = g_sea.cslot[hf][lane * 64 + tl]; }
                if (lane == 0) g.flag_()[count] = ext > 1 ? 1 : 0;
                count += 1;
                int skip = ext - 100 < unfilled() ? ext - 100 : unfilled();
                if (skip > 0) {
                    for (uint32_t i = lane; i < (uint32_t)skip; i += 64) g.flag_()[count + i] = 0;
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
            if (lane == 0) g.flag_()[count] = 0;
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
    if (WAVES > 1) {
        if (lane == 0) g_old.fw_cmd = FW_EXIT;                     // the helper waves leave
        __syncthreads();
    }
#ifdef SPNG_DEFLATE_PROF
    if (lane == 0 && blockIdx.x == 0)
        printf("deflate_full prof Mcycles: insert %llu search %llu register %llu forward %llu backward %llu trees %llu tables %llu emit %llu\n",
               g_prof[0] >> 20, g_prof[1] >> 20, g_prof[2] >> 20, g_prof[4] >> 20, g_prof[5] >> 20, g_prof[6] >> 20, g_prof[7] >> 20, g_prof[8] >> 20);
#endif
    if (lane == 0) {
        spng_result &res = results[job.image];
        res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
        res.written = b.total; res.consumed = n; res.aux[0] = res.aux[1] = 0;
    }
}

// =====================================================================================================================
// levels >= 8, round 4: the search and the parse in kernels of their own
// =====================================================================================================================
// deflate_full_kernel above keeps a stream on ONE wave: insertion, chain walks, the shortest path, the trees and the bits, one
// after the other -- a quarter of the chip's SIMDs with a single, latency-bound wave each, and a per-stream graph scratch (130
// bytes per vertex) that lets ~370 streams of 64 MiB in at a time.  But the candidates of a position are a pure function of the
// input (the design note at the top of this file), and the block boundaries of the full search are fixed vertex counts
// (2047, 4095, ... 2^21 - 1: LZ77.DeflatorMatches.swift:229): nothing about them has to wait for the parse.  So a batch now
// goes through in ROUNDS of up to 2^21 vertices per stream (the small blocks of a stream's start together, then one block at
// the cap per round), each round two launches:
//   * the search kernel (round 4: dfl2_search_kernel, links in HBM; round 5: dfl3_search_kernel below, the window in LDS) --
//     every stream's round cut into chunks, a workgroup per chunk: wave 0 inserts (32 KiB of warm-up in front of the chunk),
//     the others take 64 positions at a time behind it, walk the chains and leave, per position, the longest run
//     seen and its candidates -- ONE packed word each (position-in-batch, distance, run), only for positions that have any:
//     a batch of 64 positions takes what it needs from a pool shared by all streams.  Incompressible input leaves no words.
//   * dfl2_parse_kernel -- one wave per stream, 38 KB of LDS (four per CU): the skip rule of runs > 100 (which vertices lose
//     their edges: the only thing about the graph that IS sequential), then per block the forward passes, the back-trace,
//     the trees and the bits as before, its state (bit writer, symbol costs, block limit) kept in HBM from round to round.
//     Forward pass: what a vertex offers each target length -- the cheapest decade that reaches it -- does not depend on the
//     vertex's depth, so it is tabulated for the 64 vertices of a batch at once (LDS atomic min per candidate word, one
//     suffix-min sweep over the lengths, all lanes busy) and the dependent part per vertex shrinks to one row read, one add
//     and one ds_min_u64 per lane.
// The old kernel stays as the path for streams the pool could not serve (dfl2 marks them; api.hip runs them afterwards).
static constexpr uint32_t D2_RV = 1u << 21;                     // vertices per stream and round
static constexpr uint32_t D2_PCOLS = 64, D2_PSTRIDE = 65;       // offer table: lengths 3 .. 66, rows padded against bank conflicts
__shared__ uint32_t g_ptab[66 * D2_PSTRIDE];                    // (parse kernel only; two rows of padding: a group of three vertices is read blind)
__host__ __device__ inline uint64_t d2_round_end(uint64_t pos, uint32_t limit, uint64_t n, bool more = false)
{
    // the blocks of a round: as many whole blocks from `pos` on as fit D2_RV vertices (at least one).  more: the input goes on
    // behind n -- a block is only taken when it is full AND all of its vertices can see their whole look-ahead (none otherwise).
    const uint64_t upto = more ? (n > 261 ? n - 261 : 0) : n;
    uint64_t end = pos;
    uint32_t lim = limit;
    for (;;) {
        const uint64_t room = upto > end ? upto - end : 0, size = (uint64_t)(lim - 1) < room ? (uint64_t)(lim - 1) : room;
        if (more && size < (uint64_t)(lim - 1)) break;
        if (end > pos && end - pos + size > D2_RV) break;
        end += size;
        if (!more && end >= n) break;                           // (>=, and no room at all: a position behind the input -- a state
        if (!room) break;                                       //  that does not belong to this input -- must not spin here)
        lim = 2 * lim < (1u << 21) ? 2 * lim : 1u << 21;
    }
    return end;
}
// rounds a call takes from (pos, limit) on, and where it leaves them
uint32_t deflate2_plan(uint64_t n, bool more, uint64_t &pos, uint32_t &lim)
{
    if (!more && n < 3) { pos = n; return 1; }
    uint32_t rounds = 0;
    for (;;) {
        const uint64_t end = d2_round_end(pos, lim, n, more);
        if (end == pos) break;
        for (uint64_t at = pos; at < end;) {                    // (the limit as the blocks of the round leave it)
            const uint64_t size = (uint64_t)(lim - 1) < end - at ? (uint64_t)(lim - 1) : end - at;
            at += size;
            if (more || at < n) lim = 2 * lim < (1u << 21) ? 2 * lim : 1u << 21;
        }
        pos = end; ++rounds;
        if (!more && pos >= n) break;
    }
    return rounds ? rounds : 1;                                 // (one launch at least: it reports where the stream stands)
}
uint32_t deflate2_rounds(uint64_t n)
{
    uint64_t pos = 0; uint32_t lim = 2048;
    return deflate2_plan(n, false, pos, lim);
}
uint64_t deflate_state_bytes() { return ((sizeof(D1State) > sizeof(D2State) ? sizeof(D1State) : sizeof(D2State)) + 255) & ~(uint64_t)255; }
uint64_t deflate2_vertices(uint64_t n) { return ((n < D2_RV ? n : D2_RV) + 63) / 64 * 64 + 128; }

// =====================================================================================================================
// round 5: the search with its window in LDS (every level)
// =====================================================================================================================
// Round 4's search kernel (dfl2_search_kernel, gone) hopped through HBM: a link is a 4-byte word in a 256 KiB ring per workgroup, a hop two dependent loads
// of a microsecond each, and with 13-bit bucket heads over a 32 K window a position of incompressible input walks ~4 foreign
// bucket members to find nothing (10 GB/s with the whole chip on it).  Here a workgroup is a whole CU -- sixteen waves, 141 KB
// of LDS -- and everything a hop touches lives in LDS:
//   * `in`:   the input bytes of the last D3_R = 34816 positions (the 32 K window + 2 K of lead), a ring with its first 320
//             bytes mirrored behind its end so that a compare never wraps;
//   * `link`: per position the distance to the previous position of its bucket (16 bits, 0 = none): no tags -- a candidate is
//             verified against the input itself, four bytes at its ring slot;
//   * `head`: 2^13 bucket heads (16 bits, positions mod 2^16; the inserter moves heads that fell out of the window to a fixed
//             distance behind the present every 2^14 positions, so that none ever aliases a young one: the note in its loop).
// Wave 0 stages the bytes (global -> LDS, three 256-byte steps in flight) and inserts, 64 positions at a time: when the 64
// buckets are all different -- what a read-back of the heads tells -- a batch costs three LDS round trips; the radix match over
// the hash bits is the slow path.  The other fifteen waves (and wave 0 once it is done) claim batches of 64 positions behind
// it and walk the chains (LZ77.DeflatorWindow.match, :132-212: attempts / goal / window of the level).  Distances only grow
// along a chain, so the candidates of one distance decade are neighbours: the "longest run per decade, closest first"
// (DeflatorMatches.set(edge:), :183-194) is a running maximum in registers and a word whenever the decade changes -- no table
// of thirty slots per lane (round 4: 7.7 KB of LDS per wave, the reason for three workgroups per CU).  Words go to a small
// per-wave area in global memory and from there, when the batch's total is known, to the pool in the same lane-major order as
// before.  Levels 0-7 (FULL = false) leave one word per position instead: the first strictly longest run > 5 and its distance
// (DeflatorWindow.match :145-208 as Stream.compress greedy / lazy asks it), for dfl3_parse_kernel.
// (121 KB: a search workgroup and ONE parse wave -- 38 KB -- share a CU, so that in batches of up to 256 streams the search of
//  round r + 1 runs beside the parse of round r; 2^14 heads and 4 K of lead -- 144 KB -- measured the same on incompressible input)
static constexpr uint32_t D3_R = 34816, D3_MIR = 320;           // ring positions (a multiple of 256), mirrored bytes
// (levels 0-7 have no 38 KB parse wave to share the CU with -- the walk wave needs 8 KB --, and their searchers' batches differ
//  much more in cost (64 to 100 candidates each): 8 K positions of lead instead of 2 K keep the inserter from waiting on the
//  slowest batch and the other searchers from waiting on the inserter -- 139 KB)
#ifndef SPNG_D3_R_FAST
#define SPNG_D3_R_FAST 40960
#endif
static constexpr uint32_t D3_R_FAST = SPNG_D3_R_FAST;
static constexpr uint32_t D3_FAR = 40000;                       // where a head too old for the window is kept (+ 2^14 between two sweeps: < 2^16)
#ifndef SPNG_D3_WAVES
#define SPNG_D3_WAVES 16
#endif
#ifndef SPNG_D3_HBITS
#define SPNG_D3_HBITS 13
#endif
// ds_mskor_rtn_b32: MEM = (MEM & ~mask) | data, the old dword back -- an exchange of HALF a dword, which is what a 16-bit bucket head is.
__device__ __forceinline__ uint32_t lds_mskor(uint16_t *p, uint32_t mask, uint32_t data)
{
#ifdef SPNG_EMU
    // (fibers run one after the other between two meetings of their wave, lane 0 first: the lanes of one address in ascending order)
    uint32_t *w = (uint32_t *)((uintptr_t)p & ~(uintptr_t)3);
    const uint32_t old = *w;
    *w = (old & ~mask) | data;
    return old;
#else
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)p & ~3u;
    uint32_t r;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(mask), "v"(data) : "memory");
    return r;
#endif
}
// the bucket's old head out, `mine` in: one LDS operation
__device__ __forceinline__ uint32_t lds_exchange16(uint16_t *p, uint32_t mine)
{
    const uint32_t sh = ((uint32_t)(uintptr_t)p & 2u) << 3;
    return (lds_mskor(p, 0xffffu << sh, (mine & 0xffffu) << sh) >> sh) & 0xffffu;
}
// four of them, one wait: the LDS serves a wave's operations in the order it issued them, so the four batches of a quad of 256
// positions keep their order and the inserter pays one round trip for the quad instead of four
__device__ __forceinline__ void lds_exchange16_x4(uint16_t *p0, uint16_t *p1, uint16_t *p2, uint16_t *p3, const uint32_t (&mine)[4], uint32_t (&old)[4])
{
#ifdef SPNG_EMU
    uint16_t *p[4] = {p0, p1, p2, p3};
    for (int k = 0; k < 4; ++k) { old[k] = lds_exchange16(p[k], mine[k]); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }   // (the emulator's lanes meet here: batch k before batch k + 1)
#else
    uint16_t *p[4] = {p0, p1, p2, p3};
    uint32_t a[4], m[4], d[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t full = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)p[k];
        sh[k] = (full & 2u) << 3; a[k] = full & ~3u; m[k] = 0xffffu << sh[k]; d[k] = (mine[k] & 0xffffu) << sh[k];
    }
    uint32_t r0, r1, r2, r3;
    asm volatile("ds_mskor_rtn_b32 %0, %4, %5, %6\n\tds_mskor_rtn_b32 %1, %7, %8, %9\n\tds_mskor_rtn_b32 %2, %10, %11, %12\n\t"
                 "ds_mskor_rtn_b32 %3, %13, %14, %15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a[0]), "v"(m[0]), "v"(d[0]), "v"(a[1]), "v"(m[1]), "v"(d[1]), "v"(a[2]), "v"(m[2]), "v"(d[2]), "v"(a[3]), "v"(m[3]), "v"(d[3])
                 : "memory");
    old[0] = (r0 >> sh[0]) & 0xffffu; old[1] = (r1 >> sh[1]) & 0xffffu; old[2] = (r2 >> sh[2]) & 0xffffu; old[3] = (r3 >> sh[3]) & 0xffffu;
#endif
}
// Does this device's LDS serve the lanes that exchange on ONE halfword in ascending lane order, whatever the lanes on the dword's
// other half do?  (The ISA does not promise it; every part seen so far does.)  d3_lds_order_kernel -- launched once per context,
// spng_create -- tries five address patterns and sets the flag the inserter reads; where the answer is no, d3_insert keeps the
// read-back form of round 5.
#ifdef SPNG_EMU
static uint32_t g_d3_xchg = getenv("EMU_D3_READBACK") ? 0u : 1u;      // (tests run both forms of the inserter)
#else
__device__ uint32_t g_d3_xchg = 0;
#endif
__device__ __forceinline__ uint32_t d3_probe_cell(int pattern, uint32_t lane)
{
    return pattern == 0 ? 0u : pattern == 1 ? lane & 1u : pattern == 2 ? lane >> 5 : pattern == 3 ? (lane * 7u) & 3u : (lane * 5u) % 11u;
}
__global__ __launch_bounds__(64) void d3_lds_order_kernel(uint32_t allow)
{
    __shared__ uint16_t cell[64];
    const int lane = threadIdx.x;
    bool ok = true;
#pragma unroll 1
    for (int pattern = 0; pattern < 5; ++pattern) {
        cell[lane] = (uint16_t)(1000u + (uint32_t)lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        const uint32_t a = d3_probe_cell(pattern, (uint32_t)lane);
        const uint32_t got = lds_exchange16(&cell[a], (uint32_t)lane + 1u);
        // expected: the value of the nearest lower lane with the same cell, or the cell's first content
        uint32_t want = 1000u + a;
        for (int l = 0; l < lane; ++l)
            if (d3_probe_cell(pattern, (uint32_t)l) == a) want = (uint32_t)l + 1u;
        ok = ok && got == want;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        // ... and what stays is the highest lane's
        uint32_t last = 1000u + (uint32_t)lane;
        for (int l = 0; l < 64; ++l)
            if (d3_probe_cell(pattern, (uint32_t)l) == (uint32_t)lane) last = (uint32_t)l + 1u;
        ok = ok && cell[lane] == last;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    }
    const bool all = __ballot(!ok) == 0;
    if (lane == 0) g_d3_xchg = all && allow ? 1u : 0u;
}
#ifndef SPNG_EMU
hipError_t launch_deflate3_probe(hipStream_t stream)
{
    d3_lds_order_kernel<<<1, 64, 0, stream>>>(getenv("SPNG_D3_READBACK") ? 0u : 1u);     // (the A/B switch of tools/probe_deflate2.py)
    return hipGetLastError();
}
// what the probe found on the current device (after the stream it ran on has been waited for)
hipError_t deflate3_probe_result(uint32_t *ordered)
{
    return hipMemcpyFromSymbol(ordered, HIP_SYMBOL(g_d3_xchg), sizeof(uint32_t), 0, hipMemcpyDeviceToHost);
}
#endif

template <uint32_t R_>
struct D3LdsT {
    static constexpr uint32_t R = R_;                          // ring positions (a multiple of 256)
    union { uint8_t in[R_ + D3_MIR]; uint32_t in32[(R_ + D3_MIR) / 4]; };
    uint16_t link[R_];
    alignas(4) uint16_t head[(1u << SPNG_D3_HBITS) + 64];       // (+ a spare slot for idle lanes)
    uint32_t cur[SPNG_D3_WAVES];                                // the batch each wave is at (~0: none any more)
    uint32_t staged, inserted, next, pad;                       // positions (relative to the warm-up's first) below which bytes / links stand; batches claimed
#ifdef SPNG_D3_PROF
    unsigned long long prof[16];                                // tuning builds: cycles / counts of the phases, summed over the waves
#endif
};
#ifdef SPNG_D3_PROF
#define D3P_T0() const unsigned long long d3p_t0 = __builtin_readcyclecounter()
#define D3P_ADD(k) do { if (lane == 0) atomicAdd(&s.prof[k], __builtin_readcyclecounter() - d3p_t0); } while (0)
#define D3P_CNT(k, v) do { if (lane == 0) atomicAdd(&s.prof[k], (unsigned long long)(v)); } while (0)
#else
#define D3P_T0() ((void)0)
#define D3P_ADD(k) ((void)0)
#define D3P_CNT(k, v) ((void)0)
#endif
typedef D3LdsT<D3_R> D3Lds;                                     // levels >= 8: 121 KB, beside a parse wave
typedef D3LdsT<D3_R_FAST> D3LdsFast;                           // levels 0-7: 8 K positions of lead
__shared__ __attribute__((aligned(16))) D3Lds g_d3;
__shared__ __attribute__((aligned(16))) D3LdsFast g_d3f;

// four / eight input bytes at ring offset `off` (any alignment; the mirror makes them contiguous)
template <class L>
__device__ __forceinline__ uint32_t d3_u32(const L &s, uint32_t off)
{
    const uint32_t w = off >> 2;
    return __builtin_amdgcn_alignbyte(s.in32[w + 1], s.in32[w], off & 3);
}
template <class L>
__device__ __forceinline__ uint64_t d3_u64(const L &s, uint32_t off)
{
    const uint32_t w = off >> 2, a = s.in32[w], b = s.in32[w + 1], c = s.in32[w + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(c, b, off & 3) << 32 | __builtin_amdgcn_alignbyte(b, a, off & 3);
}
template <class L>
__device__ __forceinline__ uint32_t d3_common_prefix(const L &s, uint32_t q, uint32_t p, uint32_t limit)
{
    uint32_t i = 0;
    while (i + 8 <= limit) {
        const uint64_t x = d3_u64(s, q + i) ^ d3_u64(s, p + i);
        if (x) return i + ((uint32_t)__builtin_ctzll(x) >> 3);
        i += 8;
    }
    if (i + 4 <= limit) {
        const uint32_t x = d3_u32(s, q + i) ^ d3_u32(s, p + i);
        if (x) return i + ((uint32_t)__builtin_ctz(x) >> 3);
        i += 4;
    }
    while (i < limit && s.in[q + i] == s.in[p + i]) ++i;
    return i;
}

// Hash insertion of 64 positions (LZ77.DeflatorWindow.update, :78-128) with heads and links in LDS.  rel: the batch's first
// position relative to the warm-up's first, idx: its ring slot.  A position's link is the distance to the previous position of
// its bucket, 0 when that is 32768 or more away (heads are positions mod 2^16; the inserter's sweep keeps old ones from
// aliasing young ones).  A candidate is compared with the key itself: a bucket's chain holds other keys too.
// (the emulator runs a wave's lanes one after another between two wave builtins: where the hardware's "every lane reads, then
//  every lane writes" is relied on, its lanes have to meet; the hardware serves a wave's LDS operations in order as they are)
#ifdef SPNG_EMU
#define D3_MEET() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#else
#define D3_MEET() ((void)0)
#endif
template <class L>
__device__ __forceinline__ void d3_insert(L &s, uint32_t rel, uint32_t idx, uint32_t key, uint64_t p0, uint32_t pm, uint64_t n, bool sum, uint32_t &accS, uint32_t &accI, bool xchg, int lane)
{
    const uint64_t p = p0 + lane;
    const bool live = p + 4 <= n;                              // the last three positions never start a match
    if (p < n && sum) {                                        // Adler-32 accumulators (pm: this lane's position mod 65521; accI is reduced by the caller)
        const uint32_t byte = key & 0xff;
        accS += byte;
        accI += pm * byte;
    }
    const uint32_t h = (key * 0x9E3779B1u) >> (32 - SPNG_D3_HBITS);
    const uint32_t mine = (rel + (uint32_t)lane) & 0xffffu, spare = (1u << SPNG_D3_HBITS) + (uint32_t)lane;
    const uint32_t slot = live ? h : spare;
    // (the store and the read-back are atomic operations: what comes back is what SOME lane of the bucket stored -- plain
    //  accesses the compiler forwards from this lane's own store, and the duplicate test below is never true)
    if (xchg) {
        // Round 6: ONE LDS operation.  ds_mskor_rtn_b32 on the bucket's half of its dword hands every lane what stood in its bucket and
        // leaves its own position there; the lanes of one bucket are served in ascending lane order (what d3_lds_order_kernel has seen
        // this device do, or this path is not taken), so a lane gets the nearest lower lane of its bucket -- or the bucket's old head --
        // and the highest lane stays: the read-back, the ballot and the radix match below are this one instruction.  (The inserter
        // is ONE wave per CU and every searcher waits for it: 750 cycles of the 1465 a batch of 64 positions took were its
        // dependent LDS round trips.  Heads stay 16 bits wide: 32-bit cells for a plain exchange cost the CU its second workgroup.)
        const uint32_t old = lds_exchange16(&s.head[slot], mine);
        const uint32_t d = (mine - old) & 0xffffu;
        s.link[idx + lane] = (uint16_t)(live && d <= 32767u ? d : 0u);
        D3_MEET();
        return;
    }
    const uint32_t old = s.head[slot];
    D3_MEET();
    __hip_atomic_store(&s.head[slot], (uint16_t)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    D3_MEET();
    const uint32_t back = __hip_atomic_load(&s.head[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t d = (mine - old) & 0xffffu;
    if (__ballot(back != mine)) {
        // two positions of one bucket in the batch: the nearest lower lane of the same bucket is the previous position, the
        // highest lane of a bucket its new head (radix match over the hash bits: one ballot per bit)
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int k = 0; k < SPNG_D3_HBITS; ++k) {
            const unsigned long long bk = __ballot((h >> k) & 1);
            same &= (h >> k) & 1 ? bk : ~bk;
        }
        const unsigned long long lower = (1ull << lane) - 1;
        const unsigned long long below = same & lower, above = same & ~lower & ~(1ull << lane);
        if (live && below) d = (uint32_t)lane - (uint32_t)(63 - __clzll((long long)below));
        D3_MEET();
        s.head[live && !above ? h : spare] = (uint16_t)mine;
    }
    s.link[idx + lane] = (uint16_t)(live && d <= 32767u ? d : 0u);
    D3_MEET();
}

// The four batches of a quad of 256 positions (the exchange form; idx: a multiple of 256, so the quad does not wrap the ring): four
// keys, four hashes, four exchanges under one wait, four links.  first_sum / end_sum: the positions [first_sum, end_sum) are summed.
template <class L>
__device__ __forceinline__ void d3_insert_quad(L &s, uint32_t rel, uint32_t idx, uint64_t p0, uint32_t &pm, uint64_t n, uint32_t first_sum, uint32_t end_sum,
                                               uint32_t &accS, uint32_t &accI, int lane)
{
    uint32_t key[4], mine[4], old[4], h[4];
    bool live[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) key[k] = d3_u32(s, idx + 64u * k + (uint32_t)lane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r = rel + 64u * k + (uint32_t)lane;
        const uint64_t p = p0 + 64u * k + (uint32_t)lane;
        live[k] = p + 4 <= n;
        if (p < n && r >= first_sum && r < end_sum) {
            const uint32_t byte = key[k] & 0xff;
            accS += byte;
            accI += pm * byte;
        }
        pm = pm + 64 >= 65521 ? pm + 64 - 65521 : pm + 64;
        const uint32_t hk = (key[k] * 0x9E3779B1u) >> (32 - SPNG_D3_HBITS);
        h[k] = live[k] ? hk : (1u << SPNG_D3_HBITS) + (uint32_t)lane;
        mine[k] = r & 0xffffu;
    }
    lds_exchange16_x4(&s.head[h[0]], &s.head[h[1]], &s.head[h[2]], &s.head[h[3]], mine, old);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t d = (mine[k] - old[k]) & 0xffffu;
        s.link[idx + 64u * k + (uint32_t)lane] = (uint16_t)(live[k] && d <= 32767u ? d : 0u);
    }
    D3_MEET();
}

// One chunk [c0, c1) of a stream's positions on one workgroup.  rb: the round's first position (records are kept in round
// coordinates); extra: the last `extra` positions are searched but not summed (the next chunk's: a lazy parse looks one ahead).  FULL: vinfo / bbase / bwords / pool (what dfl2_parse_kernel reads); else match[position - rb] = run << 16 | distance
// (0: no run > 5).  temp: SPNG_D3_WAVES x 30 x 64 words of global scratch of this workgroup (FULL).
template <bool FULL, class L>
__device__ __forceinline__ void d3_search_chunk(L &s, const gbyte *in, uint64_t n, uint64_t rb, uint64_t c0, uint64_t c1, uint32_t extra, int attempts, int goal, uint32_t wmask,
                                                uint32_t *adlerS, uint32_t *adlerI, uint32_t *fail,
                                                uint16_t *vinfo, uint64_t *bbase, uint32_t *bwords, uint32_t *pool, unsigned long long *pool_next, uint64_t pool_cap,
                                                uint32_t *temp, uint32_t *match)
{
    constexpr uint32_t D3R = L::R;
    const int lane = threadIdx.x & 63, wave = (int)UNI(threadIdx.x >> 6);
    const uint64_t warm = (c0 >= 32768 ? c0 - 32768 : 0) & ~(uint64_t)255;       // first position entered into the window
    const uint32_t c0r = (uint32_t)(c0 - warm), c1r = (uint32_t)(c1 - warm);
    const uint32_t nr = n - warm < (1u << 30) ? (uint32_t)(n - warm) : 1u << 30;  // the input's end, relative (bytes behind it read as zero)
    const uint32_t stage_end = (c1r + 336 + 255) & ~255u;      // bytes staged in all (a batch looks 258 + 8 bytes ahead, + a dword of slack)
    const uint64_t last_main = n - 4 + 1;                      // positions 0 .. n-4 are searched
    const bool xchg = UNI(g_d3_xchg) != 0;                       // (read once: the inserter's loop keeps it in a scalar register)
    const uint32_t nbatches = (c1r - c0r + 63) / 64;
    for (uint32_t i = threadIdx.x; i < (1u << SPNG_D3_HBITS) + 64; i += SPNG_D3_WAVES * 64) s.head[i] = (uint16_t)(0u - D3_FAR);   // (far behind position 0: no link)
    if (threadIdx.x < SPNG_D3_WAVES) s.cur[threadIdx.x] = threadIdx.x == 0 ? ~0u : 0u;
    if (threadIdx.x == 0) { s.staged = 0; s.inserted = 0; s.next = 0; }
#ifdef SPNG_D3_PROF
    if (threadIdx.x < 16) s.prof[threadIdx.x] = 0;
    const unsigned long long d3p_start = __builtin_readcyclecounter();
#endif
    __syncthreads();

    if (wave == 0) {
        // ---- the stager and inserter
        __builtin_amdgcn_s_setprio(3);                         // (fifteen waves wait on this one)
        uint32_t accS = 0, accI = 0;
        auto fetch = [&](uint32_t at) -> uint32_t {            // the four bytes at relative offset at + 4 lane
            const uint32_t off = at + 4u * (uint32_t)lane;
            if (off + 4 <= nr) return load32(in + warm + off);
            uint32_t v = 0;
            for (uint32_t k = 0; k < 4; ++k) if (off + k < nr) v |= (uint32_t)in[warm + off + k] << (8 * k);
            return v;
        };
        auto put_step = [&](uint32_t sidx, uint32_t v) {       // 256 bytes into ring slot sidx (a multiple of 256)
            const uint32_t o = sidx + 4u * (uint32_t)lane;
            s.in32[o >> 2] = v;
            if (o < D3_MIR) s.in32[(D3R + o) >> 2] = v;
        };
        uint32_t staged = 0, sidx = 0;                         // bytes in the ring; the ring slot of the next step
        uint32_t q0 = fetch(0), q1 = fetch(256), q2 = fetch(512), q3 = fetch(768);
        put_step(0, q0); staged = 256; sidx = 256;
        q0 = q1; q1 = q2; q2 = q3; q3 = fetch(1024);
        uint32_t fetched = 1280;                               // next offset to ask for
        uint32_t smin = 0;                                     // lowest batch some wave may still be at (as last looked up)
        uint32_t iidx = 0;
        uint32_t pm = (uint32_t)((warm + (uint32_t)lane) % 65521);      // this lane's position in the batch at hand, mod 65521
        // one more step of 256 bytes into the ring -- never beyond what the slowest searcher still needs: the slot of position x
        // is x + D3R's, and a batch looks back 32767 positions
        auto stage_step = [&]() {
            {
                D3P_T0();
                SpinGuard guard;
                while (staged + 256 > c0r + 64u * smin + (D3R - 32768)) {
                    uint32_t v = (uint32_t)lane < SPNG_D3_WAVES ? __hip_atomic_load(&s.cur[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                               : lane == SPNG_D3_WAVES ? __hip_atomic_load(&s.next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : ~0u;
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64); v = o < v ? o : v; }
                    smin = UNI(v);
                    if (staged + 256 <= c0r + 64u * smin + (D3R - 32768)) break;
                    __builtin_amdgcn_s_sleep(4);
                    guard.tick();
                }
                D3P_ADD(0);
                put_step(sidx, q0);
                staged += 256; sidx = sidx + 256 >= D3R ? 0 : sidx + 256;
                q0 = q1; q1 = q2; q2 = q3; q3 = fetch(fetched); fetched += 256;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_store(&s.staged, staged, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        uint32_t key_next = 0;
        for (uint32_t i = 0; i < c1r; i += 256) {
            if (i && (i & 16383) == 0) {
                // Heads are positions mod 2^16: one that nobody has touched for 2^16 positions would read as a young one, and its
                // link would send every walk that meets it down the chain of whatever stands there -- thousands of hops where
                // that is a long run of one byte (measured: the chunks that cost 30 x the others).  So every 2^14 positions the
                // heads older than the window are moved to D3_FAR behind the present: they stay "far" for ever, and a link is
                // exactly the distance to the bucket's previous member, or none.
                for (uint32_t k = (uint32_t)lane; k < (1u << SPNG_D3_HBITS); k += 64) {
                    const uint32_t age = (i - s.head[k]) & 0xffffu;
                    if (age >= 32768u) s.head[k] = (uint16_t)(i - D3_FAR);
                }
                D3_MEET();
            }
            if (staged < stage_end) stage_step();              // the step behind this quad's positions: their keys reach three bytes into it
            D3P_T0();
            const bool quad = xchg && i + 256 <= c1r;
            if (quad) {
                d3_insert_quad(s, i, iidx, warm + i, pm, n, c0r, c1r - extra, accS, accI, lane);
                iidx = iidx + 256 >= D3R ? 0 : iidx + 256;
            } else if (i == 0 || xchg) key_next = d3_u32(s, iidx + (uint32_t)lane);
#pragma unroll 1
            for (uint32_t k = 0; !quad && k < 256 && i + k < c1r; k += 64) {
                const uint32_t rel = i + k;
                const uint32_t key = key_next, nidx = iidx + 64 >= D3R ? 0 : iidx + 64;
                key_next = d3_u32(s, nidx + (uint32_t)lane);   // (keys travel a batch ahead: bytes below rel + 131, and i + 512 are staged)
                d3_insert(s, rel, iidx, key, warm + rel, pm, n, rel + (uint32_t)lane >= c0r && rel + (uint32_t)lane < c1r - extra, accS, accI, xchg, lane);
                iidx = nidx;
                pm = pm + 64 >= 65521 ? pm + 64 - 65521 : pm + 64;
            }
            accI %= 65521;                                     // (four products < 2^24 each on top of a reduced sum)
            D3P_ADD(1);
            const uint32_t done = i + 256 < c1r ? i + 256 : c1r;
            if (lane == 0) __hip_atomic_store(&s.inserted, done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        while (staged < stage_end) stage_step();               // what the last batches look ahead at
        // Adler-32 sums of the chunk (MRC32.swift:26-50 in the closed form of inflate.hip), added to the stream's
        uint32_t S = accS % 65521, I = accI % 65521;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
        if (lane == 0) { atomicAdd(adlerS, S % 65521); atomicAdd(adlerI, I % 65521); }
        __builtin_amdgcn_s_setprio(0);
    }
    // ---- the searchers: 64 positions at a time
    uint32_t *tw = FULL ? temp + (uint64_t)wave * (30 * 64) : nullptr;
    for (;;) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&s.next, 1u);
        b = UNI(b);
        if (b >= nbatches) break;
        if (lane == 0) __hip_atomic_store(&s.cur[wave], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t p0r = c0r + 64u * b, upto = p0r + 64 < c1r ? p0r + 64 : c1r;
        const uint32_t ahead = p0r + 336 < stage_end ? p0r + 336 : stage_end;
        {
            D3P_T0();
            SpinGuard guard;
            while (__hip_atomic_load(&s.inserted, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < upto ||
                   __hip_atomic_load(&s.staged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ahead) {
                __builtin_amdgcn_s_sleep(8);
                guard.tick();
            }
            D3P_ADD(2);
        }
        D3P_T0();
        uint32_t p_hops = 0, p_cmp = 0;
        const uint32_t rel = p0r + (uint32_t)lane, idx0 = UNI(p0r % D3R) + (uint32_t)lane, idx = idx0 >= D3R ? idx0 - D3R : idx0;
        const uint64_t p = warm + rel;
        const bool inchunk = rel < c1r, live = inchunk && p < last_main;
        const uint32_t key = d3_u32(s, idx);
        const uint32_t lim = n - p < 258 ? (uint32_t)(n - p) : 258u;
        // LZ77.DeflatorWindow.match (:132-212): the candidates of the key, most recent first
        uint32_t d = live ? s.link[idx] : 0u, acc = 0;
        int rem = attempts;
        bool first = true;
        uint32_t ext = FULL ? 1u : 5u, bestd = 1;              // longest run seen (levels 0-7: it must exceed 5, and its distance)
        uint32_t cdec = 0xff, crun = 0, cdist = 0, cnt = 0;    // FULL: the decade at hand, its longest run and that run's distance; words so far
        while (d) {
#ifdef SPNG_D3_PROF
            p_hops += 1;
#endif
            acc += d;
            if (acc > wmask || (!first && acc >= wmask)) break;
            const uint32_t cidx = idx >= acc ? idx - acc : idx + D3R - acc;
            const uint32_t e = s.link[cidx];
            if (d3_u32(s, cidx) == key) {
                // A run that does not exceed the longest one seen (FULL: of its decade) changes nothing -- only a strictly longer
                // one is taken, and the goal lies above the longest (a run at the goal ends the walk) -- so the byte at that
                // length is looked at first: one byte settles most candidates of a long chain.
                const uint32_t have = FULL ? (dist_decade(acc) == cdec ? crun : 0u) : ext;
                uint32_t run = 0;
                if (!(have && (have >= lim || s.in[cidx + have] != s.in[idx + have]))) {
                    run = d3_common_prefix(s, cidx, idx, lim);
#ifdef SPNG_D3_PROF
                    p_cmp += run / 8 + 1;
#endif
                }
                if (FULL) {
                    ext = run > ext ? run : ext;
                    const uint32_t dec = dist_decade(acc);
                    if (dec != cdec) {
                        if (cdec != 0xff) { tw[cnt * 64 + (uint32_t)lane] = (uint32_t)lane << 24 | cdist << 9 | crun; ++cnt; }
                        cdec = dec; crun = run; cdist = acc;
                    } else if (run > crun) { crun = run; cdist = acc; }    // (strict: the closest candidate of a decade stays)
                } else if (ext < run) { ext = run; bestd = acc; }          // (the first strictly longest run wins, :145-208)
                first = false; rem -= 1;
                if (!(rem > 0 && goal > (int)run)) break;
            }
            d = e;
        }
        D3P_ADD(3);
#ifdef SPNG_D3_PROF
        {
            uint32_t mh = p_hops, mc = p_cmp;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const uint32_t a = (uint32_t)__shfl_xor((int)mh, m, 64), b2 = (uint32_t)__shfl_xor((int)mc, m, 64); mh = a > mh ? a : mh; mc = b2 > mc ? b2 : mc; }
            D3P_CNT(4, 1); D3P_CNT(5, mh); D3P_CNT(6, mc);
            uint32_t sh_ = p_hops + p_cmp, ms_ = p_hops + p_cmp;       // lane steps in all, against 64 x the slowest lane's
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                sh_ += (uint32_t)__shfl_xor((int)sh_, m, 64);
                const uint32_t o_ = (uint32_t)__shfl_xor((int)ms_, m, 64);
                ms_ = o_ > ms_ ? o_ : ms_;
            }
            D3P_CNT(7, sh_); D3P_CNT(8, ms_);
            if (mh + mc > 256) { D3P_CNT(9, mh); D3P_CNT(10, mc); D3P_CNT(11, 1); D3P_CNT(12, sh_); }     // the stragglers
        }
#endif
        const uint64_t v = (warm + p0r - rb) + (uint32_t)lane;  // round coordinates
        if (!FULL) {
            if (inchunk) match[v] = ext > 5 ? ext << 16 | bestd : 0u;
            continue;
        }
        if (cdec != 0xff) { tw[cnt * 64 + (uint32_t)lane] = (uint32_t)lane << 24 | cdist << 9 | crun; ++cnt; }
        // ---- the batch's record: per position candidates << 9 | longest run; the words where the pool has room
        uint32_t T;
        const uint32_t pre = wave_excl_scan(cnt, T, lane);
        uint32_t mr = ext;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mr, m, 64); mr = o > mr ? o : mr; }
        unsigned long long base = 0;
        if (T && lane == 0) base = atomicAdd(pool_next, (unsigned long long)T);
        base = uni64(base);
        const bool ok = base + T <= pool_cap;
        if (!ok && lane == 0) atomicOr(fail, 1u);
        if (inchunk) vinfo[v] = (uint16_t)(cnt << 9 | ext);
        if (lane == 0) { bbase[v >> 6] = base; bwords[v >> 6] = T | mr << 16; }
        if (ok && T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's own words, back from the scratch (L1 is write-through: read past it)
            for (uint32_t k = 0; k < cnt; ++k)
                pool[base + pre + k] = __hip_atomic_load(&tw[k * 64 + (uint32_t)lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (lane == 0) __hip_atomic_store(&s.cur[wave], ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef SPNG_D3_PROF
    __syncthreads();
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0)
        printf("d3 prof wg %u (%u positions, %u batches): total %llu kcyc; inserter: throttle wait %llu insert %llu; searchers (sum of 16 waves): wait %llu walk %llu kcyc; batches %llu, max-lane hops %llu, compare steps %llu; lane steps (hops + compares) %llu of 64 x %llu; batches over 256 steps: %llu with max-lane hops %llu, compare steps %llu, lane steps %llu\n",
               blockIdx.x, c1r - c0r, nbatches, (__builtin_readcyclecounter() - d3p_start) >> 10, s.prof[0] >> 10, s.prof[1] >> 10, s.prof[2] >> 10, s.prof[3] >> 10, s.prof[4], s.prof[5], s.prof[6], s.prof[7], s.prof[8], s.prof[11], s.prof[9], s.prof[10], s.prof[12]);
#endif
}

__global__ __launch_bounds__(SPNG_D3_WAVES * 64) void dfl3_search_kernel(const D2Stream *__restrict__ streams, uint32_t cps, uint32_t chunk_len, uint32_t *__restrict__ pool,
                                                          unsigned long long *__restrict__ pool_next, uint64_t pool_cap, uint32_t *__restrict__ temp, uint32_t parity)
{
    const D2Stream &st = streams[blockIdx.x / cps];
    D2State *state = (D2State *)uni64((uint64_t)st.state);
    if (UNI(state->done) || UNI(state->fail)) return;
    // (the search's own cursor: this round may be a round ahead of the one the parse kernel is at)
    const uint64_t n = uni64(st.src_len), rb = uni64(state->srb), re = uni64(state->sre);
    const uint64_t c0 = rb + (uint64_t)(blockIdx.x % cps) * chunk_len, c1 = c0 + chunk_len < re ? c0 + chunk_len : re;
    if (c0 >= re || n < 3) return;
    const int lv = (int)UNI(st.level) > 13 ? 13 : (int)UNI(st.level);
    // DeflatorSearch.init(level:) (:13-35), full rows
    const int attempts = lv == 8 ? 14 : lv == 9 ? 20 : lv == 10 ? 30 : lv == 11 ? 60 : lv == 12 ? 100 : 0x7fffffff;
    const int goal = lv == 8 ? 20 : lv == 9 ? 32 : lv == 10 ? 50 : lv == 11 ? 80 : lv == 12 ? 133 : 258;
    d3_search_chunk<true>(g_d3, (const gbyte *)uni64((uint64_t)st.src), n, rb, c0, c1, 0, attempts, goal, (1u << UNI(st.exponent)) - 1,
                          &state->adlerS, &state->adlerI, &state->fail,
                          (uint16_t *)uni64((uint64_t)(parity ? st.vinfo2 : st.vinfo)), (uint64_t *)uni64((uint64_t)(parity ? st.bbase2 : st.bbase)),
                          (uint32_t *)uni64((uint64_t)(parity ? st.bwords2 : st.bwords)), pool, pool_next, pool_cap,
                          temp + (uint64_t)blockIdx.x * (SPNG_D3_WAVES * 30 * 64), nullptr);
}

// ---- levels 0-7 in rounds: dfl3_search_kernel<false> + dfl3_parse_kernel ----------------------------------------------------
// deflate_kernel keeps insertion, chain walk, parse, trees and bits of a stream on ONE wave (2.6 MB/s per stream at level 6, two
// thirds of it the chain walk, 512 streams resident).  What DeflatorWindow.match answers is a function of the input alone at
// these levels too (the design note at the top of this file), so the search goes chip-wide exactly as at levels >= 8: a stream's
// positions in rounds of 2^21, every round cut into chunks for the search workgroups above, which leave ONE word per position --
// the first strictly longest run > 5 within `attempts` candidates, stopped at `goal` (:132-212) -- and a parse wave per stream
// that only walks those answers with the greedy / lazy rules (DeflatorBuffers.Stream.swift:209-342), queues terms and writes
// blocks; its state (parse position, queued terms, bit writer) is the D1State of spng_deflate_resume_batch, kept in HBM from
// round to round.  The search of round r + 1 runs beside the parse of round r.  A round's search covers one position more than
// the round (a lazy parse looks at position + 1).
static constexpr uint32_t D3_RV = 1u << 21;
uint64_t deflate3_round_positions() { return D3_RV; }
// the positions a call may parse: all of them, or -- more input to come -- those whose look-ahead (their own and that of the
// position behind them: 258 bytes + the key) is complete whatever follows
__host__ __device__ inline uint64_t d3_end(uint64_t n, bool more) { return more ? (n > 264 ? n - 264 : 0) : n; }
uint64_t deflate3_end(uint64_t n, bool more) { return d3_end(n, more); }

__global__ void dfl3_begin_kernel(const D3Stream *__restrict__ streams, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const D3Stream &st = streams[i];
    D1State &t = *st.state;
    if (!t.started) { t.started = 1; t.w = 0; t.spos = 0; t.count = 0; t.acc = 0; t.total = 0; t.nacc = 0; t.overflow = 0; t.adlerS = 0; t.adlerI = 0; t.done = 0; }
    uint64_t E = d3_end(st.src_len, st.more != 0);
    if (E < t.spos) E = t.spos;
    t.srb = t.spos; t.sre = t.srb + D3_RV < E ? t.srb + D3_RV : E;
    t.rb = t.srb; t.re = t.sre;
}
__global__ void dfl3_advance_kernel(const D3Stream *__restrict__ streams, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const D3Stream &st = streams[i];
    D1State &t = *st.state;
    uint64_t E = d3_end(st.src_len, st.more != 0);
    if (E < t.sre) E = t.sre;
    t.spos = t.sre; t.srb = t.sre; t.sre = t.srb + D3_RV < E ? t.srb + D3_RV : E;
}

__global__ __launch_bounds__(SPNG_D3_WAVES * 64) void dfl3_search_fast_kernel(const D3Stream *__restrict__ streams, uint32_t cps, uint32_t chunk_len, uint32_t parity)
{
    const D3Stream &st = streams[blockIdx.x / cps];
    D1State *state = (D1State *)uni64((uint64_t)st.state);
    if (UNI(state->done)) return;
    const uint64_t n = uni64(st.src_len), rb = uni64(state->srb), re = uni64(state->sre);
    const uint64_t c0 = rb + (uint64_t)(blockIdx.x % cps) * chunk_len;
    if (c0 >= re || n < 3) return;
    uint64_t c1 = c0 + chunk_len < re ? c0 + chunk_len : re;
    uint32_t extra = 0;
    if (c1 == re && c1 < n) { c1 += 1; extra = 1; }           // (the position behind the round: what a lazy parse looks at from the last one)
    // DeflatorSearch.init(level:) (:13-35), greedy and lazy rows
    const int level = (int)UNI(st.level) < 0 ? 0 : (int)UNI(st.level);
    const int lv = level & 7;
    const int attempts = lv == 0 ? 1 : lv == 1 ? 2 : lv == 2 ? 4 : lv == 3 ? 40 : lv == 4 ? 20 : lv == 5 ? 40 : lv == 6 ? 64 : 100;
    const int goal = lv == 0 ? 6 : lv == 1 ? 8 : lv == 2 ? 10 : lv == 3 ? 24 : lv == 4 ? 32 : lv == 5 ? 54 : lv == 6 ? 80 : 160;
    d3_search_chunk<false>(g_d3f, (const gbyte *)uni64((uint64_t)st.src), n, rb, c0, c1, extra, attempts, goal, (1u << UNI(st.exponent)) - 1,
                           &state->adlerS, &state->adlerI, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr,
                           (uint32_t *)uni64((uint64_t)st.match[parity]));
}

// Two waves per stream.  Wave 0, the parser, walks the answers and queues terms -- into one of two buffers; wave 1, the writer,
// owns the bit writer: the stream's header, every block (symbol counts, the trees, the tables, the bits: write_block), the
// trailer, the results.  A full buffer is handed over with a command word and the parser goes on in the other one: the walk of
// block k + 1 runs beside the trees and bits of block k (one wave did them one after the other: at level 6 the blocks were 60 %
// of a stream's time).  At the end of a round the parser keeps its position and the terms of the unfinished block in the
// D1State, the writer its pending bits.
enum : uint32_t { D3_CMD_BLOCK = 1u << 28, D3_CMD_FINAL = 2u << 28, D3_CMD_SAVE = 3u << 28, D3_CMD_MORE = 4u << 28, D3_CMD_MASK = 7u << 28 };

__global__ __launch_bounds__(128) void dfl3_parse_kernel(const D3Stream *__restrict__ streams, spng_result *__restrict__ results, uint32_t parity)
{
    DLds &s = g_lds;
    const D3Stream *sp = streams + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = (int)UNI(threadIdx.x >> 6);
    D1State *state = (D1State *)uni64((uint64_t)sp->state);
    if (UNI(state->done)) return;
    const gbyte *in = (const gbyte *)uni64((uint64_t)sp->src);
    const uint64_t n = uni64(sp->src_len);
    const int32_t format = (int32_t)UNI(sp->format);
    const bool lazy = (int32_t)UNI(sp->level) >= 4;            // Stream.compress lazy (:268-323) from level 4 on
    const bool more = UNI(sp->more) != 0;                      // spng_deflate_resume_batch: the input goes on behind src_len
    const uint32_t image = UNI(sp->image);
    const uint64_t rb = uni64(state->rb), re = uni64(state->re);
    uint64_t E = d3_end(n, more);
    E = E < re ? re : E;
    if (n < 3 && more) {
        // (nothing can be decided yet: not even whether this will be a stored tail)
        if (threadIdx.x == 0) {
            spng_result &res = results[image];
            res.status = SPNG_NEED_MORE_INPUT; res.reserved = 0; res.written = state->total; res.consumed = state->w; res.aux[0] = state->w; res.aux[1] = state->spos;
        }
        return;
    }
    // (what both waves need of the state is read before either of them writes to it)
    const uint64_t w0 = uni64(state->w);
    const uint32_t count0 = UNI(state->count);
    if (threadIdx.x < 2) g_trm.cmd[threadIdx.x] = 0;
    __syncthreads();

    if (wave == 0) {
        // ---- the parser
        const gword *match = (const gword *)uni64((uint64_t)sp->match[parity]);
        uint64_t w = w0;
        int count = (int)count0;
        uint32_t tb = 0;                                       // the buffer being filled
        uint32_t *terms = g_trm.terms[0];
        for (int i = lane; i < count; i += 64) terms[i] = state->terms[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        const int limit_terms = 2048;
        auto unfilled = [&]() { return limit_terms - 1 - count; };
#ifdef SPNG_D3_PROF
        unsigned long long pwait = 0, pblocks = 0;
        const unsigned long long pstart = __builtin_readcyclecounter();
#endif
        // hands the buffer's `count` terms to the writer and turns to the other buffer (once the writer has let go of it)
        auto hand_over = [&](uint32_t cmd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            if (lane == 0) __hip_atomic_store(&g_trm.cmd[tb], cmd | (uint32_t)count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            tb ^= 1; terms = g_trm.terms[tb]; count = 0;
#ifdef SPNG_D3_PROF
            const unsigned long long t0 = __builtin_readcyclecounter();
#endif
            SpinGuard guard;
            while (__hip_atomic_load(&g_trm.cmd[tb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
                __builtin_amdgcn_s_sleep(4);
                guard.tick();
            }
#ifdef SPNG_D3_PROF
            pwait += __builtin_readcyclecounter() - t0; pblocks += 1;
#endif
        };
        if (n >= 3) {
            const uint64_t last_main = n - 4 + 1;              // positions 0 .. n-4 are searched
            const uint64_t stop = re < last_main ? re : last_main; // tokens that start below `stop` are this round's
            // the answers of 128 positions at a time (two per lane), fetched a batch ahead: a wave alone has nobody to hide a load behind
            auto ask = [&](uint64_t p, uint32_t &m, uint32_t &lit) {
                m = (p <= re && p < last_main) ? match[p - rb] : 0u;
                lit = p < n ? (uint32_t)in[p] : 0u;
            };
            uint32_t nmA, nmB, nlA, nlB;
            uint64_t asked = w;
            ask(w + lane, nmA, nlA); ask(w + 64 + lane, nmB, nlB);
            while (w < stop) {
                uint32_t mA, mB, litA, litB;
                if (asked == w) { mA = nmA; mB = nmB; litA = nlA; litB = nlB; }
                else { ask(w + lane, mA, litA); ask(w + 64 + lane, mB, litB); }
                asked = w + 128;                               // (the guess: the batch is used up to its end -- a run across it asks again)
                ask(asked + lane, nmA, nlA); ask(asked + 64 + lane, nmB, nlB);
                auto at = [&](uint32_t xa, uint32_t xb, uint32_t t) -> uint32_t {
                    return t < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)xa, (int)t) : (uint32_t)__builtin_amdgcn_readlane((int)xb, (int)(t - 64));
                };
                // ---- the parse: Stream.compress greedy (:209-252) / lazy (:268-323) over these 128 answers
                uint32_t t = 0;
                while (t < 128 && w + t < stop) {
                    if (!(unfilled() > (lazy ? 1 : 0))) hand_over(D3_CMD_BLOCK);
                    const uint32_t m = at(mA, mB, t);
                    const uint32_t lit = at(litA, litB, t);
                    if (!m) { terms[count] = 0xf8000000u | lit; ++count; t += 1; continue; }
                    uint32_t use_run = m >> 16, use_dist = m & 0xffff;
                    uint32_t adv = use_run;
                    if (lazy) {
                        // the answer for position w + t + 1 is needed: start the next batch there if it is not in this one
                        if (t + 1 >= 128) break;
                        // lazy match at a + 1 (:293-299); it exists only if that position is still searched
                        const uint32_t lm = at(mA, mB, t + 1);
                        if ((lm >> 16) > use_run) {
                            terms[count] = 0xf8000000u | lit;
                            ++count;
                            use_run = lm >> 16; use_dist = lm & 0xffff;
                            adv = 1 + use_run;
                        }
                    }
                    // LZ77.DeflatorTerm.init(run:distance:) (DeflatorTerm.swift:34-56)
                    const uint32_t rd = run_decade(use_run), dd = dist_decade(use_dist);
                    terms[count] = dd << 27 | 0x100u | rd | dist_extra_value(use_dist, dd) << 14 | run_extra_value(use_run, rd) << 9;
                    ++count;
                    t += adv;
                }
                w = uni64(w + t);
            }
        } else w = n;                                          // (a stored tail: the writer's)
        if (re < E || more) {
            // on with the next round / the next push: the unfinished block's terms and the position into the state
            for (int i = lane; i < count; i += 64) state->terms[i] = terms[i];
            if (lane == 0) {
                state->w = w; state->count = (uint32_t)count;
                if (re < E) { state->rb = re; state->re = re + D3_RV < E ? re + D3_RV : E; }
                g_trm.fin_w = w;
            }
            count = 0;
            hand_over(re < E ? D3_CMD_SAVE : D3_CMD_MORE);
#ifdef SPNG_D3_PROF
            if (lane == 0 && blockIdx.x == 0) printf("d3 parse prof (round from %llu): parser total %llu kcyc, of it waiting for the writer %llu kcyc; %llu blocks\n",
                                                     (unsigned long long)rb, (__builtin_readcyclecounter() - pstart) >> 10, pwait >> 10, pblocks);
#endif
            return;
        }
        if (n >= 3) {
            // epilogue: the positions still in the window pipeline become literals (:254-265, :331-342)
            for (uint64_t p = w; p < n; ++p) {
                if (!(unfilled() > 0)) hand_over(D3_CMD_BLOCK);
                terms[count] = 0xf8000000u | UNI(in[p]);
                ++count;
            }
        }
        hand_over(D3_CMD_FINAL);
        return;
    }

    // ---- the writer
    for (int i = lane; i < OUTB / 4; i += 64) s.out32[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    Bits b = {uni64(state->acc), UNI(state->nacc), uni64(state->total), uni64(state->total), (gbyte *)uni64((uint64_t)sp->dst), uni64(sp->dst_cap),
              UNI(state->overflow) != 0};
    if (w0 == 0 && b.total == 0 && b.nacc == 0 && count0 == 0) {
        // the stream's first round
        if (format == SPNG_FORMAT_ZLIB) {
            // StreamHeader.write (StreamHeader.swift:56-62)
            const uint32_t unpaired = (UNI(sp->exponent) - 8) << 4 | 0x08;
            const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
            put(s, b, check << 8 | unpaired, 16, lane);
        } else if (format == SPNG_FORMAT_GZIP) {
            // Gzip.StreamHeader.write (Gzip.StreamHeader.swift:84-96); the trailer is appended by gzip.hip
            put(s, b, 0x8b1f, 16, lane); put(s, b, 0x0008, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0xff00, 16, lane);
        }
    }
    uint32_t tailS = 0, tailI = 0;
    if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434)
        put(s, b, 1, 3, lane);
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, (uint32_t)n, 16, lane); put(s, b, ~(uint32_t)n & 0xffff, 16, lane);
        for (uint64_t k = 0; k < n; ++k) put(s, b, in[k], 8, lane);
        if ((uint64_t)lane < n) { tailS = in[lane]; tailI = (uint32_t)lane * in[lane]; }
    }
    uint32_t tb = 0, kind = 0;
#ifdef SPNG_D3_PROF
    unsigned long long wwait = 0;
    const unsigned long long wstart = __builtin_readcyclecounter();
#endif
    for (;;) {
        uint32_t c;
#ifdef SPNG_D3_PROF
        const unsigned long long t0 = __builtin_readcyclecounter();
#endif
        SpinGuard guard;
        while ((c = __hip_atomic_load(&g_trm.cmd[tb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) {
            __builtin_amdgcn_s_sleep(4);
            guard.tick();
        }
#ifdef SPNG_D3_PROF
        wwait += __builtin_readcyclecounter() - t0;
#endif
        c = UNI(c);
        kind = c & D3_CMD_MASK;
        if (kind == D3_CMD_BLOCK || (kind == D3_CMD_FINAL && n >= 3))
            b = uni_bits(write_block(b, (int)(c & 0xfffffu), kind == D3_CMD_FINAL, lane, (int)tb));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) __hip_atomic_store(&g_trm.cmd[tb], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        tb ^= 1;
        if (kind != D3_CMD_BLOCK) break;
    }
#ifdef SPNG_D3_PROF
    if (lane == 0 && blockIdx.x == 0) printf("d3 parse prof (round from %llu): writer total %llu kcyc, of it waiting for the parser %llu kcyc\n",
                                             (unsigned long long)rb, (__builtin_readcyclecounter() - wstart) >> 10, wwait >> 10);
#endif
    if (kind != D3_CMD_FINAL) {
        // on with the next round / the next push: whole bytes out, the rest into the state
        drain(s, b, b.total, lane);
        if (lane == 0) {
            state->acc = b.acc; state->nacc = b.nacc; state->total = b.total; state->overflow = b.overflow ? 1u : 0u;
            if (kind == D3_CMD_MORE) {
                const uint64_t w = g_trm.fin_w;
                spng_result &res = results[image];
                res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_NEED_MORE_INPUT; res.reserved = 0;
                res.written = b.total; res.consumed = w; res.aux[0] = w; res.aux[1] = state->spos;
            }
        }
        return;
    }
    if (format == SPNG_FORMAT_ZLIB) {
        // Adler-32 from the sums the search kernel left (s1 = 1 + S, s2 = N + N * S - I)
        uint32_t S, I;
        if (n < 3) {
            S = tailS; I = tailI;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
        } else { S = UNI(state->adlerS); I = UNI(state->adlerI); }
        S %= 65521; I %= 65521;
        const uint32_t N = (uint32_t)(n % 65521);
        const uint32_t sum = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, sum >> 24, 8, lane); put(s, b, (sum >> 16) & 0xff, 8, lane);
        put(s, b, (sum >> 8) & 0xff, 8, lane); put(s, b, sum & 0xff, 8, lane);
    }
    if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);                // DeflatorOut.pull flushes padding bits
    drain(s, b, b.total, lane);
    if (lane == 0) {
        spng_result &res = results[image];
        res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
        res.written = b.total; res.consumed = n; res.aux[0] = res.aux[1] = 0;
        state->done = 1;
    }
}

// ---- levels 0-7, one-shot streams: the blocks side by side (dfl4_walk / dfl4_block / dfl4_scan / dfl4_place) -----------------
// With the answers in hand a stream's serial part is only the walk -- which positions are asked, which terms that gives.  The
// blocks themselves (2047 terms: symbol counts, two Huffman trees with the reference's heap replayed on one lane, the
// run-length coded tables, the bits: ~1 M cycles of a latency-bound wave) do not depend on each other at all but for WHERE
// in the stream their bits go.  So, per round and stream:
//   * dfl4_walk_kernel: a wave walks the answers (the parser half of dfl3_parse_kernel) and leaves the terms in global memory,
//     block after block, with a list of the blocks; the terms of the unfinished block go to the next round in the D1State;
//   * dfl4_block_kernel: a wave per BLOCK, the whole chip over all blocks of all streams: write_block into a scratch of the
//     block's own, from bit 0 on; its length in bits;
//   * dfl4_scan_kernel: per stream the prefix sum of the lengths = where each block starts (header, stored tail, trailer and
//     result are its business too);
//   * dfl4_place_kernel: a wave per block moves the block's bits to their place: output byte b is the eight bits from 8 b on,
//     a two-byte read and a shift per lane; a byte is written by the block its first bit lies in, which takes the missing bits
//     from the blocks behind (and, first block of a round, the bits the round before left pending).
// Streams that arrive in pieces (spng_deflate_resume_batch) keep the two-wave form above: their state outlives the call.
static constexpr uint32_t D4_BCAP = 13312;      // bytes of a block's bits: 2047 terms of <= 48 bits, the tables, the end-of-block symbol; slack
uint64_t deflate4_max_blocks(uint64_t positions) { return positions / 2046 + 4; }
uint64_t deflate4_block_bytes() { return D4_BCAP; }
__shared__ uint32_t g_wterms[2048];             // dfl4_walk_kernel: the terms of the block being filled (carried over a round's end)

__global__ __launch_bounds__(64) void dfl4_walk_kernel(const D3Stream *__restrict__ streams, uint32_t parity)
{
    const D3Stream *sp = streams + blockIdx.x;
    const int lane = threadIdx.x;
    D1State *state = (D1State *)uni64((uint64_t)sp->state);
    gword *bd = (gword *)uni64((uint64_t)sp->bdesc);
    if (UNI(state->done)) { if (lane == 0) bd[0] = 0; return; }
    const gbyte *in = (const gbyte *)uni64((uint64_t)sp->src);
    const uint64_t n = uni64(sp->src_len);
    const bool lazy = (int32_t)UNI(sp->level) >= 4;            // Stream.compress lazy (:268-323) from level 4 on
    const gword *match = (const gword *)uni64((uint64_t)sp->match[parity]);
    gword *tbuf = (gword *)uni64((uint64_t)sp->terms);
    const uint64_t rb = uni64(state->rb), re = uni64(state->re);
    const uint64_t E = n < re ? re : n;
    uint64_t w = uni64(state->w);
    uint32_t count = UNI(state->count);
    uint32_t off = 0, nblk = 0;                                // first term of the block being filled; blocks closed
    for (uint32_t i = (uint32_t)lane; i < count; i += 64) g_wterms[i] = state->terms[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    // A block closes when 2047 terms are queued (lazy: 2046 or 2047: a step may queue two), looked at before every step
    // (DeflatorBuffers.Stream.swift:219, 277): its terms leave for global memory 64 at a time, its place goes into the list.
    const uint32_t cap = lazy ? 2046u : 2047u;
    auto close = [&](bool final) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        for (uint32_t i = (uint32_t)lane; i < count; i += 64) tbuf[off + i] = g_wterms[i];
        if (lane == 0) { bd[4 + 2 * nblk] = off; bd[5 + 2 * nblk] = count | (final ? 1u << 31 : 0u); }
        off += count; count = 0; nblk += 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // (every lane has read its terms before the next block's are queued)
    };
    if (n >= 3) {
        const uint64_t last_main = n - 4 + 1;                  // positions 0 .. n-4 are searched
        const uint64_t stop = re < last_main ? re : last_main; // tokens that start below `stop` are this round's
        // the answers of 128 positions at a time (two per lane), fetched a batch ahead: a wave alone has nobody to hide a load behind
        auto ask = [&](uint64_t p, uint32_t &m, uint32_t &lit) {
            m = (p <= re && p < last_main) ? match[p - rb] : 0u;
            lit = p < n ? (uint32_t)in[p] : 0u;
        };
        uint32_t nmA, nmB, nlA, nlB;
        uint64_t asked = w;
        ask(w + lane, nmA, nlA); ask(w + 64 + lane, nmB, nlB);
        while (w < stop) {
            uint32_t mA, mB, litA, litB;
            if (asked == w) { mA = nmA; mB = nmB; litA = nlA; litB = nlB; }
            else { ask(w + lane, mA, litA); ask(w + 64 + lane, mB, litB); }
            asked = w + 128;                                   // (the guess: the batch is used up to its end -- a run across it asks again)
            ask(asked + lane, nmA, nlA); ask(asked + 64 + lane, nmB, nlB);
            // What a position would queue if the walk came by -- its literal, its match (LZ77.DeflatorTerm.init(run:distance:),
            // DeflatorTerm.swift:34-56) -- is worked out for all 128 at once; the walk itself, one position after the other on the
            // scalar unit, only picks: a run of positions without a match is one vector store of their literals.
            auto mterm = [&](uint32_t m) -> uint32_t {
                const uint32_t run = m >> 16, dist = m & 0xffff;
                const uint32_t rd = run_decade(run ? run : 3u), dd = dist_decade(dist ? dist : 1u);
                return dd << 27 | 0x100u | rd | dist_extra_value(dist, dd) << 14 | run_extra_value(run, rd) << 9;
            };
            const uint32_t tmA = mterm(mA), tmB = mterm(mB), tlA = 0xf8000000u | litA, tlB = 0xf8000000u | litB;
            const uint32_t runA = mA >> 16, runB = mB >> 16;
            const unsigned long long nzA = __ballot(runA != 0), nzB = __ballot(runB != 0);
            auto at = [&](uint32_t xa, uint32_t xb, uint32_t t) -> uint32_t {
                return t < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)xa, (int)t) : (uint32_t)__builtin_amdgcn_readlane((int)xb, (int)(t - 64));
            };
            // ---- the parse: Stream.compress greedy (:209-252) / lazy (:268-323) over these 128 answers
            const uint32_t avail = stop - w < 128 ? (uint32_t)(stop - w) : 128u;   // positions of this batch that are this round's
            uint32_t t = 0;
            while (t < avail) {
                if (count >= cap) close(false);
                // positions from t on without a match
                uint32_t L;
                if (t < 64) { const unsigned long long a = nzA >> t; L = a ? (uint32_t)__builtin_ctzll(a) : 64 - t + (nzB ? (uint32_t)__builtin_ctzll(nzB) : 64u); }
                else { const unsigned long long b2 = nzB >> (t - 64); L = b2 ? (uint32_t)__builtin_ctzll(b2) : 128 - t; }
                L = L < avail - t ? L : avail - t;
                if (L) {
                    L = L < cap - count ? L : cap - count;     // (the block closes in between: the next turn goes on)
                    const uint32_t qa = (uint32_t)lane, qb = 64u + (uint32_t)lane;
                    if (qa >= t && qa < t + L) g_wterms[count + qa - t] = tlA;
                    if (qb >= t && qb < t + L) g_wterms[count + qb - t] = tlB;
                    count += L; t += L;
                    continue;
                }
                const uint32_t run = at(runA, runB, t);
                if (lazy) {
                    // the answer for position w + t + 1 is needed: the next batch starts there if it is not in this one
                    if (t + 1 >= 128) break;
                    // lazy match at a + 1 (:293-299); it exists only if that position is still searched
                    const uint32_t lrun = at(runA, runB, t + 1);
                    if (lrun > run) {
                        const uint32_t a0 = at(tlA, tlB, t), a1 = at(tmA, tmB, t + 1);
                        if (lane == 0) { g_wterms[count] = a0; g_wterms[count + 1] = a1; }
                        count += 2; t += 1 + lrun;
                        continue;
                    }
                }
                const uint32_t a0 = at(tmA, tmB, t);
                if (lane == 0) g_wterms[count] = a0;
                count += 1; t += run;
            }
            w = uni64(w + t);
        }
    } else w = n;
    const bool last = !(re < E);
    if (!last) {
        // on with the next round: the unfinished block's terms and the position into the state
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        for (uint32_t i = (uint32_t)lane; i < count; i += 64) state->terms[i] = g_wterms[i];
        if (lane == 0) { state->w = w; state->count = count; state->rb = re; state->re = re + D3_RV < E ? re + D3_RV : E; }
    } else if (n >= 3) {
        // epilogue: the positions still in the window pipeline become literals (:254-265, :331-342), then the final block
        for (uint64_t p = w; p < n; ++p) {
            if (count >= 2047u) close(false);
            const uint32_t t = 0xf8000000u | UNI(in[p]);
            if (lane == 0) g_wterms[count] = t;
            count += 1;
        }
        close(true);
    }
    if (lane == 0) { bd[0] = nblk; bd[1] = last ? 1u : 0u; bd[2] = (uint32_t)rb; bd[3] = (uint32_t)(rb >> 32); }
}

__global__ __launch_bounds__(64) void dfl4_block_kernel(const D3Stream *__restrict__ streams)
{
    DLds &s = g_lds;
    const D3Stream *sp = streams + blockIdx.y;
    const int lane = threadIdx.x;
    const gword *bd = (const gword *)uni64((uint64_t)sp->bdesc);
    const uint32_t k = blockIdx.x;
    if (k >= UNI(bd[0])) return;
    const uint32_t first = UNI(bd[4 + 2 * k]), cw = UNI(bd[5 + 2 * k]);
    for (int i = lane; i < OUTB / 4; i += 64) s.out32[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    Bits b = {0, 0, 0, 0, (gbyte *)uni64((uint64_t)sp->scratch) + (uint64_t)k * D4_BCAP, D4_BCAP, false};
    b = uni_bits(write_block_global(b, (int)(cw & 0x7fffffffu), (cw >> 31) != 0, lane, (const gword *)uni64((uint64_t)sp->terms) + first));
    const uint64_t bits = b.total * 8 + b.nacc;
    if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);                // (the bits behind the block's last in its last byte: zero)
    put(s, b, 0, 16, lane);                                    // (dfl4_place reads two bytes at a time)
    drain(s, b, b.total, lane);
    if (lane == 0) ((unsigned long long *)sp->bbits)[k] = b.overflow ? ~0ull : bits;
}

// one workgroup per stream: where the blocks of the round start; header, stored tail, trailer, result
__global__ __launch_bounds__(256) void dfl4_scan_kernel(const D3Stream *__restrict__ streams, uint32_t maxb, spng_result *__restrict__ results)
{
    const D3Stream &st = streams[blockIdx.x];
    D1State *state = st.state;
    uint32_t *bd = st.bdesc;
    unsigned long long *bb = (unsigned long long *)st.bbits;
    const uint32_t nb = bd[0];
    const bool last = bd[1] != 0;
    const uint64_t rb = (uint64_t)bd[3] << 32 | bd[2];
    __shared__ unsigned long long part[256];
    __shared__ unsigned long long shared_p0;
    const int tid = threadIdx.x;
    if (state->done) return;
    uint8_t *dst = st.dst;
    const uint64_t cap = st.dst_cap, n = st.src_len;
    if (tid == 0) {
        if (rb == 0 && state->total == 0 && state->nacc == 0) {
            // the stream's first round: its header
            uint8_t hdr[10]; uint32_t hn = 0;
            if (st.format == SPNG_FORMAT_ZLIB) {
                // StreamHeader.write (StreamHeader.swift:56-62)
                const uint32_t unpaired = (st.exponent - 8) << 4 | 0x08;
                const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
                hdr[0] = (uint8_t)unpaired; hdr[1] = (uint8_t)check; hn = 2;
            } else if (st.format == SPNG_FORMAT_GZIP) {
                // Gzip.StreamHeader.write (Gzip.StreamHeader.swift:84-96); the trailer is appended by gzip.hip
                const uint8_t g[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
                for (int i = 0; i < 10; ++i) hdr[i] = g[i];
                hn = 10;
            }
            for (uint32_t i = 0; i < hn; ++i) if (i < cap) dst[i] = hdr[i];
            state->total = hn;
        }
        shared_p0 = state->total * 8 + state->nacc;
    }
    __syncthreads();
    const unsigned long long p0 = shared_p0;
    // prefix sum of the blocks' lengths (a thread takes a run of consecutive blocks)
    const uint32_t per = (nb + 255) / 256;
    unsigned long long mine = 0;
    bool over = false;
    for (uint32_t k = tid * per; k < nb && k < (tid + 1) * per; ++k) { const unsigned long long l = bb[k]; if (l == ~0ull) over = true; else mine += l; }
    part[tid] = mine;
    __syncthreads();
    unsigned long long before = 0, total = 0;
    for (int t = 0; t < 256; ++t) { if (t < tid) before += part[t]; total += part[t]; }
    unsigned long long at = p0 + before;
    for (uint32_t k = tid * per; k < nb && k < (tid + 1) * per; ++k) { bb[maxb + k] = at; at += bb[k] == ~0ull ? 0 : bb[k]; }
    over = __syncthreads_or(over);
    if (tid != 0) return;
    const unsigned long long pend = p0 + total;
    // what dfl4_place needs of the state as it was: the pending bits in front of the round
    bd[2] = state->nacc; bd[3] = (uint32_t)state->acc & 0xff;
    state->total = pend >> 3; state->nacc = (uint32_t)(pend & 7);
    if (!last) return;
    uint64_t written = (pend + 7) >> 3;                        // DeflatorOut.pull flushes padding bits
    uint32_t S = state->adlerS % 65521, I = state->adlerI % 65521;
    if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434): no block was queued; the bytes by hand (the header ends on a byte)
        uint8_t t[8]; uint32_t tn = 0;
        t[tn++] = 1; t[tn++] = (uint8_t)n; t[tn++] = 0; t[tn++] = (uint8_t)~n; t[tn++] = 0xff;
        S = 0; I = 0;
        for (uint64_t k = 0; k < n; ++k) { t[tn++] = st.src[k]; S += st.src[k]; I += (uint32_t)k * st.src[k]; }
        for (uint32_t i = 0; i < tn; ++i) if (written + i < cap) dst[written + i] = t[i];
        written += tn;
    }
    if (st.format == SPNG_FORMAT_ZLIB) {
        // Adler-32 from the sums the search kernel left (s1 = 1 + S, s2 = N + N * S - I)
        const uint32_t N = (uint32_t)(n % 65521);
        const uint32_t sum = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
        for (int i = 0; i < 4; ++i) if (written + i < cap) dst[written + i] = (uint8_t)(sum >> (24 - 8 * i));
        written += 4;
    }
    spng_result &res = results[st.image];
    res.status = (over || written > cap) ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 0;
    res.written = written; res.consumed = n; res.aux[0] = res.aux[1] = 0;
    state->total = written; state->nacc = 0;
    state->done = 1;
}

__global__ __launch_bounds__(64) void dfl4_place_kernel(const D3Stream *__restrict__ streams, uint32_t maxb)
{
    const D3Stream *sp = streams + blockIdx.y;
    const int lane = threadIdx.x;
    const gword *bd = (const gword *)uni64((uint64_t)sp->bdesc);
    const uint32_t nb = UNI(bd[0]), k = blockIdx.x;
    if (k >= nb) return;
    const unsigned long long *bb = (const unsigned long long *)uni64((uint64_t)sp->bbits);
    const gbyte *scr = (const gbyte *)uni64((uint64_t)sp->scratch);
    gbyte *dst = (gbyte *)uni64((uint64_t)sp->dst);
    const uint64_t cap = uni64(sp->dst_cap);
    D1State *state = (D1State *)uni64((uint64_t)sp->state);
    const uint64_t S = uni64(bb[maxb + k]), Lk = uni64(bb[k]);
    if (Lk == ~0ull) return;                                   // (a block that outgrew its scratch: the stream reports the capacity error)
    const uint64_t Eb = S + Lk;
    const uint64_t pend = uni64(bb[maxb + nb - 1]) + (uni64(bb[nb - 1]) == ~0ull ? 0 : uni64(bb[nb - 1]));   // the round's last bit + 1
    // byte b of the stream from `have` valid low bits of v on: the rest from block kk on, from its bit `o`
    auto finish = [&](uint64_t b, uint32_t v, uint32_t have, uint32_t kk, uint64_t o) {
        while (have < 8 && kk < nb) {
            const uint64_t lk = bb[kk];
            if (lk != ~0ull && o < lk) {
                const gbyte *src = scr + (uint64_t)kk * D4_BCAP;
                const uint32_t two = (uint32_t)src[o >> 3] | (uint32_t)src[(o >> 3) + 1] << 8;
                const uint32_t avail = lk - o < 8 - have ? (uint32_t)(lk - o) : 8 - have;
                v |= ((two >> (o & 7)) & ((1u << avail) - 1)) << have;
                have += avail; o += avail;
            }
            if (have < 8) { kk += 1; o = 0; }
        }
        if (b < cap) dst[b] = (gbyte)v;
        if ((b << 3) + 8 > pend && (pend & 7)) state->acc = v;          // the round's last, unfinished byte: pending for the next round
    };
    const uint64_t b0 = (S + 7) >> 3, b1 = (Eb + 7) >> 3;      // the bytes whose first bit lies in this block
    for (uint64_t b = b0 + (uint32_t)lane; b < b1; b += 64) finish(b, 0u, 0u, k, (b << 3) - S);
    if (k == 0 && lane == 0 && UNI(bd[2])) finish(S >> 3, UNI(bd[3]) & ((1u << UNI(bd[2])) - 1), UNI(bd[2]), 0u, 0);   // the byte the round before left unfinished
}

// ---- the parse kernel -------------------------------------------------------------------------------------------
struct D2Arrays {                                               // (pointers of one stream, wave-uniform)
    const uint16_t *vinfo; const uint64_t *bbase; const uint32_t *bwords; uint64_t *emask;
    gword *up, *step; gbyte *pathb, *litb;
    const uint32_t *pool;
};

__device__ __forceinline__ D2Arrays uni_arrays(const D2Arrays &a)
{
    D2Arrays r;
    r.vinfo = UNIP(const uint16_t *, a.vinfo); r.bbase = UNIP(const uint64_t *, a.bbase); r.bwords = UNIP(const uint32_t *, a.bwords);
    r.emask = UNIP(uint64_t *, a.emask); r.up = UNIP(gword *, a.up); r.step = UNIP(gword *, a.step);
    r.pathb = UNIP(gbyte *, a.pathb); r.litb = UNIP(gbyte *, a.litb); r.pool = UNIP(const uint32_t *, a.pool);
    return r;
}

// The parse kernel walks a block in batches of 64 vertices ALIGNED IN ROUND COORDINATES (the first one may start in front of the
// block: sh = (block's first vertex in the round) mod 64 lanes idle), so that a batch is exactly one batch of the search kernel:
// one candidate list, the word's position-in-batch is the lane.  Batch j holds the block's vertices 64 j - sh ... 64 j - sh + 63.
//
// Which vertices of the block keep their edges (Stream.compress full, DeflatorBuffers.Stream.swift:344-400): behind a
// SEARCHED vertex whose longest run exceeds 100 the next run - 100 vertices -- not beyond the block's capacity -- are not
// searched.  One 64-bit mask per batch.
__device__ __attribute__((noinline)) void d2_skip_rule(const D2Arrays g_, uint64_t vr0_, uint32_t count_, uint32_t cap_, int lane)
{
    const D2Arrays g = uni_arrays(g_);
    const uint64_t vr0 = uni64(vr0_);
    const uint32_t count = UNI(count_), cap = UNI(cap_);
    const int sh = (int)(vr0 & 63);
    const uint32_t nb = ((uint32_t)sh + count + 64) >> 6;      // (as the forward pass: the batch of the end vertex too)
    int skip_until = 0, base = -sh;
    auto info_of = [&](int v) -> uint32_t { return (v >= 0 && (uint32_t)v < count) ? (uint32_t)g.vinfo[vr0 + v] : 1u; };
    uint32_t iq0 = info_of(base + lane), iq1 = info_of(base + 64 + lane), iq2 = info_of(base + 128 + lane);   // (three batches ahead)
    for (uint32_t j = 0; j < nb; ++j, base += 64) {
        const int v = base + lane;
        const bool valid = v >= 0 && (uint32_t)v < count;
        const uint32_t info = iq0;
        iq0 = iq1; iq1 = iq2; iq2 = info_of(v + 192);
        const uint32_t ext = info & 0x1ff;
        unsigned long long em = __ballot(valid && (info >> 9) != 0), xm = __ballot(valid && ext > 100);
        if (xm || skip_until > base) {
            int cur = 0;                                        // (lane coordinates)
            for (;;) {
                const int su = skip_until - base;
                if (su > cur) {
                    const int hi = su < 64 ? su : 64;
                    const unsigned long long clear = (hi >= 64 ? ~0ull : (1ull << hi) - 1) & ~((1ull << cur) - 1);
                    em &= ~clear; xm &= ~clear;
                    cur = hi;
                    if (cur >= 64) break;
                }
                const unsigned long long m = xm & ~((1ull << cur) - 1);
                if (!m) break;
                const int x = __ffsll((long long)m) - 1, vx = base + x;
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ext, x);
                const uint32_t room = cap - (uint32_t)(vx + 1);         // unfilled() once vx itself is in the block
                const uint32_t skip = e - 100 < room ? e - 100 : room;
                skip_until = vx + 1 + (int)skip;
                xm &= ~(1ull << x);
                cur = x + 1;
                if (cur >= 64) break;
            }
        }
        if (lane == 0) g.emask[j] = em;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The offer table of a batch: g_ptab[lane of the vertex][L - 3] = cost << 20 | decade << 15 | distance of the cheapest decade (first
// among equals: the lowest) whose run from that vertex reaches length L, for L = 3 .. 66; ~0: none.  Returns the vertices with a
// run beyond 66 (they take the per-entry path); `cols`: the columns in use.  w4: the first 256 words of the list, fetched a batch ahead.
__device__ __forceinline__ unsigned long long d2_offers(const D2Arrays g, uint32_t bw, uint64_t lbase, const uint32_t (&w4)[4], int base, uint32_t count,
                                                        unsigned long long em, uint32_t &cols, int lane)
{
    DLds &s = g_lds;
    const uint32_t T = bw & 0xffff, maxr = bw >> 16;
    cols = (T && maxr >= 3) ? (maxr < 66 ? maxr : 66u) - 2 : 0u;       // lengths 3 .. min(maxr, 66)
    for (uint32_t j = 0; j < cols; ++j) g_ptab[lane * D2_PSTRIDE + j] = ~0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    unsigned long long longm = 0;
    for (uint32_t i0 = 0; i0 < T; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        uint32_t w = 0;
        if (i0 < 256) { w = i0 == 0 ? w4[0] : i0 == 64 ? w4[1] : i0 == 128 ? w4[2] : w4[3]; }
        else if (i < T) w = g.pool[lbase + i];
        const uint32_t ln = w >> 24, run = w & 0x1ff, dist = (w >> 9) & 0x7fff;
        const int vv = base + (int)ln;
        const bool mine = i < T && ((em >> ln) & 1) && vv >= 0 && (uint32_t)vv < count;
        const uint32_t rem = count - (uint32_t)vv;
        const uint32_t r = run < rem ? run : rem;
        const bool use = mine && r >= 3;
        if (use) {
            const uint32_t dec = dist_decade(dist);
            const uint32_t key = (uint32_t)g_full.depths[512 + dec] << 20 | dec << 15 | dist;
            atomicMin(&g_ptab[ln * D2_PSTRIDE + (r < 66 ? r : 66u) - 3], key);
        }
        // vertices with a run beyond the table
        unsigned long long lm = __ballot(use && r > 66);
        while (lm) {
            const int l = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            longm |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)ln, l);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    // suffix minimum over the lengths: what reaches L + 1 reaches L
    uint32_t t = ~0u;
    for (int j = (int)cols - 1; j >= 0; --j) {
        const uint32_t x = g_ptab[lane * D2_PSTRIDE + j];
        t = x < t ? x : t;
        g_ptab[lane * D2_PSTRIDE + j] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    return longm;
}

// a vertex with a run beyond the table: the per-entry form of the one-kernel search (forward_body), lane = candidate word
__device__ __forceinline__ void d2_relax_long(const D2Arrays g, uint64_t vr, uint32_t vv, uint32_t count, uint32_t Dk, const uint32_t (&rc)[4], int lane)
{
    DLds &s = g_lds;
    const uint64_t q = vr >> 6;
    const uint32_t vtx = (uint32_t)(vr & 63);
    const uint32_t cntl = (uint32_t)g.vinfo[(q << 6) + lane] >> 9;
    uint32_t tot;
    const uint32_t prel = wave_excl_scan(cntl, tot, lane);
    const uint32_t pre = (uint32_t)__builtin_amdgcn_readlane((int)prel, (int)vtx), cnt = (uint32_t)__builtin_amdgcn_readlane((int)cntl, (int)vtx);
    const uint32_t w = (uint32_t)lane < cnt ? g.pool[uni64(g.bbase[q]) + pre + lane] : 0u;
    const uint32_t rem = count - vv;
    const uint32_t run = w & 0x1ff, dist = (w >> 9) & 0x7fff, dec = dist ? dist_decade(dist) : 0u;
    const uint32_t r = run < rem ? run : rem;
    const uint32_t dcost = g_full.depths[512 + dec];
    unsigned long long m = __ballot((uint32_t)lane < cnt && r >= 3);
    uint32_t bc[4] = {~0u, ~0u, ~0u, ~0u}, bd[4] = {0, 0, 0, 0}, reach = 0;
    while (m) {
        const int e = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t maxlen = (uint32_t)__builtin_amdgcn_readlane((int)r, e);
        const uint32_t dc = (uint32_t)__builtin_amdgcn_readlane((int)dcost, e);
        const uint32_t dd = (uint32_t)__builtin_amdgcn_readlane((int)(dec << 15 | dist), e);
        reach = maxlen > reach ? maxlen : reach;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (3u + 64u * j > maxlen) break;
            const uint32_t L = 3u + (uint32_t)lane + 64u * j;
            if (L <= maxlen && dc < bc[j]) { bc[j] = dc; bd[j] = dd; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (3u + 64u * j > reach) break;
        const uint32_t L = 3u + (uint32_t)lane + 64u * j;
        if (bc[j] != ~0u) {
            const uint64_t key = (uint64_t)(Dk + bc[j] + rc[j]) << 32 | (258u - L) << 23 | (bd[j] + (1u << 15));
            __hip_atomic_fetch_min(&g_full.win[(vv + L) & 511], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__device__ __forceinline__ uint32_t d2_smin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// sum over the wave, in scalar registers (DPP row scans + four v_readlane: no LDS round trips -- __shfl_xor is ds_bpermute)
__device__ __forceinline__ uint32_t wave_total(uint32_t v)
{
    const uint32_t incl = row_scan(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 15) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31) +
           (uint32_t)__builtin_amdgcn_readlane((int)incl, 47) + (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}

// minimize() forwards, as full_forward; keys: depth << 32 | (258 - length) << 23 | (decade + 1) << 15 | distance (the distance
// rides along -- one per vertex and decade, it never decides -- so that the way in knows it without a look-up).
//   * A batch without edges that no earlier edge reaches is a run of literals: its depths are a plain sum, its ways in are not
//     written at all (g.litb[batch] = 1 says so to the back-trace) -- the whole of an incompressible stream but a few batches.
//   * In a batch with edges the depths are scanned over all 64 lanes only at its start and at its end; between two groups of
//     vertices (three consecutive ones are final together) only the few lanes up to the next group are recomputed, one DPP
//     step per lane.
__device__ __attribute__((noinline)) void d2_forward(const D2Arrays g_, const gbyte *in_, uint64_t bbase_, uint64_t vr0_, uint32_t count_, int lane)
{
    const D2Arrays g = uni_arrays(g_);
    const gbyte *in = UNIP(const gbyte *, in_);
    const uint64_t bbase = uni64(bbase_), vr0 = uni64(vr0_);
    const uint32_t count = UNI(count_);
    DLds &s = g_lds;
    uint32_t rc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint32_t L = 3u + (uint32_t)lane + 64u * j; rc[j] = L <= 258 ? g_full.depths[253 + L] : 0u; }
    // (packed groups: lane = 21 * (vertex of the group) + (length - 3))
    const uint32_t pq = (uint32_t)lane / 21, pl = (uint32_t)lane % 21;
    const uint32_t rcp = g_full.depths[253 + 3 + pl];
    if (lane == 0) g_full.win[0] = 0;                               // vertex 0: depth 0
    const int sh = (int)(vr0 & 63);
    const uint64_t qbase = vr0 >> 6;
    const uint32_t nb = ((uint32_t)sh + count + 64) >> 6;      // batches that hold the vertices 0 .. count (the last one: the end)
    uint32_t inited = 1, carry = DINF;
    int pend = 0;                                              // no vertex behind it holds a key yet
    // a batch ahead: literals, edge masks, list heads; the first words of the next batch's list
    auto lit_of = [&](int v) -> uint32_t { return (v >= 1 && (uint32_t)v <= count) ? (uint32_t)in[bbase + (uint32_t)v - 1] : 0u; };
    uint32_t lb_next = lit_of(-sh + lane);
    // (what a batch needs from memory is asked for two batches ahead and only looked at -- made wave-uniform -- one batch later:
    //  this wave has nobody to hide a load behind)
    unsigned long long em1 = uni64(g.emask[0]);
    uint32_t bw1 = UNI(g.bwords[qbase]);
    uint64_t bb1 = uni64(g.bbase[qbase]);
    unsigned long long r_em = nb > 1 ? g.emask[1] : 0ull;      // raw: batch j + 1
    uint32_t r_bw = g.bwords[qbase + 1];
    uint64_t r_bb = g.bbase[qbase + 1];
    uint32_t wn[4] = {0, 0, 0, 0};
    auto fetch_words = [&](unsigned long long em, uint32_t bw, uint64_t bb) {
        const uint32_t T = em ? bw & 0xffff : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) wn[c] = 64u * c + (uint32_t)lane < T ? g.pool[bb + 64u * c + lane] : 0u;
    };
    fetch_words(em1, bw1, bb1);
    int base = -sh;
    for (uint32_t j = 0; j < nb; ++j, base += 64) {
        const int v = base + lane;
        const bool act = v >= 0 && (uint32_t)v <= count;       // a vertex of the block (the end vertex included)
        const unsigned long long em = em1;
        const uint32_t bw = bw1;
        const uint64_t bb = bb1;
        const uint32_t w4[4] = {wn[0], wn[1], wn[2], wn[3]};
        const uint32_t lb = lb_next;
        em1 = uni64(r_em); bw1 = UNI(r_bw); bb1 = uni64(r_bb);  // (asked for a batch ago)
        fetch_words(em1, bw1, bb1);
        lb_next = lit_of(v + 64);
        r_em = j + 2 < nb ? g.emask[j + 2] : 0ull;
        r_bw = g.bwords[qbase + j + 2]; r_bb = g.bbase[qbase + j + 2];
        const uint32_t cin = (v >= 1 && (uint32_t)v <= count) ? g_full.depths[lb] : 0u;     // the literal edge INTO v
        if (!em && base > pend && (uint32_t)(base + 63) < count) {
            // Literals only -- and so, in incompressible data, are the batches behind it: the whole run in one tight loop (a
            // literal load, a cost look-up and an add per batch; one sum over the wave at its end).
            const uint32_t most = (count - 64 - (uint32_t)base) / 64;          // batches behind this one that lie inside the block
            const unsigned long long ahead = j + 1 + (uint32_t)lane < nb ? g.emask[j + 1 + lane] : ~0ull;
            const unsigned long long busy = __ballot(ahead != 0);
            uint32_t z = busy ? (uint32_t)__ffsll((long long)busy) - 1 : 63u;   // (at most 64 batches a run: one litb store per lane)
            z = z < 63 ? z : 63u;
            z = z < most ? z : most;
            uint32_t acc = cin;
            {
                // (batches j + 1 ..: four loads in flight -- a batch is a load, a look-up and an add)
                uint32_t q0 = lit_of(v + 64), q1 = lit_of(v + 128), q2 = lit_of(v + 192), q3 = lit_of(v + 256);
                for (uint32_t i = 1; i <= z; ++i) {
                    const uint32_t lbc = q0;
                    q0 = q1; q1 = q2; q2 = q3; q3 = lit_of(v + 64 * (int)(i + 4));
                    acc += g_full.depths[lbc];
                }
            }
            carry += wave_total(acc);
            if ((uint32_t)lane <= z) g.litb[j + lane] = 1;
#ifdef SPNG_DEFLATE_PROF
            if (threadIdx.x == 0) g_prof[0] += (uint64_t)(1 + z) << 20;       // (count of literal batches)
#endif
            if (z) {
                // on behind the run: what the loop's head expects of batch j + z + 1
                j += z; base += 64 * (int)z;
                const uint32_t jn = j + 1;
                em1 = jn < nb ? uni64(g.emask[jn]) : 0ull;
                bw1 = UNI(g.bwords[qbase + jn]); bb1 = uni64(g.bbase[qbase + jn]);
                fetch_words(em1, bw1, bb1);
                lb_next = lit_of(base + 64 + lane);
                r_em = jn + 1 < nb ? g.emask[jn + 1] : 0ull;
                r_bw = g.bwords[qbase + jn + 1]; r_bb = g.bbase[qbase + jn + 1];
            }
            continue;
        }
        if (lane == 0) g.litb[j] = 0;
        const uint32_t top = (uint32_t)(base + 64 + 258);
        const uint32_t need = (top < count ? top : count) + 1;
        if (need > inited + 512) inited = need - 512;          // (behind a stretch of literals: nothing older is alive)
        for (uint32_t q = inited + lane; q < need; q += 64) g_full.win[q & 511] = ~0ull;
        inited = inited > need ? inited : need;
        unsigned long long longm = 0;
        uint32_t cols = 0;
#ifdef SPNG_DEFLATE_PROF
        const uint64_t o_t0 = __builtin_readcyclecounter();
#endif
        if (em) longm = d2_offers(g, bw, bb, w4, base, count, em, cols, lane);
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#ifdef SPNG_DEFLATE_PROF
        if (threadIdx.x == 0) g_prof[9] += __builtin_readcyclecounter() - o_t0;
#endif
        uint64_t W = act ? g_full.win[(uint32_t)v & 511] : ~0ull;
        uint32_t Wd = (uint32_t)(W >> 32) < DINF ? (uint32_t)(W >> 32) : DINF;
        uint32_t D = minplus_scan(cin, Wd, carry, lane);
        uint32_t k = 0;
        bool dirty = false;                                    // keys were written since D was computed
        // Runs up to 23 -- most batches of image data: the offers of a group's three vertices side by side, 21 lanes each, so
        // that ONE row read, one key and one ds_min_u64 per lane serve the whole group.  A wave alone on its SIMD issues an
        // instruction every four cycles, scalar or vector: this loop is bound by their number (~145 per group in its general
        // form below, a third of that here).  Per batch: which vertices have table edges (okm) / edges beyond the table (lgm) and
        // room for a match; per lane: its row offset, key constants and whether its column is in use.  Per group: the depths of
        // its three vertices on the scalar unit -- from the group before when it ends where this one starts (the ring's depths
        // and the literal costs of three lanes by v_readlane), from a scan over the wave otherwise.
        if (cols <= 21) {
            // vertices x (lanes) with 0 <= base + x and base + x + 3 <= count
            unsigned long long range = ~0ull;
            if (base < 0) range &= ~0ull << (uint32_t)(-base);
            {
                const int top3 = (int)count - 3 - base;                         // last lane with room
                range = top3 < 0 ? 0ull : top3 >= 63 ? range : range & ((2ull << (uint32_t)top3) - 1);
            }
            const unsigned long long gm = em & range, okm = gm & ~longm, lgm = gm & longm;
            const uint32_t pqbit = (pq < 3 && pl < cols) ? 1u << pq : 0u;
            const uint32_t rowc = (pq * D2_PSTRIDE + pl) * 4;                  // byte offset of my offer in the group's rows
            const uint32_t klo_c = (258u - (3u + pl)) << 23 | 1u << 15;
            const uint32_t ac = pq + 3u + pl;                                   // my target, from the group's first vertex
            uint32_t prevD = 0, kvs = ~0u;                                      // depth of the vertex at lane kvs - 1 (kvs: none)
            for (;;) {
                const unsigned long long rest = k < 64 ? gm >> k : 0ull;
                if (!rest) break;
                const uint32_t kk = k + (uint32_t)__builtin_ctzll(rest);
                const uint32_t m3 = (uint32_t)(okm >> kk) & 7u;
                uint32_t l3 = (uint32_t)(lgm >> kk) & 7u;
#ifdef SPNG_DEFLATE_PROF
                if (threadIdx.x == 0) g_prof[7] += 1ull << 20;
#endif
                const uint32_t pk = *(const uint32_t *)((const uint8_t *)g_ptab + kk * (D2_PSTRIDE * 4) + rowc);
                if (dirty) {
                    W = act ? g_full.win[(uint32_t)v & 511] : ~0ull;
                    Wd = (uint32_t)(W >> 32) < DINF ? (uint32_t)(W >> 32) : DINF;
                }
                const uint32_t k1 = kk + 1 < 63 ? kk + 1 : 63, k2 = kk + 2 < 63 ? kk + 2 : 63;
                // (the depth in front of the group: the group before left it, or a scan over the wave finds it)
                if (kk != kvs) {
                    if (dirty) D = minplus_scan(cin, Wd, carry, lane);
                    prevD = kk ? (uint32_t)__builtin_amdgcn_readlane((int)D, (int)(kk - 1)) : carry;
                }
                const uint32_t D0 = d2_smin(prevD + (uint32_t)__builtin_amdgcn_readlane((int)cin, (int)kk), (uint32_t)__builtin_amdgcn_readlane((int)Wd, (int)kk));
                const uint32_t D1 = d2_smin(D0 + (uint32_t)__builtin_amdgcn_readlane((int)cin, (int)k1), (uint32_t)__builtin_amdgcn_readlane((int)Wd, (int)k1));
                const uint32_t D2 = d2_smin(D1 + (uint32_t)__builtin_amdgcn_readlane((int)cin, (int)k2), (uint32_t)__builtin_amdgcn_readlane((int)Wd, (int)k2));
                prevD = D2; kvs = kk + 3;                                       // (kk + 3 > 63: no group behind this one)
                if ((m3 & pqbit) && pk != ~0u) {
                    const uint32_t Dq = pq == 0 ? D0 : pq == 1 ? D1 : D2;
                    const uint64_t key = (uint64_t)(Dq + (pk >> 20) + rcp) << 32 | ((pk & 0xfffffu) + klo_c);
                    __hip_atomic_fetch_min(&g_full.win[((uint32_t)(base + (int)kk) + ac) & 511], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                {
                    const int reach = base + (int)kk + 2 + (int)cols + 2;
                    pend = reach > pend ? reach : pend;
                }
                // (vertices of the group with a run beyond the table)
                while (l3) {
                    const uint32_t b3 = (uint32_t)__builtin_ctz(l3);
                    l3 &= l3 - 1;
                    const uint32_t vl = (uint32_t)(base + (int)(kk + b3));
                    d2_relax_long(g, vr0 + vl, vl, count, b3 == 0 ? D0 : b3 == 1 ? D1 : D2, rc, lane);
                    pend = (int)vl + 258 > pend ? (int)vl + 258 : pend;
                }
                dirty = true;
                k = kk + 3;
                asm volatile("" ::: "memory");
            }
        } else
        for (;;) {
            const unsigned long long rest = k < 64 ? (em >> k) << k : 0ull;
            if (!rest) break;
            const uint32_t kk = (uint32_t)__ffsll((long long)rest) - 1;
            // A match is at least 3 long: the depths of three consecutive vertices are final together.  D is final below lane k;
            // the group needs it through lane kk + 2: a few lanes are recomputed one by one, many by the scan over all of them.
            // (their offer rows are asked for first: they do not depend on the depths)  Longer runs than the loop above takes:
            // vertex by vertex, lane = length.
#ifdef SPNG_DEFLATE_PROF
            if (threadIdx.x == 0) g_prof[8] += 1ull << 20;
#endif
            uint32_t row[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) row[q] = ((uint32_t)lane < cols && kk + q < 64) ? g_ptab[(kk + q) * D2_PSTRIDE + lane] : ~0u;
            if (dirty) {
                W = act ? g_full.win[(uint32_t)v & 511] : ~0ull;
                Wd = (uint32_t)(W >> 32) < DINF ? (uint32_t)(W >> 32) : DINF;
                if (kk + 3 - k <= 8) {
                    const uint32_t hi = kk + 2 < 63 ? kk + 2 : 63;
                    for (uint32_t x = k; x <= hi; ++x) {
                        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)D, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                        const uint32_t cand = prev + cin, nd = cand < Wd ? cand : Wd;
                        D = (uint32_t)lane == x ? nd : D;
                    }
                } else D = minplus_scan(cin, Wd, carry, lane);
            }
            {
                for (uint32_t kq = kk; kq < kk + 3; ++kq) {
                    if (!(kq < 64 && ((em >> kq) & 1))) continue;
                    const uint32_t vv = (uint32_t)(base + (int)kq);
                    if (count - vv < 3) continue;
                    const uint32_t Dk = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)kq);
                    if ((longm >> kq) & 1) {
                        d2_relax_long(g, vr0 + vv, vv, count, Dk, rc, lane);
                        pend = (int)vv + 258 > pend ? (int)vv + 258 : pend;
                        continue;
                    }
                    const uint32_t pk = kq == kk ? row[0] : kq == kk + 1 ? row[1] : row[2];
                    if (pk != ~0u) {
                        const uint32_t L = 3u + (uint32_t)lane;
                        const uint64_t key = (uint64_t)(Dk + (pk >> 20) + rc[0]) << 32 | (258u - L) << 23 | ((pk & 0xfffffu) + (1u << 15));
                        __hip_atomic_fetch_min(&g_full.win[(vv + L) & 511], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    pend = (int)(vv + cols + 2) > pend ? (int)(vv + cols + 2) : pend;
                }
            }
            dirty = true;
            k = kk + 3;
            // (LDS serves a wave's operations in order: the next read of the ring sees these keys without waiting for them here)
            asm volatile("" ::: "memory");
        }
        if (dirty) {
            W = act ? g_full.win[(uint32_t)v & 511] : ~0ull;
            Wd = (uint32_t)(W >> 32) < DINF ? (uint32_t)(W >> 32) : DINF;
            D = minplus_scan(cin, Wd, carry, lane);
        }
        // the way in: the literal only when strictly cheaper than what the matches offer.
        // run << 16 | decade << 8 | distance (low byte; high bits from bit 25 up); literal: 1 << 16 | 0xff00
        if (act && v >= 1) {
            const uint32_t low = (uint32_t)W, dist = low & 0x7fff;
            g.up[v] = D < Wd ? 0x0001ff00u : (258u - (low >> 23)) << 16 | (((low >> 15) & 31u) - 1u) << 8 | (dist & 0xff) | (dist >> 8) << 25;
        }
        const uint32_t last = count - (uint32_t)base < 63 ? count - (uint32_t)base : 63;      // (base <= count in every batch)
        carry = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)last);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// minimize() backwards (:282-320), as full_backward with the wider ways-in
__device__ __attribute__((noinline)) void d2_backward(const D2Arrays g_, const gbyte *in_, uint64_t bbase_, uint64_t vr0_, uint32_t count_, int lane)
{
    const D2Arrays g = uni_arrays(g_);
    const gbyte *in = UNIP(const gbyte *, in_);
    const uint64_t bbase = uni64(bbase_), vr0 = uni64(vr0_);
    const uint32_t count = UNI(count_);
    DLds &s = g_lds;
    const uint32_t sh = (uint32_t)(vr0 & 63);
    // (a batch of literals only: the forward pass wrote no ways in for it)
    auto way_in = [&](uint32_t c) -> uint32_t { return g.litb[(sh + c) >> 6] ? 0x0001ff00u : g.up[c]; };
    for (int i = lane; i < 320; i += 64) s.freq[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    uint32_t hi = count;
    // (the ways into the next 64 vertices -- and the literal in front of each -- are fetched ahead on the guess that the path
    //  leaves this batch exactly at its end, as it does through literals; a longer hop refetches)
    uint32_t u_next = ((uint32_t)lane < hi) ? way_in(hi - (uint32_t)lane) : 0u, hi_next = hi;
    uint32_t l_next = ((uint32_t)lane < hi) ? (uint32_t)in[bbase + hi - (uint32_t)lane - 1] : 0u;
    uint32_t litrun = 0;                                       // batches of nothing but literal steps in a row
    for (;;) {
        const bool valid = (uint32_t)lane <= hi;
        const uint32_t c = valid ? hi - (uint32_t)lane : 0u;
        uint32_t u = u_next, lit = l_next;
        if (hi_next != hi) { u = (valid && c > 0) ? way_in(c) : 0u; lit = (valid && c > 0) ? (uint32_t)in[bbase + c - 1] : 0u; }
        if (hi >= 64) {
            hi_next = hi - 64;
            u_next = ((uint32_t)lane < hi_next) ? way_in(hi_next - (uint32_t)lane) : 0u;
            l_next = ((uint32_t)lane < hi_next) ? (uint32_t)in[bbase + hi_next - (uint32_t)lane - 1] : 0u;
        }
        const uint32_t len = (u >> 16) & 0x1ff;                // 0: vertex 0 (or nothing)
        unsigned long long pm = 0;
        uint32_t pos = 0;
        if (!__ballot(valid && c > 0 && len != 1)) { pm = __ballot(valid); pos = 64; litrun += 1; }
        else {
            litrun = 0;
            while (pos < 64) {
                pm |= 1ull << pos;
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)pos);
                if (!l) break;
                pos += l;
            }
        }
        const bool on = ((pm >> lane) & 1) != 0;
        if (valid && c < count) g.pathb[c] = on ? 1 : 0;
        if (on && c > 0) {
            const uint32_t nxt = c - len;
            g.step[nxt] = u;
            if (len == 1) atomicAdd(&s.freq[lit], 1u);           // (the literal in front of c: in[bbase + c - 1])
            else { atomicAdd(&s.freq[256 | run_decade(len)], 1u); atomicAdd(&s.freq[288 + ((u >> 8) & 0xff)], 1u); }
        }
        if (hi < 64 || pos > hi) break;                        // vertex 0 was in this batch
        if (pos < 64) break;                                   // (cannot happen: a hop of length 0 above vertex 0)
        const uint32_t nhi = hi - pos;
        // vertices the leaving hop jumped over are not on the path
        for (uint32_t cc = nhi + 1 + (uint32_t)lane; cc + 64 <= hi; cc += 64) g.pathb[cc] = 0;
        hi = nhi;
        // Two batches of literals in a row -- incompressible data: the batches below that the forward pass marked "literal ways
        // in only" (litb) are walked without looking at their ways in at all: every vertex is on the path, its step is the
        // literal in front of it.  (A load per batch cannot hide behind the few instructions a batch of literals takes; here
        // the bytes travel three batches ahead.)
        if (litrun >= 2 && hi >= 128) {
            const uint32_t q_hi = (sh + hi) >> 6;
            const uint32_t lbv = (uint32_t)lane <= q_hi ? (uint32_t)g.litb[q_hi - (uint32_t)lane] : 0u;
            const unsigned long long nz = __ballot(lbv == 0);
            const uint32_t z = nz ? (uint32_t)__ffsll((long long)nz) - 1 : 64u;     // aligned batches q_hi .. q_hi - z + 1
            if (z >= 2) {
                int c_lo = (int)(64 * (q_hi - z + 1)) - (int)sh;                 // their lowest vertex (vertex 0 has no way in)
                c_lo = c_lo < 1 ? 1 : c_lo;
                auto lit_at = [&](int top) -> uint32_t { const int cc = top - lane; return cc >= c_lo ? (uint32_t)in[bbase + (uint32_t)cc - 1] : 0u; };
                uint32_t l0 = lit_at((int)hi), l1 = lit_at((int)hi - 64), l2 = lit_at((int)hi - 128);
                for (int top = (int)hi; top >= c_lo; top -= 64) {
                    const uint32_t lt = l0;
                    l0 = l1; l1 = l2; l2 = lit_at(top - 192);
                    const int cc = top - lane;
                    if (cc >= c_lo) {
                        if ((uint32_t)cc < count) g.pathb[cc] = 1;
                        g.step[cc - 1] = 0x0001ff00u;
                        atomicAdd(&s.freq[lt], 1u);
                    }
                }
                hi = (uint32_t)(c_lo - 1);
                litrun = 0;
            }
        }
    }
    if (lane == 0) s.freq[256] = 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Stream.writeBlock (DeflatorBuffers.Stream.swift:440-709), full form, over the records of the search kernel
__device__ __attribute__((noinline)) Bits d2_block(Bits b_, const D2Arrays g_, const gbyte *in_, uint64_t bbase_, uint64_t vr0_, uint32_t count_, uint32_t cap_,
                                                   bool final_, int iterations_, bool generic_, int lane)
{
    Bits b = uni_bits(b_);
    const D2Arrays g = uni_arrays(g_);
    const gbyte *in = UNIP(const gbyte *, in_);
    const uint64_t bbase = uni64(bbase_), vr0 = uni64(vr0_);
    const uint32_t count = UNI(count_), cap = UNI(cap_);
    const bool final = UB(final_), generic = UB(generic_);
    const int iterations = (int)UNI(iterations_);
    DLds &s = g_lds;
    FPROF(0);
    d2_skip_rule(g, vr0, count, cap, lane);
    FPROF(1);
    for (int i = generic ? -iterations : 0;;) {
        d2_forward(g, in, bbase, vr0, count, lane);
        FPROF(2);
        d2_backward(g, in, bbase, vr0, count, lane);
        FPROF(3);
        build_tree(s.freq, 286, 15, s.ll, lane);
        build_tree(s.freq + 288, 30, 15, s.dl, lane);
        ++i;
        FPROF(4);
        if (!(i < iterations)) break;
        full_depths_update(lane);
    }
    b = uni_bits(write_tables(b, final, lane));
    FPROF(5);
    // writeBlock(with:) (:661-707): the path's terms, 64 vertices at a time
    // (what a batch needs travels three batches ahead: a batch of literals is a few hundred cycles of work, a load two thousand)
    uint32_t pbq[3], stq[3], ltq[3];
    auto ask = [&](uint32_t vn, uint32_t &pbv, uint32_t &stv, uint32_t &ltv) {
        const bool inn = vn < count;
        pbv = inn ? g.pathb[vn] : 0u; stv = inn ? g.step[vn] : 0u; ltv = inn ? in[bbase + vn] : 0u;
    };
#pragma unroll
    for (int q = 0; q < 3; ++q) ask(64u * q + (uint32_t)lane, pbq[q], stq[q], ltq[q]);
    for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        const uint32_t v = b0 + lane;
        const bool on = v < count && pbq[0] != 0;
        const uint32_t st = stq[0], lt = ltq[0];
        pbq[0] = pbq[1]; stq[0] = stq[1]; ltq[0] = ltq[1];
        pbq[1] = pbq[2]; stq[1] = stq[2]; ltq[1] = ltq[2];
        ask(v + 192, pbq[2], stq[2], ltq[2]);
        uint64_t bits = 0; uint32_t nb = 0;
        if (on) {
            const uint32_t cnt = (st >> 16) & 0x1ff, dd = (st >> 8) & 0xff;
            if (cnt == 1) bits = literal_bits(s, lt, nb);
            else {
                const uint32_t off = (st & 0xff) | (st >> 25) << 8, rd = run_decade(cnt);
                bits = match_bits(s, rd, run_extra_value(cnt, rd), dd, dist_extra_value(off, dd), nb);
            }
        }
        bulk_put(s, b, bits, nb, lane);
        maybe_drain(s, b, lane);
    }
    put(s, b, s.lcode[256], s.ll[256], lane);
    maybe_drain(s, b, lane);
    FPROF(6);
    // resetGraph -> Depths.generalize (Depths.swift:88-98)
    for (uint32_t i = lane; i < 542; i += 64) {
        const uint32_t x = g_full.depths[i], d = depth_default(i);
        g_full.depths[i] = (uint8_t)((x & d) + ((x ^ d) >> 1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    return b;
}

__global__ __launch_bounds__(64) void dfl2_parse_kernel(const D2Stream *__restrict__ streams, const uint32_t *__restrict__ pool, spng_result *__restrict__ results, uint32_t parity)
{
    DLds &s = g_lds;
    const D2Stream *sp = streams + blockIdx.x;
    const int lane = threadIdx.x;
    D2State *state = (D2State *)uni64((uint64_t)sp->state);
    if (UNI(state->done) || UNI(state->fail)) return;
    const gbyte *in = (const gbyte *)uni64((uint64_t)sp->src);
    const uint64_t n = uni64(sp->src_len);
    const int32_t format = (int32_t)UNI(sp->format);
    D2Arrays g;
    g.vinfo = (const uint16_t *)uni64((uint64_t)(parity ? sp->vinfo2 : sp->vinfo)); g.bbase = (const uint64_t *)uni64((uint64_t)(parity ? sp->bbase2 : sp->bbase));
    g.bwords = (const uint32_t *)uni64((uint64_t)(parity ? sp->bwords2 : sp->bwords)); g.emask = (uint64_t *)uni64((uint64_t)sp->emask);
    g.up = (gword *)uni64((uint64_t)sp->up); g.step = (gword *)uni64((uint64_t)sp->step); g.pathb = (gbyte *)uni64((uint64_t)sp->pathb);
    g.litb = (gbyte *)uni64((uint64_t)sp->litb);
    g.pool = pool;
    const int lv = (int)UNI(sp->level) > 13 ? 13 : (int)UNI(sp->level);
    const int iterations = lv - 7;
    const bool more = UNI(sp->more) != 0;                      // spng_deflate_resume_batch: the input goes on behind src_len
#ifdef SPNG_DEFLATE_PROF
    if (lane < 12) g_prof[lane] = lane == 11 ? __builtin_readcyclecounter() : 0;
#endif

    for (int i = lane; i < OUTB / 4; i += 64) s.out32[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    Bits b = {uni64(state->acc), UNI(state->nacc), uni64(state->total), uni64(state->total), (gbyte *)uni64((uint64_t)sp->dst), uni64(sp->dst_cap),
              UNI(state->overflow) != 0};
    uint64_t pos = uni64(state->pos);
    uint32_t limit = UNI(state->limit);
    bool generic = UNI(state->generic) != 0;
    const uint64_t rb = uni64(state->rb), re = uni64(state->re);
    if (pos == 0 && b.total == 0 && b.nacc == 0) {
        // the stream's first round
        if (format == SPNG_FORMAT_ZLIB) {
            // StreamHeader.write (StreamHeader.swift:56-62)
            const uint32_t unpaired = (UNI(sp->exponent) - 8) << 4 | 0x08;
            const uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
            put(s, b, check << 8 | unpaired, 16, lane);
        } else if (format == SPNG_FORMAT_GZIP) {
            // Gzip.StreamHeader.write (Gzip.StreamHeader.swift:84-96); the trailer is appended by gzip.hip
            put(s, b, 0x8b1f, 16, lane); put(s, b, 0x0008, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0, 16, lane); put(s, b, 0xff00, 16, lane);
        }
        for (uint32_t i = lane; i < 542; i += 64) g_full.depths[i] = (uint8_t)depth_default(i);
    } else {
        for (uint32_t i = lane; i < 542; i += 64) g_full.depths[i] = state->depths[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");

    uint32_t tailS = 0, tailI = 0;
    if (n < 3 && more) {}                                      // (nothing can be decided yet)
    else if (n < 3) {
        // Stream.compressBlocks stored tail (:45-60, :417-434)
        put(s, b, 1, 3, lane);
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
        put(s, b, (uint32_t)n, 16, lane); put(s, b, ~(uint32_t)n & 0xffff, 16, lane);
        for (uint64_t k = 0; k < n; ++k) put(s, b, in[k], 8, lane);
        if ((uint64_t)lane < n) { tailS = in[lane]; tailI = (uint32_t)lane * in[lane]; }
        pos = n;
    } else {
        while (pos < re) {
            const uint64_t room = n - pos;
            const uint32_t count = (uint64_t)(limit - 1) < room ? limit - 1 : (uint32_t)room;
            const bool final = !more && pos + count == n;
            b = uni_bits(d2_block(b, g, in, pos, pos - rb, count, limit - 1, final, iterations, generic, lane));
            generic = false;
            pos += count;
            if (!final) limit = 2 * limit < (1u << 21) ? 2 * limit : 1u << 21;     // trees(iterations:) :229
        }
    }
    if (!more && pos >= n) {
        if (format == SPNG_FORMAT_ZLIB) {
            // Adler-32 from the sums the search kernel left (s1 = 1 + S, s2 = N + N * S - I)
            uint32_t S, I;
            if (n < 3) {
                S = tailS; I = tailI;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) { S += __shfl_xor(S, m, 64); I += __shfl_xor(I, m, 64); }
            } else { S = UNI(state->adlerS); I = UNI(state->adlerI); }
            S %= 65521; I %= 65521;
            const uint32_t N = (uint32_t)(n % 65521);
            const uint32_t sum = ((N + (uint64_t)N * S % 65521 + 65521 - I) % 65521) << 16 | (1 + S) % 65521;
            if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);
            put(s, b, sum >> 24, 8, lane); put(s, b, (sum >> 16) & 0xff, 8, lane);
            put(s, b, (sum >> 8) & 0xff, 8, lane); put(s, b, sum & 0xff, 8, lane);
        }
        if (b.nacc) put(s, b, 0, 8 - b.nacc, lane);            // DeflatorOut.pull flushes padding bits
        drain(s, b, b.total, lane);
#ifdef SPNG_DEFLATE_PROF
        if (lane == 0 && blockIdx.x == 0) printf("dfl2_parse prof Mcycles (last round, from %llu): other %llu skip-rule %llu forward %llu backward %llu trees %llu tables %llu emit %llu; offers %llu long %llu\n",
            (unsigned long long)rb, g_prof[0] >> 20, g_prof[1] >> 20, g_prof[2] >> 20, g_prof[3] >> 20, g_prof[4] >> 20, g_prof[5] >> 20, g_prof[6] >> 20, g_prof[9] >> 20, g_prof[10] >> 20);
#endif
        if (lane == 0) {
            spng_result &res = results[UNI(sp->image)];
            res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_DONE; res.reserved = 1;
            res.written = b.total; res.consumed = n; res.aux[0] = res.aux[1] = 0;
            state->done = 1;
        }
        return;
    }
#ifdef SPNG_DEFLATE_PROF
#define D2_PROF_PRINT() if (lane == 0 && blockIdx.x == 0) printf("dfl2_parse prof Mcycles (round from %llu): other %llu skip-rule %llu forward %llu backward %llu trees %llu tables %llu emit %llu; offers %llu long %llu; literal batches %llu packed groups %llu vertex-wise groups %llu\n", \
        (unsigned long long)rb, 0ull, g_prof[1] >> 20, g_prof[2] >> 20, g_prof[3] >> 20, g_prof[4] >> 20, g_prof[5] >> 20, g_prof[6] >> 20, g_prof[9] >> 20, g_prof[10] >> 20, g_prof[0] >> 20, g_prof[7] >> 20, g_prof[8] >> 20)
#else
#define D2_PROF_PRINT()
#endif
    D2_PROF_PRINT();
    // the next round
    drain(s, b, b.total, lane);
    for (uint32_t i = lane; i < 542; i += 64) state->depths[i] = g_full.depths[i];
    if (lane == 0) {
        state->acc = b.acc; state->nacc = b.nacc; state->total = b.total; state->overflow = b.overflow ? 1u : 0u;
        state->pos = pos; state->limit = limit; state->generic = generic ? 1u : 0u;
        state->rb = pos; state->re = d2_round_end(pos, limit, n, more);
        // (where the stream stands: the answer of a push that did not carry its end; overwritten by the round that finishes it)
        spng_result &res = results[UNI(sp->image)];
        res.status = b.overflow ? SPNG_E_OUTPUT_CAPACITY : SPNG_NEED_MORE_INPUT; res.reserved = 1;
        res.written = b.total; res.consumed = pos; res.aux[0] = pos; res.aux[1] = limit;
    }
}

__global__ void dfl2_failed_kernel(const D2Stream *__restrict__ streams, uint32_t count, uint32_t *__restrict__ failed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t f = streams[i].state->fail ? 2u : streams[i].state->done ? 0u : 1u;     // 2: the pool ran dry under it; 1: not finished
    failed[1 + i] = f;
    if (f) atomicAdd(&failed[0], 1u);
}
hipError_t launch_deflate2_failed(const D2Stream *d_streams, uint32_t count, uint32_t *d_failed, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_failed, 0, 4, stream);
    if (e != hipSuccess) return e;
    dfl2_failed_kernel<<<(count + 255) / 256, 256, 0, stream>>>(d_streams, count, d_failed);
    return hipGetLastError();
}

// The first round of a call: a state that arrives zeroed is a stream's beginning; the round = the blocks the input allows now.
__global__ void dfl2_begin_kernel(const D2Stream *__restrict__ streams, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const D2Stream &st = streams[i];
    D2State &t = *st.state;
    if (!t.started) { t.started = 1; t.pos = 0; t.limit = 2048; t.generic = 1; }
    t.fail = 0;
    t.rb = t.pos;
    t.re = st.src_len < 3 ? t.pos : d2_round_end(t.pos, t.limit, st.src_len, st.more != 0);
    t.spos = t.pos; t.slimit = t.limit; t.srb = t.rb; t.sre = t.re;
}
// The search's cursor, from the round it has just searched to the next: the blocks of a round and the limit they leave are a
// function of the positions alone (deflate2_plan), so the search of round r + 1 does not wait for the parse of round r.
__global__ void dfl2_advance_kernel(const D2Stream *__restrict__ streams, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const D2Stream &st = streams[i];
    D2State &t = *st.state;
    const uint64_t n = st.src_len, end = t.sre;
    const bool more = st.more != 0;
    if (n < 3 || end == t.spos) return;
    uint32_t lim = t.slimit;
    for (uint64_t at = t.spos; at < end;) {
        const uint64_t size = (uint64_t)(lim - 1) < end - at ? (uint64_t)(lim - 1) : end - at;
        at += size;
        if (more || at < n) lim = 2 * lim < (1u << 21) ? 2 * lim : 1u << 21;
    }
    t.spos = end; t.slimit = lim;
    t.srb = end; t.sre = (!more && end >= n) ? end : d2_round_end(end, lim, n, more);
}
hipError_t launch_deflate2_begin(const D2Stream *d_streams, uint32_t count, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl2_begin_kernel<<<(count + 255) / 256, 256, 0, stream>>>(d_streams, count);
    return hipGetLastError();
}

uint64_t deflate2_temp_bytes(uint32_t workgroups) { return (uint64_t)workgroups * (SPNG_D3_WAVES * 30 * 64 * 4); }
hipError_t launch_deflate2_search(const D2Stream *d_streams, uint32_t count, uint32_t cps, uint32_t chunk_len, uint32_t *d_pool, unsigned long long *d_pool_next,
                                  uint64_t pool_words, uint32_t *d_temp, uint32_t parity, hipStream_t stream)
{
    if (!count) return hipSuccess;
    hipError_t e = hipMemsetAsync(d_pool_next, 0, 8, stream);
    if (e != hipSuccess) return e;
    dfl3_search_kernel<<<count * cps, SPNG_D3_WAVES * 64, 0, stream>>>(d_streams, cps, chunk_len, d_pool, d_pool_next, pool_words, d_temp, parity);
    dfl2_advance_kernel<<<(count + 255) / 256, 256, 0, stream>>>(d_streams, count);
    return hipGetLastError();
}
hipError_t launch_deflate3_begin(const D3Stream *d_streams, uint32_t count, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl3_begin_kernel<<<(count + 255) / 256, 256, 0, stream>>>(d_streams, count);
    return hipGetLastError();
}
hipError_t launch_deflate3_search(const D3Stream *d_streams, uint32_t count, uint32_t cps, uint32_t chunk_len, uint32_t parity, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl3_search_fast_kernel<<<count * cps, SPNG_D3_WAVES * 64, 0, stream>>>(d_streams, cps, chunk_len, parity);
    dfl3_advance_kernel<<<(count + 255) / 256, 256, 0, stream>>>(d_streams, count);
    return hipGetLastError();
}
hipError_t launch_deflate3_parse(const D3Stream *d_streams, uint32_t count, spng_result *d_results, uint32_t parity, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl3_parse_kernel<<<count, 128, 0, stream>>>(d_streams, d_results, parity);
    return hipGetLastError();
}
hipError_t launch_deflate4_round(const D3Stream *d_streams, uint32_t count, uint32_t max_blocks, spng_result *d_results, uint32_t parity, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl4_walk_kernel<<<count, 64, 0, stream>>>(d_streams, parity);
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u) {           // (grid y stops at 65535)
        const uint32_t ny = count - y0 < 65535u ? count - y0 : 65535u;
        dfl4_block_kernel<<<dim3(max_blocks, ny), 64, 0, stream>>>(d_streams + y0);
    }
    dfl4_scan_kernel<<<count, 256, 0, stream>>>(d_streams, max_blocks, d_results);
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u) {
        const uint32_t ny = count - y0 < 65535u ? count - y0 : 65535u;
        dfl4_place_kernel<<<dim3(max_blocks, ny), 64, 0, stream>>>(d_streams + y0, max_blocks);
    }
    return hipGetLastError();
}
hipError_t launch_deflate2_parse(const D2Stream *d_streams, uint32_t count, const uint32_t *d_pool, spng_result *d_results, uint32_t parity, hipStream_t stream)
{
    if (!count) return hipSuccess;
    dfl2_parse_kernel<<<count, 64, 0, stream>>>(d_streams, d_pool, d_results, parity);
    return hipGetLastError();
}

// bytes of graph scratch a stream of n bytes needs at levels >= 8 (api.hip sizes the slab with it)
uint64_t deflate_graph_vertices(uint64_t n)
{
    const uint64_t cap = (1u << 21) - 1;
    return n + 2 < cap ? n + 2 : cap;
}
uint64_t deflate_graph_bytes(uint64_t vertices)
{
    return ((vertices + 1) * (30 * 4 + 4 + 4 + 1 + 1) + 1024 + 255) & ~(uint64_t)255;
}

// How often does a 4-byte key repeat close by?  Sampled (four windows of 8192 positions, a 4096-entry table of the
// last key per bucket): the full-search kernel is launched with helper waves only for streams whose vertices will
// have edges to relax -- on incompressible input the four-wave form costs a few per cent and buys nothing.
__global__ __launch_bounds__(64) void deflate_density_kernel(const DeflateJob *__restrict__ jobs, uint32_t *__restrict__ dense)
{
    __shared__ uint32_t tab[4096];
    const int lane = threadIdx.x;
    const DeflateJob *jp = jobs + blockIdx.x;
    const gbyte *in = (const gbyte *)uni64((uint64_t)jp->src);
    const uint64_t n = uni64(jp->src_len);
    uint32_t hits = 0, seen = 0;
    if (n >= 64) {
        const uint64_t win = n / 4 < 8192 ? n / 4 : 8192;
        for (int w = 0; w < 4; ++w) {
            const uint64_t from = (uint64_t)w * (n / 4);
            for (int i = lane; i < 4096; i += 64) tab[i] = 0x9e3779b9u + (uint32_t)i;       // (no key hashes to its own filler)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            for (uint64_t p0 = 0; p0 + 4 <= win; p0 += 64) {
                const uint64_t p = from + p0 + lane;
                const bool live = p0 + lane + 4 <= win && p + 4 <= n;
                const uint32_t key = live ? load32(in + p) : 0u;
                const uint32_t h = (key * 0x9E3779B1u) >> 20;
                const bool hit = live && tab[h] == key;
                if (live) tab[h] = key;
                hits += (uint32_t)__popcll(__ballot(hit));
                seen += (uint32_t)__popcll(__ballot(live));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            }
        }
    }
    if (lane == 0) dense[blockIdx.x] = (seen && hits * 20u >= seen) ? 1u : 0u;             // >= 5 % of the positions
}

hipError_t launch_deflate_density(const DeflateJob *d_jobs, uint32_t count, uint32_t *d_dense, hipStream_t stream)
{
    if (!count) return hipSuccess;
    deflate_density_kernel<<<count, 64, 0, stream>>>(d_jobs, d_dense);
    return hipGetLastError();
}

hipError_t launch_deflate_full(const DeflateJob *d_jobs, uint32_t count, bool helpers, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    if (helpers) deflate_full_kernel<4><<<count, 256, 0, stream>>>(d_jobs, d_results);
    else deflate_full_kernel<1><<<count, 64, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

hipError_t launch_deflate(const DeflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream)
{
    if (!count) return hipSuccess;
    deflate_kernel<<<count, 64, 0, stream>>>(d_jobs, d_results);
    return hipGetLastError();
}

}  // namespace spng
