// huffman.hpp -- canonical-Huffman table construction shared by the inflate kernels (gfx950 only).
//
// Restates, for a whole wavefront working on LDS-resident code lengths:
//   tree validation    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:80-174
//   length/distance    Sources/LZ77/LZ77.Composites.swift:19-111
// LUT entries carry base value + extra-bit count, a canonical first-code/count description serves the
// rare codes longer than the LUT index.
#pragma once
#include "common.hpp"

namespace spng {

// LUT entry: [3:0] code length (0 = longer than the LUT index), [7:4] extra bits,
// [9:8] kind, [10] literal / [11] back-reference half that the speculative decoder may take without
// any further check (a real code with a non-zero base), [31:16] literal / base run / base distance.
enum { K_LIT = 0, K_EOB = 1, K_MATCH = 2, K_UNDEF = 3 };
static constexpr uint32_t F_LIT = 1u << 10, F_REF = 1u << 11;
__device__ __forceinline__ uint32_t entry(uint32_t len, uint32_t extra, uint32_t kind, uint32_t value)
{
    const uint32_t fast = len == 0 ? 0u : kind == K_LIT ? F_LIT : (kind == K_MATCH && value != 0) ? F_REF : 0u;
    return len | extra << 4 | kind << 8 | fast | value << 16;
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

// Only one wave of the workgroup builds tables; LDS operations of one wave execute in order, so a
// compiler + counter fence is all the synchronisation the cooperative phases need.
#define WSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup", "local")
#define COMPILER_ORDER() asm volatile("" ::: "memory")

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// inclusive prefix sum inside each row of 16 lanes (DPP row_shr 1, 2, 4, 8; lanes without a source add 0)
__device__ __forceinline__ uint32_t row_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

// exclusive prefix sum over the wave; every lane gets the grand total too
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total, int lane)
{
    uint32_t incl = row_scan(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
    const uint32_t r1 = r0 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
    const uint32_t r2 = r1 + (uint32_t)__builtin_amdgcn_readlane((int)incl, 47);
    incl += lane < 16 ? 0u : lane < 32 ? r0 : lane < 48 ? r1 : r2;
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    return incl - v;
}

// Builds LUT + canonical fallback for `n` code lengths (lens[], LDS).  KIND: 0 lit/len, 1 distance,
// 2 code-length code.  Returns false when the code is not complete (HuffmanTree.size, :80-108).
// `normalizing` restates validate(symbols:normalizing:) (:112-135): 0 or 1 used symbol of length 1
// gives a stub whose unused half the reference leaves uninitialised (K_UNDEF here).
//
// Everything per code length lives in lane l of a vector register (or in 16-entry LDS arrays), never
// in 16-element wave-uniform arrays: those end up in scalar registers and crowd the decoder's
// loop-carried state out into spill slots.
template <int KIND>
__device__ __attribute__((always_inline)) bool build(uint32_t *hist, uint32_t *run, const uint8_t *lens, int n, uint32_t *lut, int lbits,
                                                     uint16_t *sorted, Tree *tree, bool normalizing, int lane,
                                                     uint32_t *ext = nullptr)
{
    const int size = 1 << lbits;
    // ---- histogram of the code lengths (LDS atomics), lane l <- count of length l
    if (lane < 16) { hist[lane] = 0; run[lane] = 0; }
    WSYNC();
    uint32_t used = 0;
    for (int base = 0; base < n; base += 64) {
        const int sym = base + lane;
        const uint32_t my = sym < n ? lens[sym] : 0;
        __hip_atomic_fetch_add(&hist[my], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // hist[0] is ignored
        used += (uint32_t)__popcll(__ballot(my != 0));
    }
    WSYNC();
    const uint32_t c = (lane >= 1 && lane < 16) ? hist[lane] : 0u;
    const uint32_t cnt1 = (uint32_t)__builtin_amdgcn_readlane((int)c, 1);
    if (normalizing && (used == 0 || (used == 1 && cnt1 == 1))) {
        // stub tree (HuffmanTree.swift:52-65)
        uint32_t sym = 0;
        for (int base = 0; base < n; base += 64) {
            const int at = base + lane;
            const unsigned long long m = __ballot(at < n && lens[at] == 1);
            if (m) sym = base + __ffsll((long long)m) - 1;
        }
        for (int j = lane; j - lane < size; j += 64)         // size is a multiple of 64: uniform trip count
            lut[j] = (used && !(j & 1)) ? (KIND == 1 ? dist_entry(sym, 1) : litlen_entry(sym, 1))
                                        : entry(1, 0, K_UNDEF, 0);
        WSYNC();
        return true;
    }
    // complete <=> Kraft sum is exactly 1: sum of count[l] << (15 - l) == 1 << 15 (the interior-node
    // recurrence of the reference, interior = 2 * interior - count[l], is linear in the counts)
    const uint32_t scaled = c << (15 - (lane & 15));           // c is 0 outside lanes 1..15
    if (UNI(wave_sum(scaled)) != 32768u) return false;

    // canonical first code and first sorted slot of each length: exclusive prefix sums over lanes 0..15
    const uint32_t off = row_scan(c) - c;
    const uint32_t first = (row_scan(scaled) - scaled) >> (15 - (lane & 15));
    if (lane >= 1 && lane < 16) {
        tree->first[lane] = (uint16_t)first; tree->count[lane] = (uint16_t)c; tree->offset[lane] = (uint16_t)off;
    }
    for (int j = lane; j - lane < size; j += 64) lut[j] = 0;      // 0 = "longer than lbits"
    WSYNC();

    // ---- canonical codes: rank of a symbol among the symbols of its length, in symbol order
    for (int base = 0; base < n; base += 64) {
        const int sym = base + lane;
        const uint32_t my = sym < n ? lens[sym] : 0;
        // lanes of this batch with the same length as mine: radix match over the 4 bits of a length
        unsigned long long same = ~0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long bk = __ballot((my >> k) & 1);
            same &= (my >> k) & 1 ? bk : ~bk;
        }
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0));
        const uint32_t rank = run[my] + before;             // + those of earlier batches
        const uint32_t f = tree->first[my], o = tree->offset[my];
        COMPILER_ORDER();
        if (before == 0) run[my] += (uint32_t)__popcll(same);      // one lane per length updates the tally
        if (my) {
            const uint32_t e = KIND == 0 ? litlen_entry(sym, my) : KIND == 1 ? dist_entry(sym, my) : meta_entry(sym, my);
            if (sorted) sorted[o + rank] = (uint16_t)sym;
            if (ext) ext[o + rank] = e;                       // (entries in canonical order)
            if ((int)my <= lbits) {
                const uint32_t rev = __brev(f + rank) >> (32 - my);
                for (int j = rev; j < size; j += 1 << my) lut[j] = e;
            }
        }
        WSYNC();
    }
    return true;
}

}  // namespace spng
