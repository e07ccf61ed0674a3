// crc32.hpp -- wave-parallel CRC-32 for gfx950, shared by the PNG chunk framing (chunks.hip) and the gzip
// wrapper (gzip.hip).  swift-hash 0.7.1's CRC32 (the reference's Package.resolved; source not in the checkout):
// the standard reflected CRC-32, polynomial 0xEDB88320, initial value and final xor 0xFFFFFFFF.
#pragma once
#include "common.hpp"

namespace spng {

#define LSYNC() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")

static constexpr uint32_t POLY = 0xedb88320u;

// a * b mod P, reflected bit order (bit 31 = x^0)
__host__ __device__ inline uint32_t multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    while (a) {                                               // (a == 0: the product is 0)
        if (a & m) { p ^= b; a &= ~m; }
        m >>= 1;
        b = b & 1 ? (b >> 1) ^ POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod P
__host__ __device__ inline uint32_t xpow8(uint64_t n)
{
    uint32_t r = 1u << 31, base = 1u << 23;                    // x^0, x^8
    while (n) { if (n & 1) r = multmodp(base, r); base = multmodp(base, base); n >>= 1; }
    return r;
}

// Four 256-entry tables for slicing-by-4 (tab[0] is the plain byte table; tab[k][i] = the CRC register after byte i
// and k zero bytes): four independent LDS lookups per input dword instead of four dependent ones.
static constexpr int CRC_TAB = 4 * 256;
__device__ __forceinline__ void crc_table(uint32_t *tab, int lane)
{
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = (uint32_t)i;
        for (int k = 0; k < 8; ++k) c = c & 1 ? (c >> 1) ^ POLY : c >> 1;
        tab[i] = c;
    }
    LSYNC();
    for (int t = 1; t < 4; ++t) {
        for (int i = lane; i < 256; i += 64) { const uint32_t c = tab[(t - 1) * 256 + i]; tab[t * 256 + i] = (c >> 8) ^ tab[c & 0xff]; }
        LSYNC();
    }
}

typedef uint32_t crc_v4u __attribute__((vector_size(16)));
struct __attribute__((packed)) CrcPV4 { crc_v4u v; };
typedef CrcPV4 __attribute__((address_space(1))) gCrcPV4;

// x^(8 * 2^j) mod P, j = 0 .. 47: what shifts a CRC register past 2^j bytes.  (Round 6: the pieces a wave cuts a buffer into are a
// power of two long, so every factor of the fold is one of these -- a chunk of 8 KiB spent 85 % of its instructions computing
// x^(8 L) and x^(8 n) by square-and-multiply: 8192 small PNG files lexed in 8.4 ms, a third of what their decode takes.)
struct CrcPow { uint32_t v[48]; };
constexpr uint32_t crc_mul_c(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) { if (a & m) p ^= b; b = b & 1 ? (b >> 1) ^ POLY : b >> 1; }
    return p;
}
constexpr CrcPow crc_pow_table()
{
    CrcPow t{};
    t.v[0] = 1u << 23;                                         // x^8
    for (int j = 1; j < 48; ++j) t.v[j] = crc_mul_c(t.v[j - 1], t.v[j - 1]);
    return t;
}
static constexpr CrcPow CRC_POW = crc_pow_table();
static_assert(CRC_POW.v[1] == crc_mul_c(1u << 23, 1u << 23), "x^16");

// standard CRC-32 of p[0 .. n), continuing from `crc` (a finished CRC-32 value; 0 for a fresh one); wave-uniform result.
// 64 pieces of L = 2^k bytes (k >= 4, 64 L >= n), one per lane, RIGHT-aligned: lane i takes the L bytes that end (63 - i) L bytes in
// front of the end, so the lanes in front of the first byte have nothing, the first lane with bytes has a shorter piece and starts
// from the running register (the initial value travels with it: no x^(8 n) afterwards), and every shift of the fold
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P is by L 2^level bytes: a constant of CRC_POW.  16 bytes per load,
// slicing-by-4.  copy_to: the bytes are also copied there (the IDAT payloads of the chunk lexer: one pass over the data).
__device__ __forceinline__ uint32_t wave_crc32(const uint32_t *tab, const gbyte *p, uint64_t n, uint32_t crc, int lane, gbyte *copy_to = nullptr)
{
    if (!n) return crc;
    uint32_t k = 4;
    if (n > 1024) k = 64 - (uint32_t)__builtin_clzll((unsigned long long)((n - 1) >> 6));     // smallest k with 2^k * 64 >= n
    const uint64_t L = 1ull << k;
    const uint64_t behind = (uint64_t)(63 - lane) << k;        // bytes between my piece's end and the buffer's
    uint32_t c = 0;
    if (behind < n) {
        const uint64_t end = n - behind;
        const bool first = end <= L;                           // (the piece that starts at byte 0)
        uint64_t i = first ? 0 : end - L;
        if (first) c = crc ^ 0xffffffffu;
        for (uint64_t head = (end - i) & 15; head; --head, ++i) {
            const uint32_t b = p[i];
            if (copy_to) copy_to[i] = (uint8_t)b;
            c = tab[(c ^ b) & 0xff] ^ (c >> 8);
        }
        for (; i < end; i += 16) {
            const crc_v4u v = ((const gCrcPV4 *)(p + i))->v;
            if (copy_to) ((gCrcPV4 *)(copy_to + i))->v = v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c ^= v[q];
                c = tab[768 + (c & 0xff)] ^ tab[512 + ((c >> 8) & 0xff)] ^ tab[256 + ((c >> 16) & 0xff)] ^ tab[c >> 24];
            }
        }
    }
#pragma unroll
    for (int lv = 0; lv < 6; ++lv) {
        const uint32_t other = (uint32_t)__shfl_down((int)c, 1 << lv, 64);
        const uint32_t shifted = multmodp(CRC_POW.v[k + lv], c);
        if ((lane & ((2 << lv) - 1)) == 0) c = shifted ^ other;
    }
    return UNI(c) ^ 0xffffffffu;
}


}  // namespace spng
