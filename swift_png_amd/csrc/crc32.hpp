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

typedef uint32_t crc_v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) CrcPV4 { crc_v4u v; };
typedef CrcPV4 __attribute__((address_space(1))) gCrcPV4;

// standard CRC-32 of p[0 .. n), continuing from `crc` (a finished CRC-32 value; 0 for a fresh one); wave-uniform result.
// 64 equal pieces, one per lane (16 bytes per load, slicing-by-4), folded with crc(A || B) = crc(A) * x^(8 |B|) + crc(B)
// in GF(2)[x] / P (the raw, zero-initialised CRC is linear), the per-level shift factor being the previous one squared;
// the < 64 leftover bytes go through the byte table.
__device__ __forceinline__ uint32_t wave_crc32(const uint32_t *tab, const gbyte *p, uint64_t n, uint32_t crc, int lane)
{
    const uint64_t L = n / 64;
    uint32_t c = 0;
    if (L) {
        const gbyte *q = p + (uint64_t)lane * L;
        uint64_t i = 0;
        for (; i + 16 <= L; i += 16) {
            const crc_v4u v = ((const gCrcPV4 *)(q + i))->v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c ^= v[k];
                c = tab[768 + (c & 0xff)] ^ tab[512 + ((c >> 8) & 0xff)] ^ tab[256 + ((c >> 16) & 0xff)] ^ tab[c >> 24];
            }
        }
        for (; i < L; ++i) c = tab[(c ^ q[i]) & 0xff] ^ (c >> 8);
        uint32_t pw = xpow8(L);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t other = (uint32_t)__shfl_down((int)c, 1 << k, 64);
            if ((lane & ((2 << k) - 1)) == 0) c = multmodp(pw, c) ^ other;
            pw = multmodp(pw, pw);
        }
        c = UNI(c);
    }
    for (uint64_t i = 64 * L; i < n; ++i) c = tab[(c ^ UNI(p[i])) & 0xff] ^ (c >> 8);
    // the raw CRC of the bytes; now the initial value (the running CRC, un-finalised) shifted past them
    return c ^ multmodp(xpow8(n), crc ^ 0xffffffffu) ^ 0xffffffffu;
}


}  // namespace spng
