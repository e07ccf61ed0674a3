// unfilter.hip -- PNG scanline reconstruction (None/Sub/Up/Average/Paeth) for gfx950.
//
// Replaces PNG.Decoder.defilter (Sources/PNG/Decoding/PNG.Decoder.swift:152-196), PNG.paeth
// (Sources/PNG/PNG.swift:124-147) and, through the job geometry, the row/pass walker of
// PNG.Decoder.push (:59-140).  Byte arithmetic is u8 wrap-around exactly as in the reference.
//
// Two kernels: `unfilter_pk_kernel` (bpp 4 and 8: RGBA8, VA16, RGBA16 -- the formats of every BASELINE config; second half
// of this file) and `unfilter_kernel`, the byte-wise form for bpp 1, 2, 3 and 6, described first:
//
// Parallelisation.  Pixel (x, y) of a sub-image depends on (x-1, y), (x, y-1) and (x-1, y-1), so
// the dependency DAG is a 2-D wavefront.  One wave (64 lanes) owns a band of 64 consecutive rows:
// lane r reconstructs row r and runs one pixel ("unit" = bpp bytes) behind lane r-1, so at step t
// lane r works on unit t - r.  The value lane r needs from the row above is exactly what lane r-1
// produced one step earlier, and arrives through a single DPP wave_shr:1 move -- no LDS, no
// barrier.  The left neighbour and the upper-left neighbour are register carries.
// A workgroup of NW waves owns one scanline chain (image / Adam7 sub-image): wave w takes bands
// w, w+NW, ... and the bands run as a software pipeline, band j trailing band j-1 by three tiles.
// The row above a band (last row of the previous band) is read back from the output raster with
// L1-bypassing (sc1) loads; per-wave progress counters in LDS order producer and consumer (the
// producer drains its stores with vmcnt(0) before publishing).  No workgroup barriers in the loop.
//
// Memory.  Row r of a tile covers the *skewed* window of units [T*P - r, (T+1)*P - r): since
// rows are pitch+1 bytes apart (never aligned) the loads are unaligned 16-byte loads anyway, so
// the skew costs nothing, and inside LDS every lane walks the same column index.  The loads of
// tile i+1 are issued (16 B/lane, 16 consecutive lanes cover one row segment) before tile i is
// reconstructed and are committed to LDS after it has been written back, so HBM latency hides
// under the arithmetic.  Tiles are reconstructed in place in LDS and written back with 16 B/lane
// stores.  Rows are padded to TB+16 bytes so that the per-lane 16-byte column accesses are
// bank-conflict free.
#include "common.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };       // 16 bytes, alignment 1

#ifndef SPNG_UNF_NW
#define SPNG_UNF_NW 4                                    // waves per scanline chain
#endif

#ifndef SPNG_UNF_P4
#define SPNG_UNF_P4 64                                   // tile width in units for bpp <= 4
#endif
#ifndef SPNG_UNF_PSUB
#define SPNG_UNF_PSUB 16                                 // tile width in dword units for pixels of 1 and 2 bytes
#endif
#ifndef SPNG_UNF_P8
#define SPNG_UNF_P8 32                                   // tile width in units for bpp > 4
#endif

// BPP: bytes of a unit (what a lane reconstructs per step and the lanes' windows are skewed by); PX: bytes of a pixel -- the distance
// of the left and upper-left neighbours.  PX < BPP (round 6): pixels of 1 and 2 bytes ride four and two to a unit of four bytes.
template <int BPP, int PX = BPP, int PT = 0> struct Cfg {          // (PT: tile width in units when it is not the format's own)
    static constexpr int P    = PT ? PT : (PX != BPP) ? SPNG_UNF_PSUB : (BPP <= 4) ? SPNG_UNF_P4 : SPNG_UNF_P8;   // units per tile window
    static constexpr int K    = (63 + P - 1) / P;        // producer tiles a consumer tile reaches into
    static constexpr int TB   = P * BPP;                 // bytes per row per tile (multiple of 16)
    static constexpr int ROWB = TB + 16;                 // LDS row stride: conflict-free b128 columns
    static constexpr int CPR  = TB / 16;                 // 16-byte chunks per row
    static_assert(TB % 16 == 0, "tile row must be a whole number of 16-byte chunks");
};

// 16 bytes of the window [off, off+16) of a row of `pitch` bytes at p; bytes outside the row read 0.
__device__ __forceinline__ u32x4 load_window(const uint8_t *p, int64_t off, int64_t pitch)
{
    if (off >= 0 && off + 16 <= pitch) return ((const U128u *)(p + off))->v;
    u32x4 v = {0, 0, 0, 0};
    if (off + 16 <= 0 || off >= pitch) return v;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int64_t i = off + k;
        uint32_t b = (i >= 0 && i < pitch) ? p[i] : 0u;
        w[k >> 2] |= b << (8 * (k & 3));
    }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}

// same, for bytes another wave of this workgroup wrote: bypass this CU's L1 (sc1 / nt loads are
// served by the XCD's L2, where the producer's write-through stores already are)
__device__ __forceinline__ u32x4 load_window_l2(const uint8_t *p, int64_t off, int64_t pitch)
{
    u32x4 v = {0, 0, 0, 0};
    if (off + 16 <= 0 || off >= pitch) return v;
    if (off >= 0 && off + 16 <= pitch && (((uintptr_t)(p + off)) & 7) == 0) {
        const unsigned long long *q = (const unsigned long long *)(p + off);
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
        return v;
    }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int64_t i = off + k;
        uint32_t b = (i >= 0 && i < pitch) ? __builtin_nontemporal_load(p + i) : 0u;
        w[k >> 2] |= b << (8 * (k & 3));
    }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}

__device__ __forceinline__ void store_window(uint8_t *p, int64_t off, int64_t pitch, u32x4 v)
{
    if (off >= 0 && off + 16 <= pitch) { ((U128u *)(p + off))->v = v; return; }
    if (off + 16 <= 0 || off >= pitch) return;
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int64_t i = off + k;
        if (i >= 0 && i < pitch) p[i] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
    }
}

// PNG.paeth (PNG.swift:124-147): a if pa <= pb && pa <= pc, else b if pb <= pc, else c.
__device__ __forceinline__ uint32_t paeth(uint32_t a, uint32_t b, uint32_t c)
{
    int pa = abs((int)b - (int)c), pb = abs((int)a - (int)c), pc = abs((int)a + (int)b - 2 * (int)c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// value of lane-1 (lane 0 receives `top`)
__device__ __forceinline__ uint32_t from_lane_above(uint32_t mine, uint32_t top)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)top, (int)mine, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ bool skip_status(int32_t s)
{
    // the reference assigns no rows in a push whose inflate threw (PNG.Decoder.swift:57)
    // (nor is there anything to assign when the device gave up on the stream)
    return (s >= 16 && s < 48) || s == SPNG_E_REFERENCE_UNDEFINED || s == SPNG_E_DEVICE;
}

// ---- reconstruction of one tile: generic byte-wise form (any bpp) ----------------------------
// (PAETH = false: no row of the band is filtered with Paeth -- its arithmetic is not there)
template <int BPP, int P, bool PAETH>
__device__ __forceinline__ void reconstruct_generic(uint8_t *tile, int rowb, int lane, uint32_t ft,
                                                    int64_t ux0, uint32_t (&o)[BPP], uint32_t (&bprev)[BPP])
{
    uint8_t *mine = tile + (1 + lane) * rowb;
#pragma unroll 2
    for (int t = 0; t < P; ++t) {
        const bool interior = ux0 + t > 0;               // unit 0 has no left / upper-left neighbour
        uint32_t b[BPP];
#pragma unroll
        for (int k = 0; k < BPP; ++k) b[k] = from_lane_above(o[k], tile[t * BPP + k]);
#pragma unroll
        for (int k = 0; k < BPP; ++k) {
            const uint32_t a = interior ? o[k] : 0u;
            const uint32_t c = interior ? bprev[k] : 0u;
            const uint32_t x = mine[t * BPP + k];
            uint32_t pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b[k];
            else if (ft == 3) pred = (a + b[k]) >> 1;
            else if (PAETH && ft == 4) pred = paeth(a, b[k], c);
            o[k] = (x + pred) & 0xffu;
            mine[t * BPP + k] = (uint8_t)o[k];
            bprev[k] = b[k];
        }
    }
}

// ---- reconstruction of one tile: pixels of 1 or 2 bytes, four or two to a dword unit (round 6) --------------------------------
// The byte-wise form above costs a pixel of one byte ~40 VALU operations and three LDS accesses (indexed-8 at 15 % of peak).  Here a
// unit is a dword whatever the pixel: one LDS read and one write per four bytes, the row above's dword by DPP; inside the dword the
// bytes are reconstructed one after the other (byte k's left neighbour is byte k - PX of the dword, or of the dword before).
template <int PX, int P, bool PAETH>
__device__ __forceinline__ void reconstruct_subunit(uint8_t *tile, int rowb, int lane, uint32_t ft, int64_t ux0, uint32_t &o, uint32_t &bprev)
{
    const uint32_t m_sub = ft == 1 ? 0xffu : 0u, m_up = ft == 2 ? 0xffu : 0u, m_avg = ft == 3 ? 0xffu : 0u, m_pae = ft == 4 ? 0xffu : 0u;
    uint8_t *mine = tile + (1 + lane) * rowb;
#pragma unroll 2
    for (int t = 0; t < P; ++t) {
        const bool interior = ux0 + t > 0;               // unit 0's first pixel has no left / upper-left neighbour
        const uint32_t b4 = from_lane_above(o, *(const uint32_t *)(tile + 4 * t));
        const uint32_t x4 = *(const uint32_t *)(mine + 4 * t);
        uint32_t a[PX], c[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            a[k] = interior ? (o >> (8 * (4 - PX + k))) & 0xffu : 0u;
            c[k] = interior ? (bprev >> (8 * (4 - PX + k))) & 0xffu : 0u;
        }
        uint32_t r4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = (x4 >> (8 * k)) & 0xffu, b = (b4 >> (8 * k)) & 0xffu;
            const uint32_t l = a[k % PX], ul = c[k % PX];
            uint32_t pred = (l & m_sub) | (b & m_up) | (((l + b) >> 1) & m_avg);
            if (PAETH) pred |= paeth(l, b, ul) & m_pae;
            const uint32_t v = (x + pred) & 0xffu;
            r4 |= v << (8 * k);
            a[k % PX] = v; c[k % PX] = b;
        }
        o = r4; bprev = b4;
        *(uint32_t *)(mine + 4 * t) = r4;
    }
}

// ---- reconstruction of one tile: 4 or 8 bytes per unit, two packed-u16 halves per dword ------
// even bytes live in `lo` (x & 0x00ff00ff), odd bytes in `hi` ((x >> 8) & 0x00ff00ff); sums of two
// bytes cannot carry across the 16-bit lanes, signed differences use the packed-i16 VALU ops.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t paeth_pk(uint32_t a, uint32_t b, uint32_t c)
{
    const s16x2 va = __builtin_bit_cast(s16x2, a), vb = __builtin_bit_cast(s16x2, b), vc = __builtin_bit_cast(s16x2, c);
    const s16x2 d0 = vb - vc, d1 = va - vc, ds = d0 + d1;
    const s16x2 pa = __builtin_elementwise_max(d0, -d0), pb = __builtin_elementwise_max(d1, -d1),
                pc = __builtin_elementwise_max(ds, -ds);
    const s16x2 fifteen = {15, 15};
    // sign masks per 16-bit lane; `opaque` keeps them bit masks (v_bfi) instead of per-half compares
    const uint32_t nota = opaque(__builtin_bit_cast(uint32_t, (__builtin_elementwise_min(pb, pc) - pa) >> fifteen));   // pa > min(pb, pc)
    const uint32_t usec = opaque(__builtin_bit_cast(uint32_t, (pc - pb) >> fifteen));
    const uint32_t bc = (c & usec) | (b & ~usec);
    return (bc & nota) | (a & ~nota);
}

// ---- reconstruction of one tile: 3 or 6 bytes per unit, a unit in one or two dwords (round 5) ----------------------------------
// The byte-wise form above spends ~40 VALU operations and two LDS accesses per BYTE.  Here a lane takes 48 bytes of its row per
// step (three aligned ds_read_b128: 16 RGB8 or 8 RGB16 units), cuts them into units with v_alignbyte -- an RGB8 unit rides in the
// low three bytes of a dword, an RGB16 unit in one and a half, the spare byte(s) carry a neighbour's bytes and are never looked
// at: every operation below is byte-wise or per 16-bit half -- runs the dword arithmetic of the 4- and 8-byte kernel on them
// (~50 operations per dword: see phase_pk), and packs the results with v_perm.  The row above still arrives by DPP, one unit late.
template <int BPP> struct Pu {
    static constexpr int DW = (BPP + 3) / 4;             // dwords a unit rides in
    static constexpr int G = 48 / BPP;                   // units per step
};

template <int BPP, int P, bool PAETH>
__device__ __forceinline__ void reconstruct_packed(uint8_t *tile, int rowb, int lane, uint32_t ft,
                                                   uint32_t (&o)[Pu<BPP>::DW], uint32_t (&bprev)[Pu<BPP>::DW])
{
    constexpr int DW = Pu<BPP>::DW, G = Pu<BPP>::G, STEPS = P * BPP / 48;
    static_assert(P * BPP % 48 == 0 && (BPP == 3 || BPP == 6), "a tile row is a whole number of 48-byte steps");
    constexpr uint32_t M = 0x00ff00ffu, SEL_LO = 0x0c020c00u, SEL_HI = 0x0c030c01u;
    const uint32_t m_sub = ft == 1 ? ~0u : 0u, m_up = ft == 2 ? ~0u : 0u, m_avg = ft == 3 ? ~0u : 0u, m_pae = ft == 4 ? ~0u : 0u;
    uint8_t *mine = tile + (1 + lane) * rowb;
    uint32_t a_lo[DW], a_hi[DW], c_lo[DW], c_hi[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) { a_lo[d] = o[d] & M; a_hi[d] = (o[d] >> 8) & M; c_lo[d] = bprev[d] & M; c_hi[d] = (bprev[d] >> 8) & M; }
    // unit j of the twelve dwords w: dword d of it (the bytes past the unit's end are whatever follows)
    auto cut = [](const uint32_t (&w)[12], int j, int d) -> uint32_t {
        const int at = BPP * j + 4 * d, k = at >> 2, sh = at & 3;
        if (sh == 0) return w[k];
        return k + 1 < 12 ? __builtin_amdgcn_alignbyte(w[k + 1], w[k], (uint32_t)sh) : w[k] >> (8 * sh);
    };
    u32x4 xa[3], xb[3], ta[3], tb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { xa[i] = *(const u32x4 *)(mine + 16 * i); ta[i] = *(const u32x4 *)(tile + 16 * i); }
#pragma unroll
    for (int g = 0; g < STEPS; ++g) {
        if (g + 1 < STEPS) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { xb[i] = *(const u32x4 *)(mine + 48 * (g + 1) + 16 * i); tb[i] = *(const u32x4 *)(tile + 48 * (g + 1) + 16 * i); }
        }
        const uint32_t w[12] = {xa[0].x, xa[0].y, xa[0].z, xa[0].w, xa[1].x, xa[1].y, xa[1].z, xa[1].w, xa[2].x, xa[2].y, xa[2].z, xa[2].w};
        const uint32_t t[12] = {ta[0].x, ta[0].y, ta[0].z, ta[0].w, ta[1].x, ta[1].y, ta[1].z, ta[1].w, ta[2].x, ta[2].y, ta[2].z, ta[2].w};
        uint32_t v[G][DW];
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int d = 0; d < DW; ++d) {
                // (lane 0: the row above the band, straight out of the tile's first LDS row -- its window is not skewed)
                const uint32_t b = from_lane_above(o[d], cut(t, j, d));
                uint32_t p = (o[d] & m_sub) | (b & m_up) | (__builtin_amdgcn_lerp(o[d], b, 0u) & m_avg);
                const uint32_t b_lo = __builtin_amdgcn_perm(0u, b, SEL_LO), b_hi = __builtin_amdgcn_perm(0u, b, SEL_HI);
                if (PAETH) p |= (paeth_pk(a_lo[d], b_lo, c_lo[d]) | paeth_pk(a_hi[d], b_hi, c_hi[d]) << 8) & m_pae;
                const uint32_t x = cut(w, j, d);
                const uint32_t r = ((x & 0x7f7f7f7fu) + (p & 0x7f7f7f7fu)) ^ ((x ^ p) & 0x80808080u);
                o[d] = r;
                v[j][d] = r;
                c_lo[d] = b_lo; c_hi[d] = b_hi;
                a_lo[d] = __builtin_amdgcn_perm(0u, r, SEL_LO); a_hi[d] = __builtin_amdgcn_perm(0u, r, SEL_HI);
                bprev[d] = b;
            }
        }
        uint32_t r[12];
        if (BPP == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                              // four units = three dwords
                r[3 * q + 0] = __builtin_amdgcn_perm(v[4 * q + 1][0], v[4 * q + 0][0], 0x04020100u);
                r[3 * q + 1] = __builtin_amdgcn_perm(v[4 * q + 2][0], v[4 * q + 1][0], 0x05040201u);
                r[3 * q + 2] = __builtin_amdgcn_perm(v[4 * q + 3][0], v[4 * q + 2][0], 0x06050402u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                              // two units = three dwords
                r[3 * q + 0] = v[2 * q][0];
                r[3 * q + 1] = __builtin_amdgcn_perm(v[2 * q + 1][0], v[2 * q][DW - 1], 0x05040100u);
                r[3 * q + 2] = __builtin_amdgcn_perm(v[2 * q + 1][DW - 1], v[2 * q + 1][0], 0x05040302u);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) *(u32x4 *)(mine + 48 * g + 16 * i) = u32x4{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
#pragma unroll
        for (int i = 0; i < 3; ++i) { xa[i] = xb[i]; ta[i] = tb[i]; }
    }
}

template <int BPP, int PX = BPP, int PT = 0>
__global__ __launch_bounds__(SPNG_UNF_NW * 64) void unfilter_kernel(const UnfJob *__restrict__ jobs,
                                                                     const spng_result *__restrict__ results,
                                                                     uint32_t sb_rows)
{
    using C = Cfg<BPP, PX, PT>;
    static_assert(PX == BPP || (BPP == 4 && (PX == 1 || PX == 2)), "sub-unit pixels ride in dwords");
    constexpr int NW = SPNG_UNF_NW;
    __shared__ __attribute__((aligned(16))) uint8_t tiles[NW][65 * C::ROWB];
    __shared__ uint32_t done[NW];                        // tiles completed by each wave

    UnfJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (results && skip_status(results[job.image].status)) return;
    if (threadIdx.x < NW) done[threadIdx.x] = 0;
    __syncthreads();

    uint32_t rows = job.rows;
    if (job.rows_len) {
        // a short stream silently yields an incomplete image (PNG.Decoder.swift:88-94)
        const uint64_t len = *job.rows_len;
        const uint64_t avail = len > job.stream_off ? (len - job.stream_off) / job.in_stride : 0;
        rows = avail < rows ? (uint32_t)avail : rows;
    }
    // A chain is cut into pieces that different workgroups take: a row filtered with None or Sub does not
    // look at the row above (PNG.Decoder.swift:160-168), so a piece may start there as if it were a first
    // row.  Workgroup y owns the rows from the first such row at or after y * sb_rows up to the first one at
    // or after (y + 1) * sb_rows.  (The filter bytes are never written, also when rows are defiltered in place.)
    // (rows that arrived with a later push, spng_unfilter_resume_batch: the row above the first one was defiltered by an
    //  earlier call and sits in front of `out`)
    bool has_top = job.has_prev != 0;
    if (gridDim.y > 1) {
        auto cut = [&](uint64_t x) -> uint32_t {
            if (x == 0) return 0;
            for (uint64_t r0 = x; r0 < rows; r0 += 64) {
                const uint64_t r = r0 + lane;
                const uint32_t ft = r < rows ? job.in[r * job.in_stride] : 0u;
                const unsigned long long m = __ballot(r < rows && ft <= 1);
                if (m) return (uint32_t)(r0 + __ffsll((long long)m) - 1);
            }
            return rows;
        };
        const uint32_t first = cut((uint64_t)blockIdx.y * sb_rows);
        const uint32_t last = blockIdx.y + 1 == gridDim.y ? rows : cut((uint64_t)(blockIdx.y + 1) * sb_rows);
        if (first >= last) return;
        if (first) has_top = false;                            // (a piece behind the first starts on a row that looks at nothing above)
        job.in += (uint64_t)first * job.in_stride;
        job.out += (uint64_t)first * job.out_stride;
        rows = last - first;
    }
    const int64_t pitch = job.pitch;
    const uint32_t W = (job.pitch + BPP - 1) / BPP;             // (PX < BPP: the last unit may reach past the row -- loads read zeros there, stores stop)
    const uint32_t ntiles = (W + 63 + C::P - 1) / C::P;
    const uint32_t nbands = (rows + 63) / 64;
    uint8_t *tile = tiles[wave];

    // staging registers for the next tile
    u32x4 R[C::CPR], Rtop;
    auto issue = [&](uint32_t band, uint32_t T) {
#pragma unroll
        for (int m = 0; m < C::CPR; ++m) {
            const int i = lane + 64 * m, r = i / C::CPR, cj = i % C::CPR;
            const uint32_t rw = band * 64 + r;
            u32x4 v = {0, 0, 0, 0};
            if (rw < rows)
                v = load_window(job.in + (uint64_t)rw * job.in_stride + 1,
                                ((int64_t)T * C::P - r) * BPP + 16 * cj, pitch);
            R[m] = v;
        }
        Rtop = u32x4{0, 0, 0, 0};
        if ((band || has_top) && lane < C::CPR)
            Rtop = load_window_l2(job.out + ((int64_t)band * 64 - 1) * (int64_t)job.out_stride,
                                  (int64_t)T * C::TB + 16 * lane, pitch);
    };
    auto commit = [&]() {
#pragma unroll
        for (int m = 0; m < C::CPR; ++m) {
            const int i = lane + 64 * m, r = i / C::CPR, cj = i % C::CPR;
            *(u32x4 *)(tile + (1 + r) * C::ROWB + 16 * cj) = R[m];
        }
        if (lane < C::CPR) *(u32x4 *)(tile + 16 * lane) = Rtop;
    };
    // band j tile T needs the last row of band j-1 on units [T*P, T*P+P): lane 63 of band j-1 is 63
    // units behind, so they come from its tiles T .. T+K.  `done` of the producing wave counts the
    // tiles (over all of its bands) whose stores have been drained.
    auto ready = [&](uint32_t band, uint32_t T) -> bool {
        if (!band) return true;
        const uint32_t pw = (band - 1) % NW, pk = (band - 1) / NW;
        const uint32_t need = pk * ntiles + (T + C::K < ntiles ? T + C::K : ntiles - 1) + 1;
        return __hip_atomic_load(&done[pw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need;
    };
    auto wait_ready = [&](uint32_t band, uint32_t T) {
        SpinGuard guard;
        while (!ready(band, T)) {
            __builtin_amdgcn_s_sleep(8);
            guard.tick();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    uint32_t count = 0;                                  // tiles this wave has completed
    bool staged = false;                                 // R holds the tile about to be processed
    for (uint32_t band = wave; band < nbands; band += NW) {
        const uint32_t row = band * 64 + lane;
        const uint32_t ft = row < rows ? job.in[(uint64_t)row * job.in_stride] : 0u;
        constexpr bool PACKED = BPP == 3 || BPP == 6;          // (a unit in dwords: reconstruct_packed)
        constexpr bool SUBUNIT = PX != BPP;                    // (pixels inside a dword unit: reconstruct_subunit)
        constexpr int NS = SUBUNIT ? 1 : PACKED ? (BPP + 3) / 4 : BPP;
        const bool any_pae = __any(ft == 4);                   // no Paeth row in this band: skip its arithmetic
        // every row of the band filtered with None (what libpng writes for palette and low-depth images): the tiles only pass
        // through (the band below reads this one's last row back from the output, not from registers)
        const bool all_none = !__any(row < rows && ft != 0);
        uint32_t o[NS], bprev[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { o[k] = 0; bprev[k] = 0; }

        for (uint32_t T = 0; T < ntiles; ++T) {
            if (!staged) { wait_ready(band, T); issue(band, T); }
            commit();
            staged = false;
            // prefetch the next tile of this wave while this one is reconstructed.  If its producer
            // (another wave) is not far enough ahead yet, fall back behind it *now*: publish what is
            // pending and wait, so that from here on the prefetch always overlaps the arithmetic.
            uint32_t nb = band, nT = T + 1;
            if (nT == ntiles) { nb = band + NW; nT = 0; }
            if (nb < nbands) {
                bool ok = ready(nb, nT);
                // Blocking here is deadlock-free only if nothing the awaited tile depends on is a tile
                // this wave has not published yet: (nb, nT) reaches back to tile nT + NW*K of this
                // wave's band nb - NW.  Same band: that band is complete.  Next band (we are in the
                // last tile of the current one): only when the chain stops short of this tile.
                const bool may_block = NW > 1 && nb && (nb == band || (uint32_t)(NW * C::K) + 1 < ntiles);
                if (!ok && may_block) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0 && count)
                        __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    wait_ready(nb, nT);
                    ok = true;
                }
                if (ok) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    issue(nb, nT);
                    staged = true;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS tile written before it is read

            const int64_t ux0 = (int64_t)T * C::P - lane;
#ifndef SPNG_UNF_NOCOMPUTE        // tuning builds only: measure the memory pipeline alone
            if (all_none) {
            } else if constexpr (SUBUNIT) {
                if (any_pae) reconstruct_subunit<PX, C::P, true>(tile, C::ROWB, lane, ft, ux0, o[0], bprev[0]);
                else         reconstruct_subunit<PX, C::P, false>(tile, C::ROWB, lane, ft, ux0, o[0], bprev[0]);
            } else if constexpr (PACKED) {
                (void)ux0;
                if (any_pae) reconstruct_packed<BPP, C::P, true>(tile, C::ROWB, lane, ft, o, bprev);
                else         reconstruct_packed<BPP, C::P, false>(tile, C::ROWB, lane, ft, o, bprev);
            } else if (any_pae) reconstruct_generic<BPP, C::P, true>(tile, C::ROWB, lane, ft, ux0, o, bprev);
            else                reconstruct_generic<BPP, C::P, false>(tile, C::ROWB, lane, ft, ux0, o, bprev);
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

            // The stores of the previous tile (and the prefetch loads) have had the whole
            // reconstruction to complete: drain, then publish the previous tile.  Publishing one
            // tile late keeps the store latency off the critical path.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && count)
                __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // write back the 64 row windows
#pragma unroll
            for (int m = 0; m < C::CPR; ++m) {
                const int i = lane + 64 * m, r = i / C::CPR, cj = i % C::CPR;
                const uint32_t rw = band * 64 + r;
                if (rw < rows)
                    store_window(job.out + (uint64_t)rw * job.out_stride,
                                 ((int64_t)T * C::P - r) * BPP + 16 * cj, pitch,
                                 *(const u32x4 *)(tile + (1 + r) * C::ROWB + 16 * cj));
            }
            ++count;
            // a tile that could not be prefetched depends on this one (or a later one): publish now
            if (!staged) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    // last tile of this wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// =====================================================================================================================
// bpp 4 and 8 (RGBA8, VA16, RGBA16 ...): the line-aligned form (round 4)
// =====================================================================================================================
// What the kernel above pays for is not arithmetic and not its loads: `tools/probe_copy.hip` (profiles/r04_probe_copy.log)
// moves the same bytes with the same skewed 64-row tiles at 3.3 TB/s -- and at 5.0 TB/s as soon as the STORES cover whole,
// aligned 128-byte lines, whatever the loads look like (skewed or not, rows pitch + 1 apart or padded).  A row window that
// trails the row above by bpp bytes never ends on a line boundary, so every line of PNG.Image.storage was written in two
// pieces a tile apart, and the L2 -- 4 MiB per XCD, 8 MiB of tiles in flight -- had usually let go of the first piece by then.
//
// So here the skew lives in LDS only.  Global memory is read and written in row-aligned tiles of TB = 128 bytes (16 bytes per
// lane, 8 lanes per row, 8 rows per instruction; storage rows are whole lines, scanline rows start wherever pitch + 1 puts them,
// which loads do not mind).  A row keeps a ring of two tiles in LDS; lane r of a ramp reconstructs unit s - r at step s, reading
// and writing its own row's ring one unit at a time (the upper neighbour still arrives from lane r - 1 by DPP).  A ramp must fit
// a tile -- its last lane finishes tile T - 1 during phase T, which is when that tile is stored -- so a ramp is RR = 128 / bpp
// rows: a wave carries two (bpp 4) or four (bpp 8) independent scanline chains side by side, each with its own top row, its own
// piece of the image (pieces: see above) and RR-row bands pipelined over the waves of the workgroup as before.
template <int BPP> struct Pk {
    static constexpr int DW = BPP / 4;                   // dwords per unit
    static constexpr int RR = 128 / BPP;                 // rows per ramp = units per tile
    static constexpr int NCH = 64 / RR;                  // chains per wave
    static constexpr int TB = 128, RING = 256;
};
#ifndef SPNG_UNF_PK_NW
#define SPNG_UNF_PK_NW 4                                 // waves per workgroup (16.9 KB of LDS each; 2 and 3 measured 16 % slower, 1 the same)
#endif
struct __attribute__((aligned(16))) PkWave {
    uint8_t rows[64][256];                               // [LDS row = lane][ring of two 128-byte tiles: tile T at (T & 1) * 128]
    uint8_t top[4][2][128];                              // [chain][T & 1]: the row above the chain's band
};

// one phase = RR steps of every lane: 32 dwords of its row, eight at a time (the reads of the next eight are issued before
// the arithmetic of these, so that no step waits for LDS).  Per dword ~50 VALU operations: the two 16-bit halves only where
// nine bits are needed (Paeth); Average is v_lerp_u8, the filter select and the byte-wise add run on whole dwords
// (x + p per byte = ((x & 7f..) + (p & 7f..)) ^ ((x ^ p) & 80..)), the halves come from v_perm_b32.
// WRAP: the lane's 128-byte window may cross the end of the ring (even phases: tile T - 1 sits in the upper slot).
template <int BPP, bool FIRST, bool PAETH, bool WRAP>
__device__ __forceinline__ void phase_pk(uint8_t *rowp, const uint8_t *topp, uint32_t a0, int u0, bool head, uint32_t ft,
                                         uint32_t *o, uint32_t *bprev)
{
    constexpr int DW = Pk<BPP>::DW;
    constexpr uint32_t M = 0x00ff00ffu, SEL_LO = 0x0c020c00u, SEL_HI = 0x0c030c01u;
    const uint32_t m_sub = ft == 1 ? ~0u : 0u, m_up = ft == 2 ? ~0u : 0u, m_avg = ft == 3 ? ~0u : 0u, m_pae = ft == 4 ? ~0u : 0u;
    uint32_t a_lo[DW], a_hi[DW], c_lo[DW], c_hi[DW];
#pragma unroll
    for (int d = 0; d < DW; ++d) { a_lo[d] = o[d] & M; a_hi[d] = (o[d] >> 8) & M; c_lo[d] = bprev[d] & M; c_hi[d] = (bprev[d] >> 8) & M; }
    auto at = [&](int dw) -> uint8_t * { return WRAP ? rowp + ((a0 + 4u * (uint32_t)dw) & 255u) : rowp + a0 + 4 * dw; };
    uint32_t xa[8], xb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = *(const uint32_t *)at(i);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        if (blk < 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[i] = *(const uint32_t *)at(8 * (blk + 1) + i);
        }
        const u32x4 t0 = ((const u32x4 *)topp)[2 * blk], t1 = ((const u32x4 *)topp)[2 * blk + 1];
        const uint32_t tq[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        uint32_t r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            constexpr int dmask = DW - 1;
            const int d = i & dmask;                           // which dword of its unit
            uint32_t b = from_lane_above(o[d], 0u);
            b = head ? tq[i] : b;                              // (first lane of a ramp: the row above the band)
            uint32_t p = (o[d] & m_sub) | (b & m_up) | (__builtin_amdgcn_lerp(o[d], b, 0u) & m_avg);
            const uint32_t b_lo = __builtin_amdgcn_perm(0u, b, SEL_LO), b_hi = __builtin_amdgcn_perm(0u, b, SEL_HI);
            if (PAETH) p |= (paeth_pk(a_lo[d], b_lo, c_lo[d]) | paeth_pk(a_hi[d], b_hi, c_hi[d]) << 8) & m_pae;
            const uint32_t x = xa[i];
            uint32_t v = ((x & 0x7f7f7f7fu) + (p & 0x7f7f7f7fu)) ^ ((x ^ p) & 0x80808080u);
            if (FIRST) v = (u0 + (8 * blk + i) / DW >= 0) ? v : 0u;   // units left of the row start produce zeros: unit 0 sees a = c = 0
            o[d] = v;
            r[i] = v;
            c_lo[d] = b_lo; c_hi[d] = b_hi;
            a_lo[d] = __builtin_amdgcn_perm(0u, v, SEL_LO); a_hi[d] = __builtin_amdgcn_perm(0u, v, SEL_HI);
            bprev[d] = b;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *(uint32_t *)at(8 * blk + i) = r[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = xb[i];
    }
}

template <int BPP>
__global__ __launch_bounds__(SPNG_UNF_PK_NW * 64) void unfilter_pk_kernel(const UnfJob *__restrict__ jobs, const spng_result *__restrict__ results,
                                                                           uint32_t sb_rows, uint32_t npieces)
{
    using C = Pk<BPP>;
    constexpr int NW = SPNG_UNF_PK_NW, RR = C::RR, NCH = C::NCH;
    __shared__ PkWave lds[NW];
    __shared__ uint32_t done[NW];                        // phases completed (stores drained) by each wave

    UnfJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = (int)UNI(threadIdx.x >> 6);
    if (results && skip_status(results[job.image].status)) return;
    if (threadIdx.x < NW) done[threadIdx.x] = 0;
    __syncthreads();

    uint32_t rows = job.rows;
    if (job.rows_len) {
        // a short stream silently yields an incomplete image (PNG.Decoder.swift:88-94)
        const uint64_t len = *job.rows_len;
        const uint64_t avail = len > job.stream_off ? (len - job.stream_off) / job.in_stride : 0;
        rows = avail < rows ? (uint32_t)avail : rows;
    }
    // the pieces of this workgroup's chains (see unfilter_kernel: a piece starts on a row filtered with None or Sub)
    auto cut = [&](uint64_t x) -> uint32_t {
        if (x == 0) return 0;
        for (uint64_t r0 = x; r0 < rows; r0 += 64) {
            const uint64_t r = r0 + lane;
            const uint32_t ft = r < rows ? job.in[r * job.in_stride] : 0u;
            const unsigned long long m = __ballot(r < rows && ft <= 1);
            if (m) return (uint32_t)(r0 + __ffsll((long long)m) - 1);
        }
        return rows;
    };
    uint32_t first[NCH], nrows[NCH];
    bool htop[NCH];
    uint32_t maxb = 0;
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
        const uint32_t q = blockIdx.y * NCH + h;
        uint32_t f = 0, l = 0;
        if (q < npieces) {
            f = npieces > 1 ? UNI(cut((uint64_t)q * sb_rows)) : 0u;
            l = q + 1 == npieces ? rows : UNI(cut((uint64_t)(q + 1) * sb_rows));
        }
        first[h] = f; nrows[h] = l > f ? l - f : 0u;
        htop[h] = f == 0 && job.has_prev != 0;          // (rows that arrived with a later push: the row above is in front of `out`)
        const uint32_t nb = (nrows[h] + RR - 1) / RR;
        maxb = nb > maxb ? nb : maxb;
    }
    if (!maxb) return;
    const int64_t pitch = job.pitch;
    const uint32_t nta = (uint32_t)((pitch + C::TB - 1) / C::TB), nph = nta + 1;   // row tiles; phases per band
    PkWave &w = lds[wave];

    // lane as a loader: chunk cj of LDS rows r0, r0 + 8, ... (8 rows per instruction);  lane as a worker: LDS row = lane
    const int r0 = lane >> 3, cj = lane & 7;
    const int hh = lane / RR, rp = lane % RR;            // my chain, my place in its ramp
    uint32_t my_first = 0, my_rows = 0;
#pragma unroll
    for (int h = 0; h < NCH; ++h) if (hh == h) { my_first = first[h]; my_rows = nrows[h]; }
    const bool head = rp == 0;

    u32x4 R[8], Rtop;
    auto issue = [&](uint32_t band, uint32_t T) {
        const int64_t off = (int64_t)T * C::TB + 16 * cj;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            constexpr int per = RR / 8;                          // load instructions per chain
            const int h = m / per;
            const uint32_t rw = band * RR + (uint32_t)((m % per) * 8 + r0);      // row inside the chain's piece
            u32x4 v = {0, 0, 0, 0};
            if (rw < nrows[h]) v = load_window(job.in + (uint64_t)(first[h] + rw) * job.in_stride + 1, off, pitch);
            R[m] = v;
        }
        Rtop = u32x4{0, 0, 0, 0};
        if (lane < NCH * 8) {
            uint32_t f = 0, n = 0; bool ht = false;
#pragma unroll
            for (int h = 0; h < NCH; ++h) if (r0 == h) { f = first[h]; n = nrows[h]; ht = htop[h]; }
            if ((band || ht) && band * RR < n)
                Rtop = load_window_l2(job.out + ((int64_t)f + (int64_t)band * RR - 1) * (int64_t)job.out_stride, off, pitch);
        }
    };
    auto commit = [&](uint32_t T) {
        const uint32_t slot = (T & 1) * 128;
#pragma unroll
        for (int m = 0; m < 8; ++m) *(u32x4 *)(&w.rows[r0 + 8 * m][slot + 16 * cj]) = R[m];
        if (lane < NCH * 8) *(u32x4 *)(&w.top[r0][T & 1][16 * cj]) = Rtop;
    };
    // Band j tile T needs the last row of band j - 1 on bytes [T * TB, T * TB + TB): that tile is stored at the end of the
    // producer's phase T + 1.  `done` of the producing wave counts the phases (over all of its bands) whose stores have drained.
    auto ready = [&](uint32_t band, uint32_t T) -> bool {
        if (!band) return true;
        const uint32_t pw = (band - 1) % NW, pk = (band - 1) / NW;
        const uint32_t need = pk * nph + (T + 1 < nph ? T + 1 : nph - 1) + 1;
        return __hip_atomic_load(&done[pw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= need;
    };
    auto wait_ready = [&](uint32_t band, uint32_t T) {
        SpinGuard guard;
        while (!ready(band, T)) {
            __builtin_amdgcn_s_sleep(8);
            guard.tick();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    uint32_t count = 0;                                  // phases this wave has completed
    bool staged = false;                                 // R holds the tile about to be committed
    for (uint32_t band = wave; band < maxb; band += NW) {
        const uint32_t row = band * RR + (uint32_t)rp;
        const bool mine = row < my_rows;
        const uint32_t ft = mine ? job.in[(uint64_t)(my_first + row) * job.in_stride] : 0u;
        const bool any_pae = __any(ft == 4);            // no Paeth row in this band: skip its arithmetic
        uint32_t o[C::DW], bprev[C::DW];
#pragma unroll
        for (int k = 0; k < C::DW; ++k) { o[k] = 0; bprev[k] = 0; }

        for (uint32_t T = 0; T < nph; ++T) {
            if (T < nta) {
                if (!staged) { wait_ready(band, T); issue(band, T); }
                commit(T);
            }
            staged = false;
            // prefetch the next tile of this wave while this phase is worked on.  If its producer (another wave) is not far
            // enough ahead yet, fall back behind it *now*: publish what is pending and wait, so that from here on the prefetch
            // always overlaps the arithmetic.
            uint32_t nb = band, nT = T + 1;
            if (nT >= nta) { nb = band + NW; nT = 0; }
            if (nb < maxb && (T + 1 < nta || T + 1 == nph)) {
                bool ok = ready(nb, nT);
                // Blocking here is deadlock-free only if nothing the awaited tile depends on is a phase this wave has not
                // published yet: (nb, nT) reaches back to phase nT + 1 + 2 (NW - 1) of this wave's band nb - NW.  Same band:
                // that band is complete.  Next band (we are in the last phase of the current one): only when the chain stops
                // short of this phase.
                const bool may_block = NW > 1 && nb && (nb == band || (uint32_t)(2 * NW) + 1 < nph);
                if (!ok && may_block) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0 && count)
                        __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    wait_ready(nb, nT);
                    ok = true;
                }
                if (ok) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    issue(nb, nT);
                    staged = true;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // LDS tile written before it is read

            const int u0 = (int)(T * RR) - rp;           // my first unit of this phase
            const uint32_t a0 = (uint32_t)(u0 * BPP) & 255u;
#ifndef SPNG_UNF_NOCOMPUTE        // tuning builds only: measure the memory pipeline alone
            // (odd phases: the window [128 - bpp rp, 256 - bpp rp) of the ring never wraps)
            if (T == 0)            phase_pk<BPP, true, true, true>(w.rows[lane], w.top[hh][0], a0, u0, head, ft, o, bprev);
            else if (T & 1) {
                if (any_pae)       phase_pk<BPP, false, true, false>(w.rows[lane], w.top[hh][1], a0, u0, head, ft, o, bprev);
                else               phase_pk<BPP, false, false, false>(w.rows[lane], w.top[hh][1], a0, u0, head, ft, o, bprev);
            } else {
                if (any_pae)       phase_pk<BPP, false, true, true>(w.rows[lane], w.top[hh][0], a0, u0, head, ft, o, bprev);
                else               phase_pk<BPP, false, false, true>(w.rows[lane], w.top[hh][0], a0, u0, head, ft, o, bprev);
            }
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

            // The stores of the previous phase (and the prefetch loads) have had the whole reconstruction to complete:
            // drain, then publish the previous phase.  Publishing one phase late keeps the store latency off the critical path.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && count)
                __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // tile T - 1 is complete in every row: whole lines to the output
            if (T) {
                const uint32_t slot = ((T - 1) & 1) * 128;
                const int64_t off = (int64_t)(T - 1) * C::TB + 16 * cj;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    constexpr int per = RR / 8;
                    const int h = m / per;
                    const uint32_t rw = band * RR + (uint32_t)((m % per) * 8 + r0);
                    if (rw < nrows[h])
                        store_window(job.out + (uint64_t)(first[h] + rw) * job.out_stride, off, pitch,
                                     *(const u32x4 *)(&w.rows[r0 + 8 * m][slot + 16 * cj]));
                }
            }
            ++count;
            // a tile that could not be prefetched depends on this phase (or a later one): publish now
            if (!staged) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    // last phase of this wave
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&done[wave], count, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

hipError_t launch_unfilter(const UnfJob *d_jobs, uint32_t count, uint32_t bpp, spng_result *d_results,
                           uint32_t pieces, uint32_t piece_rows, hipStream_t stream, uint32_t widest)
{
    if (!count) return hipSuccess;
    constexpr int T = SPNG_UNF_NW * 64;
    const dim3 grid(count, pieces ? pieces : 1);
    if (bpp == 4 || bpp == 8) {
        // the line-aligned form: a workgroup's waves carry NCH chains each, i.e. NCH pieces per workgroup
        const uint32_t np = pieces ? pieces : 1, nch = bpp == 4 ? Pk<4>::NCH : Pk<8>::NCH;
        const dim3 g2(count, (np + nch - 1) / nch);
        if (bpp == 4) unfilter_pk_kernel<4><<<g2, SPNG_UNF_PK_NW * 64, 0, stream>>>(d_jobs, d_results, piece_rows, np);
        else unfilter_pk_kernel<8><<<g2, SPNG_UNF_PK_NW * 64, 0, stream>>>(d_jobs, d_results, piece_rows, np);
        return hipGetLastError();
    }
    switch (bpp) {
    // (pixels of 1 and 2 bytes: tiles of 32 dword units for rows of 2 KiB and more, of SPNG_UNF_PSUB = 16 for narrower ones -- wide
    //  tiles cost rows of 512 bytes a third and save rows of 4 KiB a sixth: profiles/r06_tuning.md 14)
    case 1:
        if (widest >= 2048) unfilter_kernel<4, 1, 32><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows);
        else unfilter_kernel<4, 1><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows);
        break;
    case 2:
        if (widest >= 2048) unfilter_kernel<4, 2, 32><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows);
        else unfilter_kernel<4, 2><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows);
        break;
    case 3: unfilter_kernel<3><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows); break;
    case 6: unfilter_kernel<6><<<grid, T, 0, stream>>>(d_jobs, d_results, piece_rows); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The copy ceiling (spng_copy_ceiling): what a kernel of this library can move between two HBM buffers with nothing to do
// in between -- the denominator next to the 8 TB/s spec peak.  pattern 0: 16 bytes per lane, grid-stride.  pattern 1: the
// round-3 unfilter's access pattern (a wave owns 64 rows of `pitch` bytes, 256-byte tiles, row r trailing row r - 1 by 4 bytes on
// BOTH sides): what unaligned stores cost, and a known byte count in that pattern to calibrate FETCH_SIZE / WRITE_SIZE against.
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy_skewed_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint32_t pitch, uint64_t rows)
{
    const int lane = threadIdx.x & 63;
    const uint64_t band = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (band * 64 >= rows) return;
    const int r0 = lane >> 4, cj = lane & 15;
    const uint32_t ntiles = (pitch + 63 * 4 + 255) / 256;
    for (uint32_t T = 0; T < ntiles; ++T) {
        u32x4 v[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint64_t row = band * 64 + (uint64_t)(r0 + 4 * m);
            const int64_t off = (int64_t)T * 256 - (int64_t)(r0 + 4 * m) * 4 + 16 * cj;
            v[m] = u32x4{0, 0, 0, 0};
            if (row < rows && off >= 0 && off + 16 <= (int64_t)pitch) v[m] = ((const U128u *)(in + row * pitch + off))->v;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const uint64_t row = band * 64 + (uint64_t)(r0 + 4 * m);
            const int64_t off = (int64_t)T * 256 - (int64_t)(r0 + 4 * m) * 4 + 16 * cj;
            if (row < rows && off >= 0 && off + 16 <= (int64_t)pitch) ((U128u *)(out + row * pitch + off))->v = v[m];
        }
    }
}
hipError_t launch_copy_probe(const void *d_src, void *d_dst, uint64_t bytes, int pattern, hipStream_t stream)
{
    if (pattern == 0) copy16_kernel<<<256 * 32, 256, 0, stream>>>((const u32x4 *)d_src, (u32x4 *)d_dst, bytes / 16);
    else {
        const uint32_t pitch = 16384;
        const uint64_t rows = bytes / pitch;
        copy_skewed_kernel<<<(uint32_t)((rows + 255) / 256), 256, 0, stream>>>((const uint8_t *)d_src, (uint8_t *)d_dst, pitch, rows);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Scatter of defiltered sub-image rows into PNG.Image.storage (PNG.Image.assign,
// Sources/PNG/PNG.Image.swift:186-285): Adam7 placement (base + i*stride) and MSB-first expansion
// of 1/2/4-bit samples to one unscaled byte each.  One thread per destination pixel.
__global__ __launch_bounds__(256) void scatter_kernel(const ScatterJob *__restrict__ jobs,
                                                      const spng_result *__restrict__ results,
                                                      const uint32_t *__restrict__ job_image)
{
    const ScatterJob job = jobs[blockIdx.y];
    if (results && skip_status(results[job_image[blockIdx.y]].status)) return;
    uint32_t rows = job.sub_h;
    if (job.rows_len) {
        const uint64_t len = *job.rows_len;
        const uint64_t avail = len > job.stream_off ? (len - job.stream_off) / job.row_stride : 0;
        rows = avail < rows ? (uint32_t)avail : rows;
    }
    const uint64_t total = (uint64_t)rows * job.sub_w;
    const uint32_t volume = job.depth * job.channels;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t y = (uint32_t)(idx / job.sub_w), i = (uint32_t)(idx % job.sub_w);
        const uint8_t *scan = job.rows + (uint64_t)y * job.row_stride;
        const uint64_t d = (uint64_t)(job.by + y * job.sy) * job.width + (job.bx + i * job.sx);
        if (volume < 8) {
            const uint32_t per = 8 / job.depth, mask = (1u << job.depth) - 1;
            const uint32_t sh = (~i & (per - 1)) * job.depth;
            job.storage[d] = (uint8_t)((scan[i / per] >> sh) & mask);
        } else {
            // one load and one store per pixel where the pixel is a power-of-two bytes (the scanline side is
            // never aligned: rows are pitch + 1 apart)
            const uint32_t bpp = volume >> 3;
            const uint8_t *from = scan + (uint64_t)i * bpp;
            uint8_t *to = job.storage + d * bpp;
            struct __attribute__((packed)) P16 { uint16_t v; };
            struct __attribute__((packed)) P32 { uint32_t v; };
            struct __attribute__((packed)) P64 { uint64_t v; };
            if (bpp == 8) ((P64 *)to)->v = ((const P64 *)from)->v;
            else if (bpp == 4) ((P32 *)to)->v = ((const P32 *)from)->v;
            else if (bpp == 2) ((P16 *)to)->v = ((const P16 *)from)->v;
            else for (uint32_t k = 0; k < bpp; ++k) to[k] = from[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PNG.Image.overdraw (Sources/PNG/PNG.Image.swift:134-183) as PNG.Context.push(data:overdraw: true) applies it
// (PNG.Context.swift:88-102): after the scanline of pass z at `base` is assigned, with stride (sx, sy),
//     s = (base.x == 0 ? 0 : 1, base.y & 7 == 0 ? 0 : 1),  brush = (sx >> s.x, sy >> s.y),
// every pixel of rows [base.y, base.y + brush.y) and columns [x, x + brush.x), x = base.x, base.x + brush.x, ... takes the value
// storage holds at (x, base.y) -- brushes of one pixel do nothing.  (So pass 3 paints 2 x 4 cells from rows 0 mod 8 and
// 2 x 2 cells from rows 4 mod 8, and pass 5 1 x 2 cells from rows 0 mod 8 only: the reference's `base.y & 0b111` test is taken
// as it stands.)  The reference does this scanline by scanline; what storage holds after any prefix of the scanlines has a
// closed form, which is what lets every pixel be done on its own: the cells of a pass never reach an assigned pixel other
// than their own source (the rows and columns strictly inside a cell belong to later passes), every source (x, base.y) is an
// assigned pixel of a pass <= z, scanlines of one pass paint disjoint rows, and passes come in order -- so an unassigned
// pixel shows the source of the LAST pass that has an assigned scanline whose cell covers it, and an assigned pixel itself.
// One thread per pixel of the rows the call may have changed: reads assigned pixels only, writes unassigned ones only.
__global__ __launch_bounds__(256) void overdraw_kernel(const OverdrawJob *__restrict__ jobs)
{
    const OverdrawJob job = jobs[blockIdx.y];
    const uint32_t BX[7] = {0, 4, 0, 2, 0, 1, 0}, BY[7] = {0, 0, 4, 0, 2, 0, 1};
    const uint32_t EX[7] = {3, 3, 2, 2, 1, 1, 0}, EY[7] = {3, 3, 3, 2, 2, 1, 1};
    const uint64_t total = (uint64_t)(job.y1 - job.y0) * job.width;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t Y = job.y0 + (uint32_t)(idx / job.width), X = (uint32_t)(idx % job.width);
        // the pixel's own pass and scanline (PNG.adam7, PNG.Decoder.swift:6-15)
        const uint32_t own = (Y & 1) ? 6 : (X & 1) ? 5 : (Y & 2) ? 4 : (X & 2) ? 3 : (Y & 4) ? 2 : (X & 4) ? 1 : 0;
        if (((Y - BY[own]) >> EY[own]) < job.done[own]) continue;          // assigned: final
        for (int q = 6; q >= 0; --q) {
            if (!job.done[q] || X < BX[q] || Y < BY[q]) continue;
            const uint32_t yq = (Y - BY[q]) >> EY[q];
            if (yq >= job.done[q]) continue;                               // that scanline has not come yet
            const uint32_t B = BY[q] + (yq << EY[q]);
            const uint32_t bx = (1u << EX[q]) >> (BX[q] ? 1 : 0), by = (1u << EY[q]) >> ((B & 7) ? 1 : 0);
            if (bx * by <= 1 || Y >= B + by) continue;
            const uint32_t x = BX[q] + (X - BX[q]) / bx * bx;
            const uint8_t *from = job.storage + ((uint64_t)B * job.width + x) * job.elem;
            uint8_t *to = job.storage + ((uint64_t)Y * job.width + X) * job.elem;
            for (uint32_t k = 0; k < job.elem; ++k) to[k] = from[k];
            break;
        }
    }
}

hipError_t launch_overdraw(const OverdrawJob *d_jobs, uint32_t count, uint32_t blocks_x, hipStream_t stream)
{
    if (!count) return hipSuccess;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)
        overdraw_kernel<<<dim3(blocks_x ? blocks_x : 1, count - y0 < 65535u ? count - y0 : 65535u), 256, 0, stream>>>(d_jobs + y0);
    return hipGetLastError();
}

hipError_t launch_scatter(const ScatterJob *d_jobs, uint32_t count, const uint32_t *d_job_image,
                          const spng_result *d_results, uint32_t blocks_x, hipStream_t stream)
{
    if (!count) return hipSuccess;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)             // (grid y stops at 65535; a batch of interlaced images has 7 jobs each)
        scatter_kernel<<<dim3(blocks_x, count - y0 < 65535u ? count - y0 : 65535u), 256, 0, stream>>>(d_jobs + y0, d_results, d_job_image + y0);
    return hipGetLastError();
}

}  // namespace spng
