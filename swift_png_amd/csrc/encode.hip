// encode.hip -- encode-side scanline kernels for gfx950: gather + filter selection, and Adler-32.
//
// Replaces PNG.Encoder.filter (Sources/PNG/Encoding/PNG.Encoder.swift:132-204) with its score
// (:229-234), and PNG.Image.collect (Sources/PNG/PNG.Image.swift:431-544) through the job
// geometry.  Unlike reconstruction, filtering has no serial dependency: every predictor reads raw
// neighbours only, so the parallelisation is one wave per scanline with the lanes striding over
// the bytes of the row; the five sum|int8| scores are reduced across the wave and the *first*
// strict minimum in the order None, Sub, Up, Average, Paeth wins, exactly as the reference loop
// (:186-193) does.  Only the winning candidate is written (filter byte + pitch bytes).
#include "common.hpp"

namespace spng {

__device__ __forceinline__ uint32_t paeth_u(uint32_t a, uint32_t b, uint32_t c)
{
    // PNG.paeth, PNG.swift:124-147
    int pa = abs((int)b - (int)c), pb = abs((int)a - (int)c), pc = abs((int)a + (int)b - 2 * (int)c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// byte j of the gathered scanline y of a (sub-)image: PNG.Image.collect (PNG.Image.swift:431-544)
__device__ __forceinline__ uint32_t raw_byte(const FilterJob &job, uint32_t y, uint32_t j, bool direct,
                                             uint32_t volume, uint32_t bpp)
{
    if (direct) return job.storage[(uint64_t)y * job.pitch + j];
    const uint64_t line = (uint64_t)(job.by + y * job.sy) * job.width;
    if (volume >= 8) {
        const uint32_t i = j / bpp, k = j % bpp;
        return job.storage[(line + job.bx + (uint64_t)i * job.sx) * bpp + k];
    }
    const uint32_t per = 8 / job.depth, mask = (1u << job.depth) - 1;
    uint32_t v = 0;
    for (uint32_t q = 0; q < per; ++q) {
        const uint32_t i = j * per + q;
        if (i < job.sub_w) {
            const uint32_t sh = (~i & (per - 1)) * job.depth;
            v |= (job.storage[line + job.bx + (uint64_t)i * job.sx] & mask) << sh;
        }
    }
    return v;
}

__device__ __forceinline__ uint32_t abs8(uint32_t v)   // |Int8(bitPattern:)|, PNG.Encoder.swift:233
{
    const int s = (int)(int8_t)(uint8_t)v;
    return (uint32_t)(s < 0 ? -s : s);
}

// ---- fast path: 8/16-bit non-interlaced rows whose pitch is a multiple of 16 ---------------------
// 16 bytes per lane per step, two 16-byte loads (this row, row above); the left / upper-left
// neighbours are the same bytes shifted by bpp, taken from the previous lane with DPP instead of a
// second pair of loads.  All five residuals are formed with byte-SWAR arithmetic in registers
// (v_lerp_u8 for Average, packed-i16 Paeth), scored with v_sad_u8; only the winner is recomputed
// and stored.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed)) U128u { u32x4 v; };

__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t sub8(uint32_t x, uint32_t p)              // per-byte x - p (mod 256)
{
    constexpr uint32_t Hb = 0x80808080u;
    return ((x | Hb) - (p & ~Hb)) ^ ((x ^ ~p) & Hb);
}
// acc + sum |Int8(byte)| over the four bytes of r: a signed byte s and the unsigned byte u = s + 128 (its pattern with the top bit
// flipped) satisfy |s| = |u - 128|, so the sum is ONE sum of absolute differences against 0x80 (round 6: seven instructions before)
__device__ __forceinline__ uint32_t sumabs8(uint32_t r, uint32_t acc)
{
    constexpr uint32_t Hb = 0x80808080u;
    return __builtin_amdgcn_sad_u8(r ^ Hb, Hb, acc);
}
// the same for the residual x - p, whose top bits the subtraction can leave flipped for free: sub8 ends in `^ ((x ^ ~p) & Hb)`;
// ending in `^ ((x ^ p) & Hb)` instead gives the residual with every byte's top bit flipped
__device__ __forceinline__ uint32_t sumabs8_sub(uint32_t x, uint32_t p, uint32_t acc)
{
    constexpr uint32_t Hb = 0x80808080u;
    const uint32_t biased = ((x | Hb) - (p & ~Hb)) ^ ((x ^ p) & Hb);
    return __builtin_amdgcn_sad_u8(biased, Hb, acc);
}
__device__ __forceinline__ uint32_t paeth_pk16(uint32_t a, uint32_t b, uint32_t c)
{
    const s16x2 va = __builtin_bit_cast(s16x2, a), vb = __builtin_bit_cast(s16x2, b), vc = __builtin_bit_cast(s16x2, c);
    const s16x2 d0 = vb - vc, d1 = va - vc, ds = d0 + d1;
    const s16x2 pa = __builtin_elementwise_max(d0, -d0), pb = __builtin_elementwise_max(d1, -d1),
                pc = __builtin_elementwise_max(ds, -ds);
    const s16x2 fifteen = {15, 15};
    const uint32_t nota = opaque(__builtin_bit_cast(uint32_t, ((pb - pa) | (pc - pa)) >> fifteen));
    const uint32_t usec = opaque(__builtin_bit_cast(uint32_t, (pc - pb) >> fifteen));
    const uint32_t bc = (c & usec) | (b & ~usec);
    return (bc & nota) | (a & ~nota);
}
__device__ __forceinline__ uint32_t paeth8(uint32_t a, uint32_t b, uint32_t c)   // PNG.paeth on 4 bytes
{
    constexpr uint32_t M = 0x00ff00ffu;
    const uint32_t lo = paeth_pk16(a & M, b & M, c & M);
    const uint32_t hi = paeth_pk16((a >> 8) & M, (b >> 8) & M, (c >> 8) & M);
    return lo | hi << 8;
}
__device__ __forceinline__ uint32_t lane_above(uint32_t mine, uint32_t first)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)mine, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// the 16 bytes that start BPP bytes before `x`, given the 8 bytes (pz, pw) that precede x
template <int BPP>
__device__ __forceinline__ u32x4 shifted(u32x4 x, uint32_t pz, uint32_t pw)
{
    u32x4 r;
    if constexpr (BPP == 4) { r.x = pw; r.y = x.x; r.z = x.y; r.w = x.z; }
    else if constexpr (BPP == 8) { r.x = pz; r.y = pw; r.z = x.x; r.w = x.y; }
    else if constexpr (BPP < 4) {
        constexpr int sh = 4 - BPP;
        r.x = __builtin_amdgcn_alignbyte(x.x, pw, sh); r.y = __builtin_amdgcn_alignbyte(x.y, x.x, sh);
        r.z = __builtin_amdgcn_alignbyte(x.z, x.y, sh); r.w = __builtin_amdgcn_alignbyte(x.w, x.z, sh);
    } else {                                                   // BPP == 6
        r.x = __builtin_amdgcn_alignbyte(pw, pz, 2); r.y = __builtin_amdgcn_alignbyte(x.x, pw, 2);
        r.z = __builtin_amdgcn_alignbyte(x.y, x.x, 2); r.w = __builtin_amdgcn_alignbyte(x.z, x.y, 2);
    }
    return r;
}

template <int BPP>
__device__ void filter_row_fast(const uint8_t *cur, const uint8_t *up, uint32_t pitch, uint8_t *out, int lane)
{
    uint32_t sc[5] = {0, 0, 0, 0, 0};
    const uint32_t steps = pitch / 1024 + (pitch % 1024 ? 1 : 0);
    // carry of the last 8 bytes of the previous 1 KiB step (lane 63 -> lane 0)
    uint32_t cz = 0, cw = 0, uz = 0, uw = 0;
    for (uint32_t it = 0; it < steps; ++it) {
        const uint32_t j = (it * 64 + lane) * 16;
        const bool live = j < pitch;
        u32x4 x = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (live) { x = ((const U128u *)(cur + j))->v; if (up) b = ((const U128u *)(up + j))->v; }
        const uint32_t pz = lane_above(x.z, cz), pw = lane_above(x.w, cw);
        const uint32_t qz = lane_above(b.z, uz), qw = lane_above(b.w, uw);
        cz = (uint32_t)__builtin_amdgcn_readlane((int)x.z, 63); cw = (uint32_t)__builtin_amdgcn_readlane((int)x.w, 63);
        uz = (uint32_t)__builtin_amdgcn_readlane((int)b.z, 63); uw = (uint32_t)__builtin_amdgcn_readlane((int)b.w, 63);
        const u32x4 a = shifted<BPP>(x, pz, pw), c = shifted<BPP>(b, qz, qw);
        if (live) {
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, as[4] = {a.x, a.y, a.z, a.w};
            const uint32_t bs[4] = {b.x, b.y, b.z, b.w}, cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sc[0] = sumabs8(xs[k], sc[0]);
                sc[1] = sumabs8_sub(xs[k], as[k], sc[1]);
                sc[2] = sumabs8_sub(xs[k], bs[k], sc[2]);
                sc[3] = sumabs8_sub(xs[k], __builtin_amdgcn_lerp(as[k], bs[k], 0u), sc[3]);
                sc[4] = sumabs8_sub(xs[k], paeth8(as[k], bs[k], cs[k]), sc[4]);
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sc[f] += __shfl_xor(sc[f], m, 64);
    uint32_t best = 0, minimum = sc[0];
#pragma unroll
    for (int f = 1; f < 5; ++f) if (sc[f] < minimum) { minimum = sc[f]; best = f; }   // first strict minimum
    if (lane == 0) out[0] = (uint8_t)best;
    cz = cw = uz = uw = 0;
    const uint32_t sft = UNI((uint32_t)((uintptr_t)(out + 1) & 15));     // the row's data starts this far behind a 16-byte boundary
    uint32_t carry[4] = {0, 0, 0, 0};
    const uint32_t steps2 = (pitch + 16) / 1024 + ((pitch + 16) % 1024 ? 1 : 0);   // (one unit more: the row's tail sits in the lane behind its last)
    for (uint32_t it = 0; it < steps2; ++it) {
        const uint32_t j = (it * 64 + lane) * 16;
        const bool live = j < pitch;
        u32x4 x = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (live) { x = ((const U128u *)(cur + j))->v; if (up) b = ((const U128u *)(up + j))->v; }
        const uint32_t pz = lane_above(x.z, cz), pw = lane_above(x.w, cw);
        const uint32_t qz = lane_above(b.z, uz), qw = lane_above(b.w, uw);
        cz = (uint32_t)__builtin_amdgcn_readlane((int)x.z, 63); cw = (uint32_t)__builtin_amdgcn_readlane((int)x.w, 63);
        uz = (uint32_t)__builtin_amdgcn_readlane((int)b.z, 63); uw = (uint32_t)__builtin_amdgcn_readlane((int)b.w, 63);
        const u32x4 a = shifted<BPP>(x, pz, pw), c = shifted<BPP>(b, qz, qw);
        uint32_t rr[4] = {0, 0, 0, 0};
        if (live) {
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, as[4] = {a.x, a.y, a.z, a.w};
            const uint32_t bs[4] = {b.x, b.y, b.z, b.w}, cs[4] = {c.x, c.y, c.z, c.w};
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t pred = 0;
                if (best == 1) pred = as[k];
                else if (best == 2) pred = bs[k];
                else if (best == 3) pred = __builtin_amdgcn_lerp(as[k], bs[k], 0u);
                else if (best == 4) pred = paeth8(as[k], bs[k], cs[k]);
                r[k] = sub8(xs[k], pred);
            }
            rr[0] = r[0]; rr[1] = r[1]; rr[2] = r[2]; rr[3] = r[3];
        }
        // ---- the store.  The scanline stream puts a row pitch + 1 bytes behind the one before, so a lane's 16 bytes never start on
        // a 16-byte boundary: stored as they are, every 128-byte line is written in pieces by two instructions (the pattern that
        // cost the unfilter kernel a third of its bandwidth).  Instead every lane stores the ALIGNED 16 bytes that end
        // inside its own: the last `sft` bytes of the lane above (DPP; lane 63 of the step before for lane 0) and its own first
        // 16 - sft.  The row's first and last partial units -- shared with the neighbouring rows, other waves' -- go byte by byte.
        {
            const uint32_t p0 = lane_above(rr[0], carry[0]), p1 = lane_above(rr[1], carry[1]), p2 = lane_above(rr[2], carry[2]), p3 = lane_above(rr[3], carry[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) carry[k] = (uint32_t)__builtin_amdgcn_readlane((int)rr[k], 63);
            if (sft == 0) {
                if (live) ((U128u *)(out + 1 + j))->v = u32x4{rr[0], rr[1], rr[2], rr[3]};
            } else {
                // X = the 32 bytes (lane above : mine); the unit = X shifted right by 16 - sft bytes
                const uint32_t X[9] = {p0, p1, p2, p3, rr[0], rr[1], rr[2], rr[3], 0u};
                const uint32_t a = (16 - sft) >> 2, b = (16 - sft) & 3;          // dwords, bytes (wave-uniform)
                u32x4 u;
                auto pick = [&](int aa) {
                    u.x = __builtin_amdgcn_alignbyte(X[aa + 1], X[aa], b); u.y = __builtin_amdgcn_alignbyte(X[aa + 2], X[aa + 1], b);
                    u.z = __builtin_amdgcn_alignbyte(X[aa + 3], X[aa + 2], b); u.w = __builtin_amdgcn_alignbyte(X[aa + 4], X[aa + 3], b);
                };
                if (a == 0) pick(0); else if (a == 1) pick(1); else if (a == 2) pick(2); else pick(3);
                // the unit ends at row byte j + 16 - sft, i.e. it holds row bytes [j - sft, j + 16 - sft)
                if (j == 0) {
                    // (the row's first unit: its first sft - 1 bytes are the row above's, then the filter byte, stored by lane 0 above)
                    for (uint32_t k = 0; k < 16 - sft; ++k) out[1 + k] = (uint8_t)(rr[k >> 2] >> (8 * (k & 3)));
                } else if (j <= pitch) {
                    if (j < pitch) *(u32x4 *)(out + 1 + j - sft) = u;
                    else for (uint32_t k = 0; k < sft; ++k) out[1 + pitch - sft + k] = (uint8_t)(X[(16 - sft + k) >> 2] >> (8 * ((16 - sft + k) & 3)));   // the row's last sft bytes: the lane behind the last one
                }
            }
        }
    }
}

static constexpr uint32_t PACKED_ROW = 2048;                   // bytes of a sub-byte scanline that go through LDS (16384 one-bit samples; 16 KB per workgroup: the other formats keep their six workgroups per CU)
__global__ __launch_bounds__(256) void filter_kernel(const FilterJob *__restrict__ jobs)
{
    __shared__ uint8_t packed[4][2][PACKED_ROW];              // [wave][this row, the row above]
    const FilterJob job = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t volume = job.depth * job.channels, bpp = (volume + 7) >> 3;
    const bool direct = volume >= 8 && job.sx == 1 && job.sy == 1 && job.bx == 0 && job.by == 0 &&
                        job.sub_w == job.width;
    const bool fast = direct && (job.pitch & 15) == 0 && bpp != 5 && bpp != 7;
    for (uint32_t y = blockIdx.x * 4 + wave; y < job.sub_h; y += gridDim.x * 4) {
        if (fast) {
            const uint8_t *cur = job.storage + (uint64_t)y * job.pitch;
            const uint8_t *up = y ? cur - job.pitch : nullptr;
            uint8_t *out = job.rows + (uint64_t)y * job.row_stride;
            switch (bpp) {
            case 1: filter_row_fast<1>(cur, up, job.pitch, out, lane); break;
            case 2: filter_row_fast<2>(cur, up, job.pitch, out, lane); break;
            case 3: filter_row_fast<3>(cur, up, job.pitch, out, lane); break;
            case 4: filter_row_fast<4>(cur, up, job.pitch, out, lane); break;
            case 6: filter_row_fast<6>(cur, up, job.pitch, out, lane); break;
            default: filter_row_fast<8>(cur, up, job.pitch, out, lane); break;
            }
            continue;
        }
        if (volume < 8 && job.pitch <= PACKED_ROW) {
            // Samples of 1, 2 or 4 bits (round 6): a scanline byte is 8 / depth storage bytes, and the loops below ask for each of
            // this row's and the row above's up to twice per candidate and pass -- 64 byte loads per byte of a 1-bit row (the
            // `scanline_formats` leg: 3 % of peak).  The two rows are packed ONCE into LDS and the filters read them there.
            uint8_t *cur = packed[wave][0], *up = packed[wave][1];
            // (samples side by side in storage -- every image that is not an Adam7 sub-image: the 8 / depth storage bytes of a scanline
            //  byte in one load, their low bits gathered by a multiplication: bytes b0 .. b3 of x = w & 0x01010101 meet in the top byte
            //  of x * 0x08040201 as b0 << 3 | b1 << 2 | b2 << 1 | b3)
            const uint32_t per = 8 / job.depth;
            const bool side_by_side = job.sx == 1;
            auto pack_row = [&](uint32_t yy, uint32_t j) -> uint32_t {
                if (!(side_by_side && (uint64_t)j * per + per <= job.sub_w)) return raw_byte(job, yy, j, direct, volume, bpp);
                const uint8_t *p = job.storage + (uint64_t)(job.by + yy * job.sy) * job.width + job.bx + (uint64_t)j * per;
                if (job.depth == 1) {
                    uint32_t lo, hi;
                    __builtin_memcpy(&lo, p, 4); __builtin_memcpy(&hi, p + 4, 4);
                    return (((lo & 0x01010101u) * 0x08040201u) >> 24) << 4 | ((hi & 0x01010101u) * 0x08040201u) >> 24;
                }
                if (job.depth == 2) {
                    uint32_t w;
                    __builtin_memcpy(&w, p, 4);
                    return ((w & 0x03030303u) * 0x40100401u) >> 24;
                }
                return ((uint32_t)p[0] & 15u) << 4 | ((uint32_t)p[1] & 15u);
            };
            for (uint32_t j = lane; j < job.pitch; j += 64) {
                cur[j] = (uint8_t)pack_row(y, j);
                up[j] = y ? (uint8_t)pack_row(y - 1, j) : (uint8_t)0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            uint32_t sc[5] = {0, 0, 0, 0, 0};
            for (uint32_t j = lane; j < job.pitch; j += 64) {          // (bpp = 1)
                const uint32_t x = cur[j], a = j ? cur[j - 1] : 0u, b = up[j], c = j ? up[j - 1] : 0u;
                sc[0] += abs8(x);
                sc[1] += abs8(x - a);
                sc[2] += abs8(x - b);
                sc[3] += abs8(x - ((a + b) >> 1));
                sc[4] += abs8(x - paeth_u(a, b, c));
            }
#pragma unroll
            for (int f = 0; f < 5; ++f)
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) sc[f] += __shfl_xor(sc[f], m, 64);
            uint32_t best = 0, minimum = sc[0];
#pragma unroll
            for (int f = 1; f < 5; ++f) if (sc[f] < minimum) { minimum = sc[f]; best = f; }
            uint8_t *out = job.rows + (uint64_t)y * job.row_stride;
            if (lane == 0) out[0] = (uint8_t)best;
            for (uint32_t j = lane; j < job.pitch; j += 64) {
                const uint32_t x = cur[j], a = j ? cur[j - 1] : 0u, b = up[j], c = j ? up[j - 1] : 0u;
                uint32_t pred = 0;
                if (best == 1) pred = a;
                else if (best == 2) pred = b;
                else if (best == 3) pred = (a + b) >> 1;
                else if (best == 4) pred = paeth_u(a, b, c);
                out[1 + j] = (uint8_t)(x - pred);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     // (the next row of this wave packs over these)
            continue;
        }
        uint32_t sc[5] = {0, 0, 0, 0, 0};
        for (uint32_t j = lane; j < job.pitch; j += 64) {
            const uint32_t x = raw_byte(job, y, j, direct, volume, bpp);
            const uint32_t a = j >= bpp ? raw_byte(job, y, j - bpp, direct, volume, bpp) : 0;
            const uint32_t b = y ? raw_byte(job, y - 1, j, direct, volume, bpp) : 0;
            const uint32_t c = (y && j >= bpp) ? raw_byte(job, y - 1, j - bpp, direct, volume, bpp) : 0;
            sc[0] += abs8(x);
            sc[1] += abs8(x - a);
            sc[2] += abs8(x - b);
            sc[3] += abs8(x - ((a + b) >> 1));
            sc[4] += abs8(x - paeth_u(a, b, c));
        }
#pragma unroll
        for (int f = 0; f < 5; ++f)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) sc[f] += __shfl_xor(sc[f], m, 64);
        uint32_t best = 0, minimum = sc[0];
#pragma unroll
        for (int f = 1; f < 5; ++f) if (sc[f] < minimum) { minimum = sc[f]; best = f; }
        uint8_t *out = job.rows + (uint64_t)y * job.row_stride;
        if (lane == 0) out[0] = (uint8_t)best;
        for (uint32_t j = lane; j < job.pitch; j += 64) {
            const uint32_t x = raw_byte(job, y, j, direct, volume, bpp);
            const uint32_t a = j >= bpp ? raw_byte(job, y, j - bpp, direct, volume, bpp) : 0;
            const uint32_t b = y ? raw_byte(job, y - 1, j, direct, volume, bpp) : 0;
            const uint32_t c = (y && j >= bpp) ? raw_byte(job, y - 1, j - bpp, direct, volume, bpp) : 0;
            uint32_t pred = 0;
            if (best == 1) pred = a;
            else if (best == 2) pred = b;
            else if (best == 3) pred = (a + b) >> 1;
            else if (best == 4) pred = paeth_u(a, b, c);
            out[1 + j] = (uint8_t)(x - pred);
        }
    }
}

hipError_t launch_filter(const FilterJob *d_jobs, uint32_t count, uint32_t max_rows, hipStream_t stream)
{
    if (!count) return hipSuccess;
    uint32_t bx = (max_rows + 3) / 4;
    if (bx > 4096) bx = 4096;
    if (!bx) bx = 1;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u)             // (grid y stops at 65535)
        filter_kernel<<<dim3(bx, count - y0 < 65535u ? count - y0 : 65535u), 256, 0, stream>>>(d_jobs + y0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Adler-32 partial sums (LZ77.MRC32, Sources/LZ77/Wrappers/LZ77.MRC32.swift:26-50): block i
// reduces bytes [i*CH, (i+1)*CH) to (sum b, sum (len-k) b_k); the host folds the partials in order.
__global__ __launch_bounds__(256) void adler_partial_kernel(const uint8_t *__restrict__ p, uint64_t n,
                                                            uint32_t chunk, uint64_t *__restrict__ out)
{
    const uint64_t from = (uint64_t)blockIdx.x * chunk;
    const uint64_t len = n - from < chunk ? n - from : chunk;
    uint64_t s1 = 0, s2 = 0;
    for (uint64_t k = threadIdx.x; k < len; k += 256) { const uint64_t b = p[from + k]; s1 += b; s2 += (len - k) * b; }
    __shared__ uint64_t r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = r1[0]; out[2 * blockIdx.x + 1] = r2[0]; }
}

hipError_t launch_adler_partial(const uint8_t *d, uint64_t n, uint32_t chunk, uint64_t *d_out, uint32_t blocks,
                                hipStream_t stream)
{
    if (!blocks) return hipSuccess;
    adler_partial_kernel<<<blocks, 256, 0, stream>>>(d, n, chunk, d_out);
    return hipGetLastError();
}

}  // namespace spng
