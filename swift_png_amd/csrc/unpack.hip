// unpack.hip -- PNG.Image.storage -> colour-target pixels for gfx950: the step right behind the decode path (the
// reference's own decode benchmark times it: Benchmarks/Decompression/Swift/Main.swift:105-106).
//
// Replaces PNG.RGBA<T>.unpack(_:of:deindexer:) and PNG.VA<T>.unpack(_:of:deindexer:) for T = UInt8 / UInt16, with the
// default deindexers:
//   format dispatch     Sources/PNG/ColorTargets/PNG.RGBA.swift:259-365, PNG.VA.swift:184-290
//   depth rescaling     Sources/PNG/PNG.swift:255-261 (quantum), :286-312, :495-524 (convolve)
//   premultiplication   Sources/PNG/PNG.swift:55-66 (premultiply), PNG.RGBA.swift:121-158, PNG.VA.swift:57-87
// i.e. samples widened by quantum = T.max / (2^depth - 1) (or shifted right when T is narrower), grey replicated to
// r = g = b, alpha T.max when the format has none, 0 for a pixel that equals the tRNS chroma key (compared at the source
// depth), palette entries dereferenced (r, g, b, a as UInt8, then widened), bgr / bgra (CgBI) swizzled to rgb.  A VA
// target keeps (r, a) of that -- the grey value, or the red channel of a colour format.  Premultiplied targets
// (.premultiplied, or .premultiplied(as: UInt8.self) for T = UInt16: what the reference's iOS goldens are compared in).
// HBM-bound: reads S, writes the target.  Four pixels per thread; RGBA8 -> RGBA<UInt8> moves 16 bytes per lane each way.
#include "common.hpp"

namespace spng {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) PV4 { v4u v; };

// PNG.premultiply (PNG.swift:55-66): (color * alpha + (T.max >> 1)) / T.max in full width
__device__ __forceinline__ uint32_t premul(uint32_t c, uint32_t a, uint32_t tmax) { return (c * a + (tmax >> 1)) / tmax; }

template <typename T>
__global__ __launch_bounds__(256) void unpack_kernel(const UnpackJob *__restrict__ jobs)
{
    const UnpackJob job = jobs[blockIdx.y];
    const uint64_t n = (uint64_t)job.width * job.height;
    constexpr uint32_t TB = sizeof(T) * 8;
    constexpr uint32_t TMAX = TB == 8 ? 0xffu : 0xffffu;
    const uint32_t depth = job.depth;
    // quantum(source: depth, destination: T.bitWidth)   (depth <= TB), else shift right
    const uint32_t quantum = depth <= TB ? TMAX / ((1u << depth) - 1) : 0u;
    const uint32_t shift = depth > TB ? depth - TB : 0u;
    const uint32_t pq = TB == 16 ? 257u : 1u;                  // palette atoms are UInt8
    const uint32_t bps = depth == 16 ? 2 : 1;                  // storage bytes per sample
    const bool fast = TB == 8 && depth == 8 && job.channels == 4 && !job.indexed && !job.has_key && job.layout == 0 &&
                      ((uintptr_t)job.storage & 3) == 0;
    T *out = (T *)job.out;
    const uint64_t quads = (n + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = q * 4;
        const uint32_t m = n - i0 < 4 ? (uint32_t)(n - i0) : 4u;
        uint32_t px[4][4];                                     // r, g, b, a of up to four pixels
        if (fast && m == 4) {
            const v4u v = ((const PV4 *)(job.storage + i0 * 4))->v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = v[k];
                px[k][0] = (w >> (job.bgr ? 16 : 0)) & 0xff; px[k][1] = (w >> 8) & 0xff;
                px[k][2] = (w >> (job.bgr ? 0 : 16)) & 0xff; px[k][3] = w >> 24;
            }
        } else {
            for (uint32_t k = 0; k < m; ++k) {
                const uint64_t i = i0 + k;
                uint32_t r, g, b, a = TMAX;
                if (job.indexed) {
                    const uint32_t idx = job.storage[i];
                    uint32_t e[4] = {0, 0, 0, 0};
                    if (idx < job.palette_count) { const uint8_t *p = job.palette + 4 * idx; e[0] = p[0]; e[1] = p[1]; e[2] = p[2]; e[3] = p[3]; }
                    r = e[0] * pq; g = e[1] * pq; b = e[2] * pq; a = e[3] * pq;
                } else {
                    uint32_t c[4] = {0, 0, 0, 0};
                    const uint8_t *p = job.storage + i * job.channels * bps;
                    for (uint32_t z = 0; z < job.channels; ++z)
                        c[z] = bps == 2 ? (uint32_t)p[2 * z] << 8 | p[2 * z + 1] : p[z];     // samples are big-endian
                    bool keyed = false;
                    if (job.has_key) {
                        const uint32_t colors = job.channels >= 3 ? 3 : 1;
                        keyed = true;
                        for (uint32_t z = 0; z < colors; ++z) keyed = keyed && c[z] == job.key[z];
                    }
                    uint32_t v[4];
                    for (uint32_t z = 0; z < 4; ++z) v[z] = depth <= TB ? (c[z] * quantum) & TMAX : c[z] >> shift;
                    if (job.channels <= 2) { r = g = b = v[0]; if (job.channels == 2) a = v[1]; }
                    else { r = v[job.bgr ? 2 : 0]; g = v[1]; b = v[job.bgr ? 0 : 2]; if (job.channels == 4) a = v[3]; }
                    if (keyed) a = 0;
                }
                px[k][0] = r; px[k][1] = g; px[k][2] = b; px[k][3] = a;
            }
        }
        if (job.premultiply) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (job.premultiply == 2 && TB == 16) {
                    // .premultiplied(as: UInt8.self): in eight bits, scaled back by T.max / 255 (alpha too)
                    const uint32_t a8 = px[k][3] >> 8;
                    for (int z = 0; z < 3; ++z) px[k][z] = premul(px[k][z] >> 8, a8, 0xff) * 257u;
                    px[k][3] = a8 * 257u;
                } else {
                    for (int z = 0; z < 3; ++z) px[k][z] = premul(px[k][z], px[k][3], TMAX);
                }
            }
        }
        if (job.layout == 0) {
            if (TB == 8) {
                uint32_t w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) w[k] = px[k][0] | px[k][1] << 8 | px[k][2] << 16 | px[k][3] << 24;
                if (m == 4) { v4u v = {w[0], w[1], w[2], w[3]}; ((PV4 *)((uint32_t *)out + i0))->v = v; }
                else for (uint32_t k = 0; k < m; ++k) ((uint32_t *)out)[i0 + k] = w[k];
            } else {
                for (uint32_t k = 0; k < m; k += 2) {
                    v4u v = {px[k][0] | px[k][1] << 16, px[k][2] | px[k][3] << 16, 0, 0};
                    if (k + 1 < m) { v[2] = px[k + 1][0] | px[k + 1][1] << 16; v[3] = px[k + 1][2] | px[k + 1][3] << 16; ((PV4 *)((uint2 *)out + i0 + k))->v = v; }
                    else { uint2 w; w.x = v[0]; w.y = v[1]; ((uint2 *)out)[i0 + k] = w; }
                }
            }
        } else if (job.layout == 1) {
            // PNG.VA<T>: (v, a)
            for (uint32_t k = 0; k < m; ++k) {
                if (TB == 8) ((uint16_t *)out)[i0 + k] = (uint16_t)(px[k][0] | px[k][3] << 8);
                else ((uint32_t *)out)[i0 + k] = px[k][0] | px[k][3] << 16;
            }
        } else {
            // scalar T (PNG.Image.unpack<T>(as:), PNG.Image.swift:682-760): the grey value / the red channel / palette[i].r
            if (TB == 8) {
                if (m == 4) ((uint32_t *)out)[q] = px[0][0] | px[1][0] << 8 | px[2][0] << 16 | px[3][0] << 24;
                else for (uint32_t k = 0; k < m; ++k) ((uint8_t *)out)[i0 + k] = (uint8_t)px[k][0];
            } else {
                if (m == 4) { uint2 w; w.x = px[0][0] | px[1][0] << 16; w.y = px[2][0] | px[3][0] << 16; ((uint2 *)out)[q] = w; }
                else for (uint32_t k = 0; k < m; ++k) ((uint16_t *)out)[i0 + k] = (uint16_t)px[k][0];
            }
        }
    }
}

// ---- pack: colour-target pixels -> PNG.Image.storage, the step in front of the encode path ------------------------------
// Replaces PNG.RGBA<T>.pack(_:as:indexer:) (PNG.RGBA.swift:409-478), PNG.VA<T>.pack (PNG.VA.swift:334-403) and the scalar
// PNG.Image.pack<T> (PNG.Image.swift:767-834) for T = UInt8 / UInt16 with the default indexers (PNG.Color.swift:158-226,
// PNG.Image.swift:1043-1062): what PNG.Image.init(packing:size:layout:) stores.
//   depth rescaling     Sources/PNG/PNG.swift:1064-1285 (deconvolve): T.bitWidth == depth: as is; < depth: times
//                       quantum(source: T.bitWidth, destination: depth) (:255-261); > depth: shifted right by the difference;
//                       samples stored big-endian (:699-745), sub-byte depths one unscaled byte per sample
//   component choice    v formats take r (v), va (r | v, a | T.max), rgb (r, g, b | v, v, v), bgr / bgra swizzled; a colour
//                       format without alpha drops it, chroma keys play no part (they are metadata of the format)
//   indexed formats     components reduced to UInt8 (T == UInt16: >> 8, PNG.swift:819-878), then the default indexer: the entry
//                       equal to (r, g, b, a) | (v, v, v, a) | (v, v, v, 255), entry 0 when there is none.  The reference builds a
//                       Dictionary(uniqueKeysWithValues:) and traps on a palette that holds a colour twice; here the lowest index
//                       of a repeated colour wins (the mirror refuses such palettes like the reference).
// HBM-bound: reads the pixels, writes S.  Four pixels per thread; RGBA<UInt8> -> rgba8 moves 16 bytes per lane each way.
static constexpr uint32_t PACK_SLOTS = 1024;                    // open addressing, <= 256 entries: load factor <= 1/4
__device__ __forceinline__ uint32_t pack_hash(uint32_t c) { return (c * 0x9E3779B1u) >> 22; }

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const PackJob *__restrict__ jobs)
{
    const PackJob job = jobs[blockIdx.y];
    const uint64_t n = (uint64_t)job.width * job.height;
    constexpr uint32_t TB = sizeof(T) * 8;
    const uint32_t depth = job.depth;
    const uint32_t bps = depth == 16 ? 2 : 1;                  // storage bytes per sample
    const uint32_t ch = job.channels;
    // transform(T) -> A: quantum(source: TB, destination: depth) = (2^depth - 1) / (2^TB - 1) when TB < depth (8 -> 16: 257)
    const uint32_t mul = TB < depth ? 257u : 1u;
    const uint32_t shr = TB > depth ? TB - depth : 0u;
    constexpr uint32_t TMAX = TB == 8 ? 0xffu : 0xffffu;
    __shared__ unsigned long long table[PACK_SLOTS];            // indexed formats: colour << 32 | index, ~0 = empty
    __shared__ __attribute__((aligned(16))) uint32_t stage[4][64 * 6];   // per wave: the storage dwords of a wave's 256 pixels (3- and 6-byte pixels)
    if (job.indexed) {
        for (uint32_t i = threadIdx.x; i < PACK_SLOTS; i += blockDim.x) table[i] = ~0ull;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < job.palette_count && i < 256; i += blockDim.x) {
            const uint8_t *p = job.palette + 4 * i;
            const uint32_t c = p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24;
            const unsigned long long mine = (unsigned long long)c << 32 | i;
            for (uint32_t h = pack_hash(c);; h = (h + 1) & (PACK_SLOTS - 1)) {
                const unsigned long long old = atomicCAS(&table[h], ~0ull, mine);
                if (old == ~0ull) break;
                if ((uint32_t)(old >> 32) == c) { atomicMin(&table[h], mine); break; }     // a repeated colour: lowest index
            }
        }
        __syncthreads();
    }
    const bool fast = TB == 8 && depth == 8 && ch == 4 && !job.indexed && job.layout == 0 &&
                      (((uintptr_t)job.storage | (uintptr_t)job.pixels) & 3) == 0;
    const bool words = ((uintptr_t)job.storage & 3) == 0;
    const T *in = (const T *)job.pixels;
    const uint64_t quads = (n + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = q * 4;
        const uint32_t m = n - i0 < 4 ? (uint32_t)(n - i0) : 4u;
        if (fast && m == 4) {
            v4u v = ((const PV4 *)((const uint32_t *)in + i0))->v;
            if (job.bgr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (v[k] & 0xff00ff00u) | (v[k] >> 16 & 0xff) | (v[k] & 0xff) << 16;
            }
            ((PV4 *)(job.storage + i0 * 4))->v = v;
            continue;
        }
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};               // the storage bytes of up to four pixels (<= 32), in memory order
        uint32_t at = 0;
        const bool rgb_fast = TB == 8 && depth == 8 && ch == 3 && !job.indexed && job.layout == 0 && ((uintptr_t)job.pixels & 3) == 0;
        if (rgb_fast && m == 4) {
            // [RGBA<UInt8>] -> rgb8 / bgr8: four pixels = one 16-byte load, twelve bytes out (alpha dropped)
            v4u v = ((const PV4 *)((const uint32_t *)in + i0))->v;
            if (job.bgr) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (v[k] & 0xff00ff00u) | (v[k] >> 16 & 0xff) | (v[k] & 0xff) << 16;
            }
            w[0] = (v[0] & 0x00ffffffu) | v[1] << 24;
            w[1] = (v[1] >> 8 & 0xffffu) | v[2] << 16;
            w[2] = (v[2] >> 16 & 0xffu) | v[3] << 8;
        } else
        for (uint32_t k = 0; k < m; ++k) {
            const uint64_t i = i0 + k;
            uint32_t r, g, b, a = TMAX;
            if (job.layout == 0) { const T *p = in + i * 4; r = p[0]; g = p[1]; b = p[2]; a = p[3]; }
            else if (job.layout == 1) { const T *p = in + i * 2; r = g = b = p[0]; a = p[1]; }
            else { r = g = b = in[i]; }
            if (job.indexed) {
                const uint32_t s8 = TB - 8;
                const uint32_t c = (r >> s8) | (g >> s8) << 8 | (b >> s8) << 16 | (a >> s8) << 24;
                uint32_t idx = 0;
                for (uint32_t h = pack_hash(c);; h = (h + 1) & (PACK_SLOTS - 1)) {
                    const unsigned long long e = table[h];
                    if (e == ~0ull) break;
                    if ((uint32_t)(e >> 32) == c) { idx = (uint32_t)e; break; }
                }
                w[at >> 2] |= idx << 8 * (at & 3); ++at;
                continue;
            }
            uint32_t c[4];
            if (ch <= 2) { c[0] = r; c[1] = a; }
            else { c[0] = job.bgr ? b : r; c[1] = g; c[2] = job.bgr ? r : b; c[3] = a; }
            for (uint32_t z = 0; z < ch; ++z) {
                const uint32_t v = (c[z] * mul) >> shr;
                if (bps == 2) { w[at >> 2] |= (v >> 8) << 8 * (at & 3); ++at; w[at >> 2] |= (v & 0xff) << 8 * (at & 3); ++at; }
                else { w[at >> 2] |= v << 8 * (at & 3); ++at; }
            }
        }
        const uint32_t per = job.indexed ? 1 : ch * bps;        // storage bytes per pixel
        uint8_t *dst = job.storage + i0 * per;
        // Pixels of 3 or 6 bytes (rgb8 -- the most common PNG format -- and rgb16): a lane's four pixels are 12 / 24 bytes, 12 / 24
        // apart from its neighbour's -- three dword stores per lane touch every line three times.  When the whole wave has full
        // quads its 768 / 1536 bytes are contiguous: through LDS they leave as 16 bytes per lane (1.9 -> ... TB/s on 4096^2 rgb8).
        const uint64_t q0 = q - (threadIdx.x & 63);            // the wave's first quad (a multiple of 64: the same for all its lanes)
        if ((per == 3 || per == 6) && (q0 + 64) * 4 <= n && ((uintptr_t)(job.storage + q0 * 4 * per) & 15) == 0) {
            const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            uint32_t *st = stage[wv];
            for (uint32_t z = 0; z < per; ++z) st[lane * per + z] = w[z];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            PV4 *out16 = (PV4 *)(job.storage + q0 * 4 * per);
            for (uint32_t u = lane; u < 16 * per; u += 64) out16[u].v = *(const v4u *)(st + 4 * u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            continue;
        }
        if (m == 4 && words) {
            const uint32_t nw = per;                            // 4 pixels x per bytes = per dwords
            if (nw == 4) { v4u v = {w[0], w[1], w[2], w[3]}; ((PV4 *)dst)->v = v; }
            else if (nw == 8) { v4u v = {w[0], w[1], w[2], w[3]}, u = {w[4], w[5], w[6], w[7]}; ((PV4 *)dst)->v = v; ((PV4 *)dst + 1)->v = u; }
            else for (uint32_t z = 0; z < nw; ++z) ((uint32_t *)dst)[z] = w[z];
        } else {
            for (uint32_t z = 0; z < m * per; ++z) dst[z] = (uint8_t)(w[z >> 2] >> 8 * (z & 3));
        }
    }
}

hipError_t launch_pack(const PackJob *d_jobs, uint32_t count, uint32_t blocks_x, int source, hipStream_t stream)
{
    if (!count) return hipSuccess;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u) {           // (grid y stops at 65535)
        const dim3 grid(blocks_x ? blocks_x : 1, count - y0 < 65535u ? count - y0 : 65535u);
        if (source == 8) pack_kernel<uint8_t><<<grid, 256, 0, stream>>>(d_jobs + y0);
        else pack_kernel<uint16_t><<<grid, 256, 0, stream>>>(d_jobs + y0);
    }
    return hipGetLastError();
}

hipError_t launch_unpack(const UnpackJob *d_jobs, uint32_t count, uint32_t blocks_x, int target, hipStream_t stream)
{
    if (!count) return hipSuccess;
    for (uint32_t y0 = 0; y0 < count; y0 += 65535u) {           // (grid y stops at 65535)
        const dim3 grid(blocks_x ? blocks_x : 1, count - y0 < 65535u ? count - y0 : 65535u);
        if (target == 8) unpack_kernel<uint8_t><<<grid, 256, 0, stream>>>(d_jobs + y0);
        else unpack_kernel<uint16_t><<<grid, 256, 0, stream>>>(d_jobs + y0);
    }
    return hipGetLastError();
}

}  // namespace spng
