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
        } else {
            // PNG.VA<T>: (v, a)
            for (uint32_t k = 0; k < m; ++k) {
                if (TB == 8) ((uint16_t *)out)[i0 + k] = (uint16_t)(px[k][0] | px[k][3] << 8);
                else ((uint32_t *)out)[i0 + k] = px[k][0] | px[k][3] << 16;
            }
        }
    }
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
