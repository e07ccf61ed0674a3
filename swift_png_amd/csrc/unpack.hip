// unpack.hip -- PNG.Image.storage -> RGBA pixels for gfx950: the step right behind the decode path (the
// reference's own decode benchmark times it: Benchmarks/Decompression/Swift/Main.swift:105-106).
//
// Replaces PNG.RGBA<T>.unpack(_:of:deindexer:) for T = UInt8 / UInt16:
//   format dispatch     Sources/PNG/ColorTargets/PNG.RGBA.swift:259-365
//   depth rescaling     Sources/PNG/PNG.swift:255-261 (quantum), :286-312, :495-524 (convolve)
// i.e. samples widened by quantum = T.max / (2^depth - 1) (or shifted right when T is narrower), grey
// replicated to r = g = b, alpha T.max when the format has none, 0 for a pixel that equals the tRNS chroma
// key (compared at the source depth), palette entries dereferenced (r, g, b, a as UInt8, then widened),
// bgr / bgra (CgBI) swizzled to rgb.  One thread per pixel; HBM-bound: reads S, writes 4 * sizeof(T) per pixel.
#include "common.hpp"

namespace spng {

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
    T *out = (T *)job.out;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t r, g, b, a = TMAX;
        if (job.indexed) {
            const uint32_t idx = job.storage[i];
            uint32_t q[4] = {0, 0, 0, 0};
            if (idx < job.palette_count) { const uint8_t *p = job.palette + 4 * idx; q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3]; }
            r = q[0] * pq; g = q[1] * pq; b = q[2] * pq; a = q[3] * pq;
        } else {
            uint32_t c[4] = {0, 0, 0, 0};
            const uint8_t *p = job.storage + i * job.channels * bps;
            for (uint32_t k = 0; k < job.channels; ++k)
                c[k] = bps == 2 ? (uint32_t)p[2 * k] << 8 | p[2 * k + 1] : p[k];     // samples are big-endian
            bool keyed = false;
            if (job.has_key) {
                const uint32_t colors = job.channels >= 3 ? 3 : 1;
                keyed = true;
                for (uint32_t k = 0; k < colors; ++k) keyed = keyed && c[k] == job.key[k];
            }
            uint32_t v[4];
            for (uint32_t k = 0; k < 4; ++k) v[k] = depth <= TB ? (c[k] * quantum) & TMAX : c[k] >> shift;
            if (job.channels <= 2) { r = g = b = v[0]; if (job.channels == 2) a = v[1]; }
            else { r = v[job.bgr ? 2 : 0]; g = v[1]; b = v[job.bgr ? 0 : 2]; if (job.channels == 4) a = v[3]; }
            if (keyed) a = 0;
        }
        if (TB == 8) ((uint32_t *)out)[i] = r | g << 8 | b << 16 | a << 24;
        else { uint2 w; w.x = r | g << 16; w.y = b | a << 16; ((uint2 *)out)[i] = w; }
    }
}

hipError_t launch_unpack(const UnpackJob *d_jobs, uint32_t count, uint32_t blocks_x, int target, hipStream_t stream)
{
    if (!count) return hipSuccess;
    const dim3 grid(blocks_x ? blocks_x : 1, count);
    if (target == 8) unpack_kernel<uint8_t><<<grid, 256, 0, stream>>>(d_jobs);
    else unpack_kernel<uint16_t><<<grid, 256, 0, stream>>>(d_jobs);
    return hipGetLastError();
}

}  // namespace spng
