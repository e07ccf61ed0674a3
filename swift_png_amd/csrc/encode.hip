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

__global__ __launch_bounds__(256) void filter_kernel(const FilterJob *__restrict__ jobs)
{
    const FilterJob job = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t volume = job.depth * job.channels, bpp = (volume + 7) >> 3;
    const bool direct = volume >= 8 && job.sx == 1 && job.sy == 1 && job.bx == 0 && job.by == 0 &&
                        job.sub_w == job.width;
    for (uint32_t y = blockIdx.x * 4 + wave; y < job.sub_h; y += gridDim.x * 4) {
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
    filter_kernel<<<dim3(bx, count), 256, 0, stream>>>(d_jobs);
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
