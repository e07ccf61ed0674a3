// unfilter.hip -- PNG scanline reconstruction (None/Sub/Up/Average/Paeth) for gfx950.
//
// Replaces PNG.Decoder.defilter (Sources/PNG/Decoding/PNG.Decoder.swift:152-196), PNG.paeth
// (Sources/PNG/PNG.swift:124-147) and, through the job geometry, the row/pass walker of
// PNG.Decoder.push (:59-140).  Byte arithmetic is u8 wrap-around exactly as in the reference.
//
// Parallelisation.  Pixel (x, y) of a sub-image depends on (x-1, y), (x, y-1) and (x-1, y-1), so
// the dependency DAG is a 2-D wavefront.  One wave (64 lanes) owns a band of 64 consecutive rows:
// lane r reconstructs row r and runs one pixel ("unit" = bpp bytes) behind lane r-1, so at step t
// lane r works on unit t - r.  The value lane r needs from the row above is exactly what lane r-1
// produced one step earlier, and arrives through a single DPP wave_shr:1 move -- no LDS, no
// barrier.  The left neighbour and the upper-left neighbour are register carries.
//
// Memory.  Row r of a tile covers the *skewed* window of units [T*P - r, (T+1)*P - r): since
// rows are pitch+1 bytes apart (never aligned) the loads are unaligned 16-byte loads anyway, so
// the skew costs nothing, and inside LDS every lane walks the same column index.  Tiles are
// staged with 16 B/lane coalesced global loads (16 consecutive lanes cover one 256-byte row
// segment), reconstructed in place in LDS, and written back with 16 B/lane stores.  The row above
// the band (the band's own previous output row, or zeros for the first row of a pass:
// PNG.Decoder.swift:83-84) is staged as LDS row 0 and feeds lane 0.
#include "common.hpp"

namespace spng {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };       // 16 bytes, alignment 1

template <int BPP> struct Cfg {
    static constexpr int P    = (BPP <= 4) ? 64 : 32;    // units per tile window
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
    return (s >= 16 && s < 48) || s == SPNG_E_REFERENCE_UNDEFINED;
}

template <int BPP>
__global__ __launch_bounds__(64) void unfilter_kernel(const UnfJob *__restrict__ jobs,
                                                      const spng_result *__restrict__ results)
{
    using C = Cfg<BPP>;
    __shared__ __attribute__((aligned(16))) uint8_t tile[65 * C::ROWB];

    const UnfJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    if (results && skip_status(results[job.image].status)) return;

    uint32_t rows = job.rows;
    if (job.rows_len) {
        // a short stream silently yields an incomplete image (PNG.Decoder.swift:88-94)
        const uint64_t len = *job.rows_len;
        const uint64_t avail = len > job.stream_off ? (len - job.stream_off) / job.in_stride : 0;
        rows = avail < rows ? (uint32_t)avail : rows;
    }
    const int64_t pitch = job.pitch;
    const uint32_t W = job.pitch / BPP;
    const uint32_t ntiles = (W + 63 + C::P - 1) / C::P;

    for (uint32_t band = 0; band * 64 < rows; ++band) {
        const uint32_t row = band * 64 + lane;
        const bool active = row < rows;
        const uint32_t ft = active ? job.in[(uint64_t)row * job.in_stride] : 0u;
        if (band) {
            // the row above this band was written by this wave's previous band: wait for those
            // stores and drop this CU's (possibly stale) L1 lines before re-reading them.
            // (same CU, hence same XCD L2: no L2 write-back needed, only vmcnt(0) + buffer_inv.)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        uint32_t o[BPP], bprev[BPP];
#pragma unroll
        for (int k = 0; k < BPP; ++k) { o[k] = 0; bprev[k] = 0; }

        for (uint32_t T = 0; T < ntiles; ++T) {
            // ---- stage: 65 row windows -> LDS
            for (int i = lane; i < 65 * C::CPR; i += 64) {
                const int rr = i / C::CPR, cj = i % C::CPR;
                u32x4 v = {0, 0, 0, 0};
                if (rr == 0) {
                    if (band)
                        v = load_window(job.out + (uint64_t)(band * 64 - 1) * job.out_stride,
                                        (int64_t)T * C::TB + 16 * cj, pitch);
                } else {
                    const int r = rr - 1;
                    const uint32_t rw = band * 64 + r;
                    if (rw < rows)
                        v = load_window(job.in + (uint64_t)rw * job.in_stride + 1,
                                        ((int64_t)T * C::P - r) * BPP + 16 * cj, pitch);
                }
                *(u32x4 *)(tile + rr * C::ROWB + 16 * cj) = v;
            }
            __syncthreads();

            // ---- reconstruct P units per lane, in place
            uint8_t *mine = tile + (1 + lane) * C::ROWB;
            const int64_t ux0 = (int64_t)T * C::P - lane;
#pragma unroll 4
            for (int t = 0; t < C::P; ++t) {
                const bool interior = ux0 + t > 0;          // unit 0 has no left / upper-left neighbour
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
                    else if (ft == 4) pred = paeth(a, b[k], c);
                    o[k] = (x + pred) & 0xffu;
                    mine[t * BPP + k] = (uint8_t)o[k];
                    bprev[k] = b[k];
                }
            }
            __syncthreads();

            // ---- write back the 64 row windows
            for (int i = lane; i < 64 * C::CPR; i += 64) {
                const int r = i / C::CPR, cj = i % C::CPR;
                const uint32_t rw = band * 64 + r;
                if (rw < rows)
                    store_window(job.out + (uint64_t)rw * job.out_stride,
                                 ((int64_t)T * C::P - r) * BPP + 16 * cj, pitch,
                                 *(const u32x4 *)(tile + (1 + r) * C::ROWB + 16 * cj));
            }
            __syncthreads();
        }
    }
}

hipError_t launch_unfilter(const UnfJob *d_jobs, uint32_t count, uint32_t bpp, spng_result *d_results,
                           hipStream_t stream)
{
    if (!count) return hipSuccess;
    switch (bpp) {
    case 1: unfilter_kernel<1><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    case 2: unfilter_kernel<2><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    case 3: unfilter_kernel<3><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    case 4: unfilter_kernel<4><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    case 6: unfilter_kernel<6><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    case 8: unfilter_kernel<8><<<count, 64, 0, stream>>>(d_jobs, d_results); break;
    default: return hipErrorInvalidValue;
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
            const uint32_t bpp = volume >> 3;
            for (uint32_t k = 0; k < bpp; ++k) job.storage[d * bpp + k] = scan[(uint64_t)i * bpp + k];
        }
    }
}

hipError_t launch_scatter(const ScatterJob *d_jobs, uint32_t count, const uint32_t *d_job_image,
                          const spng_result *d_results, uint32_t blocks_x, hipStream_t stream)
{
    if (!count) return hipSuccess;
    scatter_kernel<<<dim3(blocks_x, count), 256, 0, stream>>>(d_jobs, d_results, d_job_image);
    return hipGetLastError();
}

}  // namespace spng
