// probe_copy.hip -- which access pattern of a 64-rows-per-wave scanline kernel reaches what share of HBM bandwidth?
//
// A calibration tool ("a real copy ceiling and a calibrated counter"), not part of the library:
//   hipcc --offload-arch=gfx950 -O3 -o probe_copy tools/probe_copy.hip && ./probe_copy
// (a) a plain 16-byte-per-lane grid-stride copy: the ceiling next to the 8 TB/s spec peak;
// (b) the unfilter kernel's pattern with nothing but the memory pipeline: a wave owns 64 rows, walks them tile by tile
//     (TW bytes per row and tile, 16 bytes per lane, TW/16 lanes per row), row r trailing row r-1 by `skew` bytes, loads of
//     tile i+1 in flight while tile i is stored -- over the row pitch of the source (pitch+1 as the inflator leaves it, or a
//     padded, aligned pitch), the tile width, the skew, the waves per SIMD and an amount of dependent VALU work per tile.
// Prints one line per configuration; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same binary calibrates the counters.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) U128u { u32x4 v; };

__global__ __launch_bounds__(256) void copy16_kernel(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = in[i];
}

struct RowJob {
    const uint8_t *in; uint8_t *out;
    int64_t in_stride, out_stride;      // bytes between rows
    int32_t pitch, rows, skew, valu;    // skew: bytes row r trails row r-1 by on the LOAD side
    int32_t sskew, rmod, ntl, nts;      // the same on the store side; skew index = r % rmod; non-temporal loads / stores    // row bytes, rows of the image, bytes row r trails row r-1 by, dependent VALU operations per 4 bytes of a row
    int64_t in_image, out_image;        // bytes between images
    int32_t pieces, piece_rows;         // an image is cut into pieces that different workgroups take
};

__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// Addressing as lean as the real kernel can make it: a wave-uniform base per load (scalar registers) plus ONE 32-bit lane
// offset for all loads of a tile and one for all stores (lane = (row r0 = lane / CPR of each group of 64 / CPR rows, chunk cj));
// the bounds are only looked at in the tiles where some window leaves its row.
template <int TW, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void rowtile_kernel(RowJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // 4 x 64 x (TW + 16) used; asking for more lowers the workgroups per CU
    constexpr int CPR = TW / 16, CH = CPR, RS = 64 / CPR;  // chunks per row, chunks per lane and tile, rows per load instruction
    constexpr int ROWB = TW + 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *tile = lds + wave * 64 * ROWB;
    const int img = blockIdx.x / job.pieces, piece = blockIdx.x % job.pieces;
    const int row0 = piece * job.piece_rows;
    const int rows = min(job.rows - row0, job.piece_rows);
    const uint8_t *in = job.in + (int64_t)img * job.in_image + (int64_t)row0 * job.in_stride;
    uint8_t *out = job.out + (int64_t)img * job.out_image + (int64_t)row0 * job.out_stride;
    const int nbands = (rows + 63) / 64;
    const int ntiles = (job.pitch + (job.rmod - 1) * (job.skew > job.sskew ? job.skew : job.sskew) + TW - 1) / TW;
    const int r0 = lane / CPR, cj = lane % CPR;
    const uint32_t lds0 = (uint32_t)(r0 * ROWB + 16 * cj);
    const int maxskew = (job.rmod - 1) * (job.skew > job.sskew ? job.skew : job.sskew);
    u32x4 R[CH];
    uint32_t sink = 0;
    for (int band = wave; band < nbands; band += 4) {
        const uint8_t *bin = (const uint8_t *)uni64((uint64_t)(in + (int64_t)band * 64 * job.in_stride));
        uint8_t *bout = (uint8_t *)uni64((uint64_t)(out + (int64_t)band * 64 * job.out_stride));
        // (per-load offsets: the skew index wraps at rmod, so the lean one-offset-per-tile form of the first version does not apply)
        auto issue = [&](int T) {
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int r = r0 + m * RS;
                const int off = T * TW + 16 * cj - (r % job.rmod) * job.skew;
                u32x4 v = {0, 0, 0, 0};
                if (off >= 0 && off + 16 <= job.pitch) {
                    const U128u *q = (const U128u *)(bin + (int64_t)r * job.in_stride + off);
                    v = NTL ? __builtin_nontemporal_load(&q->v) : q->v;
                }
                R[m] = v;
            }
        };
        issue(0);
        for (int T = 0; T < ntiles; ++T) {
            // as the unfilter kernel: the tile goes to LDS, the next one's loads are issued, the tile is worked on (lane = row)
            // and stored from LDS 16 bytes per lane
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < CH; ++m) *(u32x4 *)(tile + lds0 + m * RS * ROWB) = R[m];
            if (T + 1 < ntiles) issue(T + 1);
            // stand-in for the reconstruction: a chain of dependent VALU operations on the lane's own row
            uint32_t x = *(const uint32_t *)(tile + lane * ROWB);
            for (int k = 0; k < job.valu * (TW / 4); ++k) x = (x << 1) + (uint32_t)k;      // (one full-rate VALU operation per step)
            sink += x;
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int r = r0 + m * RS;
                const int off = T * TW + 16 * cj - (r % job.rmod) * job.sskew;
                if (off >= 0 && off + 16 <= job.pitch) {
                    U128u *q = (U128u *)(bout + (int64_t)r * job.out_stride + off);
                    const u32x4 v = *(const u32x4 *)(tile + lds0 + m * RS * ROWB);
                    if (NTS) __builtin_nontemporal_store(v, &q->v); else q->v = v;
                }
            }
        }
    }
    if (sink == 0x12345678u) out[0] = 1;                   // (keeps the arithmetic alive)
}

static float time_kernel(void (*launch)(void *), void *arg, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(arg); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch(arg);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms / reps;
}

struct RowLaunch { RowJob job; int tw, nimg, lds; };
static void launch_row(void *p)
{
    RowLaunch &l = *(RowLaunch *)p;
    const int grid = l.nimg * l.job.pieces;
#define ROWCASE(TWV) case TWV: \
        if (l.job.ntl && l.job.nts) rowtile_kernel<TWV, true, true><<<grid, 256, l.lds>>>(l.job); \
        else if (l.job.ntl) rowtile_kernel<TWV, true, false><<<grid, 256, l.lds>>>(l.job); \
        else if (l.job.nts) rowtile_kernel<TWV, false, true><<<grid, 256, l.lds>>>(l.job); \
        else rowtile_kernel<TWV, false, false><<<grid, 256, l.lds>>>(l.job); \
        break;
    switch (l.tw) { ROWCASE(128) ROWCASE(256) ROWCASE(512) }
}
struct CopyLaunch { const u32x4 *in; u32x4 *out; uint64_t n; int grid; };
static void launch_copy(void *p) { CopyLaunch &l = *(CopyLaunch *)p; copy16_kernel<<<l.grid, 256>>>(l.in, l.out, l.n); }

int main(int argc, char **argv)
{
    const int nimg = argc > 1 ? atoi(argv[1]) : 128;
    const char *only = argc > 2 ? argv[2] : "";           // "pmc": one configuration of each kind, one launch each
    const int W = 4096, H = 4096, pitch = W * 4;
    const int64_t in_bytes = (int64_t)nimg * (pitch + 128) * H + 4096, out_bytes = (int64_t)nimg * pitch * H;
    uint8_t *in, *out;
    CHECK(hipMalloc(&in, in_bytes)); CHECK(hipMalloc(&out, out_bytes));
    CHECK(hipMemset(in, 0x5a, in_bytes)); CHECK(hipMemset(out, 0, out_bytes));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<128, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<128, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<256, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<256, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<512, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<512, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<512, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CHECK(hipFuncSetAttribute((const void *)rowtile_kernel<512, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const bool pmc = !strcmp(only, "pmc");
    const int reps = pmc ? 1 : 5;
    const double alg = 2.0 * (double)out_bytes;
    {
        CopyLaunch c{(const u32x4 *)in, (u32x4 *)out, (uint64_t)out_bytes / 16, 256 * 16};
        for (int g : {256 * 8, 256 * 16, 256 * 32}) {
            c.grid = g;
            const float ms = time_kernel(launch_copy, &c, reps);
            printf("copy16 grid=%d: %.3f ms  %.0f GB/s  (%.1f%% of 8 TB/s)\n", g, ms, alg / ms / 1e6, alg / ms / 1e6 / 80.0);
            if (pmc) break;
        }
    }
    // {load skew, store skew, rows sharing a skew ramp, nt loads, nt stores}
    struct Cfg { int tw, skew, sskew, rmod, ntl, nts; };
    const Cfg cfgs[] = {
        {256, 0, 0, 64, 0, 0}, {256, 4, 4, 64, 0, 0}, {256, 4, 0, 64, 0, 0}, {256, 0, 4, 64, 0, 0},      // which side does the skew hurt on?
        {256, 4, 4, 64, 1, 0}, {256, 4, 4, 64, 0, 1}, {256, 4, 4, 64, 1, 1}, {256, 0, 0, 64, 1, 1},      // cache policy
        {256, 32, 32, 64, 0, 0}, {256, 64, 64, 64, 0, 0}, {256, 16, 16, 64, 1, 0},
        {128, 4, 0, 32, 0, 0}, {128, 4, 0, 32, 1, 0}, {128, 4, 0, 64, 0, 0}, {128, 0, 0, 64, 1, 0},      // half-wave bands: skewed loads, aligned line stores
        {128, 4, 4, 32, 0, 0}, {128, 4, 4, 32, 1, 0}, {256, 4, 4, 32, 0, 0}, {256, 4, 0, 32, 1, 0},
    };
    const int64_t stride = pitch + 1;
    for (const Cfg &c : cfgs) {
        if (pmc && !(c.tw == 256 && c.rmod == 64 && !c.ntl && !c.nts && c.skew == c.sskew && (c.skew == 0 || c.skew == 4))) continue;
        RowLaunch l;
        l.tw = c.tw; l.nimg = nimg;
        l.job.in = in + 1; l.job.out = out;
        l.job.in_stride = stride; l.job.out_stride = pitch;
        l.job.pitch = pitch; l.job.rows = H; l.job.skew = c.skew; l.job.sskew = c.sskew; l.job.rmod = c.rmod; l.job.ntl = c.ntl; l.job.nts = c.nts; l.job.valu = 0;
        l.job.in_image = stride * H; l.job.out_image = (int64_t)pitch * H;
        l.job.pieces = 4; l.job.piece_rows = H / 4;
        l.lds = 4 * 64 * (c.tw + 16);
        const float ms = time_kernel(launch_row, &l, reps);
        printf("rows(pitch+1 source) tile=%3d load-skew=%3d store-skew=%3d ramp-rows=%2d nt-loads=%d nt-stores=%d: %.3f ms  %.0f GB/s  (%.1f%% of 8 TB/s)\n", c.tw, c.skew, c.sskew,
               c.rmod, c.ntl, c.nts, ms, alg / ms / 1e6, alg / ms / 1e6 / 80.0);
        fflush(stdout);
    }
    return 0;
}
