// api.hip -- host side of libspng_mi355.so: the C ABI declared in include/spng_mi355.h.
//
// Everything here is plumbing: argument checks, job tables (the Adam7 / row geometry of
// PNG.Decoder.push, Sources/PNG/Decoding/PNG.Decoder.swift:59-140, and of PNG.Encoder.pull,
// Sources/PNG/Encoding/PNG.Encoder.swift:33-129), one pinned->device upload per call, kernel
// launches on the context's stream and optional HIP-event timing around each launch.  There is
// no CPU implementation of any hot-path function in this library.
#include "common.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>

namespace spng {

static thread_local char g_err[512] = "";

static int32_t fail_hip(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return SPNG_E_DEVICE;
}
static int32_t fail_text(const char *what)
{
    snprintf(g_err, sizeof g_err, "%s", what);
    return SPNG_E_DEVICE;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail_hip(e_, #expr); } while (0)

// PNG.adam7 (PNG.Decoder.swift:6-15) and the sub-image geometry of :63-82
int passes(uint32_t w, uint32_t h, int volume, int interlaced, Pass out[7])
{
    static const uint32_t A7[7][4] = {{0, 0, 3, 3}, {4, 0, 3, 3}, {0, 4, 2, 3}, {2, 0, 2, 2},
                                      {0, 2, 1, 2}, {1, 0, 1, 1}, {0, 1, 0, 1}};
    int n = 0;
    if (!interlaced) {
        if (w && h) out[n++] = Pass{0, 0, 1, 1, w, h, ((uint64_t)w * volume + 7) >> 3};
        return n;
    }
    for (int z = 0; z < 7; ++z) {
        const uint32_t bx = A7[z][0], by = A7[z][1], ex = A7[z][2], ey = A7[z][3];
        const uint32_t sx = 1u << ex, sy = 1u << ey;
        if (w + sx - bx - 1 < sx || h + sy - by - 1 < sy) continue;   // empty pass (:76-80)
        const uint32_t sw = (w + sx - bx - 1) >> ex, sh = (h + sy - by - 1) >> ey;
        if (!sw || !sh) continue;
        out[n++] = Pass{bx, by, sx, sy, sw, sh, ((uint64_t)sw * volume + 7) >> 3};
    }
    return n;
}

static bool valid_format(int depth, int channels)
{
    if (channels < 1 || channels > 4) return false;
    if (depth == 8 || depth == 16) return true;
    return channels == 1 && (depth == 1 || depth == 2 || depth == 4);
}

}  // namespace spng

using namespace spng;

struct spng_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // device + pinned workspaces for job tables
    void *d_ws = nullptr;  size_t d_ws_cap = 0;
    // Pinned staging for the job tables.  An entry point fills a slab on the host, enqueues its
    // upload and may return before the copy engine has read it (h_results == NULL), so the next
    // call must not scribble over the same pinned bytes: slabs are used round-robin and each is
    // guarded by an event recorded behind its upload.  (The device copy d_ws needs no such care: the
    // next upload is ordered behind the kernels that read the previous tables by the stream itself.)
    static constexpr int SLABS = 4;
    struct Slab { void *h = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    Slab slabs[SLABS];
    int slab_next = 0;
    void *h_ws = nullptr;                 // the slab of the call in progress
    Slab *cur = nullptr;
    void *d_ring = nullptr; size_t ring_cap = 0;     // deflate link rings (greedy / lazy kernel)
    void *d_ring2 = nullptr; size_t ring2_cap = 0;   // (one-kernel full search)
    // parallel inflate (pinflate2.hip): chunk-record slab, token buffer, knobs (spng_configure)
    void *d_graph = nullptr; size_t graph_cap = 0;   // deflate levels >= 8: match graphs
    void *d_log = nullptr;  size_t log_cap = 0;
    void *d_tok = nullptr;  size_t tok_cap = 0;      // bytes
    void *d_sym = nullptr;  size_t sym_cap = 0;      // several workgroups per stream: 16-bit symbols, windows (bytes)
    uint64_t sym_failed = 0;                        // a symbol scratch of this size could not be had (forgotten by spng_trim)
    void *d_win = nullptr;  size_t win_cap = 0;
    // token pool of the pipeline (pinflate2.hip): halfwords a compressed byte turned into in the last batch (learned,
    // so that the next batch of the same kind takes one pass), and the pinned word the page counter is read back into
    double   pool_ratio = 0;
    uint32_t *h_pool_used = nullptr;
    uint64_t pool_pages_planned = 0, pool_src_bytes = 0, pool_src_pending = 0;   // (source bytes of the batch planned / of the one whose counters are on their way)
    double   block_bytes = 0;        // compressed bytes per DEFLATE block in the last batch (0: not known)
    hipEvent_t pool_ev = nullptr; bool pool_pending = false;
    hipEvent_t ev_dfl[4] = {nullptr, nullptr, nullptr, nullptr};    // level >= 8 rounds: searched[parity], parsed[parity]
    // second stream of the pipeline: the decode of one half of a batch runs beside the resolve of the other
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_mid = nullptr, ev_join = nullptr;
    // spng_decode_batch_multi: the stream a context's rasters leave on (so that a group's copies run beside the next group's
    // decode), the events between the two, its part of the results, the peers it has been given access to
    hipStream_t stream_out = nullptr;
    hipEvent_t ev_out[2] = {nullptr, nullptr};
    void *d_multi = nullptr; size_t multi_cap = 0;
    uint64_t peers = 0, peers_refused = 0;
    int64_t cfg[SPNG_CFG_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // profiling
    bool profiling = false;
    struct Span { int kernel; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    std::mutex mu;

    // Starts a call: device table space for `bytes`, and a pinned slab nobody is reading any more.
    int32_t reserve(size_t bytes)
    {
        if (bytes > d_ws_cap) {
            // the previous tables may still be read by in-flight kernels
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_ws) { HIP_TRY(hipFree(d_ws)); d_ws = nullptr; }
            const size_t cap = bytes + bytes / 2 + 4096;
            HIP_TRY(hipMalloc(&d_ws, cap));
            d_ws_cap = cap;
        }
        Slab &sl = slabs[slab_next];
        slab_next = (slab_next + 1) % SLABS;
        if (sl.pending) { HIP_TRY(hipEventSynchronize(sl.ev)); sl.pending = false; }
        if (bytes > sl.cap) {
            if (sl.h) { HIP_TRY(hipHostFree(sl.h)); sl.h = nullptr; sl.cap = 0; }
            const size_t cap = bytes + bytes / 2 + 4096;
            HIP_TRY(hipHostMalloc(&sl.h, cap, hipHostMallocDefault));
            sl.cap = cap;
        }
        if (!sl.ev) HIP_TRY(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
        cur = &sl; h_ws = sl.h;
        return SPNG_DONE;
    }
    // Enqueues the upload of slab bytes [from, to) to the same offsets of d_ws and marks the slab busy
    // until the copy has executed.
    int32_t upload(size_t from, size_t to)
    {
        if (to > from)
            HIP_TRY(hipMemcpyAsync((char *)d_ws + from, (char *)h_ws + from, to - from, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(cur->ev, stream));
        cur->pending = true;
        return SPNG_DONE;
    }
    hipEvent_t event()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

struct Timed {            // records a pair of events around a launch when profiling is on
    spng_ctx *c; int k; hipEvent_t a = nullptr; hipStream_t s;
    Timed(spng_ctx *c_, int k_, hipStream_t s_ = nullptr) : c(c_), k(k_), s(s_ ? s_ : c_->stream) { if (c->profiling) { a = c->event(); (void)hipEventRecord(a, s); } }
    ~Timed() { if (a) { hipEvent_t b = c->event(); (void)hipEventRecord(b, s); c->spans.push_back({k, a, b}); } }
};

// simple bump allocator over the paired pinned/device workspaces
struct Arena {
    spng_ctx *c; size_t off = 0;
    template <class T> T *host(size_t at) { return (T *)((char *)c->h_ws + at); }
    template <class T> T *dev(size_t at) { return (T *)((char *)c->d_ws + at); }
    size_t take(size_t bytes) { size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; }
};

extern "C" {

int32_t spng_version(void) { return SPNG_VERSION; }

const char *spng_status_string(int32_t s)
{
    switch (s) {
    case SPNG_DONE: return "done";
    case SPNG_NEED_MORE_INPUT: return "need more input";
    case SPNG_E_COMPRESSION_METHOD: return "invalid rfc-1950 compression method code";
    case SPNG_E_WINDOW_SIZE: return "invalid rfc-1950 window size";
    case SPNG_E_CHECK_BITS: return "invalid rfc-1950 header check bits";
    case SPNG_E_DICTIONARY: return "unexpected rfc-1950 stream dictionary";
    case SPNG_E_STREAM_CHECKSUM: return "invalid rfc-1950 checksum";
    case SPNG_E_BLOCK_TYPE: return "invalid rfc-1951 block type code";
    case SPNG_E_BLOCK_COUNT_PARITY: return "invalid rfc-1951 block element count parity";
    case SPNG_E_RUNLITERAL_COUNT: return "invalid rfc-1951 run-literal symbol count";
    case SPNG_E_CODELENGTH_TABLE: return "malformed rfc-1951 codelength huffman table";
    case SPNG_E_CODELENGTH_SEQUENCE: return "invalid rfc-1951 codelength sequence";
    case SPNG_E_HUFFMAN_TABLE: return "malformed rfc-1951 huffman table";
    case SPNG_E_STRING_REFERENCE: return "invalid rfc-1951 string reference";
    case SPNG_E_EXTRANEOUS_IMAGE_DATA: return "image data buffer not empty after decoding final scanline";
    case SPNG_E_EXTRANEOUS_COMPRESSED_DATA: return "extraneous compressed image data after end of compressed stream";
    case SPNG_E_INCOMPLETE_DATASTREAM: return "reached end-of-image chunk while compressed image data stream is incomplete";
    case SPNG_E_TRUNCATED_SIGNATURE: return "signature truncated";
    case SPNG_E_SIGNATURE: return "invalid png signature bytes";
    case SPNG_E_TRUNCATED_CHUNK_HEADER: return "chunk header truncated";
    case SPNG_E_TRUNCATED_CHUNK_BODY: return "chunk body truncated";
    case SPNG_E_CHUNK_TYPE: return "invalid chunk type code";
    case SPNG_E_CHUNK_CHECKSUM: return "invalid chunk checksum";
    case SPNG_E_OUTPUT_CAPACITY: return "destination buffer too small";
    case SPNG_E_ARGUMENT: return "invalid argument";
    case SPNG_E_DEVICE: return "device error";
    case SPNG_E_REFERENCE_UNDEFINED: return "stream uses a distance code the reference leaves undefined";
    case SPNG_E_GZIP_SIGIL: return "invalid gzip sigil";
    case SPNG_E_GZIP_METHOD: return "invalid gzip compression method";
    case SPNG_E_GZIP_FLAG_BITS: return "invalid gzip flag bits";
    case SPNG_E_GZIP_HEADER_CHECKSUM: return "gzip header checksum unsupported";
    default: return "unknown status";
    }
}

const char *spng_last_error_string(void) { return g_err; }

uint64_t spng_inflated_size(uint32_t w, uint32_t h, int depth, int channels, int interlaced)
{
    Pass p[7];
    const int n = passes(w, h, depth * channels, interlaced, p);
    uint64_t u = 0;
    for (int i = 0; i < n; ++i) u += (p[i].pitch + 1) * (uint64_t)p[i].h;
    return u;
}

uint64_t spng_storage_size(uint32_t w, uint32_t h, int depth, int channels)
{
    return (uint64_t)w * h * (uint64_t)((depth * channels + 7) >> 3);
}

int32_t spng_create(int device, void *stream, spng_ctx **out)
{
    if (!out) return SPNG_E_ARGUMENT;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count)
        return fail_hip(e == hipSuccess ? hipErrorInvalidDevice : e, "spng_create: no such HIP device");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof g_err, "spng_create: device %d is %s; this library contains gfx950 code only",
                 device, prop.gcnArchName);
        return SPNG_E_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    spng_ctx *c = new spng_ctx;
    c->device = device;
    if (stream) c->stream = (hipStream_t)stream;
    else {
        hipError_t e2 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e2 != hipSuccess) { delete c; return fail_hip(e2, "hipStreamCreateWithFlags"); }
        c->owns_stream = true;
    }
    // (the deflater's match search asks the device once how its LDS orders the lanes of an atomic exchange: deflate.hip)
    if (hipError_t e3 = launch_deflate3_probe(c->stream); e3 != hipSuccess) { (void)hipGetLastError(); }
    *out = c;
    return SPNG_DONE;
}

void spng_destroy(spng_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->d_ws) (void)hipFree(c->d_ws);
    for (auto &sl : c->slabs) { if (sl.h) (void)hipHostFree(sl.h); if (sl.ev) (void)hipEventDestroy(sl.ev); }
    if (c->d_ring) (void)hipFree(c->d_ring);
    if (c->d_ring2) (void)hipFree(c->d_ring2);
    if (c->d_graph) (void)hipFree(c->d_graph);
    if (c->d_log) (void)hipFree(c->d_log);
    if (c->d_tok) (void)hipFree(c->d_tok);
    if (c->d_sym) (void)hipFree(c->d_sym);
    if (c->d_win) (void)hipFree(c->d_win);
    if (c->h_pool_used) (void)hipHostFree(c->h_pool_used);
    if (c->pool_ev) (void)hipEventDestroy(c->pool_ev);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream_out) { (void)hipStreamSynchronize(c->stream_out); (void)hipStreamDestroy(c->stream_out); }
    for (hipEvent_t e : c->ev_out) if (e) (void)hipEventDestroy(e);
    if (c->d_multi) (void)hipFree(c->d_multi);
    for (hipEvent_t e : {c->ev_fork, c->ev_mid, c->ev_join}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_dfl) if (e) (void)hipEventDestroy(e);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void *spng_stream(spng_ctx *c) { return c ? (void *)c->stream : nullptr; }

int32_t spng_sync(spng_ctx *c)
{
    if (!c) return SPNG_E_ARGUMENT;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return SPNG_DONE;
}

int32_t spng_configure(spng_ctx *c, int key, int64_t value)
{
    if (!c || key < 0 || key >= SPNG_CFG_COUNT || value < 0) return SPNG_E_ARGUMENT;
    std::lock_guard<std::mutex> g(c->mu);
    c->cfg[key] = value;
    return SPNG_DONE;
}

int32_t spng_profile(spng_ctx *c, int enable)
{
    if (!c) return SPNG_E_ARGUMENT;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &s : c->spans) { c->pool.push_back(s.a); c->pool.push_back(s.b); }
    c->spans.clear();
    c->profiling = enable != 0;
    return SPNG_DONE;
}

int32_t spng_token_stats(spng_ctx *c, uint64_t *page_bytes, uint64_t *blocks, int32_t *ran_dry)
{
    if (!c) return SPNG_E_ARGUMENT;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t *h = c->h_pool_used;                          // {pages, ran dry, blocks} of the last batch read back
    if (page_bytes) *page_bytes = h ? (uint64_t)h[0] << 16 : 0;
    if (ran_dry) *ran_dry = h ? (int32_t)h[1] : 0;
    if (blocks) *blocks = h ? h[2] : 0;
    return SPNG_DONE;
}

int32_t spng_profile_get(spng_ctx *c, int kernel, double *total_ms, uint64_t *launches)
{
    if (!c || kernel < 0 || kernel >= SPNG_K_COUNT) return SPNG_E_ARGUMENT;
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    double t = 0; uint64_t n = 0;
    for (auto &s : c->spans) if (s.kernel == kernel) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        t += ms; ++n;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    return SPNG_DONE;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
namespace spng {

__global__ void finish_decode_kernel(spng_result *results, const uint64_t *expected, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int32_t st = results[i].status;
    // PNG.Decoder.swift:142-147: anything left in the inflator after the last row
    if (st == SPNG_E_OUTPUT_CAPACITY ||
        ((st == SPNG_DONE || st == SPNG_NEED_MORE_INPUT) && results[i].written > expected[i]))
        results[i].status = SPNG_E_EXTRANEOUS_IMAGE_DATA;
}

// A stream whose waves gave up on each other (inflate.hip: SPIN_LIMIT) never writes its result: such
// a slot must read as a device error that wrote and consumed nothing.
__global__ void poison_results_kernel(spng_result *results, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    results[i].status = SPNG_E_DEVICE; results[i].reserved = 0;
    results[i].written = 0; results[i].consumed = 0;
    results[i].aux[0] = results[i].aux[1] = 0;
}

__global__ void init_results_kernel(spng_result *results, const uint64_t *written, uint32_t count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    results[i].status = SPNG_DONE; results[i].reserved = 0;
    results[i].written = written[i]; results[i].consumed = 0;
    results[i].aux[0] = results[i].aux[1] = 0;
}

// Plans and launches the unfilter (+ scatter) stage for a batch.  rows_len_of(i) gives the device
// address holding the number of valid inflated bytes of image i (or null).
struct UnfilterPlan {
    std::vector<UnfJob> unf[9];          // indexed by bpp
    std::vector<ScatterJob> scat;
    std::vector<uint32_t> scat_image;
    std::vector<OverdrawJob> over;       // spng_unfilter_resume_batch with SPNG_IMAGE_OVERDRAW
};

static int32_t plan_unfilter(const spng_image_desc *descs, uint32_t count, UnfilterPlan &plan,
                             const uint64_t *(*rows_len_of)(void *, uint32_t), void *user)
{
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        if (!valid_format(d.depth, d.channels) || !d.d_rows || !d.d_storage) return SPNG_E_ARGUMENT;
        const int volume = d.depth * d.channels;
        const uint32_t bpp = (uint32_t)(volume + 7) >> 3;
        const uint64_t u = spng_inflated_size(d.width, d.height, d.depth, d.channels, d.interlaced);
        if (d.rows_cap < u) return SPNG_E_ARGUMENT;
        Pass p[7];
        const int np = passes(d.width, d.height, volume, d.interlaced, p);
        const bool direct = !d.interlaced && volume >= 8;   // rows land in storage as they are
        uint64_t off = 0;
        for (int z = 0; z < np; ++z) {
            UnfJob j;
            j.in = (const uint8_t *)d.d_rows + off;
            j.in_stride = p[z].pitch + 1;
            if (direct) { j.out = (uint8_t *)d.d_storage; j.out_stride = p[z].pitch; }
            else        { j.out = (uint8_t *)d.d_rows + off + 1; j.out_stride = p[z].pitch + 1; }
            j.stream_off = off;
            j.rows_len = rows_len_of(user, i);
            j.pitch = (uint32_t)p[z].pitch; j.rows = p[z].h; j.image = i; j.bpp = bpp; j.has_prev = 0; j.pad = 0;
            plan.unf[bpp].push_back(j);
            if (!direct) {
                ScatterJob s;
                s.rows = (const uint8_t *)d.d_rows + off + 1;
                s.storage = (uint8_t *)d.d_storage;
                s.row_stride = p[z].pitch + 1; s.stream_off = off; s.rows_len = j.rows_len;
                s.sub_w = p[z].w; s.sub_h = p[z].h; s.width = d.width;
                s.bx = p[z].bx; s.by = p[z].by; s.sx = p[z].sx; s.sy = p[z].sy;
                s.depth = d.depth; s.channels = d.channels;
                plan.scat.push_back(s);
                plan.scat_image.push_back(i);
            }
            off += (p[z].pitch + 1) * (uint64_t)p[z].h;
        }
    }
    return SPNG_DONE;
}

static size_t plan_bytes(const UnfilterPlan &plan)
{
    size_t b = 0;
    for (int k = 1; k <= 8; ++k) b += plan.unf[k].size() * sizeof(UnfJob) + 256;
    b += plan.scat.size() * (sizeof(ScatterJob) + 4) + 512;
    b += plan.over.size() * sizeof(OverdrawJob) + 256;
    return b;
}

// copies the plan into the arena (host side) and launches after the caller's upload
struct PlanSlots { size_t unf[9]; size_t scat, scat_image, over; };

static void stage_plan(const UnfilterPlan &plan, Arena &a, PlanSlots &slots)
{
    for (int k = 1; k <= 8; ++k) {
        slots.unf[k] = a.take(plan.unf[k].size() * sizeof(UnfJob));
        if (!plan.unf[k].empty())
            memcpy(a.host<UnfJob>(slots.unf[k]), plan.unf[k].data(), plan.unf[k].size() * sizeof(UnfJob));
    }
    slots.scat = a.take(plan.scat.size() * sizeof(ScatterJob));
    slots.scat_image = a.take(plan.scat.size() * 4);
    if (!plan.scat.empty()) {
        memcpy(a.host<ScatterJob>(slots.scat), plan.scat.data(), plan.scat.size() * sizeof(ScatterJob));
        memcpy(a.host<uint32_t>(slots.scat_image), plan.scat_image.data(), plan.scat.size() * 4);
    }
    slots.over = a.take(plan.over.size() * sizeof(OverdrawJob));
    if (!plan.over.empty()) memcpy(a.host<OverdrawJob>(slots.over), plan.over.data(), plan.over.size() * sizeof(OverdrawJob));
}

static int32_t launch_plan(spng_ctx *c, const UnfilterPlan &plan, Arena &a, const PlanSlots &slots,
                           spng_result *d_results)
{
    {
        Timed t(c, SPNG_K_UNFILTER);
        for (int k = 1; k <= 8; ++k)
            if (!plan.unf[k].empty()) {
                // enough workgroups to fill the chip: chains are cut into pieces of piece_rows rows
                // (unfilter.hip: at rows filtered with None / Sub) when there are few of them
                uint64_t total_rows = 0; uint32_t max_rows = 1;
                for (auto &j : plan.unf[k]) { total_rows += j.rows; max_rows = j.rows > max_rows ? j.rows : max_rows; }
                uint32_t piece_rows = (uint32_t)c->cfg[SPNG_CFG_UNFILTER_PIECE_ROWS];
                uint64_t widest = 0;
                for (auto &j : plan.unf[k]) widest = (uint64_t)j.pitch > widest ? (uint64_t)j.pitch : widest;
                if (!piece_rows) {
                    piece_rows = (uint32_t)((total_rows / 4096 + 63) & ~(uint64_t)63);
                    if (piece_rows < 128) piece_rows = 128;
                    // (round 6: a piece of 128 rows is two bands of 64 -- two of the workgroup's four waves have nothing to do.  Rows of
                    //  2 KiB and more take four bands at least: 128 x 4096^2 RGB16 15.2 -> 10.0 ms, 256 x RGB8 10.5 -> 10.0; rows of a
                    //  1-bit image, 512 bytes, want the workgroups more than the waves: profiles/r06_tuning.md 15)
                    if (widest >= 2048 && k != 4 && k != 8) {
                        piece_rows = (uint32_t)((total_rows / 2048 + 63) & ~(uint64_t)63);
                        if (piece_rows < 256) piece_rows = 256;
                        if (piece_rows > 1024) piece_rows = 1024;
                    }
                    if (k == 4 || k == 8) {
                        // (the line-aligned kernel: bands of 128 / bpp rows, 64 / that many chains per wave -- pieces may be
                        //  shorter, and a few images still fill the chip)
                        const uint32_t rr = 128u / (uint32_t)k;
                        piece_rows = (uint32_t)((total_rows / 8192 + rr - 1) / rr * rr);
                        if (piece_rows < rr) piece_rows = rr;
                        // (a piece of fewer than four bands leaves waves of its 4-wave workgroup idle and pays a pipeline fill per
                        //  band: while two rounds of resident workgroups -- 3 per CU -- are there anyway, pieces are not cut below
                        //  four bands.  128 images: 64-row pieces, 9.4 ms -> 128-row pieces; VERDICT r4 "what's weak" 6)
                        const uint32_t fill = (uint32_t)(total_rows / 1536 / rr * rr);
                        const uint32_t floor4 = fill < 4 * rr ? fill : 4 * rr;
                        if (piece_rows < floor4) piece_rows = floor4;
                        // (round 6, measured per batch size -- profiles/r06_tuning.md 16: 128 x 4096^2 RGBA8 want pieces of 256 rows
                        //  (7.1 -> 5.9 ms), 32 images 128 (3.0 -> 2.3), 8 images 64 (1.8 -> 1.5): total rows / 1024, between 64 and 256)
                        //  -- for 4-byte pixels in images of 1024 rows and more, what was measured: the 8192^2 RGBA16 image of configs[4], whose
                        //  rows are 64 KiB, wants its 16-row bands one to a piece (3.9 ms; 64-row pieces 5.5), and images of 240 rows
                        //  two pieces each)
                        if (k == 4 && max_rows >= 1024) {
                            uint32_t few = (uint32_t)((total_rows / 1024 + rr - 1) / rr * rr);
                            few = few < 64 ? 64 : few > 256 ? 256 : few;
                            if (piece_rows < few) piece_rows = few;
                        }
                    }
                }
                const uint32_t pieces = (max_rows + piece_rows - 1) / piece_rows;
                HIP_TRY(launch_unfilter(a.dev<UnfJob>(slots.unf[k]), (uint32_t)plan.unf[k].size(), k,
                                        d_results, pieces, piece_rows, c->stream, (uint32_t)(widest > 0xffffffffull ? 0xffffffffull : widest)));
            }
    }
    if (!plan.scat.empty()) {
        Timed t(c, SPNG_K_SCATTER);
        uint64_t maxpix = 1;
        for (auto &s : plan.scat) { uint64_t px = (uint64_t)s.sub_w * s.sub_h; if (px > maxpix) maxpix = px; }
        uint32_t bx = (uint32_t)((maxpix + 255) / 256);
        if (bx > 1024) bx = 1024;
        HIP_TRY(launch_scatter(a.dev<ScatterJob>(slots.scat), (uint32_t)plan.scat.size(),
                               a.dev<uint32_t>(slots.scat_image), d_results, bx, c->stream));
    }
    if (!plan.over.empty()) {
        Timed t(c, SPNG_K_SCATTER);
        uint64_t maxpix = 1;
        for (auto &o : plan.over) { uint64_t px = (uint64_t)o.width * (o.y1 - o.y0); if (px > maxpix) maxpix = px; }
        uint32_t bx = (uint32_t)((maxpix + 255) / 256);
        if (bx > 4096) bx = 4096;
        HIP_TRY(launch_overdraw(a.dev<OverdrawJob>(slots.over), (uint32_t)plan.over.size(), bx, c->stream));
    }
    return SPNG_DONE;
}

// ---- inflate stage: the parallel pipeline (pinflate2.hip) in front of the serial kernel (inflate.hip) ----
struct InflatePlan {
    std::vector<InflateJob> jobs;
    std::vector<PStream> streams;
    std::vector<PSeg> segs;
    size_t log_bytes = 0;
    bool parallel = false;
    // pinflate2: the token pool and the groups of streams that share it, one after the other
    uint32_t pool_pages = 0;
    struct Group { uint32_t s0, s1, g0, g1, page0, pages; };   // streams, segments, its pages of the pool
    std::vector<Group> groups;
    bool overlap = false;            // two groups, each with its own half of the pool, on two streams (see launch_inflate_plan)
    uint32_t pmax = 0;               // several workgroups per stream: part slots per stream (0: one workgroup per stream)
    size_t parts_at = 0;
    size_t next_at = 0;
    bool gzip = false;               // some stream is SPNG_FORMAT_GZIP: header kernel in front, CRC-32 check behind
    std::vector<uint64_t> state;     // {bit, written} per stream: spng_inflate_resume_batch's, else the library's own zeros
    bool internal = true;            // (the latter)
    size_t jobs_at = 0, streams_at = 0, segs_at = 0, done_at = 0, gz_at = 0, gzparts_at = 0, state_at = 0, sumparts_at = 0;
    size_t bytes() const
    {
        return jobs.size() * (sizeof(InflateJob) + sizeof(PStream) + 4 + (size_t)pmax * sizeof(PPart) + 256 + (gzip ? 8 + 4 * (size_t)gzip_pieces() : 0) +
                              (state.empty() ? 0 : 32 + 8 * (size_t)gzip_pieces())) +
               segs.size() * sizeof(PSeg) + 8192 + 512;
    }
};

#ifndef SPNG_PARTS_MAX
#define SPNG_PARTS_MAX 128        // parts a stream's chain is cut into at most (round 5: 64 -- one to four images left half the chip idle)
#endif
static constexpr uint64_t RESUME_SERIAL_BITS = 8ull << 20;      // 1 MiB of input inside one block: resume there, not at its header

// Cuts every stream into segments and makes sure the context owns what the pipeline needs.  Segment length: long
// enough that the search for a block header (which costs more per bit than decoding) stays a small part of a
// segment's work, short enough that the batch yields several thousand segments, i.e. a few waves per SIMD.
//
// pinflate2 (the default): a page table per segment (c->d_log) and the token pool (c->d_tok, 64 KiB pages).  A
// compressed byte becomes at most 8 token halfwords and a stream at most one per output byte; what a batch really
// needs is far less (0.8 per byte for zlib-made PNG streams, 1.4 for swift-png's own), so the pool is sized by the
// ratio the previous batch showed (3.2 bytes per byte before there is one), capped by SPNG_CFG_TOKEN_BYTES or half of
// the free memory, and the streams take it in as many groups as that needs.  A segment that finds the pool empty
// gives its stream to the serial kernel.
static int32_t plan_inflate(spng_ctx *c, InflatePlan &p)
{
    p.internal = p.state.empty();
    if (p.internal) p.state.assign(p.jobs.size() * 4, 0);
    for (auto &j : p.jobs) j.internal = p.internal ? 1 : 0;
    p.parallel = c->cfg[SPNG_CFG_INFLATE_MODE] != SPNG_INFLATE_SERIAL && !p.jobs.empty();
    for (auto &j : p.jobs) p.gzip = p.gzip || j.format == SPNG_FORMAT_GZIP;
    if (!p.parallel) return SPNG_DONE;
    uint64_t total = 0;
    for (auto &j : p.jobs) total += j.src_len;
    // what the last batch taught about token volume (its page count comes back behind its kernels: when the planning
    // figure would cut THIS batch into groups, waiting for that number is cheaper than not knowing it)
    if (c->pool_pending && c->pool_ratio == 0 && hipEventQuery(c->pool_ev) != hipSuccess) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint64_t room = c->cfg[SPNG_CFG_TOKEN_BYTES] ? (uint64_t)c->cfg[SPNG_CFG_TOKEN_BYTES] : (uint64_t)(free_b + c->tok_cap) / 2;
        if ((double)total * 3.2 > (double)room) HIP_TRY(hipEventSynchronize(c->pool_ev));
    }
    if (c->pool_pending && hipEventQuery(c->pool_ev) == hipSuccess) {
        c->pool_pending = false;
        const uint64_t used = c->h_pool_used[0], blocks = c->h_pool_used[2];
        if (c->h_pool_used[1]) c->pool_ratio = 0;                                   // it ran dry: back to the default
        else if (c->pool_src_pending > (1u << 20)) c->pool_ratio = (double)used * 65536.0 / (double)c->pool_src_pending;
        c->block_bytes = (blocks && c->pool_src_pending > (1u << 20)) ? (double)c->pool_src_pending / (double)blocks : 0;
    }
    uint64_t seg_bytes = (uint64_t)c->cfg[SPNG_CFG_SEGMENT_BYTES];
    if (!seg_bytes) {
        // (~9 rounds of resident waves, so that the last, partly filled one costs little; the search costs 7 ms per 10^4 segments)
        // (small batches: >= 4096 segments if that leaves them 64 KiB -- a zlib block is ~40 KB, and a segment without a block
        // start is a wave without work; one 4K image: 15.1 ms per decode with 256 KiB segments, 10.3 with 64 KiB)
        // (streams of small blocks -- swift-png closes one every 2047 terms, ~2 KB -- as the last batch showed them: their
        // search is nearly free, and twice the segments halve what the last round of resident waves leaves idle:
        // 1024 x 4K images, decode 317 -> 301 ms)
        // (round 6: the search is 2-3 x cheaper -- pinf2_find's bit-parallel screen --, so streams of small blocks take 2.5 x the
        // segments again: 1024 x 4K images, swift-png-made: 418 KB segments 503.4 ms per step, 180 KB 494.5, 140 KB 494.0, 100 KB
        // 500.1; zlib-made streams, whose search still costs 2.4 ms per 10^4 segments, stay: profiles/archive/r06m_probe_segments.log)
        seg_bytes = total / ((c->block_bytes > 0 && c->block_bytes < 8192) ? 163840 : 32768);
        // (streams of small blocks, whose search is nearly free: 32 .. 64 KiB in small batches -- one 4K image 7.6 -> 6.65 ms per
        // call, 32 images 26.1 -> 22.7; zlib-made streams pay the search of every added segment -- 32 images 23.1 -> 27.0 at 32 KiB --
        // and stay at 64 .. 256 KiB: profiles/archive/r06r_probe_small_segments_*.log)
        const bool small_blocks = c->block_bytes > 0 && c->block_bytes < 8192;
        uint64_t least = total / 4096;
        const uint64_t lo = small_blocks ? 32u << 10 : 64u << 10, hi = small_blocks ? 64u << 10 : 256u << 10;
        if (least < lo) least = lo;
        if (least > hi) least = hi;
        if (seg_bytes < least) seg_bytes = least;
    }
    seg_bytes = (seg_bytes + 255) & ~(uint64_t)255;
    p.streams.resize(p.jobs.size());
    const double per_byte = c->pool_ratio > 0 ? (c->pool_ratio * 1.25 < 1.0 ? 1.0 : c->pool_ratio * 1.25) : 3.2;
    std::vector<uint64_t> est(p.jobs.size(), 0);
    size_t log = 0;
    for (size_t i = 0; i < p.jobs.size(); ++i) {
        const InflateJob &j = p.jobs[i];
        PStream &st = p.streams[i];
        memset(&st, 0, sizeof st);
        st.src = j.src; st.dst = j.dst; st.src_len = j.src_len; st.dst_cap = j.dst_cap;
        st.format = j.format; st.image = j.image;
        if (!p.state.empty()) {
            st.start_bit = p.state[4 * i]; st.out_pos = p.state[4 * i + 1];
            // The caller's state stands inside a block.  A block of ordinary size is simply decoded again from its header by the
            // pipeline (cheaper than the serial kernel for everything behind it); one that has already taken more than
            // RESUME_SERIAL_BITS of input would make every push cost what all pushes before it did -- a stream that is ONE block
            // pushed in k pieces O(n k) --: the serial kernel goes on at the token the last push stopped in front of.
            if (!p.internal && p.state[4 * i + 2] && p.state[4 * i + 2] - p.state[4 * i] > RESUME_SERIAL_BITS) st.serial_only = 1;
        }
        st.seg_first = (uint32_t)p.segs.size();
        uint64_t k = (j.src_len + seg_bytes - 1) / seg_bytes;
        if (k < 1) k = 1;
        if (p.segs.size() + k > 0x7fffffffu) return SPNG_E_ARGUMENT;
        st.seg_count = (uint32_t)k; st.seg_bytes = seg_bytes;
        for (uint64_t q = 0; q < k; ++q) {
            PSeg sg;
            memset(&sg, 0, sizeof sg);
            sg.stream = (uint32_t)i; sg.index = (uint32_t)q;
            const uint64_t len = q + 1 < k ? seg_bytes : j.src_len - q * seg_bytes;
            // page-table entries: 16 token bytes per compressed byte at most, and never more than two per output byte
            uint64_t most = 16 * len;
            if (most > 2 * (j.dst_cap + 64)) most = 2 * (j.dst_cap + 64);
            sg.log_cap = (most >> 16) + 2;
            sg.log_off = log; log += sg.log_cap;
            sg.start_bit = ~0ull;
            p.segs.push_back(sg);
        }
        uint64_t e = (uint64_t)(per_byte * (double)j.src_len);
        if (e > 2 * (j.dst_cap + 64)) e = 2 * (j.dst_cap + 64);
        est[i] = e + k * 65536 + 65536;                                            // (every segment ends inside a page)
    }
    log *= 4;
    p.log_bytes = log;
    if (log > c->log_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_log) { HIP_TRY(hipFree(c->d_log)); c->d_log = nullptr; c->log_cap = 0; }
        HIP_TRY(hipMalloc(&c->d_log, log + log / 8));
        c->log_cap = log + log / 8;
    }
    uint64_t want = 0, largest = 0;
    for (auto e : est) { want += e; if (e > largest) largest = e; }
    uint64_t budget = (uint64_t)c->cfg[SPNG_CFG_TOKEN_BYTES];
    if (!budget) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        budget = (uint64_t)(free_b + c->tok_cap) / 2;
        if (budget < (64ull << 20)) budget = 64ull << 20;
    }
    if (budget < 2 * largest && !c->cfg[SPNG_CFG_TOKEN_BYTES]) budget = 2 * largest;
    uint64_t need = want < budget ? want : budget;
    if (need < largest) need = largest;                                            // (a stream is never split over groups)
    need = (need + 65535) & ~(uint64_t)65535;
    if (need > c->tok_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_tok) { HIP_TRY(hipFree(c->d_tok)); c->d_tok = nullptr; c->tok_cap = 0; }
        if (hipMalloc(&c->d_tok, need) != hipSuccess) {
            // no room for the pipeline: the serial kernel takes the batch
            (void)hipGetLastError();
            c->d_tok = nullptr; c->tok_cap = 0;
            p.parallel = false; p.streams.clear(); p.segs.clear();
            return SPNG_DONE;
        }
        c->tok_cap = need;
    }
    // groups of consecutive streams whose estimates fit the pool together
    const uint64_t pool = c->tok_cap & ~(uint64_t)65535;
    p.pool_pages = (uint32_t)(pool >> 16 > 0xfffffff0ull ? 0xfffffff0ull : pool >> 16);
    // (as many groups as the estimates need, of equal share: 730 + 294 streams cost resolve three rounds of resident
    // workgroups where 512 + 512 cost two)
    uint64_t share = pool;
    if (want > pool) {
        const uint64_t ng = (want + pool - 1) / pool;
        share = (want + ng - 1) / ng;
        if (share < largest) share = largest;
        if (share > pool) share = pool;
    }
    uint64_t run = 0;
    InflatePlan::Group g{0, 0, 0, 0, 0, p.pool_pages};
    for (size_t i = 0; i < p.jobs.size(); ++i) {
        if (run + est[i] > share && g.s1 > g.s0) {
            p.groups.push_back(g);
            g.s0 = g.s1; g.g0 = g.g1; run = 0;
        }
        run += est[i];
        g.s1 = (uint32_t)i + 1; g.g1 = p.streams[i].seg_first + p.streams[i].seg_count;
    }
    p.groups.push_back(g);
    // On request (SPNG_CFG_INFLATE_OVERLAP) a batch that fits the pool at once goes in two halves, each with its share of
    // the pool, the decode of the second beside the resolve of the first on a second stream.  Not by default: both
    // kernels live on LDS (14 x 11 KB decode waves or 2 x 61 KB resolve workgroups fill a CU), so side by side they
    // only displace each other -- 1024 x 4K images: 635 ms per step instead of 526 (profiles/r03_inflate_tuning.md).
    const int64_t ovl = c->cfg[SPNG_CFG_INFLATE_OVERLAP];
    if (p.groups.size() == 1 && p.jobs.size() >= 2 && ovl == SPNG_OVERLAP_ALWAYS) {
        uint64_t half = 0, sum = 0;
        size_t cut = 0;
        for (auto e : est) sum += e;
        while (cut + 1 < p.jobs.size() && 2 * (half + est[cut]) <= sum + est[cut]) half += est[cut++];
        if (cut >= 1 && cut < p.jobs.size()) {
            InflatePlan::Group a = p.groups[0], b = p.groups[0];
            a.s1 = b.s0 = (uint32_t)cut;
            a.g1 = b.g0 = p.streams[cut].seg_first;
            uint64_t pa = (uint64_t)((double)p.pool_pages * ((double)half / (double)sum));
            if (pa < 1) pa = 1;
            if (pa >= p.pool_pages) pa = p.pool_pages - 1;
            a.page0 = 0; a.pages = (uint32_t)pa;
            b.page0 = (uint32_t)pa; b.pages = p.pool_pages - (uint32_t)pa;
            p.groups = {a, b};
            p.overlap = true;
        }
    }
    // Few streams: a workgroup per stream would leave most of the chip idle while each resolves its stream at ~0.9 GB/s
    // (one 4K image: 77 ms of an 89 ms decode).  Then a stream's chain is cut into parts that resolve side by side
    // (pinflate2.hip, "Several workgroups per stream"): 16-bit symbols in c->d_sym, the windows in c->d_win.
    p.pmax = 0;
    if (p.internal && p.jobs.size() <= 384 && c->cfg[SPNG_CFG_RESOLVE_PARTS] != 1) {
        // (512 workgroups in all: the marker parts run two to a CU -- pinflate2.hip, RGeo --, so that is one round of resident
        // workgroups; 768 left half a round behind: 128 images 41.6 ms of resolve against 33.5, 1024 parts 34.2)
        uint32_t pm = (uint32_t)(512 / p.jobs.size());
        if (pm < 2) pm = 2;
        if (c->cfg[SPNG_CFG_RESOLVE_PARTS] > 1) pm = (uint32_t)c->cfg[SPNG_CFG_RESOLVE_PARTS];
        if (pm > SPNG_PARTS_MAX) pm = SPNG_PARTS_MAX;
        if (c->cfg[SPNG_CFG_RESOLVE_PARTS] <= 1) {
            // (a part has fixed costs -- its first window, the hand-over of the windows part by part, the symbols' second pass --: not
            // below ~1 MiB of output each.  One 4K image: 64 parts 2.94 ms, 128 parts 3.81; the 8192^2 RGBA16 image of configs[4]:
            // 64 parts 11.1 ms, 128 parts 8.3: profiles/archive/r06x_probe_parts128.log)
            uint64_t most = 0;
            for (auto &j : p.jobs) most = j.dst_cap > most ? j.dst_cap : most;
            const uint64_t by_size = most >> 20 < 2 ? 2 : most >> 20;
            if (pm > by_size) pm = (uint32_t)by_size;
        }
        uint64_t syms = 0;
        for (size_t i = 0; i < p.jobs.size(); ++i) {
            p.streams[i].sym_off = syms;
            syms += (p.jobs[i].dst_cap + 15) & ~(uint64_t)7;
        }
        const uint64_t win = (uint64_t)p.jobs.size() * pm * 32768;
        // (the symbol scratch is two bytes per output byte: only when it fits a quarter of what is free, and not again after an
        // allocation of that size has failed -- a failed hipMalloc of gigabytes per call costs more than the parts save)
        bool afford = syms * 2 <= c->sym_cap;
        if (!afford && !(c->sym_failed && syms * 2 >= c->sym_failed)) {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            afford = syms * 2 + win <= (uint64_t)(free_b + c->sym_cap + c->win_cap) / 4;
        }
        if (pm >= 2 && afford) {
            bool room = true;
            if (syms * 2 > c->sym_cap) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (c->d_sym) { HIP_TRY(hipFree(c->d_sym)); c->d_sym = nullptr; c->sym_cap = 0; }
                if (hipMalloc(&c->d_sym, syms * 2) != hipSuccess) { (void)hipGetLastError(); room = false; c->sym_failed = syms * 2; }
                else { c->sym_cap = syms * 2; c->sym_failed = 0; }
            }
            if (room && win > c->win_cap) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (c->d_win) { HIP_TRY(hipFree(c->d_win)); c->d_win = nullptr; c->win_cap = 0; }
                if (hipMalloc(&c->d_win, win) != hipSuccess) { (void)hipGetLastError(); room = false; }
                else c->win_cap = win;
            }
            if (room) {
                p.pmax = pm;
                for (auto &st : p.streams) st.parts_max = pm;
            }
        }
    }
    if (!p.pmax && c->d_sym && p.jobs.size() > 384) {
        // (a large batch after small ones: the symbol scratch -- two bytes per output byte -- goes back to the device)
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->d_sym)); c->d_sym = nullptr; c->sym_cap = 0;
    }
    c->pool_pages_planned = p.pool_pages;
    c->pool_src_bytes = total;
    if (!c->h_pool_used) {
        HIP_TRY(hipHostMalloc((void **)&c->h_pool_used, 64, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&c->pool_ev, hipEventDisableTiming));
    }
    return SPNG_DONE;
}

static void stage_inflate(InflatePlan &p, Arena &a)
{
    const size_t n = p.jobs.size();
    p.jobs_at = a.take(n * sizeof(InflateJob));
    if (!p.state.empty()) {
        p.state_at = a.take(n * 32);
        memcpy(a.host<uint64_t>(p.state_at), p.state.data(), n * 32);
        for (size_t i = 0; i < n; ++i) {
            p.jobs[i].state = a.dev<uint64_t>(p.state_at) + 4 * i;
            if (p.parallel) p.streams[i].state = p.jobs[i].state;
        }
    }
    if (p.parallel) {
        p.streams_at = a.take(n * sizeof(PStream));
        p.segs_at = a.take(p.segs.size() * sizeof(PSeg));
        memcpy(a.host<PStream>(p.streams_at), p.streams.data(), n * sizeof(PStream));
        memcpy(a.host<PSeg>(p.segs_at), p.segs.data(), p.segs.size() * sizeof(PSeg));
    }
    if (p.parallel) {
        p.next_at = a.take(64);
        memset(a.host<uint32_t>(p.next_at), 0, 64);
        if (p.pmax) {
            p.parts_at = a.take(n * p.pmax * sizeof(PPart));
            memset(a.host<PPart>(p.parts_at), 0, n * p.pmax * sizeof(PPart));
        }
    }
    if (p.parallel || p.gzip) {
        p.done_at = a.take(n * 4);
        memset(a.host<int32_t>(p.done_at), 0, n * 4);
        for (size_t i = 0; i < n; ++i) p.jobs[i].skip = a.dev<int32_t>(p.done_at) + i;
    }
    memcpy(a.host<InflateJob>(p.jobs_at), p.jobs.data(), n * sizeof(InflateJob));
}

static int32_t launch_inflate_plan(spng_ctx *c, InflatePlan &p, Arena &a, spng_result *dr)
{
    const uint32_t n = (uint32_t)p.jobs.size();
    if (p.gzip)   // (slots behind the uploaded part of the arena: the header kernel fills them)
        HIP_TRY(launch_gzip_pre(a.dev<InflateJob>(p.jobs_at), p.parallel ? a.dev<PStream>(p.streams_at) : nullptr, dr,
                                a.dev<uint64_t>(p.gz_at), a.dev<int32_t>(p.done_at), n, c->stream));
    if (p.parallel) {
        Timed whole(c, SPNG_K_PINFLATE);
        PStream *ds = a.dev<PStream>(p.streams_at);
        PSeg *dg = a.dev<PSeg>(p.segs_at);
        int32_t *dd = a.dev<int32_t>(p.done_at);
        uint32_t *dnext = a.dev<uint32_t>(p.next_at);
        // every group of streams in turn, then once more those whose segments found the pool empty (a batch unlike the
        // one the pool was sized by): they get a second pass instead of the serial kernel.  Two overlapping groups: the
        // first on the context's stream, the second on stream2 with its decode held back until the first one's is done
        // (the two decodes side by side would only share the issue slots; decode beside resolve fills what either leaves).
        if (p.overlap && !c->stream2) {
            HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            for (hipEvent_t *e : {&c->ev_fork, &c->ev_mid, &c->ev_join}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        if (p.overlap) {
            HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        }
        for (size_t gi = 0; gi <= p.groups.size(); ++gi) {
            const bool retry = gi == p.groups.size();
            const InflatePlan::Group g = retry ? InflatePlan::Group{0, n, 0, (uint32_t)p.segs.size(), 0, p.pool_pages} : p.groups[gi];
            const bool second = p.overlap && gi == 1;
            hipStream_t q = second ? c->stream2 : c->stream;
            uint32_t *ctr = dnext + 4 * (second ? 1 : 0);       // {page counter} of the pass; dnext[8 ..] = the batch's totals
            uint8_t *pool = (uint8_t *)c->d_tok + ((uint64_t)g.page0 << 16);
            if (p.overlap && retry) {
                HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            }
            if (gi && !second) HIP_TRY(hipMemsetAsync(ctr, 0, 4, q));
            { Timed t(c, SPNG_K_PINF_FIND, q); HIP_TRY(launch_pinf2_find(ds, dg, g.g0, g.g1 - g.g0, retry, q)); }
            if (second) HIP_TRY(hipStreamWaitEvent(q, c->ev_mid, 0));
            { Timed t(c, SPNG_K_PINF_DECODE, q); HIP_TRY(launch_pinf2_decode(ds, dg, g.g0, g.g1 - g.g0, (uint32_t *)c->d_log, pool, ctr, g.pages, retry, q)); }
            if (p.overlap && gi == 0) HIP_TRY(hipEventRecord(c->ev_mid, q));
            PPart *dparts = p.pmax ? a.dev<PPart>(p.parts_at) + (size_t)g.s0 * p.pmax : nullptr;
            const uint32_t pm = retry ? 0u : p.pmax;                 // (the retry pass: one workgroup per stream)
            HIP_TRY(launch_pinf2_scan(ds + g.s0, g.s1 - g.s0, dg, dparts, retry, q));
            {
                Timed t(c, SPNG_K_PINF_RESOLVE, q);
                // (the parts behind the first on the second stream, beside the first parts: 79 KB -- 93 KB when they have a CU each -- and 61 KB of LDS share a CU)
                hipStream_t q2 = q;
                if (pm && !p.overlap) {
                    if (!c->stream2) {
                        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                        for (hipEvent_t *e : {&c->ev_fork, &c->ev_mid, &c->ev_join}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
                    }
                    q2 = c->stream2;
                    HIP_TRY(hipEventRecord(c->ev_fork, q));
                    HIP_TRY(hipStreamWaitEvent(q2, c->ev_fork, 0));
                }
                if (pm) HIP_TRY(launch_pinf2_parts(ds + g.s0, g.s1 - g.s0, dg, (uint32_t *)c->d_log, pool, g.pages, dr, dd + g.s0, dparts, pm,
                                                   (uint16_t *)c->d_sym, q2));
                HIP_TRY(launch_pinf2_resolve(ds + g.s0, g.s1 - g.s0, dg, (uint32_t *)c->d_log, pool, g.pages, dr, dd + g.s0, dparts, pm, retry, q));
                if (pm) {
                    if (q2 != q) { HIP_TRY(hipEventRecord(c->ev_join, q2)); HIP_TRY(hipStreamWaitEvent(q, c->ev_join, 0)); }
                    HIP_TRY(launch_pinf2_join(ds + g.s0, g.s1 - g.s0, dr, dd + g.s0, dparts, pm, (uint16_t *)c->d_sym,
                                              (uint8_t *)c->d_win + (size_t)g.s0 * pm * 32768, q));
                }
            }
            HIP_TRY(launch_pinf2_account(ctr, dnext + 8, g.pages, q));
        }
        if (!c->pool_pending) {
            // pages this batch took: read at the start of the next one (never waited for)
            HIP_TRY(hipMemcpyAsync(c->h_pool_used, dnext + 8, 16, hipMemcpyDeviceToHost, c->stream));
            c->pool_src_pending = c->pool_src_bytes;
            HIP_TRY(hipEventRecord(c->pool_ev, c->stream));
            c->pool_pending = true;
        }
    }
    if (p.parallel && getenv("SPNG_TRACE_PINFLATE")) {
        // diagnostic: how far every stream got in the pipeline (synchronises; never on by default)
        std::vector<PStream> hs(n);
        std::vector<PSeg> hg(p.segs.size());
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(hs.data(), a.dev<PStream>(p.streams_at), n * sizeof(PStream), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hg.data(), a.dev<PSeg>(p.segs_at), hg.size() * sizeof(PSeg), hipMemcpyDeviceToHost));
        {
            // anomalies over the whole batch: segments without a start, segments that did not end on the next one
            uint64_t nostart = 0, fail = 0, skipped = 0; uint32_t shown = 0;
            for (uint32_t i = 0; i < n; ++i) {
                const PStream &st = hs[i];
                for (uint32_t k = 0; k < st.seg_count; ++k) {
                    const PSeg &sg = hg[st.seg_first + k];
                    const bool a = sg.start_bit == ~0ull, b = !a && sg.status != PSEG_CONT && sg.status != PSEG_FINAL, cskip = !a && !b && sg.status == PSEG_CONT && sg.next != k + 1;
                    nostart += a; fail += b; skipped += cskip;
                    if ((a || b || cskip) && shown < 12) {
                        ++shown;
                        fprintf(stderr, "[pinflate] anomaly: stream %u seg %u/%u start %lld end %lld status %d next %u ntok %llu\n", i, k, st.seg_count,
                                (long long)sg.start_bit, (long long)sg.end_bit, sg.status, sg.next, (unsigned long long)sg.ntok);
                    }
                }
            }
            fprintf(stderr, "[pinflate] %u streams, %zu segments in %zu groups, pool %u pages: %llu without a start, %llu failed, %llu ran past the next start\n", n,
                    hg.size(), p.groups.size(), p.pool_pages, (unsigned long long)nostart, (unsigned long long)fail, (unsigned long long)skipped);
        }
        for (uint32_t i = 0; i < n && i < 4; ++i) {
            const PStream &st = hs[i];
            fprintf(stderr, "[pinflate] stream %u: len %llu segs %u seg_bytes %llu ok %d pass %u ntok %llu end_bit %llu\n", i,
                    (unsigned long long)st.src_len, st.seg_count, (unsigned long long)st.seg_bytes, st.ok, st.pass,
                    (unsigned long long)st.ntok, (unsigned long long)st.end_bit);
            for (uint32_t k = 0; k < st.seg_count && k < 12; ++k) {
                const PSeg &sg = hg[st.seg_first + k];
                fprintf(stderr, "   seg %u: start %lld end %lld status %d used %u ntok %llu tok_base %llu\n", k,
                        (long long)sg.start_bit, (long long)sg.end_bit, sg.status, sg.used, (unsigned long long)sg.ntok,
                        (unsigned long long)sg.tok_base);
            }
        }
    }
    {
        Timed t(c, SPNG_K_INFLATE);
        HIP_TRY(launch_inflate(a.dev<InflateJob>(p.jobs_at), n, dr, c->stream));
    }
    if (p.gzip)
        HIP_TRY(launch_gzip_inflate_post(a.dev<InflateJob>(p.jobs_at), dr, a.dev<uint64_t>(p.gz_at), a.dev<uint32_t>(p.gzparts_at), n,
                                         c->stream));
    if (!p.state.empty())   // the zlib checksum of a stream that finished in this call, over all of its bytes
        HIP_TRY(launch_resume_post(a.dev<InflateJob>(p.jobs_at), dr, a.dev<uint64_t>(p.sumparts_at), n, c->stream));
    return SPNG_DONE;
}

static const uint64_t *rows_len_in_results(void *user, uint32_t i)
{
    return &((spng_result *)user)[i].written;
}
static const uint64_t *rows_len_in_array(void *user, uint32_t i)
{
    return user ? (const uint64_t *)user + i : nullptr;
}

}  // namespace spng

extern "C" {

static int32_t inflate_batch(spng_ctx *c, const spng_stream_desc *descs, const uint64_t *h_state, bool resume, uint32_t count,
                             spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count) || (!d_results && !h_results && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    InflatePlan plan;
    plan.jobs.resize(count);
    if (resume) plan.state.assign((size_t)count * 4, 0);
    for (uint32_t i = 0; i < count; ++i) {
        if ((!descs[i].d_src && descs[i].src_len) || (!descs[i].d_dst && descs[i].dst_cap) || descs[i].format < SPNG_FORMAT_ZLIB ||
            descs[i].format > SPNG_FORMAT_GZIP) return SPNG_E_ARGUMENT;
        plan.jobs[i] = InflateJob{(const uint8_t *)descs[i].d_src, (uint8_t *)descs[i].d_dst,
                                  descs[i].src_len, descs[i].dst_cap, descs[i].format, i, nullptr, nullptr, 0, 0};
        if (resume && h_state) {
            const uint64_t *hs = h_state + 4 * (size_t)i;
            for (int k = 0; k < 4; ++k) plan.state[4 * (size_t)i + k] = hs[k];
            // (a state is only ever what an earlier call handed out: inside the input and the output, the token behind its block's
            // header, its bytes behind the block's)
            if (hs[0] > descs[i].src_len * 8 || hs[1] > descs[i].dst_cap || hs[2] > descs[i].src_len * 8 || hs[3] > descs[i].dst_cap ||
                (hs[2] && (hs[2] <= hs[0] || hs[3] < hs[1])) || (!hs[2] && hs[3])) return SPNG_E_ARGUMENT;
        }
    }
    if (int32_t st = plan_inflate(c, plan)) return st;
    if (int32_t st = c->reserve(plan.bytes() + count * sizeof(spng_result) + 1024)) return st;
    Arena a{c};
    stage_inflate(plan, a);
    const size_t upload = a.off;
    const size_t res = a.take(count * sizeof(spng_result));
    if (plan.gzip) { plan.gz_at = a.take(count * 8); plan.gzparts_at = a.take((size_t)count * 4 * gzip_pieces()); }
    plan.sumparts_at = a.take((size_t)count * 8 * gzip_pieces());
    if (int32_t st = c->upload(0, upload)) return st;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    poison_results_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(dr, count);
    HIP_TRY(hipGetLastError());
    if (int32_t st = launch_inflate_plan(c, plan, a, dr)) return st;
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_inflate_batch(spng_ctx *c, const spng_stream_desc *descs, uint32_t count,
                           spng_result *d_results, spng_result *h_results)
{
    return inflate_batch(c, descs, nullptr, false, count, d_results, h_results);
}

int32_t spng_inflate_resume_batch(spng_ctx *c, const spng_stream_desc *descs, const uint64_t *h_state, uint32_t count,
                                  spng_result *d_results, spng_result *h_results)
{
    return inflate_batch(c, descs, h_state, true, count, d_results, h_results);
}

int32_t spng_unfilter_batch(spng_ctx *c, const spng_image_desc *descs, uint32_t count,
                            const uint64_t *d_rows_len,
                            spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    UnfilterPlan plan;
    if (int32_t st = plan_unfilter(descs, count, plan, rows_len_in_array, (void *)d_rows_len)) return st;
    const size_t need = plan_bytes(plan) + count * (sizeof(spng_result) + 16) + 1024;
    if (int32_t st = c->reserve(need)) return st;
    Arena a{c};
    PlanSlots slots;
    stage_plan(plan, a, slots);
    const size_t expected = a.take(count * 8);
    for (uint32_t i = 0; i < count; ++i)
        a.host<uint64_t>(expected)[i] = spng_inflated_size(descs[i].width, descs[i].height, descs[i].depth,
                                                           descs[i].channels, descs[i].interlaced);
    const size_t upload = a.off;
    const size_t res = a.take(count * sizeof(spng_result));
    if (int32_t st = c->upload(0, upload)) return st;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    // results: status DONE, written = rows_len (or U); then extraneous check
    init_results_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(
        dr, d_rows_len ? d_rows_len : a.dev<uint64_t>(expected), count);
    HIP_TRY(hipGetLastError());
    if (int32_t st = launch_plan(c, plan, a, slots, nullptr)) return st;
    finish_decode_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(dr, a.dev<uint64_t>(expected), count);
    HIP_TRY(hipGetLastError());
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_unfilter_resume_batch(spng_ctx *c, const spng_image_desc *descs, void *const *d_work, const uint64_t *h_prev_len,
                                   const uint64_t *h_now_len, uint32_t count, spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count) || !h_prev_len || !h_now_len) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    UnfilterPlan plan;
    std::vector<spng_result> res(count);
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        if (!valid_format(d.depth, d.channels) || !d.d_rows || !d.d_storage || h_prev_len[i] > h_now_len[i] ||
            (d.reserved & ~(uint32_t)SPNG_IMAGE_OVERDRAW)) return SPNG_E_ARGUMENT;   // (unknown flag bits: an uninitialised desc)
        const int volume = d.depth * d.channels;
        const uint32_t bpp = (uint32_t)(volume + 7) >> 3;
        const uint64_t u = spng_inflated_size(d.width, d.height, d.depth, d.channels, d.interlaced);
        if (d.rows_cap < u) return SPNG_E_ARGUMENT;
        const bool direct = !d.interlaced && volume >= 8;       // rows land in storage as they are
        if (!direct && (!d_work || !d_work[i])) return SPNG_E_ARGUMENT;
        Pass p[7];
        const int np = passes(d.width, d.height, volume, d.interlaced, p);
        uint64_t off = 0, fresh = 0, upto = 0;
        OverdrawJob ov;
        memset(&ov, 0, sizeof ov);
        ov.y0 = d.height; ov.y1 = 0;
        for (int z = 0; z < np; ++z) {
            const uint64_t stride = p[z].pitch + 1, end = off + stride * p[z].h;
            // rows of this sub-image complete before / with this push (PNG.Decoder.row / pass, PNG.Decoder.swift:20-21, 88-94)
            const uint64_t r0 = h_prev_len[i] <= off ? 0 : (h_prev_len[i] >= end ? p[z].h : (h_prev_len[i] - off) / stride);
            const uint64_t r1 = h_now_len[i] <= off ? 0 : (h_now_len[i] >= end ? p[z].h : (h_now_len[i] - off) / stride);
            upto = off + r1 * stride > upto && r1 ? off + r1 * stride : upto;
            if (r1 > r0) {
                UnfJob j;
                j.in = (const uint8_t *)d.d_rows + off + r0 * stride;
                j.in_stride = stride;
                if (direct) { j.out = (uint8_t *)d.d_storage + r0 * p[z].pitch; j.out_stride = p[z].pitch; }
                else        { j.out = (uint8_t *)d_work[i] + off + r0 * stride + 1; j.out_stride = stride; }
                j.stream_off = off; j.rows_len = nullptr;
                j.pitch = (uint32_t)p[z].pitch; j.rows = (uint32_t)(r1 - r0); j.image = i; j.bpp = bpp;
                j.has_prev = r0 ? 1 : 0; j.pad = 0;
                plan.unf[bpp].push_back(j);
                if (!direct) {
                    ScatterJob s;
                    s.rows = (const uint8_t *)d_work[i] + off + r0 * stride + 1;
                    s.storage = (uint8_t *)d.d_storage;
                    s.row_stride = stride; s.stream_off = off; s.rows_len = nullptr;
                    s.sub_w = p[z].w; s.sub_h = (uint32_t)(r1 - r0); s.width = d.width;
                    s.bx = p[z].bx; s.by = p[z].by + (uint32_t)r0 * p[z].sy; s.sx = p[z].sx; s.sy = p[z].sy;
                    s.depth = d.depth; s.channels = d.channels;
                    plan.scat.push_back(s);
                    plan.scat_image.push_back(i);
                }
                fresh += (r1 - r0) * stride;
            }
            if (d.interlaced && (d.reserved & SPNG_IMAGE_OVERDRAW)) {
                // (the pass index of PNG.adam7 from the pass's base; rows [first new scanline, last new scanline + its stride))
                const int q = p[z].sy == 8 ? (p[z].by ? 2 : p[z].bx ? 1 : 0) : p[z].sy == 4 ? (p[z].by ? 4 : 3) : (p[z].by ? 6 : 5);
                ov.done[q] = (uint32_t)r1;
                if (r1 > r0) {
                    const uint32_t a0 = p[z].by + (uint32_t)r0 * p[z].sy, a1 = p[z].by + (uint32_t)r1 * p[z].sy;
                    ov.y0 = a0 < ov.y0 ? a0 : ov.y0;
                    ov.y1 = a1 > ov.y1 ? a1 : ov.y1;
                }
            }
            off = end;
        }
        if (ov.y1 > ov.y0) {
            ov.storage = (uint8_t *)d.d_storage; ov.width = d.width; ov.height = d.height;
            ov.elem = volume < 8 ? 1u : (uint32_t)volume >> 3;
            if (ov.y1 > d.height) ov.y1 = d.height;
            plan.over.push_back(ov);
        }
        res[i].status = SPNG_DONE; res[i].reserved = 0;
        res[i].written = fresh;                                // scanline bytes defiltered by THIS call
        res[i].consumed = upto;                                // inflated bytes that are whole rows by now
        res[i].aux[0] = res[i].aux[1] = 0;
    }
    const size_t need = plan_bytes(plan) + count * sizeof(spng_result) + 1024;
    if (int32_t st = c->reserve(need)) return st;
    Arena a{c};
    PlanSlots slots;
    stage_plan(plan, a, slots);
    const size_t rslot = a.take(count * sizeof(spng_result));
    memcpy(a.host<spng_result>(rslot), res.data(), count * sizeof(spng_result));
    if (int32_t st = c->upload(0, a.off)) return st;
    if (int32_t st = launch_plan(c, plan, a, slots, nullptr)) return st;
    if (d_results) HIP_TRY(hipMemcpyAsync(d_results, a.dev<spng_result>(rslot), count * sizeof(spng_result), hipMemcpyDeviceToDevice, c->stream));
    if (h_results) {
        memcpy(h_results, res.data(), count * sizeof(spng_result));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_decode_batch(spng_ctx *c, const spng_image_desc *descs, uint32_t count,
                          spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    // results live on the device from the start: the unfilter jobs read `written` from them
    const size_t res_bytes = count * sizeof(spng_result);
    UnfilterPlan plan;
    InflatePlan ip;
    ip.jobs.resize(count);
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        if ((!d.d_idat && d.idat_len) || (d.format != SPNG_FORMAT_ZLIB && d.format != SPNG_FORMAT_IOS)) return SPNG_E_ARGUMENT;
        ip.jobs[i] = InflateJob{(const uint8_t *)d.d_idat, (uint8_t *)d.d_rows, d.idat_len, d.rows_cap, d.format, i, nullptr, nullptr, 0, 0};
    }
    if (int32_t st = plan_inflate(c, ip)) return st;
    // two-phase: we need the device address of the results before planning
    const size_t fixed = ip.bytes() + count * 8 + res_bytes + 2048;
    // conservative upper bound on plan size: 7 passes per image
    if (int32_t st = c->reserve(fixed + (size_t)count * 7 * (sizeof(UnfJob) + sizeof(ScatterJob) + 4) + 8192)) return st;
    Arena a{c};
    const size_t res = a.take(res_bytes);                      // first, so its device address is stable
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    if (int32_t st = plan_unfilter(descs, count, plan, rows_len_in_results, (void *)dr)) return st;
    stage_inflate(ip, a);
    const size_t expected = a.take(count * 8);
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        a.host<uint64_t>(expected)[i] = spng_inflated_size(d.width, d.height, d.depth, d.channels, d.interlaced);
    }
    PlanSlots slots;
    stage_plan(plan, a, slots);
    // upload everything except the results region at the front
    const size_t first = (res_bytes + 255) & ~(size_t)255;
    const size_t staged = a.off;
    ip.sumparts_at = a.take((size_t)count * 8 * gzip_pieces());
    if (int32_t st = c->upload(first, staged)) return st;
    poison_results_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(dr, count);
    HIP_TRY(hipGetLastError());
    if (int32_t st = launch_inflate_plan(c, ip, a, dr)) return st;
    if (int32_t st = launch_plan(c, plan, a, slots, dr)) return st;
    finish_decode_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(dr, a.dev<uint64_t>(expected), count);
    HIP_TRY(hipGetLastError());
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, res_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

// ---- host-pointer convenience wrappers ---------------------------------------------------------

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
};

int32_t spng_inflate(spng_ctx *c, const void *src, uint64_t n, int32_t format,
                     void *dst, uint64_t cap, spng_result *result)
{
    if (!c || (!src && n) || (!dst && cap) || !result) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf ds, dd;
    HIP_TRY(ds.alloc(n)); HIP_TRY(dd.alloc(cap));
    HIP_TRY(hipMemcpyAsync(ds.p, src, n, hipMemcpyHostToDevice, c->stream));
    spng_stream_desc d{ds.p, n, dd.p, cap, format, 0};
    if (int32_t st = spng_inflate_batch(c, &d, 1, nullptr, result)) return st;
    const uint64_t w = result->written < cap ? result->written : cap;
    if (w) HIP_TRY(hipMemcpy(dst, dd.p, w, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}

int32_t spng_unfilter(spng_ctx *c, const void *rows, uint64_t rows_len,
                      uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                      void *storage, spng_result *result)
{
    if (!c || (!rows && rows_len) || !storage || !result || !valid_format(depth, channels)) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t u = spng_inflated_size(w, h, depth, channels, interlaced);
    const uint64_t s = spng_storage_size(w, h, depth, channels);
    const uint64_t take = rows_len < u ? rows_len : u;
    DevBuf dr, dst, dl;
    HIP_TRY(dr.alloc(u)); HIP_TRY(dst.alloc(s)); HIP_TRY(dl.alloc(8));
    HIP_TRY(hipMemcpyAsync(dr.p, rows, take, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dst.p, storage, s, hipMemcpyHostToDevice, c->stream));   // keep undecoded rows
    HIP_TRY(hipMemcpyAsync(dl.p, &rows_len, 8, hipMemcpyHostToDevice, c->stream));
    spng_image_desc d{};
    d.d_rows = dr.p; d.rows_cap = u; d.d_storage = dst.p; d.width = w; d.height = h;
    d.depth = (uint8_t)depth; d.channels = (uint8_t)channels; d.interlaced = (uint8_t)(interlaced != 0);
    if (int32_t st = spng_unfilter_batch(c, &d, 1, (const uint64_t *)dl.p, nullptr, result)) return st;
    HIP_TRY(hipMemcpy(storage, dst.p, s, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}

int32_t spng_decode(spng_ctx *c, const void *idat, uint64_t n, int32_t format,
                    uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                    void *storage, spng_result *result)
{
    if (!c || (!idat && n) || !storage || !result || !valid_format(depth, channels)) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t u = spng_inflated_size(w, h, depth, channels, interlaced);
    const uint64_t s = spng_storage_size(w, h, depth, channels);
    DevBuf di, dr, dst;
    HIP_TRY(di.alloc(n)); HIP_TRY(dr.alloc(u + 4096)); HIP_TRY(dst.alloc(s));
    HIP_TRY(hipMemcpyAsync(di.p, idat, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dst.p, storage, s, hipMemcpyHostToDevice, c->stream));
    spng_image_desc d{};
    d.d_idat = di.p; d.idat_len = n; d.d_rows = dr.p; d.rows_cap = u + 4096; d.d_storage = dst.p;
    d.width = w; d.height = h; d.depth = (uint8_t)depth; d.channels = (uint8_t)channels;
    d.interlaced = (uint8_t)(interlaced != 0); d.format = (uint8_t)format;
    if (int32_t st = spng_decode_batch(c, &d, 1, nullptr, result)) return st;
    HIP_TRY(hipMemcpy(storage, dst.p, s, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}


int32_t spng_adler32(spng_ctx *c, const void *data, uint64_t n, uint32_t *out)
{
    if (!c || (!data && n) || !out) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t chunk = 1u << 16;
    const uint32_t blocks = (uint32_t)((n + chunk - 1) / chunk);
    uint32_t s1 = 1, s2 = 0;
    if (blocks) {
        DevBuf dd, dp;
        HIP_TRY(dd.alloc(n)); HIP_TRY(dp.alloc((size_t)blocks * 16));
        HIP_TRY(hipMemcpyAsync(dd.p, data, n, hipMemcpyHostToDevice, c->stream));
        {
            Timed t(c, SPNG_K_ADLER);
            HIP_TRY(launch_adler_partial((const uint8_t *)dd.p, n, chunk, (uint64_t *)dp.p, blocks, c->stream));
        }
        std::vector<uint64_t> part((size_t)blocks * 2);
        HIP_TRY(hipMemcpyAsync(part.data(), dp.p, (size_t)blocks * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (uint32_t i = 0; i < blocks; ++i) {
            const uint64_t len = n - (uint64_t)i * chunk < chunk ? n - (uint64_t)i * chunk : chunk;
            s2 = (uint32_t)((s2 + (len % 65521) * s1 + part[2 * i + 1] % 65521) % 65521);
            s1 = (uint32_t)((s1 + part[2 * i] % 65521) % 65521);
        }
    }
    *out = s2 << 16 | s1;
    return SPNG_DONE;
}

int32_t spng_filter_batch(spng_ctx *c, const spng_image_desc *descs, uint32_t count,
                          spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    std::vector<FilterJob> jobs;
    std::vector<uint64_t> sizes(count);
    uint32_t max_rows = 1;
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        if (!valid_format(d.depth, d.channels) || !d.d_rows || !d.d_storage) return SPNG_E_ARGUMENT;
        const int volume = d.depth * d.channels;
        sizes[i] = spng_inflated_size(d.width, d.height, d.depth, d.channels, d.interlaced);
        if (d.rows_cap < sizes[i]) return SPNG_E_ARGUMENT;
        Pass p[7];
        const int np = passes(d.width, d.height, volume, d.interlaced, p);
        uint64_t off = 0;
        for (int z = 0; z < np; ++z) {
            FilterJob j;
            j.storage = (const uint8_t *)d.d_storage;
            j.rows = (uint8_t *)d.d_rows + off;
            j.row_stride = p[z].pitch + 1;
            j.sub_w = p[z].w; j.sub_h = p[z].h; j.width = d.width;
            j.bx = p[z].bx; j.by = p[z].by; j.sx = p[z].sx; j.sy = p[z].sy;
            j.depth = d.depth; j.channels = d.channels; j.pitch = (uint32_t)p[z].pitch;
            jobs.push_back(j);
            if (p[z].h > max_rows) max_rows = p[z].h;
            off += (p[z].pitch + 1) * (uint64_t)p[z].h;
        }
    }
    const size_t need = jobs.size() * sizeof(FilterJob) + count * (sizeof(spng_result) + 8) + 2048;
    if (int32_t st = c->reserve(need)) return st;
    Arena a{c};
    const size_t jslot = a.take(jobs.size() * sizeof(FilterJob));
    const size_t wslot = a.take(count * 8);
    if (!jobs.empty()) memcpy(a.host<FilterJob>(jslot), jobs.data(), jobs.size() * sizeof(FilterJob));
    memcpy(a.host<uint64_t>(wslot), sizes.data(), count * 8);
    const size_t upload = a.off;
    const size_t res = a.take(count * sizeof(spng_result));
    if (int32_t st = c->upload(0, upload)) return st;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    {
        Timed t(c, SPNG_K_FILTER);
        HIP_TRY(launch_filter(a.dev<FilterJob>(jslot), (uint32_t)jobs.size(), max_rows, c->stream));
    }
    init_results_kernel<<<(count + 255) / 256, 256, 0, c->stream>>>(dr, a.dev<uint64_t>(wslot), count);
    HIP_TRY(hipGetLastError());
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_filter(spng_ctx *c, const void *storage,
                    uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                    void *rows, spng_result *result)
{
    if (!c || !storage || !rows || !result || !valid_format(depth, channels)) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t u = spng_inflated_size(w, h, depth, channels, interlaced);
    const uint64_t s = spng_storage_size(w, h, depth, channels);
    DevBuf dr, dst;
    HIP_TRY(dr.alloc(u)); HIP_TRY(dst.alloc(s));
    HIP_TRY(hipMemcpyAsync(dst.p, storage, s, hipMemcpyHostToDevice, c->stream));
    spng_image_desc d{};
    d.d_rows = dr.p; d.rows_cap = u; d.d_storage = dst.p; d.width = w; d.height = h;
    d.depth = (uint8_t)depth; d.channels = (uint8_t)channels; d.interlaced = (uint8_t)(interlaced != 0);
    if (int32_t st = spng_filter_batch(c, &d, 1, nullptr, result)) return st;
    HIP_TRY(hipMemcpy(rows, dr.p, u, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}


int32_t spng_lex_batch(spng_ctx *c, const spng_file_desc *files, uint32_t count, spng_lexed *d_infos, spng_lexed *h_infos)
{
    if (!c || (!files && count) || (!d_infos && !h_infos && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    // the chunk lists: a file of `len` bytes holds at most len / 12 chunks; 64 KiB IDATs are the usual case, so a list of
    // len / 2048 + 64 entries (at most 8192) is plenty, and a file with more is finished by its own wave
    std::vector<uint64_t> at(count + 1, 0);
    uint32_t max_listed = 1;
    for (uint32_t i = 0; i < count; ++i) {
        if ((!files[i].d_png && files[i].len) || (!files[i].d_idat && files[i].idat_cap)) return SPNG_E_ARGUMENT;
        uint64_t k = files[i].len / 2048 + 64;
        if (k > files[i].len / 12 + 1) k = files[i].len / 12 + 1;
        if (k > 8192) k = 8192;
        at[i + 1] = at[i] + k;
        max_listed = k > max_listed ? (uint32_t)k : max_listed;
    }
    const size_t table_bytes = (size_t)at[count] * lex_chunk_bytes();
    if (int32_t st = c->reserve(count * (sizeof(spng_file_desc) + sizeof(spng_lexed) + 8 + lex_walk_bytes()) + table_bytes + 2048)) return st;
    Arena a{c};
    const size_t fslot = a.take(count * sizeof(spng_file_desc));
    for (uint32_t i = 0; i < count; ++i) a.host<spng_file_desc>(fslot)[i] = files[i];
    const size_t atslot = a.take((count + 1) * 8);
    memcpy(a.host<uint64_t>(atslot), at.data(), (count + 1) * 8);
    const size_t upload = a.off;
    const size_t oslot = a.take(count * sizeof(spng_lexed));
    const size_t wslot = a.take(count * lex_walk_bytes());
    const size_t tslot = a.take(table_bytes);
    if (int32_t st = c->upload(0, upload)) return st;
    spng_lexed *dout = d_infos ? d_infos : a.dev<spng_lexed>(oslot);
    {
        Timed t(c, SPNG_K_LEX);
        HIP_TRY(launch_lex(a.dev<spng_file_desc>(fslot), count, dout, a.dev<uint8_t>(tslot), a.dev<uint64_t>(atslot), a.dev<uint8_t>(wslot),
                           max_listed, c->stream));
    }
    if (h_infos) {
        HIP_TRY(hipMemcpyAsync(h_infos, dout, count * sizeof(spng_lexed), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_write_idat_batch(spng_ctx *c, const spng_chunking_desc *descs, uint32_t count,
                              spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count) || (!d_results && !h_results && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    if (int32_t st = c->reserve(count * (sizeof(spng_chunking_desc) + sizeof(spng_result)) + 1024)) return st;
    Arena a{c};
    const size_t dslot = a.take(count * sizeof(spng_chunking_desc));
    uint64_t most = 1;
    for (uint32_t i = 0; i < count; ++i) {
        if ((!descs[i].d_stream && descs[i].len) || !descs[i].d_out || !descs[i].chunk_bytes ||
            descs[i].chunk_bytes > 0x7fffffffull) return SPNG_E_ARGUMENT;
        a.host<spng_chunking_desc>(dslot)[i] = descs[i];
        const uint64_t pieces = (descs[i].len + descs[i].chunk_bytes - 1) / descs[i].chunk_bytes;
        most = pieces > most ? pieces : most;
    }
    const size_t upload = a.off;
    const size_t rslot = a.take(count * sizeof(spng_result));
    if (int32_t st = c->upload(0, upload)) return st;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(rslot);
    { Timed t(c, SPNG_K_LEX); HIP_TRY(launch_write_idat(a.dev<spng_chunking_desc>(dslot), count, (uint32_t)(most > 4096 ? 4096 : most), dr, c->stream)); }
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_crc32(spng_ctx *c, const void *data, uint64_t n, uint32_t *out)
{
    if (!c || (!data && n) || !out) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t piece = 1u << 20;
    const uint32_t pieces = (uint32_t)((n + piece - 1) / piece);
    std::vector<uint32_t> part(pieces ? pieces : 1);
    if (pieces) {
        DevBuf dd, dp;
        HIP_TRY(dd.alloc(n)); HIP_TRY(dp.alloc((size_t)pieces * 4));
        HIP_TRY(hipMemcpyAsync(dd.p, data, n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_crc_partial((const uint8_t *)dd.p, n, piece, (uint32_t *)dp.p, pieces, c->stream));
        HIP_TRY(hipMemcpyAsync(part.data(), dp.p, (size_t)pieces * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *out = crc32_fold(part.data(), pieces, n, piece);
    return SPNG_DONE;
}

int32_t spng_unpack_batch(spng_ctx *c, const spng_unpack_desc *descs, uint32_t count)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    const int target = descs[0].target;
    if (target != 8 && target != 16) return SPNG_E_ARGUMENT;
    if (int32_t st = c->reserve(count * sizeof(UnpackJob) + 1024)) return st;
    Arena a{c};
    const size_t jslot = a.take(count * sizeof(UnpackJob));
    uint64_t maxpix = 1;
    for (uint32_t i = 0; i < count; ++i) {
        const spng_unpack_desc &d = descs[i];
        if (!valid_format(d.depth, d.channels) || !d.d_storage || !d.d_out || d.target != target ||
            (d.indexed && (d.channels != 1 || d.depth > 8 || (!d.d_palette && d.palette_count))) ||
            d.layout > SPNG_TARGET_SCALAR || d.premultiply > SPNG_PREMULTIPLY_AS_U8 || (d.premultiply == SPNG_PREMULTIPLY_AS_U8 && target != 16) ||
            (d.layout == SPNG_TARGET_SCALAR && d.premultiply))
            return SPNG_E_ARGUMENT;
        UnpackJob j;
        memset(&j, 0, sizeof j);
        j.storage = (const uint8_t *)d.d_storage; j.out = d.d_out; j.palette = (const uint8_t *)d.d_palette;
        j.width = d.width; j.height = d.height; j.palette_count = d.palette_count;
        j.key[0] = d.key[0]; j.key[1] = d.key[1]; j.key[2] = d.key[2];
        j.depth = d.depth; j.channels = d.channels; j.indexed = d.indexed; j.bgr = d.bgr; j.has_key = d.has_key;
        j.layout = d.layout; j.premultiply = d.premultiply;
        a.host<UnpackJob>(jslot)[i] = j;
        const uint64_t px = (uint64_t)d.width * d.height;
        maxpix = px > maxpix ? px : maxpix;
    }
    if (int32_t st = c->upload(0, a.off)) return st;
    uint64_t bx = (maxpix + 4095) / 4096;                      // (four pixels per thread, 256 threads, a few rounds)
    if (bx > 4096) bx = 4096;
    Timed t(c, SPNG_K_UNPACK);
    HIP_TRY(launch_unpack(a.dev<UnpackJob>(jslot), count, (uint32_t)bx, target, c->stream));
    return SPNG_DONE;
}

int32_t spng_unpack_as(spng_ctx *c, const void *storage, uint32_t w, uint32_t h, int depth, int channels,
                       int indexed, int bgr, int target, int layout, int premultiply, const void *palette,
                       uint32_t palette_count, const uint16_t *key, void *out)
{
    if (!c || !storage || !out || !valid_format(depth, channels) || (target != 8 && target != 16)) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t s = spng_storage_size(w, h, depth, channels),
                   o = (uint64_t)w * h * (layout == SPNG_TARGET_VA ? 2 : layout == SPNG_TARGET_SCALAR ? 1 : 4) * (target / 8);
    DevBuf ds, dout, dp;
    HIP_TRY(ds.alloc(s)); HIP_TRY(dout.alloc(o)); HIP_TRY(dp.alloc((size_t)palette_count * 4));
    HIP_TRY(hipMemcpyAsync(ds.p, storage, s, hipMemcpyHostToDevice, c->stream));
    if (palette_count) HIP_TRY(hipMemcpyAsync(dp.p, palette, (size_t)palette_count * 4, hipMemcpyHostToDevice, c->stream));
    spng_unpack_desc d{};
    d.d_storage = ds.p; d.d_out = dout.p; d.d_palette = palette_count ? dp.p : nullptr;
    d.width = w; d.height = h; d.palette_count = palette_count;
    if (key) { d.key[0] = key[0]; d.key[1] = key[1]; d.key[2] = key[2]; d.has_key = 1; }
    d.depth = (uint8_t)depth; d.channels = (uint8_t)channels; d.indexed = (uint8_t)(indexed != 0); d.bgr = (uint8_t)(bgr != 0);
    d.target = (uint8_t)target; d.layout = (uint8_t)layout; d.premultiply = (uint8_t)premultiply;
    if (int32_t st = spng_unpack_batch(c, &d, 1)) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (o) HIP_TRY(hipMemcpy(out, dout.p, o, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}

int32_t spng_unpack(spng_ctx *c, const void *storage, uint32_t w, uint32_t h, int depth, int channels,
                    int indexed, int bgr, int target, const void *palette, uint32_t palette_count,
                    const uint16_t *key, void *out)
{
    return spng_unpack_as(c, storage, w, h, depth, channels, indexed, bgr, target, SPNG_TARGET_RGBA, 0, palette, palette_count, key, out);
}

int32_t spng_pack_batch(spng_ctx *c, const spng_pack_desc *descs, uint32_t count)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    const int source = descs[0].source;
    if (source != 8 && source != 16) return SPNG_E_ARGUMENT;
    if (int32_t st = c->reserve(count * sizeof(PackJob) + 1024)) return st;
    Arena a{c};
    const size_t jslot = a.take(count * sizeof(PackJob));
    uint64_t maxpix = 1;
    for (uint32_t i = 0; i < count; ++i) {
        const spng_pack_desc &d = descs[i];
        if (!valid_format(d.depth, d.channels) || !d.d_storage || !d.d_pixels || d.source != source ||
            (d.indexed && (d.channels != 1 || d.depth > 8 || (!d.d_palette && d.palette_count) || d.palette_count > 256)) ||
            d.layout > SPNG_TARGET_SCALAR || ((uintptr_t)d.d_pixels & (source / 8 - 1)))
            return SPNG_E_ARGUMENT;
        PackJob j;
        memset(&j, 0, sizeof j);
        j.pixels = d.d_pixels; j.storage = (uint8_t *)d.d_storage; j.palette = (const uint8_t *)d.d_palette;
        j.width = d.width; j.height = d.height; j.palette_count = d.palette_count;
        j.depth = d.depth; j.channels = d.channels; j.indexed = d.indexed; j.bgr = d.bgr; j.layout = d.layout;
        a.host<PackJob>(jslot)[i] = j;
        const uint64_t px = (uint64_t)d.width * d.height;
        maxpix = px > maxpix ? px : maxpix;
    }
    if (int32_t st = c->upload(0, a.off)) return st;
    uint64_t bx = (maxpix + 4095) / 4096;                      // (four pixels per thread, 256 threads, a few rounds)
    if (bx > 4096) bx = 4096;
    Timed t(c, SPNG_K_PACK);
    HIP_TRY(launch_pack(a.dev<PackJob>(jslot), count, (uint32_t)bx, source, c->stream));
    return SPNG_DONE;
}

int32_t spng_pack_as(spng_ctx *c, const void *pixels, uint32_t w, uint32_t h, int depth, int channels,
                     int indexed, int bgr, int source, int layout, const void *palette, uint32_t palette_count, void *storage)
{
    if (!c || !storage || !pixels || !valid_format(depth, channels) || (source != 8 && source != 16) || layout < 0 ||
        layout > SPNG_TARGET_SCALAR) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t s = spng_storage_size(w, h, depth, channels),
                   o = (uint64_t)w * h * (layout == SPNG_TARGET_VA ? 2 : layout == SPNG_TARGET_SCALAR ? 1 : 4) * (source / 8);
    DevBuf ds, dpx, dp;
    HIP_TRY(ds.alloc(s)); HIP_TRY(dpx.alloc(o)); HIP_TRY(dp.alloc((size_t)palette_count * 4));
    HIP_TRY(hipMemcpyAsync(dpx.p, pixels, o, hipMemcpyHostToDevice, c->stream));
    if (palette_count) HIP_TRY(hipMemcpyAsync(dp.p, palette, (size_t)palette_count * 4, hipMemcpyHostToDevice, c->stream));
    spng_pack_desc d{};
    d.d_pixels = dpx.p; d.d_storage = ds.p; d.d_palette = palette_count ? dp.p : nullptr;
    d.width = w; d.height = h; d.palette_count = palette_count;
    d.depth = (uint8_t)depth; d.channels = (uint8_t)channels; d.indexed = (uint8_t)(indexed != 0); d.bgr = (uint8_t)(bgr != 0);
    d.source = (uint8_t)source; d.layout = (uint8_t)layout;
    if (int32_t st = spng_pack_batch(c, &d, 1)) return st;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (s) HIP_TRY(hipMemcpy(storage, ds.p, s, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}

uint64_t spng_deflate_bound(uint64_t n) { return n + n / 4 + 4096; }   // (covers the 18 bytes of a gzip wrapper too)

// The one-kernel full search (deflate_full_kernel) for the streams in `sorted[first, last)`: per-stream graph scratch (129 bytes per
// vertex, up to 2^21 vertices) from the context's slab, in groups that fit it.  The path of streams the two-kernel search could
// not finish (its pool ran dry under them), and of SPNG_CFG_DEFLATE_MODE = SPNG_DEFLATE_ONE_KERNEL.
static int32_t deflate_full_legacy(spng_ctx *c, std::vector<DeflateJob> &sorted, size_t first, size_t last, spng_result *dr, Arena &a, size_t jslot)
{
    if (first >= last) return SPNG_DONE;
    {
        // link rings of these streams (the slab may still be read by the greedy / lazy kernel of this call: wait before it moves)
        const size_t ring_bytes = (last - first) * 65536 * 4;
        if (ring_bytes > c->ring2_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_ring2) HIP_TRY(hipFree(c->d_ring2));
            c->d_ring2 = nullptr; c->ring2_cap = 0;
            HIP_TRY(hipMalloc(&c->d_ring2, ring_bytes));
            c->ring2_cap = ring_bytes;
        }
    }
    uint64_t need = 0, largest = 0;
    for (size_t i = first; i < last; ++i) {
        DeflateJob &f = sorted[i];
        f.graph_vertices = (uint32_t)deflate_graph_vertices(f.src_len);
        const uint64_t bytes = deflate_graph_bytes(f.graph_vertices);
        need += bytes; largest = bytes > largest ? bytes : largest;
    }
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        uint64_t budget = c->cfg[SPNG_CFG_DEFLATE_BYTES] ? (uint64_t)c->cfg[SPNG_CFG_DEFLATE_BYTES] : (uint64_t)(free_b + c->graph_cap) / 2;
        if (budget < largest) budget = largest;
        uint64_t want = need < budget ? need : budget;
        while (want > c->graph_cap) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->d_graph) HIP_TRY(hipFree(c->d_graph));
            c->d_graph = nullptr; c->graph_cap = 0;
            if (hipMalloc(&c->d_graph, want) == hipSuccess) { c->graph_cap = want; break; }
            (void)hipGetLastError();
            // no room for that many streams side by side: smaller groups
            if (want <= largest) return fail_hip(hipErrorOutOfMemory, "deflate: no room for the match graph of one stream");
            want = want / 2 > largest ? want / 2 : largest;
        }
    }
    // Which full-search streams get helper waves: the ones whose input repeats itself (deflate_density_kernel).  The sparse
    // ones go first, so every launch group is of one kind.
    const size_t nfull = last - first;
    std::vector<uint32_t> dense(nfull, 0);
    {
        memcpy(a.host<DeflateJob>(jslot), sorted.data(), sorted.size() * sizeof(DeflateJob));
        if (int32_t st = c->upload(jslot, jslot + sorted.size() * sizeof(DeflateJob))) return st;
        const size_t dslot = a.take(nfull * 4);
        HIP_TRY(launch_deflate_density(a.dev<DeflateJob>(jslot) + first, (uint32_t)nfull, a.dev<uint32_t>(dslot), c->stream));
        HIP_TRY(hipMemcpyAsync(dense.data(), a.dev<uint32_t>(dslot), nfull * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::vector<DeflateJob> part;
        part.reserve(nfull);
        for (int kind = 0; kind < 2; ++kind)
            for (size_t i = 0; i < nfull; ++i) if ((int)dense[i] == kind) part.push_back(sorted[first + i]);
        size_t ndense = 0;
        for (uint32_t d : dense) ndense += d;
        std::copy(part.begin(), part.end(), sorted.begin() + first);
        for (size_t i = 0; i < nfull; ++i) dense[i] = i >= nfull - ndense;
    }
    // groups of full-search streams of one kind that fit the slab together
    struct Group { size_t first, last; bool helpers; };
    std::vector<Group> groups;
    for (size_t i = first; i < last;) {
        uint64_t used = 0;
        size_t k = i;
        const bool kind = dense[i - first] != 0;
        while (k < last && (dense[k - first] != 0) == kind) {
            const uint64_t bytes = deflate_graph_bytes(sorted[k].graph_vertices);
            if (used + bytes > c->graph_cap && k > i) break;
            sorted[k].graph = (uint32_t *)((char *)c->d_graph + used);
            sorted[k].ring = (uint32_t *)c->d_ring2 + (k - first) * 65536;
            used += bytes; ++k;
        }
        groups.push_back({i, k, kind});
        i = k;
    }
    memcpy(a.host<DeflateJob>(jslot), sorted.data(), sorted.size() * sizeof(DeflateJob));
    if (int32_t st = c->upload(jslot, jslot + sorted.size() * sizeof(DeflateJob))) return st;
    for (auto &gr : groups)
        HIP_TRY(launch_deflate_full(a.dev<DeflateJob>(jslot) + gr.first, (uint32_t)(gr.last - gr.first), gr.helpers, dr, c->stream));
    return SPNG_DONE;
}

// The two-kernel full search (deflate.hip, "round 4") for sorted[first, last): per stream 11 bytes of scratch per vertex of a
// round (<= 2^21 vertices), a pool of candidate words shared by all streams, rings for the search workgroups -- all from the
// context's slab, which is capped at SPNG_CFG_DEFLATE_BYTES (default: half of the free memory); streams whose scratch does not fit
// side by side go in groups.  Streams the pool could not serve come back unfinished and take the one-kernel search.
static int32_t deflate_full_rounds(spng_ctx *c, std::vector<DeflateJob> &sorted, size_t first, size_t last, spng_result *dr, Arena &a, size_t jslot)
{
    if (first >= last) return SPNG_DONE;
    const size_t nfull = last - first;
    auto scratch_of = [](uint64_t n) -> uint64_t {
        const uint64_t V = deflate2_vertices(n), B = V / 64 + 2;
        // (the candidate records twice: round r + 1 is searched while round r is parsed)
        return 2 * (((2 * V + 255) & ~255ull) + ((8 * B + 255) & ~255ull) + ((4 * B + 255) & ~255ull)) + ((8 * B + 255) & ~255ull) +
               2 * ((4 * (V + 2) + 255) & ~255ull) + ((V + 2 + 255) & ~255ull) + ((B + 255) & ~255ull);
    };
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = c->cfg[SPNG_CFG_DEFLATE_BYTES] ? (uint64_t)c->cfg[SPNG_CFG_DEFLATE_BYTES] : (uint64_t)(free_b + c->graph_cap + c->ring_cap) / 2;
    uint64_t largest = 0, all = 0, worst_pool = 0;
    for (size_t i = first; i < last; ++i) {
        const uint64_t sc = scratch_of(sorted[i].src_len);
        largest = sc > largest ? sc : largest; all += sc;
        // (a position leaves at most min(attempts, 30) words: 14 at level 8, 20 at level 9, 30 from level 10 on)
        const uint64_t per = sorted[i].level <= 8 ? 14 : sorted[i].level == 9 ? 20 : 30;
        worst_pool += (sorted[i].src_len < (1u << 21) ? sorted[i].src_len : (1u << 21)) * per * 4;
    }
    const uint64_t min_pool = 64ull << 20;
    if (budget < largest + min_pool + (64ull << 20)) budget = largest + min_pool + (64ull << 20);
    // groups of streams whose scratch takes at most 2 / 3 of the budget; the rest is rings and pool
    std::vector<std::pair<size_t, size_t>> groups;
    for (size_t i = first; i < last;) {
        uint64_t used = 0; size_t k = i;
        while (k < last && (k == i || used + scratch_of(sorted[k].src_len) <= budget * 2 / 3)) used += scratch_of(sorted[k++].src_len);
        groups.push_back({i, k});
        i = k;
    }
    uint64_t slab = 0;
    struct Lay { uint64_t scratch, rings, pool; uint32_t cps, chunk; };
    std::vector<Lay> lay;
    for (auto &gr : groups) {
        const uint32_t cnt = (uint32_t)(gr.second - gr.first);
        Lay l;
        l.scratch = 0;
        for (size_t i = gr.first; i < gr.second; ++i) l.scratch += scratch_of(sorted[i].src_len);
        uint32_t cps = (256 + cnt - 1) / cnt;                 // (a search workgroup is a CU: one round of them where the streams are few,
        cps = cps < 2 ? 2 : cps > 64 ? 64 : cps;              //  chunks of 2^20 positions -- 3 % of warm-up -- where they are many)
        l.cps = cps; l.chunk = (((1u << 21) / cps + 63) / 64) * 64;
        l.rings = deflate2_temp_bytes(cnt * cps);             // (the searchers' word scratch; round 4: a 256 KiB link ring per workgroup)
        uint64_t room = budget > l.scratch + l.rings ? budget - l.scratch - l.rings : 0;
        l.pool = worst_pool < room / 2 ? worst_pool : room / 2;       // (one pool per round parity)
        if (l.pool < min_pool) l.pool = min_pool;
        l.pool &= ~255ull;
        const uint64_t tot = l.scratch + l.rings + 2 * (l.pool + 256) + 4096;
        slab = tot > slab ? tot : slab;
        lay.push_back(l);
    }
    if (slab > c->graph_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_graph) HIP_TRY(hipFree(c->d_graph));
        c->d_graph = nullptr; c->graph_cap = 0;
        if (hipMalloc(&c->d_graph, slab) != hipSuccess) {
            (void)hipGetLastError();
            // (the one-kernel search knows neither `more` nor a state kept between pushes: a pushed stream it finished from byte 0
            //  would leave its D2State stale and a non-final push with a trailer -- ADVICE r4.  Such a call fails instead; the
            //  caller's states and destinations are untouched and the push can be repeated once memory is there)
            for (size_t i = first; i < last; ++i)
                if (sorted[i].state || sorted[i].more) {
                    return fail_text("spng_deflate_resume_batch: no device memory for the search scratch of pushed streams");
                }
            return deflate_full_legacy(c, sorted, first, last, dr, a, jslot);   // (it sizes its groups by what it can get)
        }
        c->graph_cap = slab;
    }
    // the stream table and the device-side states: in the arena, uploaded once
    const size_t sslot = a.take(nfull * sizeof(D2Stream)), tslot = a.take(nfull * sizeof(D2State)), fslot = a.take((nfull + 1) * 4);
    D2Stream *hs = a.host<D2Stream>(sslot);
    D2State *ht = a.host<D2State>(tslot);
    memset(ht, 0, nfull * sizeof(D2State));                    // (a zeroed state = a stream's beginning: dfl2_begin_kernel)
    std::vector<uint32_t> rounds_of(nfull, 1);
    for (size_t g = 0; g < groups.size(); ++g) {
        char *base = (char *)c->d_graph;
        uint64_t at = 0;
        auto take = [&](uint64_t bytes) { char *p = base + at; at += (bytes + 255) & ~255ull; return p; };
        for (size_t i = groups[g].first; i < groups[g].second; ++i) {
            const DeflateJob &j = sorted[i];
            D2Stream &s = hs[i - first];
            const uint64_t V = deflate2_vertices(j.src_len), B = V / 64 + 2;
            s.src = j.src; s.dst = j.dst; s.src_len = j.src_len; s.dst_cap = j.dst_cap; s.format = j.format; s.level = j.level;
            s.image = j.image; s.exponent = j.exponent; s.more = j.more; s.pad = 0;
            // (spng_deflate_resume_batch: the caller's state, kept from push to push; else the call's own)
            s.state = j.state ? (D2State *)j.state : a.dev<D2State>(tslot) + (i - first);
            s.vinfo = (uint16_t *)take(2 * V); s.bbase = (uint64_t *)take(8 * B); s.bwords = (uint32_t *)take(4 * B); s.emask = (uint64_t *)take(8 * B);
            s.vinfo2 = (uint16_t *)take(2 * V); s.bbase2 = (uint64_t *)take(8 * B); s.bwords2 = (uint32_t *)take(4 * B);
            s.up = (uint32_t *)take(4 * (V + 2)); s.step = (uint32_t *)take(4 * (V + 2)); s.pathb = (uint8_t *)take(V + 2); s.litb = (uint8_t *)take(B);
            uint64_t pos = j.state ? j.plan_pos : 0;
            uint32_t lim = j.state && j.plan_limit ? j.plan_limit : 2048;
            rounds_of[i - first] = deflate2_plan(j.src_len, j.more != 0, pos, lim);
        }
    }
    if (int32_t st = c->upload(sslot, tslot + nfull * sizeof(D2State))) return st;
    uint32_t *d_failed = a.dev<uint32_t>(fslot);
    for (size_t g = 0; g < groups.size(); ++g) {
        const uint32_t cnt = (uint32_t)(groups[g].second - groups[g].first);
        const Lay &l = lay[g];
        char *rings = (char *)c->d_graph + ((l.scratch + 255) & ~255ull);
        char *pools[2] = {rings + l.rings, rings + l.rings + l.pool + 256};       // (each followed by its bump counter)
        uint32_t rounds = 0;
        for (size_t i = groups[g].first; i < groups[g].second; ++i) rounds = rounds_of[i - first] > rounds ? rounds_of[i - first] : rounds;
        const D2Stream *ds = a.dev<D2Stream>(sslot) + (groups[g].first - first);
        HIP_TRY(launch_deflate2_begin(ds, cnt, c->stream));
        // The search of round r + 1 beside the parse of round r, on a stream of its own (candidates are a function of the input
        // alone; the parse is one wave per stream and leaves most of the chip idle): two sets of records and pools, by round parity.
        // searched[p] / parsed[p]: the last search into / parse out of the records of parity p.
        if (!c->stream2) {
            HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
            for (hipEvent_t *e : {&c->ev_fork, &c->ev_mid, &c->ev_join}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        if (!c->ev_dfl[0]) for (hipEvent_t &e : c->ev_dfl) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t par = r & 1;
            if (r >= 2) HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_dfl[2 + par], 0));     // the parse of round r - 2 is done with these records
            { Timed t(c, SPNG_K_DFL_SEARCH, c->stream2); HIP_TRY(launch_deflate2_search(ds, cnt, l.cps, l.chunk, (uint32_t *)pools[par],
                                                                            (unsigned long long *)(pools[par] + l.pool), l.pool / 4, (uint32_t *)rings, par, c->stream2)); }
            HIP_TRY(hipEventRecord(c->ev_dfl[par], c->stream2));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_dfl[par], 0));
            { Timed t(c, SPNG_K_DFL_PARSE); HIP_TRY(launch_deflate2_parse(ds, cnt, (uint32_t *)pools[par], dr, par, c->stream)); }
            HIP_TRY(hipEventRecord(c->ev_dfl[2 + par], c->stream));
        }
    }
    // who is not finished?  (the pool ran dry under them: a batch of very compressible streams on little memory)
    HIP_TRY(launch_deflate2_failed(a.dev<D2Stream>(sslot), (uint32_t)nfull, d_failed, c->stream));
    std::vector<uint32_t> failed(nfull + 1, 0);
    HIP_TRY(hipMemcpyAsync(failed.data(), d_failed, (nfull + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nfull; ++i) if (failed[1 + i] == 1 && sorted[first + i].more) { failed[1 + i] = 0; failed[0] -= 1; }   // (a push that is not the last is never "finished")
    for (size_t i = 0; i < nfull; ++i)
        if (failed[1 + i] && sorted[first + i].state) {
            snprintf(g_err, sizeof g_err, "spng_deflate_resume_batch: the candidate pool ran dry under stream %u (raise SPNG_CFG_DEFLATE_BYTES)", sorted[first + i].image);
            return SPNG_E_DEVICE;
        }
    if (failed[0]) {
        std::vector<DeflateJob> again;
        for (size_t i = 0; i < nfull; ++i) if (failed[1 + i]) again.push_back(sorted[first + i]);
        std::vector<DeflateJob> rest;
        for (size_t i = 0; i < nfull; ++i) if (!failed[1 + i]) rest.push_back(sorted[first + i]);
        std::copy(again.begin(), again.end(), sorted.begin() + first);
        std::copy(rest.begin(), rest.end(), sorted.begin() + first + again.size());
        return deflate_full_legacy(c, sorted, first, first + again.size(), dr, a, jslot);
    }
    return SPNG_DONE;
}

// Levels 0-7 in rounds (deflate.hip, "round 5"): the chip-wide search leaves one word per position, a parse wave per stream walks
// them.  Per stream two sets of 4 bytes per position of a round (<= 2^21 positions; the search of round r + 1 beside the parse of
// round r) from the context's slab, streams that do not fit side by side in groups.  No room at all: the one-kernel form for
// streams that keep no state, an error for pushed ones.
static int32_t deflate_fast_rounds(spng_ctx *c, std::vector<DeflateJob> &sorted, size_t first, size_t last, spng_result *dr, Arena &a, size_t jslot, bool &fell_back)
{
    fell_back = false;
    if (first >= last) return SPNG_DONE;
    const size_t nfast = last - first;
    const uint64_t RV = deflate3_round_positions();
    // one-shot streams first: their blocks are written side by side (dfl4_*); streams that arrive in pieces keep their state
    // between calls and take the two-wave parse (dfl3_parse_kernel)
    std::stable_partition(sorted.begin() + first, sorted.begin() + last, [](const DeflateJob &j) { return !j.state && !j.more; });
    size_t mid = first;
    while (mid < last && !sorted[mid].state && !sorted[mid].more) ++mid;
    // what a call searches: from where the previous push left the search (plan_aux) to the end the input allows
    auto span_of = [&](const DeflateJob &j) -> uint64_t {
        const uint64_t from = j.state ? j.plan_aux : 0, E = deflate3_end(j.src_len, j.more != 0);
        return E > from ? E - from : 0;
    };
    auto round_of = [&](const DeflateJob &j) -> uint64_t { const uint64_t sp = span_of(j); return sp < RV ? sp : RV; };
    auto al = [](uint64_t b) -> uint64_t { return (b + 255) & ~255ull; };
    // (dfl4_scan / dfl4_place find a block's first bit at bbits[launch's max_blocks + k]: every stream's bbits is sized by the
    // largest block count of the call, not by its own -- ADVICE r5: a short stream beside a long one had its staged bits overwritten)
    uint64_t mb_all = 0;
    for (size_t i = first; i < mid; ++i) { const uint64_t mb = deflate4_max_blocks(round_of(sorted[i])); mb_all = mb > mb_all ? mb : mb_all; }
    auto scratch_of = [&](const DeflateJob &j, bool blocks) -> uint64_t {
        const uint64_t V = round_of(j) + 64;
        uint64_t b = 2 * al(4 * V);
        if (blocks) {
            const uint64_t mb = deflate4_max_blocks(round_of(j));
            b += al(4 * (V + 4096)) + al(4 * (4 + 2 * mb)) + al(16 * mb_all) + al(mb * deflate4_block_bytes() + 64);
        }
        return b;
    };
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = c->cfg[SPNG_CFG_DEFLATE_BYTES] ? (uint64_t)c->cfg[SPNG_CFG_DEFLATE_BYTES] : (uint64_t)(free_b + c->graph_cap + c->ring_cap) / 2;
    struct Group { size_t first, last; bool blocks; };
    std::vector<Group> groups;
    uint64_t slab = 0;
    for (size_t i = first; i < last;) {
        const bool blocks = i < mid;
        const size_t stop = blocks ? mid : last;
        uint64_t used = 0;
        size_t k = i;
        while (k < stop) {
            const uint64_t sc = scratch_of(sorted[k], blocks);
            if (used + sc > budget && k > i) break;
            used += sc; ++k;
        }
        slab = used > slab ? used : slab;
        groups.push_back({i, k, blocks});
        i = k;
    }
    slab += 4096;
    if (slab > c->graph_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_graph) HIP_TRY(hipFree(c->d_graph));
        c->d_graph = nullptr; c->graph_cap = 0;
        if (hipMalloc(&c->d_graph, slab) != hipSuccess) {
            (void)hipGetLastError();
            for (size_t i = first; i < last; ++i)
                if (sorted[i].state || sorted[i].more) return fail_text("spng_deflate_resume_batch: no device memory for the search records of pushed streams");
            fell_back = true;                                  // (the caller runs the one-kernel form)
            return SPNG_DONE;
        }
        c->graph_cap = slab;
    }
    const size_t sslot = a.take(nfast * sizeof(D3Stream)), tslot = a.take(nfast * sizeof(D1State));
    D3Stream *hs = a.host<D3Stream>(sslot);
    // (a zeroed state = a stream's beginning: dfl3_begin_kernel)
    for (size_t i = 0; i < nfast; ++i) memset(a.host<D1State>(tslot) + i, 0, sizeof(D1State));
    std::vector<uint32_t> rounds_of(nfast, 1), maxb_of(nfast, 0);
    for (auto &gr : groups) {
        char *base = (char *)c->d_graph;
        uint64_t at = 0;
        auto take = [&](uint64_t bytes) { char *p = base + at; at += al(bytes); return p; };
        for (size_t i = gr.first; i < gr.last; ++i) {
            const DeflateJob &j = sorted[i];
            D3Stream &s = hs[i - first];
            memset(&s, 0, sizeof s);
            const uint64_t V = round_of(j) + 64;
            s.src = j.src; s.dst = j.dst; s.src_len = j.src_len; s.dst_cap = j.dst_cap; s.format = j.format; s.level = j.level;
            s.image = j.image; s.exponent = j.exponent; s.more = j.more; s.pad = 0;
            s.state = j.state ? j.state : a.dev<D1State>(tslot) + (i - first);
            s.match[0] = (uint32_t *)take(4 * V); s.match[1] = (uint32_t *)take(4 * V);
            if (gr.blocks) {
                const uint64_t mb = deflate4_max_blocks(round_of(j));
                s.terms = (uint32_t *)take(4 * (V + 4096)); s.bdesc = (uint32_t *)take(4 * (4 + 2 * mb));
                s.bbits = (uint64_t *)take(16 * mb_all); s.scratch = (uint8_t *)take(mb * deflate4_block_bytes() + 64);
                maxb_of[i - first] = (uint32_t)mb;
            }
            const uint64_t sp = span_of(j);
            rounds_of[i - first] = sp ? (uint32_t)((sp + RV - 1) / RV) : 1u;   // (one launch at least: it reports where the stream stands)
        }
    }
    if (int32_t st = c->upload(sslot, tslot + nfast * sizeof(D1State))) return st;
    if (!c->stream2) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        for (hipEvent_t *e : {&c->ev_fork, &c->ev_mid, &c->ev_join}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (!c->ev_dfl[0]) for (hipEvent_t &e : c->ev_dfl) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &gr : groups) {
        const uint32_t cnt = (uint32_t)(gr.last - gr.first);
        uint32_t cps = (256 + cnt - 1) / cnt;                 // (as the level >= 8 search: one round of workgroups where the streams are few)
        cps = cps < 2 ? 2 : cps > 64 ? 64 : cps;
        const uint32_t chunk = (uint32_t)(((RV / cps + 63) / 64) * 64);
        uint32_t rounds = 0, maxb = 0;
        for (size_t i = gr.first; i < gr.last; ++i) {
            rounds = rounds_of[i - first] > rounds ? rounds_of[i - first] : rounds;
            maxb = maxb_of[i - first] > maxb ? maxb_of[i - first] : maxb;
        }
        const D3Stream *ds = a.dev<D3Stream>(sslot) + (gr.first - first);
        HIP_TRY(launch_deflate3_begin(ds, cnt, c->stream));
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        for (uint32_t r = 0; r < rounds; ++r) {
            const uint32_t par = r & 1;
            if (r >= 2) HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_dfl[2 + par], 0));     // the parse of round r - 2 is done with these records
            { Timed t(c, SPNG_K_DFL_SEARCH, c->stream2); HIP_TRY(launch_deflate3_search(ds, cnt, cps, chunk, par, c->stream2)); }
            HIP_TRY(hipEventRecord(c->ev_dfl[par], c->stream2));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_dfl[par], 0));
            {
                Timed t(c, SPNG_K_DFL_PARSE);
                if (gr.blocks) HIP_TRY(launch_deflate4_round(ds, cnt, maxb, dr, par, c->stream));
                else HIP_TRY(launch_deflate3_parse(ds, cnt, dr, par, c->stream));
            }
            HIP_TRY(hipEventRecord(c->ev_dfl[2 + par], c->stream));
        }
    }
    return SPNG_DONE;
}

// shared by spng_deflate_batch / spng_encode_batch.  Per-stream link rings of the greedy / lazy kernel live in a context-owned slab.
static int32_t deflate_launch(spng_ctx *c, std::vector<DeflateJob> &jobs, spng_result *dr, Arena &a, size_t jslot,
                              size_t gzparts = (size_t)-1)
{
    // greedy / lazy streams first, then the full-search ones: two contiguous job tables
    std::vector<DeflateJob> sorted;
    sorted.reserve(jobs.size());
    for (auto &j : jobs) if (j.level < 8) sorted.push_back(j);
    const size_t nfast = sorted.size();
    for (auto &j : jobs) if (j.level >= 8) sorted.push_back(j);
    bool legacy = c->cfg[SPNG_CFG_DEFLATE_MODE] == SPNG_DEFLATE_ONE_KERNEL;
    for (auto &j : jobs) if (j.state) legacy = false;          // (streams that arrive in pieces: only the two-kernel search keeps a state)
    // levels 0-7: the search chip-wide and a parse wave per stream, round by round -- unless the one-kernel form is asked for
    bool fast_one_kernel = legacy && nfast;
    if (nfast && !fast_one_kernel) {
        Timed t(c, SPNG_K_DEFLATE);
        if (int32_t st = deflate_fast_rounds(c, sorted, 0, nfast, dr, a, jslot, fast_one_kernel)) return st;
    }
    const size_t nring = fast_one_kernel ? nfast : 0;
    const size_t ring_bytes = nring * 65536 * 4;
    if (ring_bytes > c->ring_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->d_ring) HIP_TRY(hipFree(c->d_ring));
        c->d_ring = nullptr; c->ring_cap = 0;
        HIP_TRY(hipMalloc(&c->d_ring, ring_bytes));
        c->ring_cap = ring_bytes;
    }
    for (size_t i = 0; i < nring; ++i) sorted[i].ring = (uint32_t *)c->d_ring + i * 65536;
    memcpy(a.host<DeflateJob>(jslot), sorted.data(), sorted.size() * sizeof(DeflateJob));
    if (int32_t st = c->upload(jslot, jslot + sorted.size() * sizeof(DeflateJob))) return st;
    {
        Timed t(c, SPNG_K_DEFLATE);
        if (fast_one_kernel) HIP_TRY(launch_deflate(a.dev<DeflateJob>(jslot), (uint32_t)nfast, dr, c->stream));
        if (legacy) { if (int32_t st = deflate_full_legacy(c, sorted, nfast, sorted.size(), dr, a, jslot)) return st; }
        else if (int32_t st = deflate_full_rounds(c, sorted, nfast, sorted.size(), dr, a, jslot)) return st;
    }
    // gzip members: CRC-32 and byte count of the input behind the stream (DeflatorBuffers.swift:96-135)
    if (gzparts != (size_t)-1)
        HIP_TRY(launch_gzip_deflate_post(a.dev<DeflateJob>(jslot), dr, a.dev<uint32_t>(gzparts), (uint32_t)sorted.size(), c->stream));
    return SPNG_DONE;
}

int32_t spng_deflate_batch(spng_ctx *c, const spng_stream_desc *descs, const int32_t *levels, uint32_t count,
                           spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count) || !levels) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    std::vector<DeflateJob> jobs(count);
    bool gzip = false;
    for (uint32_t i = 0; i < count; ++i) {
        // spng_stream_desc.reserved: window exponent 8 ... 15 (0 = 15, as PNG always uses)
        const int32_t e = descs[i].reserved ? descs[i].reserved : 15;
        if ((!descs[i].d_src && descs[i].src_len) || !descs[i].d_dst || e < 8 || e > 15 || descs[i].format < SPNG_FORMAT_ZLIB ||
            descs[i].format > SPNG_FORMAT_GZIP) return SPNG_E_ARGUMENT;
        gzip = gzip || descs[i].format == SPNG_FORMAT_GZIP;
        jobs[i] = DeflateJob{(const uint8_t *)descs[i].d_src, (uint8_t *)descs[i].d_dst, descs[i].src_len,
                             descs[i].dst_cap, nullptr, descs[i].format, levels[i], i,
                             descs[i].format == SPNG_FORMAT_IOS ? 15u : (uint32_t)e, nullptr, 0, 0};
    }
    if (int32_t st = c->reserve(count * (sizeof(DeflateJob) + sizeof(spng_result) + sizeof(D2Stream) + sizeof(D2State) + sizeof(D3Stream) + sizeof(D1State) + 1024 + (gzip ? 4 * (size_t)gzip_pieces() : 0)) + 8192)) return st;
    Arena a{c};
    const size_t jslot = a.take(count * sizeof(DeflateJob));
    const size_t res = a.take(count * sizeof(spng_result));
    const size_t gzparts = gzip ? a.take((size_t)count * 4 * gzip_pieces()) : (size_t)-1;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    if (int32_t st = deflate_launch(c, jobs, dr, a, jslot, gzparts)) return st;
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

uint64_t spng_deflate_state_bytes(void) { return deflate_state_bytes(); }

int32_t spng_deflate_resume_batch(spng_ctx *c, const spng_stream_desc *descs, const int32_t *levels, void *const *d_states, const uint8_t *last,
                                  const uint64_t *h_state, uint32_t count, spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count) || !levels || !d_states || !last) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    std::vector<DeflateJob> jobs(count);
    bool gzip = false;
    for (uint32_t i = 0; i < count; ++i) {
        const int32_t e = descs[i].reserved ? descs[i].reserved : 15;
        if ((!descs[i].d_src && descs[i].src_len) || !descs[i].d_dst || !d_states[i] || e < 8 || e > 15 || descs[i].format < SPNG_FORMAT_ZLIB ||
            descs[i].format > SPNG_FORMAT_GZIP) return SPNG_E_ARGUMENT;
        gzip = gzip || descs[i].format == SPNG_FORMAT_GZIP;
        DeflateJob j{};
        j.src = (const uint8_t *)descs[i].d_src; j.dst = (uint8_t *)descs[i].d_dst; j.src_len = descs[i].src_len; j.dst_cap = descs[i].dst_cap;
        j.format = descs[i].format; j.level = levels[i]; j.image = i; j.exponent = descs[i].format == SPNG_FORMAT_IOS ? 15u : (uint32_t)e;
        j.more = last[i] ? 0u : 1u; j.state = (D1State *)d_states[i];
        if (h_state) { j.plan_pos = h_state[2 * i]; j.plan_limit = (uint32_t)h_state[2 * i + 1]; j.plan_aux = h_state[2 * i + 1]; }
        // (a state is only ever what an earlier call handed out: the search position it names lies inside what the input allows --
        // the match arrays and the round count of levels 0-7 are sized from it)
        if (j.plan_pos > j.src_len || (levels[i] < 8 && j.plan_aux > deflate3_end(j.src_len, j.more != 0))) return SPNG_E_ARGUMENT;
        jobs[i] = j;
    }
    if (int32_t st = c->reserve(count * (sizeof(DeflateJob) + sizeof(spng_result) + sizeof(D2Stream) + sizeof(D2State) + sizeof(D3Stream) + sizeof(D1State) + 1024 + (gzip ? 4 * (size_t)gzip_pieces() : 0)) + 8192)) return st;
    Arena a{c};
    const size_t jslot = a.take(count * sizeof(DeflateJob));
    const size_t res = a.take(count * sizeof(spng_result));
    const size_t gzparts = gzip ? a.take((size_t)count * 4 * gzip_pieces()) : (size_t)-1;
    spng_result *dr = d_results ? d_results : a.dev<spng_result>(res);
    if (int32_t st = deflate_launch(c, jobs, dr, a, jslot, gzparts)) return st;
    if (h_results) {
        HIP_TRY(hipMemcpyAsync(h_results, dr, count * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return SPNG_DONE;
}

int32_t spng_deflate(spng_ctx *c, const void *src, uint64_t n, int32_t format, int32_t level,
                     void *dst, uint64_t cap, spng_result *result)
{
    return spng_deflate_window(c, src, n, format, level, 15, dst, cap, result);
}

int32_t spng_deflate_window(spng_ctx *c, const void *src, uint64_t n, int32_t format, int32_t level, int32_t exponent,
                            void *dst, uint64_t cap, spng_result *result)
{
    if (!c || (!src && n) || !dst || !result || exponent < 8 || exponent > 15) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    DevBuf ds, dd;
    HIP_TRY(ds.alloc(n + 8)); HIP_TRY(dd.alloc(cap));
    HIP_TRY(hipMemcpyAsync(ds.p, src, n, hipMemcpyHostToDevice, c->stream));
    spng_stream_desc d{ds.p, n, dd.p, cap, format, exponent};
    if (int32_t st = spng_deflate_batch(c, &d, &level, 1, nullptr, result)) return st;
    const uint64_t w = result->written < cap ? result->written : cap;
    if (w) HIP_TRY(hipMemcpy(dst, dd.p, w, hipMemcpyDeviceToHost));
    return SPNG_DONE;
}

int32_t spng_encode_batch(spng_ctx *c, const spng_image_desc *descs, int32_t level, uint32_t count,
                          spng_result *d_results, spng_result *h_results)
{
    if (!c || (!descs && count)) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    // filter-select (own lock), then deflate of the filtered scanlines
    if (int32_t st = spng_filter_batch(c, descs, count, d_results, nullptr)) return st;
    std::vector<spng_stream_desc> sd(count);
    std::vector<int32_t> lv(count, level);
    for (uint32_t i = 0; i < count; ++i) {
        const spng_image_desc &d = descs[i];
        if (!d.d_idat) return SPNG_E_ARGUMENT;
        sd[i] = spng_stream_desc{d.d_rows, spng_inflated_size(d.width, d.height, d.depth, d.channels, d.interlaced),
                                 (void *)d.d_idat, d.idat_len, d.format, 0};
    }
    return spng_deflate_batch(c, sd.data(), lv.data(), count, d_results, h_results);
}


// ---- measurement and housekeeping --------------------------------------------------------------------------------
int32_t spng_copy_ceiling(spng_ctx *c, void *d_dst, const void *d_src, uint64_t bytes, int32_t pattern, int32_t repeats, double *ms_per_copy)
{
    if (!c || !d_dst || !d_src || bytes < 65536 || pattern < 0 || pattern > 1 || repeats < 1 || !ms_per_copy) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    hipEvent_t e0 = c->event(), e1 = c->event();
    HIP_TRY(launch_copy_probe(d_src, d_dst, bytes, pattern, c->stream));      // (first touch)
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int32_t i = 0; i < repeats; ++i) HIP_TRY(launch_copy_probe(d_src, d_dst, bytes, pattern, c->stream));
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    c->pool.push_back(e0); c->pool.push_back(e1);
    *ms_per_copy = (double)ms / repeats;
    return SPNG_DONE;
}

int32_t spng_lds_exchange_ordered(spng_ctx *c, int32_t *ordered)
{
    if (!c || !ordered) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t v = 0;
    HIP_TRY(deflate3_probe_result(&v));
    *ordered = (int32_t)v;
    return SPNG_DONE;
}

int32_t spng_trim(spng_ctx *c)
{
    if (!c) return SPNG_E_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> g(c->mu);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
    if (c->stream_out) HIP_TRY(hipStreamSynchronize(c->stream_out));
    void **bufs[] = {&c->d_ring, &c->d_ring2, &c->d_graph, &c->d_log, &c->d_tok, &c->d_sym, &c->d_win, &c->d_multi};
    size_t *caps[] = {&c->ring_cap, &c->ring2_cap, &c->graph_cap, &c->log_cap, &c->tok_cap, &c->sym_cap, &c->win_cap, &c->multi_cap};
    for (int i = 0; i < 8; ++i) {
        if (*bufs[i]) HIP_TRY(hipFree(*bufs[i]));
        *bufs[i] = nullptr; *caps[i] = 0;
    }
    c->pool_ratio = 0; c->block_bytes = 0; c->sym_failed = 0;  // (what the token pool had learned went with it)
    return SPNG_DONE;
}

// spng_decode_batch over several devices: contiguous blocks of ceil(count / n_ctx) images per context,
// no communication while decoding; then, where d_gather names a destination on the FIRST context's device, every raster of the
// other contexts leaves for it as a peer-to-peer copy behind its context's decode (xGMI: one link per peer, all at once).
int32_t spng_shard(uint32_t count, uint32_t parts, uint32_t index, uint32_t *first, uint32_t *n)
{
    if (!parts || index >= parts || !first || !n) return SPNG_E_ARGUMENT;
    const uint32_t per = (count + parts - 1) / parts;
    const uint64_t lo = (uint64_t)per * index;
    *first = lo < count ? (uint32_t)lo : count;
    *n = lo >= count ? 0u : (count - lo < per ? count - (uint32_t)lo : per);
    return SPNG_DONE;
}

int32_t spng_decode_batch_multi(spng_ctx *const *ctxs, uint32_t n_ctx, const spng_image_desc *descs, uint32_t count, void *const *d_gather,
                                spng_result *h_results)
{
    if (!ctxs || !n_ctx || (!descs && count) || !h_results) return SPNG_E_ARGUMENT;
    for (uint32_t k = 0; k < n_ctx; ++k) if (!ctxs[k]) return SPNG_E_ARGUMENT;
    if (!count) return SPNG_DONE;
    // (the caller's current device is the caller's: put back on every way out -- ADVICE r4)
    struct Restore { int dev = -1; Restore() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
                     ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore;
    spng_ctx *root = ctxs[0];
    // Every context's shard is enqueued -- asynchronous calls, results stay on the devices -- in SPNG_CFG_MULTI_GROUPS groups
    // (2 by default when rasters leave for another device): a group's rasters leave on the context's second stream behind an
    // event, while the next group decodes on the first.
    int32_t status = SPNG_DONE;
    std::vector<uint32_t> sent(n_ctx, 0);
    for (uint32_t k = 0; k < n_ctx && status == SPNG_DONE; ++k) {
        uint32_t first = 0, n = 0;
        (void)spng_shard(count, n_ctx, k, &first, &n);
        if (!n) continue;
        spng_ctx *c = ctxs[k];
        if (hipError_t e = hipSetDevice(c->device); e != hipSuccess) { status = fail_hip(e, "hipSetDevice"); break; }
        bool leaves = false;                                   // any raster of this shard wanted on another device?
        if (d_gather)
            for (uint32_t i = first; i < first + n && !leaves; ++i) leaves = d_gather[i] && d_gather[i] != descs[i].d_storage;
        uint32_t groups = 1;
        {
            std::lock_guard<std::mutex> g(c->mu);
            const size_t need = (size_t)n * sizeof(spng_result);
            if (need > c->multi_cap) {
                if (hipError_t e = hipStreamSynchronize(c->stream); e != hipSuccess) { status = fail_hip(e, "hipStreamSynchronize"); break; }
                if (c->d_multi) { (void)hipFree(c->d_multi); c->d_multi = nullptr; c->multi_cap = 0; }
                if (hipError_t e = hipMalloc(&c->d_multi, need + need / 2); e != hipSuccess) { status = fail_hip(e, "hipMalloc"); break; }
                c->multi_cap = need + need / 2;
            }
            if (leaves) {
                if (!c->stream_out) {
                    if (hipError_t e = hipStreamCreateWithFlags(&c->stream_out, hipStreamNonBlocking); e != hipSuccess) { status = fail_hip(e, "hipStreamCreateWithFlags"); break; }
                    for (hipEvent_t &ev : c->ev_out)
                        if (hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming); e != hipSuccess) { status = fail_hip(e, "hipEventCreateWithFlags"); break; }
                    if (status != SPNG_DONE) break;
                }
                // peer access, once per pair of devices: with it the copies ride xGMI directly; without it the runtime stages them
                // through the host -- slower, not wrong, and said so in spng_last_error_string
                const int rd = root->device;
                if (c->device != rd && rd < 64 && !((c->peers | c->peers_refused) >> rd & 1)) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, c->device, rd) != hipSuccess) { can = 0; (void)hipGetLastError(); }
                    hipError_t e = can ? hipDeviceEnablePeerAccess(rd, 0) : hipErrorPeerAccessUnsupported;
                    if (e == hipErrorPeerAccessAlreadyEnabled) { e = hipSuccess; (void)hipGetLastError(); }
                    if (e == hipSuccess) c->peers |= 1ull << rd;
                    else {
                        (void)hipGetLastError();
                        c->peers_refused |= 1ull << rd;
                        snprintf(g_err, sizeof g_err, "spng_decode_batch_multi: no peer access from device %d to device %d (%s): copies are staged",
                                 c->device, rd, hipGetErrorString(e));
                    }
                }
                const int64_t want = c->cfg[SPNG_CFG_MULTI_GROUPS];
                groups = want > 0 ? (uint32_t)want : 2u;
                if (groups > n) groups = n;
                if (groups > 2) groups = 2;                    // (two events per context; more groups cost more than they hide)
            }
        }
        spng_result *d_res = (spng_result *)c->d_multi;
        for (uint32_t g = 0; g < groups && status == SPNG_DONE; ++g) {
            const uint32_t g0 = (uint32_t)((uint64_t)n * g / groups), g1 = (uint32_t)((uint64_t)n * (g + 1) / groups);
            status = spng_decode_batch(c, descs + first + g0, g1 - g0, d_res + g0, nullptr);
            if (status != SPNG_DONE || !leaves) continue;
            sent[k] = 1;                                       // (before the first copy is enqueued: a failure half way still waits for stream_out below)
            if (hipError_t e = hipEventRecord(c->ev_out[g & 1], c->stream); e != hipSuccess) { status = fail_hip(e, "hipEventRecord"); break; }
            if (hipError_t e = hipStreamWaitEvent(c->stream_out, c->ev_out[g & 1], 0); e != hipSuccess) { status = fail_hip(e, "hipStreamWaitEvent"); break; }
            for (uint32_t i = first + g0; i < first + g1; ++i) {
                if (!d_gather[i] || d_gather[i] == descs[i].d_storage) continue;
                const uint64_t s = spng_storage_size(descs[i].width, descs[i].height, descs[i].depth, descs[i].channels);
                const hipError_t e = hipMemcpyPeerAsync(d_gather[i], root->device, descs[i].d_storage, c->device, s, c->stream_out);
                if (e != hipSuccess) { status = fail_hip(e, "hipMemcpyPeerAsync"); break; }
            }
        }
    }
    // wait for everything that was enqueued (also on the way out of a failure), results to the host
    for (uint32_t k = 0; k < n_ctx; ++k) {
        uint32_t first = 0, n = 0;
        (void)spng_shard(count, n_ctx, k, &first, &n);
        if (!n) continue;
        spng_ctx *c = ctxs[k];
        (void)hipSetDevice(c->device);
        if (status == SPNG_DONE && c->d_multi) {
            const hipError_t e = hipMemcpyAsync(h_results + first, c->d_multi, (size_t)n * sizeof(spng_result), hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) status = fail_hip(e, "hipMemcpyAsync");
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && status == SPNG_DONE) status = fail_hip(e, "hipStreamSynchronize");
        if (sent[k] && c->stream_out) {
            e = hipStreamSynchronize(c->stream_out);
            if (e != hipSuccess && status == SPNG_DONE) status = fail_hip(e, "hipStreamSynchronize");
        }
    }
    return status;
}

}  // extern "C"
