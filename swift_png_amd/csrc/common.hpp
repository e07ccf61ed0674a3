// common.hpp -- shared host/device declarations for libspng_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spng_mi355.h"

namespace spng {

// ---- device-side vocabulary shared by every kernel file --------------------------------------------------------
// Wave-uniform values loaded through the vector path are pinned to scalar registers (the bit readers and
// symbol-boundary chains then run on the scalar unit).
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return (uint64_t)UNI(v >> 32) << 32 | UNI((uint32_t)v); }
// A wave-uniform condition, said so to the compiler: its divergence analysis is conservative at control-flow joins,
// and one branch it takes for lane-dependent turns every loop around it into exec-mask bookkeeping.
#define UB(c) (UNI((c) ? 1u : 0u) != 0u)
#ifndef SPNG_EMU
typedef uint8_t __attribute__((address_space(1))) gbyte;      // a byte in global memory (global_load, not flat_load)
#else
typedef uint8_t gbyte;
#endif

// A bound on the spin waits of the wave-to-wave protocols (unfilter bands, the search kernel's inserter and searchers): a
// protocol bug must fault, not hang the GPU -- but a wave held up from outside (a profiler serialising the kernel, a debugger)
// is not a bug, so the bound is wall time (s_memrealtime, 100 MHz), looked at every 4096 spins: 20 s.
struct SpinGuard {
    uint32_t spins = 0;
    uint64_t t0 = 0;
    __device__ __forceinline__ void tick()
    {
        if ((++spins & 0xfffu) != 0) return;
#ifndef SPNG_EMU
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        if (!t0) t0 = now;
        else if (now - t0 > 2000000000ull) __builtin_trap();
#else
        if (spins > (1u << 28)) __builtin_trap();
#endif
    }
};

// One unfilter job = one dependency chain of scanlines: a whole non-interlaced image or one
// Adam7 sub-image (PNG.Decoder.swift:59-140).  Rows are `in_stride` apart starting at `in`
// (which points at the first row's filter byte); defiltered bytes of row y go to
// out + y*out_stride.  For 8/16-bit non-interlaced images `out` is PNG.Image.storage itself;
// otherwise the rows are defiltered in place (out = in + 1) and a scatter kernel follows.
struct UnfJob {
    const uint8_t *in;
    uint8_t       *out;
    uint64_t       in_stride;     // pitch + 1
    uint64_t       out_stride;
    uint64_t       stream_off;    // offset of `in` inside the image's inflated stream
    const uint64_t *rows_len;     // device pointer to the number of valid inflated bytes, or null
    uint32_t       pitch;         // bytes per row without the filter byte
    uint32_t       rows;
    uint32_t       image;         // index into the result array
    uint32_t       bpp;           // "delay": ceil(volume / 8)
    uint32_t       has_prev;      // spng_unfilter_resume_batch: the defiltered row above row 0 is at out - out_stride
    uint32_t       pad;
};

// One scatter job: defiltered rows of one (sub-)image -> PNG.Image.storage (PNG.Image.assign,
// PNG.Image.swift:186-285), including MSB-first expansion of 1/2/4-bit samples.
struct ScatterJob {
    const uint8_t *rows;          // first row's first data byte
    uint8_t       *storage;
    uint64_t       row_stride;    // pitch + 1
    uint64_t       stream_off;
    const uint64_t *rows_len;
    uint32_t       sub_w, sub_h;  // sub-image size in pixels
    uint32_t       width;         // full image width
    uint32_t       bx, by, sx, sy;
    uint32_t       depth, channels;
};

// One image whose unfinished Adam7 cells are filled in for progressive display (PNG.Image.overdraw, PNG.Image.swift:134-183, as
// PNG.Context.push(data:overdraw: true) calls it after every scanline, PNG.Context.swift:88-102).
struct OverdrawJob {
    uint8_t *storage;
    uint32_t width, height;
    uint32_t elem;                // storage bytes per pixel (1, 2, 3, 4, 6, 8)
    uint32_t y0, y1;              // storage rows this call may have changed
    uint32_t done[7];             // scanlines of each Adam7 pass assigned so far
};

// One filter job (encode): storage -> filtered rows of one (sub-)image
// (PNG.Encoder.pull + PNG.Image.collect + PNG.Encoder.filter).
struct FilterJob {
    const uint8_t *storage;
    uint8_t       *rows;          // first row's filter byte
    uint64_t       row_stride;    // pitch + 1
    uint32_t       sub_w, sub_h;
    uint32_t       width;
    uint32_t       bx, by, sx, sy;
    uint32_t       depth, channels;
    uint32_t       pitch;
};

// One image to unpack: PNG.Image.storage -> RGBA<UInt8 / UInt16> (unpack.hip)
struct UnpackJob {
    const uint8_t *storage;
    void          *out;
    const uint8_t *palette;       // indexed formats: palette_count x (r, g, b, a)
    uint32_t       width, height;
    uint32_t       palette_count;
    uint16_t       key[3];        // tRNS chroma key, at the source depth
    uint8_t        depth, channels, indexed, bgr, has_key, pad;
    uint8_t        layout, premultiply;      // spng_unpack_desc.layout / .premultiply
};

// One image to pack: RGBA<T> / VA<T> / T pixels -> PNG.Image.storage (unpack.hip, pack_kernel)
struct PackJob {
    const void    *pixels;
    uint8_t       *storage;
    const uint8_t *palette;       // indexed formats: palette_count x (r, g, b, a)
    uint32_t       width, height;
    uint32_t       palette_count;
    uint8_t        depth, channels, indexed, bgr;
    uint8_t        layout, pad[3];           // spng_pack_desc.layout
};

struct InflateJob {
    const uint8_t *src;
    uint8_t       *dst;
    uint64_t       src_len;
    uint64_t       dst_cap;
    int32_t        format;
    uint32_t       image;
    const int32_t *skip;          // device flag: non-zero = the parallel pipeline already produced this stream
    // FOUR words: {first bit of the first block not decoded completely yet, inflated bytes in front of it, first bit INSIDE that block
    // that is still to decode (0: its header), inflated bytes in front of that}: the pipeline moves the first pair forward (and
    // clears the second), the serial kernel starts there.  spng_inflate_resume_batch: what the previous call returned; every other
    // call: the library's own slot, {0, 0} (`internal`), so that a stream the pipeline cannot finish -- truncated,
    // corrupt -- costs the serial kernel one block, not the whole stream.
    uint64_t      *state;
    uint32_t       internal, pad;
};

// ---- the parallel inflate pipeline (pinflate2.hip) -----------------------------------------------
// One stream of a batch.  The host fills the first group of fields, the kernels the second.
struct PStream {
    const uint8_t *src;
    uint8_t       *dst;
    uint64_t       src_len, dst_cap;
    int32_t        format;
    uint32_t       image;
    uint32_t       seg_first, seg_count;   // its segments in the PSeg table
    uint64_t       seg_bytes;              // nominal segment length (multiple of 256)
    // device side
    uint64_t       tok_base, ntok;         // its tokens in the token buffer (of its pass)
    uint64_t       end_bit;                // first bit after the final block
    int32_t        ok;                     // 1: the segment chain holds from the first bit to a final block; 2 (resumable
                                           // streams): up to the first block the input does not hold completely
    uint32_t       pass;
    // resumable streams (spng_inflate_resume_batch): where to start, and where to note how far the chain got
    uint64_t       start_bit, out_pos;
    uint64_t      *state;
    uint32_t       serial_only, pad_;      // the caller's state stands deep inside a huge block: the serial kernel goes on THERE (the pipeline would decode the block again)
    // several workgroups per stream: parts_max slots in the part table (0: one workgroup), how many the chain was cut into
    uint32_t       parts_max, parts;
    uint64_t       out_total;              // scan: bytes of the whole chain
    uint64_t       sym_off;                // its 16-bit symbols in the symbol scratch (in symbols)
};
enum { PSEG_FAIL = 0, PSEG_CONT = 1, PSEG_FINAL = 2, PSEG_PARTIAL = 3, PSEG_NOPAGE = 4 };
// One segment: the blocks that start in [index * seg_bytes, (index + 1) * seg_bytes).
struct PSeg {
    uint32_t stream, index;
    uint64_t log_off, log_cap;             // its page table in the slab (first entry, entries)
    // device side
    uint64_t start_bit;                    // find: first block header at or after the nominal start (~0: none)
    uint64_t end_bit;                      // count: where decoding stopped (start of the next block)
    uint64_t ntok;                         // decode: token halfwords
    uint64_t tok_base;                     // scan: first token, relative to the stream's
    int32_t  status;                       // count: PSEG_*
    uint32_t used;                         // scan: part of the chain
    uint32_t next, pad;                    // count: index of the segment that starts where this one stopped
    uint64_t nbytes;                       // decode: bytes its tokens stand for
    uint64_t out_base;                     // scan: its first byte, counted from the stream's first byte of this call
};
// A part of a stream's segment chain that one workgroup resolves (pinflate2.hip, "Several workgroups per stream").
struct PPart {
    uint32_t seg, seg_end;                 // first chain segment (index in the stream), the next part's (~0: to the chain's end)
    uint64_t out_pos;                      // first byte, counted from the stream's first byte of this call
    uint64_t out_len;                      // bytes
    unsigned long long sumS, sumI;         // Adler-32 partial sums over its bytes (not reduced)
    uint32_t present, failed;
};

struct DeflateJob {
    const uint8_t *src;
    uint8_t       *dst;
    uint64_t       src_len;
    uint64_t       dst_cap;
    uint32_t      *ring;          // 65536-entry link ring (scratch, HBM)
    int32_t        format, level;
    uint32_t       image;
    uint32_t       exponent;      // window = 2^exponent (LZ77.Deflator(exponent:); PNG: 15)
    uint32_t      *graph;         // levels >= 8: match-graph scratch (deflate_graph_bytes)
    uint32_t       graph_vertices;
    uint32_t       more;          // spng_deflate_resume_batch: more input will follow (src_len is what arrived so far)
    struct D1State *state;        // ... and where the stream keeps itself between pushes (levels 0-7: a D1State, 8 and up: a D2State; null: one-shot)
    uint64_t       plan_pos;      // (host side: where the previous push left the parse; levels >= 8: its block limit, 0-7: how far
    uint32_t       plan_limit, pad2;   //  the input has been searched -- what the previous call returned in aux[0], aux[1])
    uint64_t       plan_aux;
};

// Greedy / lazy kernel between two pushes (spng_deflate_resume_batch, levels 0-7): the parse position, the terms queued for the
// block being filled, the bit writer, the Adler sums.  The hash window is not kept: the next push enters the last 32 KiB again.
struct D1State {
    uint64_t w, inserted;         // first unparsed position; (one-kernel form) positions below `inserted` are in the Adler sums
    uint64_t acc, total;          // the bit writer: pending bits, bytes produced
    uint32_t nacc, overflow, count, started;
    uint32_t adlerS, adlerI, pad[2];
    uint32_t terms[2048];
    // the two-kernel form (round 5: dfl3_search_kernel + dfl3_parse_kernel): the positions of the round the parse is at, the
    // search's own cursor (it runs a round ahead) and how far the input has been searched -- rounds partition the positions, so the
    // Adler sums the search workgroups add up (unreduced, adlerS / adlerI) count every byte once whatever the pushes were
    uint64_t rb, re, srb, sre, spos;
    uint32_t done, pad3;
};
// One stream of the greedy / lazy levels in the two-kernel form
struct D3Stream {
    const uint8_t *src; uint8_t *dst;
    uint64_t src_len, dst_cap;
    int32_t  format, level;
    uint32_t image, exponent;
    uint32_t more, pad;
    D1State *state;
    uint32_t *match[2];           // by round parity: per position of the round (and one behind it) run << 16 | distance, 0: no run > 5
    // the block-parallel form (one-shot streams: dfl4_walk / dfl4_block / dfl4_scan / dfl4_place), null otherwise
    uint32_t *terms;              // the round's terms, block after block
    uint32_t *bdesc;              // [0] blocks of the round, [1] the stream's last round, [2..3] the bit the round starts at; then per block {first term, terms | final << 31}
    uint64_t *bbits;              // per block: the bits it takes (dfl4_block), its first bit in the stream (dfl4_scan) at [max blocks + k]
    uint8_t  *scratch;            // per block D4_BCAP bytes: its bits from bit 0 on
};

// levels >= 8, the two-kernel form (deflate.hip, "round 4"): what the search and the parse kernel share per stream.
struct D2State {                  // device side, kept from round to round; arrives zeroed except rb / re / limit / generic
    uint64_t rb, re;              // the positions of the current round
    uint64_t pos;                 // first position not parsed yet
    uint32_t limit, generic;      // block limit of the next block; the first block of a stream iterates twice as often
    uint64_t acc, total;          // the bit writer: pending bits, bytes produced
    uint32_t nacc, overflow;
    uint32_t adlerS, adlerI;      // Adler-32 partial sums of the chunks searched so far (mod 65521 each time)
    uint32_t fail, done;          // the pool ran dry under this stream (the one-kernel search takes it afterwards); finished
    uint8_t  depths[544];         // LZ77.DeflatorMatches.Depths between blocks
    uint32_t started, pad;        // (a state arrives zeroed: the first round of the first call sets the block limit)
    // the search kernel's own cursor: it runs a round ahead of the parse (block boundaries are a function of the positions alone)
    uint64_t srb, sre, spos;
    uint32_t slimit, spad;
};
struct D2Stream {
    const uint8_t *src; uint8_t *dst;
    uint64_t src_len, dst_cap;
    int32_t  format, level;
    uint32_t image, exponent;
    uint32_t more, pad;           // spng_deflate_resume_batch: more input will follow -- only blocks whose every vertex has its whole
                                  // look-ahead (258 bytes, + 3 for the key) are taken now
    D2State *state;
    // scratch, round coordinates (vertex = position - rb): per vertex candidates << 9 | longest run; per batch of 64 the first
    // word of its list in the pool and words | longest run << 16; block coordinates: which vertices keep their edges, the
    // ways in, the path
    uint16_t *vinfo; uint64_t *bbase; uint32_t *bwords; uint64_t *emask;
    uint16_t *vinfo2; uint64_t *bbase2; uint32_t *bwords2;   // (the candidate records of the odd rounds: round r + 1 is searched while round r is parsed)
    uint32_t *up, *step; uint8_t *pathb, *litb;   // (litb: per batch of the block, nothing but literal ways in)
};

// PNG.adam7, PNG.Decoder.swift:6-15
struct Pass { uint32_t bx, by, sx, sy, w, h; uint64_t pitch; };
int passes(uint32_t w, uint32_t h, int volume, int interlaced, Pass out[7]);

// kernel launchers (each returns the hipError_t of the launch)
hipError_t launch_unfilter(const UnfJob *d_jobs, uint32_t count, uint32_t bpp, spng_result *d_results,
                           uint32_t pieces, uint32_t piece_rows, hipStream_t stream, uint32_t widest = 0);   // widest: the longest row of the batch in bytes
hipError_t launch_copy_probe(const void *d_src, void *d_dst, uint64_t bytes, int pattern, hipStream_t stream);
hipError_t launch_scatter(const ScatterJob *d_jobs, uint32_t count, const uint32_t *d_job_image,
                          const spng_result *d_results, uint32_t blocks_x, hipStream_t stream);
hipError_t launch_overdraw(const OverdrawJob *d_jobs, uint32_t count, uint32_t blocks_x, hipStream_t stream);
hipError_t launch_inflate(const InflateJob *d_jobs, uint32_t count, spng_result *d_results,
                          hipStream_t stream);
// pinflate2.hip: the parallel inflate pipeline
hipError_t launch_pinf2_find(PStream *d_streams, PSeg *d_segs, uint32_t seg0, uint32_t nsegs, uint32_t retry, hipStream_t stream);
hipError_t launch_pinf2_decode(PStream *d_streams, PSeg *d_segs, uint32_t seg0, uint32_t nsegs, uint32_t *d_pt, uint8_t *d_pool, uint32_t *d_next,
                               uint32_t pages, uint32_t retry, hipStream_t stream);
hipError_t launch_pinf2_scan(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, PPart *d_parts, uint32_t retry, hipStream_t stream);
hipError_t launch_pinf2_resolve(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint32_t *d_pt, uint8_t *d_pool, uint32_t pages,
                                spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax, uint32_t retry, hipStream_t stream);
hipError_t launch_pinf2_parts(PStream *d_streams, uint32_t nstreams, PSeg *d_segs, uint32_t *d_pt, uint8_t *d_pool, uint32_t pages,
                              spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax, uint16_t *d_sym, hipStream_t stream);
hipError_t launch_pinf2_join(PStream *d_streams, uint32_t nstreams, spng_result *d_results, int32_t *d_done, PPart *d_parts, uint32_t pmax,
                             uint16_t *d_sym, uint8_t *d_win, hipStream_t stream);
hipError_t launch_pinf2_account(uint32_t *d_ctr, uint32_t *d_totals, uint32_t pages, hipStream_t stream);
hipError_t launch_deflate(const DeflateJob *d_jobs, uint32_t count, spng_result *d_results, hipStream_t stream);
// gzip.hip
static constexpr uint64_t GZ_NONE = ~0ull;
uint32_t   gzip_pieces();
hipError_t launch_gzip_pre(InflateJob *d_jobs, PStream *d_streams, spng_result *d_results, uint64_t *d_gz, int32_t *d_done,
                           uint32_t count, hipStream_t stream);
hipError_t launch_gzip_inflate_post(const InflateJob *d_jobs, spng_result *d_results, const uint64_t *d_gz, uint32_t *d_parts,
                                    uint32_t count, hipStream_t stream);
hipError_t launch_gzip_deflate_post(const DeflateJob *d_jobs, spng_result *d_results, uint32_t *d_parts, uint32_t count,
                                    hipStream_t stream);
hipError_t launch_resume_post(const InflateJob *d_jobs, spng_result *d_results, uint64_t *d_parts, uint32_t count, hipStream_t stream);
hipError_t launch_deflate_full(const DeflateJob *d_jobs, uint32_t count, bool helpers, spng_result *d_results, hipStream_t stream);
hipError_t launch_deflate_density(const DeflateJob *d_jobs, uint32_t count, uint32_t *d_dense, hipStream_t stream);
uint32_t deflate2_rounds(uint64_t n);
uint64_t deflate_state_bytes();
uint32_t deflate2_plan(uint64_t n, bool more, uint64_t &pos, uint32_t &lim);
hipError_t launch_deflate2_begin(const D2Stream *d_streams, uint32_t count, hipStream_t stream);
uint64_t deflate2_vertices(uint64_t n);
hipError_t launch_deflate2_search(const D2Stream *d_streams, uint32_t count, uint32_t cps, uint32_t chunk_len, uint32_t *d_pool, unsigned long long *d_pool_next,
                                  uint64_t pool_words, uint32_t *d_temp, uint32_t parity, hipStream_t stream);
uint64_t deflate2_temp_bytes(uint32_t workgroups);
// levels 0-7 in rounds (deflate.hip, "round 5")
uint64_t deflate3_round_positions();
uint64_t deflate3_end(uint64_t n, bool more);
hipError_t launch_deflate3_begin(const D3Stream *d_streams, uint32_t count, hipStream_t stream);
hipError_t launch_deflate3_search(const D3Stream *d_streams, uint32_t count, uint32_t cps, uint32_t chunk_len, uint32_t parity, hipStream_t stream);
hipError_t launch_deflate3_probe(hipStream_t stream);
hipError_t deflate3_probe_result(uint32_t *ordered);      // once per context: may the match search's inserter use the one-exchange form?
hipError_t launch_deflate3_parse(const D3Stream *d_streams, uint32_t count, spng_result *d_results, uint32_t parity, hipStream_t stream);
// ... with the blocks written side by side (one-shot streams)
uint64_t deflate4_max_blocks(uint64_t positions);
uint64_t deflate4_block_bytes();
hipError_t launch_deflate4_round(const D3Stream *d_streams, uint32_t count, uint32_t max_blocks, spng_result *d_results, uint32_t parity, hipStream_t stream);
hipError_t launch_deflate2_parse(const D2Stream *d_streams, uint32_t count, const uint32_t *d_pool, spng_result *d_results, uint32_t parity, hipStream_t stream);
hipError_t launch_deflate2_failed(const D2Stream *d_streams, uint32_t count, uint32_t *d_failed, hipStream_t stream);
uint64_t deflate_graph_vertices(uint64_t n);
uint64_t deflate_graph_bytes(uint64_t vertices);
hipError_t launch_unpack(const UnpackJob *d_jobs, uint32_t count, uint32_t blocks_x, int target, hipStream_t stream);
hipError_t launch_pack(const PackJob *d_jobs, uint32_t count, uint32_t blocks_x, int source, hipStream_t stream);
size_t lex_chunk_bytes();
size_t lex_walk_bytes();
hipError_t launch_lex(const spng_file_desc *d_files, uint32_t count, spng_lexed *d_out, void *d_table, const uint64_t *d_table_at,
                      void *d_walks, uint32_t max_listed, hipStream_t stream);
hipError_t launch_write_idat(const spng_chunking_desc *d_descs, uint32_t count, uint32_t blocks_x, spng_result *d_results, hipStream_t stream);
hipError_t launch_crc_partial(const uint8_t *d, uint64_t n, uint64_t piece, uint32_t *d_partial, uint32_t pieces, hipStream_t stream);
uint32_t crc32_fold(const uint32_t *partial, uint64_t pieces, uint64_t n, uint64_t piece);
hipError_t launch_filter(const FilterJob *d_jobs, uint32_t count, uint32_t max_rows, hipStream_t stream);
hipError_t launch_adler_partial(const uint8_t *d, uint64_t n, uint32_t chunk, uint64_t *d_out, uint32_t blocks,
                                hipStream_t stream);

}  // namespace spng
