/*
 * spng_mi355.h -- C ABI of the MI355X-native PNG hot path (libspng_mi355.so).
 *
 * This is the drop-in boundary for swift-png's decode/encode hot path.  swift-png has no FFI of
 * its own (it is pure Swift); the entry points below are what a Swift host would bind (module
 * map / @_silgen_name) from inside the reference functions cited on each
 * declaration.  Plain pointers and sizes only; every function returns an int32 status and never
 * throws; the caller owns all memory; pointers are only used for the duration of the call
 * (asynchronous calls: until spng_sync returns).
 *
 * Pointers named d_* are DEVICE pointers (HBM of the context's GPU); h_* / unprefixed are host
 * pointers.  The compute path is HIP for gfx950 only; there is no CPU fallback: without a GPU
 * spng_create fails with SPNG_E_DEVICE.
 */
#ifndef SPNG_MI355_H
#define SPNG_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPNG_VERSION 0x000100

/* ---- status codes --------------------------------------------------------------------------
 * One code per case of the reference's error enums; payloads travel in spng_result.aux.      */
enum {
    SPNG_DONE = 0,                    /* LZ77.Inflator.push returned nil (LZ77.Inflator.swift:39-40) */
    SPNG_NEED_MORE_INPUT = 1,         /* ... returned ()  (LZ77.Inflator.swift:46-47): truncated stream */
    /* LZ77.StreamHeaderError, Sources/LZ77/Inflator/LZ77.StreamHeaderError.swift:6-28 */
    SPNG_E_COMPRESSION_METHOD = 16,   /* invalidCompressionMethod(aux0) */
    SPNG_E_WINDOW_SIZE = 17,          /* invalidWindowSize(exponent: aux0) */
    SPNG_E_CHECK_BITS = 18,           /* invalidCheckBits */
    SPNG_E_DICTIONARY = 19,           /* unexpectedDictionary */
    /* LZ77.DecompressionError, Sources/LZ77/Inflator/LZ77.DecompressionError.swift:19-59 */
    SPNG_E_STREAM_CHECKSUM = 32,      /* invalidStreamChecksum(declared: aux0, computed: aux1) */
    SPNG_E_BLOCK_TYPE = 33,           /* invalidBlockTypeCode(aux0) */
    SPNG_E_BLOCK_COUNT_PARITY = 34,   /* invalidBlockElementCountParity(aux0, aux1) */
    SPNG_E_RUNLITERAL_COUNT = 35,     /* invalidHuffmanRunLiteralSymbolCount(aux0) */
    SPNG_E_CODELENGTH_TABLE = 36,     /* invalidHuffmanCodelengthHuffmanTable */
    SPNG_E_CODELENGTH_SEQUENCE = 37,  /* invalidHuffmanCodelengthSequence */
    SPNG_E_HUFFMAN_TABLE = 38,        /* invalidHuffmanTable */
    SPNG_E_STRING_REFERENCE = 39,     /* invalidStringReference */
    /* Gzip.StreamHeaderError, Sources/LZ77/Gzip/Gzip.StreamHeaderError.swift:4-11 (SPNG_FORMAT_GZIP) */
    SPNG_E_GZIP_SIGIL = 24,           /* invalidSigil */
    SPNG_E_GZIP_METHOD = 25,          /* invalidCompressionMethod(aux0) */
    SPNG_E_GZIP_FLAG_BITS = 26,       /* invalidFlagBits(aux0) */
    SPNG_E_GZIP_HEADER_CHECKSUM = 27, /* _headerChecksumUnsupported (FHCRC set) */
    /* PNG.DecodingError, Sources/PNG/Decoding/PNG.DecodingError.swift:5-44 */
    SPNG_E_EXTRANEOUS_IMAGE_DATA = 48,         /* extraneousImageData */
    SPNG_E_EXTRANEOUS_COMPRESSED_DATA = 49,    /* extraneousImageDataCompressedData (host side) */
    SPNG_E_INCOMPLETE_DATASTREAM = 50,         /* incompleteImageDataCompressedDatastream (host side) */
    /* PNG.LexingError, Sources/PNG/Lexing/PNG.LexingError.swift:9-35 (spng_lex_batch) */
    SPNG_E_TRUNCATED_SIGNATURE = 80,           /* truncatedSignature */
    SPNG_E_SIGNATURE = 81,                     /* invalidSignature(aux0 = the eight bytes, big-endian) */
    SPNG_E_TRUNCATED_CHUNK_HEADER = 82,        /* truncatedChunkHeader (also: no IEND before the end) */
    SPNG_E_TRUNCATED_CHUNK_BODY = 83,          /* truncatedChunkBody(expected: aux0) */
    SPNG_E_CHUNK_TYPE = 84,                    /* invalidChunkTypeCode(aux0) */
    SPNG_E_CHUNK_CHECKSUM = 85,                /* invalidChunkChecksum(declared: aux0, computed: aux1) */
    /* boundary-level conditions with no reference counterpart */
    SPNG_E_OUTPUT_CAPACITY = 64,      /* destination buffer too small */
    SPNG_E_ARGUMENT = 65,             /* bad argument (null pointer, depth/channels combination, ...) */
    SPNG_E_DEVICE = 66,               /* HIP error / no gfx950 device; see spng_last_error_string */
    SPNG_E_REFERENCE_UNDEFINED = 67   /* malformed input on which the reference reads uninitialised
                                         memory (unused distance code); no defined answer to match */
};

enum { SPNG_FORMAT_ZLIB = 0,          /* LZ77.Format.zlib  (PNG.Standard.common) */
       SPNG_FORMAT_IOS = 1,           /* LZ77.Format.ios   (raw DEFLATE, CgBI)   */
       SPNG_FORMAT_GZIP = 2 };        /* Gzip.Format.gzip  (Gzip.Inflator / Gzip.Deflator, Sources/LZ77/Gzip; one member:
                                         header with FEXTRA / FNAME / FCOMMENT skipped, CRC-32 checked, ISIZE read;
                                         spng_inflate_batch / spng_deflate_batch and their host-pointer forms only) */

typedef struct spng_ctx spng_ctx;     /* owns one device, one HIP stream, its workspaces; re-entrant per handle */

/* Result of one unit of work (one stream / one image). */
typedef struct spng_result {
    int32_t  status;
    int32_t  reserved;                /* inflate / decode: 1 = produced by the parallel pipeline, 0 = serial kernel */
    uint64_t written;                 /* bytes produced (inflate: inflated bytes; deflate: stream bytes) */
    uint64_t consumed;                /* compressed bytes consumed through the end of the stream */
    uint64_t aux[2];                  /* error payload, see the status table */
} spng_result;

/* One zlib/raw-DEFLATE stream to inflate.  All pointers are device pointers. */
typedef struct spng_stream_desc {
    const void *d_src;  uint64_t src_len;     /* whole stream = concatenated IDAT payloads */
    void       *d_dst;  uint64_t dst_cap;     /* inflated bytes */
    int32_t     format;
    int32_t     reserved;                     /* deflate: window exponent 8 ... 15 (LZ77.Deflator(exponent:)); 0 = 15 */
} spng_stream_desc;

/* One image.  d_rows is the inflated scanline stream (filter byte + pitch bytes per row, pass
 * after pass; U bytes, spng_inflated_size); d_storage is PNG.Image.storage (S bytes,
 * spng_storage_size).  d_rows is scratch for the decoder (it may be overwritten). */
typedef struct spng_image_desc {
    const void *d_idat;    uint64_t idat_len;   /* decode: input;  encode: unused            */
    void       *d_rows;    uint64_t rows_cap;   /* >= U; decode: scratch, encode: output      */
    void       *d_storage;                      /* S bytes                                     */
    uint32_t    width, height;
    uint8_t     depth;                          /* 1,2,4,8,16                                  */
    uint8_t     channels;                       /* 1 (v / indexed), 2 (va), 3 (rgb), 4 (rgba)  */
    uint8_t     interlaced;                     /* Adam7                                       */
    uint8_t     format;                         /* SPNG_FORMAT_*                               */
    uint32_t    reserved;                       /* flags, MUST be zero-initialised: SPNG_IMAGE_OVERDRAW (spng_unfilter_resume_batch
                                                 * acts on it and refuses unknown bits with SPNG_E_ARGUMENT) */
} spng_image_desc;
/* PNG.Context.push(data:overdraw: true) (PNG.Context.swift:88-102, PNG.Image.overdraw, PNG.Image.swift:134-183): while an
 * interlaced image is incomplete, every assigned pixel is replicated over the cell of storage the later passes will refine,
 * for progressive display.  Honoured by spng_unfilter_resume_batch: after every call d_storage equals the reference's
 * image.storage after the same scanlines (pixels no scanline has reached yet keep what the caller put there). */
enum { SPNG_IMAGE_OVERDRAW = 1 };

/* ---- utilities ----------------------------------------------------------------------------- */
int32_t     spng_version(void);
const char *spng_status_string(int32_t status);                 /* PNG.Error.message analogue      */
const char *spng_last_error_string(void);                       /* last HIP error text, thread-local */
/* U = sum over passes of (pitch+1)*rows   (PNG.Decoder.swift:59-84) */
uint64_t    spng_inflated_size(uint32_t w, uint32_t h, int depth, int channels, int interlaced);
/* S = w*h*ceil(volume/8)                  (PNG.Image.swift:73-74)   */
uint64_t    spng_storage_size(uint32_t w, uint32_t h, int depth, int channels);

/* ---- context ------------------------------------------------------------------------------- */
/* device: HIP ordinal.  stream: an existing hipStream_t to launch on (e.g. the caller's), or
 * NULL to let the context create its own non-blocking stream. */
int32_t spng_create(int device, void *stream, spng_ctx **out);
void    spng_destroy(spng_ctx *ctx);
void   *spng_stream(spng_ctx *ctx);                             /* the hipStream_t kernels launch on */
int32_t spng_sync(spng_ctx *ctx);                               /* hipStreamSynchronize */

/* Tuning knobs of a context (none changes results).  value 0 = automatic. */
enum { SPNG_CFG_INFLATE_MODE = 0,   /* SPNG_INFLATE_AUTO: parallel pipeline, serial kernel for what it declines;
                                       SPNG_INFLATE_SERIAL: serial kernel only */
       SPNG_CFG_SEGMENT_BYTES = 1,  /* parallel inflate: nominal segment length in compressed bytes */
       SPNG_CFG_TOKEN_BYTES = 2,    /* parallel inflate: size limit of the token page pool in bytes */
       SPNG_CFG_UNFILTER_PIECE_ROWS = 3,   /* unfilter: rows per piece a scanline chain is cut into */
       SPNG_CFG_INFLATE_OVERLAP = 4,       /* parallel inflate: SPNG_OVERLAP_ALWAYS = every batch of >= 2 streams in two halves
                                              on two streams, the decode of one beside the resolve of the other
                                              (an experiment: slower than one pass on MI355X, so never by default) */
       SPNG_CFG_RESOLVE_PARTS = 5,         /* parallel inflate, batches of <= 384 streams: workgroups that resolve ONE stream side by
                                              side (0: as many as fill the chip, at most 128 and not below ~1 MiB of output each; 1: one, as in large batches; n: n, at most 128) */
       SPNG_CFG_DEFLATE_MODE = 6,          /* levels >= 8: SPNG_DEFLATE_AUTO = search and parse in kernels of their own, rounds of 2^21 vertices
                                              (streams its candidate pool cannot serve: the one-kernel search afterwards);
                                              SPNG_DEFLATE_ONE_KERNEL = one wave per stream does everything (round 2's kernel) */
       SPNG_CFG_DEFLATE_BYTES = 7,         /* levels >= 8: size limit of the context's scratch slab (per-stream vertex arrays, candidate
                                              pool, link rings) in bytes; 0 = half of the free device memory.  Streams that do not
                                              fit side by side go in groups */
       SPNG_CFG_MULTI_GROUPS = 8,          /* spng_decode_batch_multi: calls a context's shard is cut into when its rasters leave for another
                                              device (a group's copies run beside the next group's decode); 0 = 2, at most 2 */
       SPNG_CFG_COUNT = 9 };
enum { SPNG_INFLATE_AUTO = 0, SPNG_INFLATE_SERIAL = 1 };
enum { SPNG_DEFLATE_AUTO = 0, SPNG_DEFLATE_ONE_KERNEL = 1 };
enum { SPNG_OVERLAP_AUTO = 0, SPNG_OVERLAP_ALWAYS = 1, SPNG_OVERLAP_NEVER = 2 };
int32_t spng_configure(spng_ctx *ctx, int key, int64_t value);

/* Per-kernel timing with HIP events recorded on the context's stream around every launch. */
enum { SPNG_K_INFLATE = 0,          /* the serial inflate kernel (streams the parallel pipeline left to it) */
       SPNG_K_UNFILTER = 1, SPNG_K_SCATTER = 2, SPNG_K_FILTER = 3,
       SPNG_K_DEFLATE = 4, SPNG_K_ADLER = 5,
       SPNG_K_PINFLATE = 6,         /* the parallel inflate pipeline as a whole: find + decode + scan + resolve */
       SPNG_K_UNPACK = 7,
       SPNG_K_PACK = 10,            /* spng_pack_batch */
       SPNG_K_LEX = 12,             /* chunk lexing + CRC-32 / IDAT chunk emission */
       SPNG_K_PINF_FIND = 8, SPNG_K_PINF_DECODE = 9, SPNG_K_PINF_RESOLVE = 11,   /* its stages */
       SPNG_K_DFL_SEARCH = 13, SPNG_K_DFL_PARSE = 14,   /* levels >= 8: the two kernels of a round (inside SPNG_K_DEFLATE) */
       SPNG_K_COUNT = 16 };
int32_t spng_profile(spng_ctx *ctx, int enable);                /* enable/disable + reset counters  */
int32_t spng_profile_get(spng_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);
/* Token volume of the most recent parallel-inflate call whose figures have come back (they travel behind its kernels; this call
 * waits for the context's stream): bytes of the 64 KiB token pages its segments took -- what pinf2_decode writes and pinf2_resolve
 * reads by design, rounded up to pages --, the DEFLATE blocks it decoded, and whether a pass found the pool empty.  No reference
 * counterpart (measurement: bench.py's per-kernel design bytes).  Any pointer may be NULL. */
int32_t spng_token_stats(spng_ctx *ctx, uint64_t *page_bytes, uint64_t *blocks, int32_t *ran_dry);

/* ---- decode: device batch entry points (asynchronous on the context's stream) -------------- */
/* replaces LZ77.Inflator.push/pull over whole streams: LZ77.Inflator.swift:30-61,
 * LZ77.InflatorBuffers.swift:25-137, LZ77.InflatorBuffers.Stream.swift:59-429.
 * d_results: device array of `count` spng_result, or NULL when h_results is given.
 * h_results: host array filled after an implicit sync, or NULL for a fully asynchronous call. */
int32_t spng_inflate_batch(spng_ctx *ctx, const spng_stream_desc *descs, uint32_t count,
                           spng_result *d_results, spng_result *h_results);

/* LZ77.Inflator.push for streams that arrive in pieces (LZ77.Inflator.swift:30-61; PNG.Context.push(data:), one call per
 * IDAT chunk, PNG.Context.swift:88-102): the device-side counterpart of the reference's resumable state machine
 * (LZ77.InflatorState / BlockState), which stops and goes on at any byte (LZ77.InflatorBuffers.Stream.swift:61-65, 284-288,
 * 352-356).  d_src / src_len: ALL compressed bytes received so far (the caller appends to its device buffer); d_dst: the output so
 * far, kept between calls.  h_state: FOUR words per stream as the previous call returned them for a SPNG_NEED_MORE_INPUT result --
 * {aux[0]: first bit of the block the input ended in, aux[1]: inflated bytes in front of that block, consumed: first bit INSIDE the
 * block that was not complete yet (a token; a byte of a stored block; 0: the input ended in the block's header), written: inflated
 * bytes in front of that bit (hand back 0 with a 0 bit)}; all zero or a NULL array: nothing seen yet.
 * Blocks the input now holds completely are decoded by the parallel pipeline exactly once.  The block the input ends in: while it is
 * of ordinary size it is decoded again from its header by the next call (cheaper than leaving everything behind it to one wave);
 * once more than 1 MiB of input lies inside ONE block, the next call goes on at the token the last one stopped in front of (the
 * serial kernel, its tables rebuilt from the block's header): a stream that is a single block pushed in k pieces costs O(n), not
 * O(n k).  (Whatever the blocks are: the pipeline's first segment starts at the resume point and takes stored and fixed blocks
 * like dynamic ones.)  Results as spng_inflate_batch, except `consumed` of a SPNG_NEED_MORE_INPUT result (above; gzip: a bit of the
 * member's DEFLATE payload, like aux[0]); written / consumed of finished streams count from the start of the stream; the zlib
 * checksum -- the CRC-32 of a gzip member -- is verified over the whole output by the call that reports SPNG_DONE); formats
 * SPNG_FORMAT_ZLIB, SPNG_FORMAT_IOS and SPNG_FORMAT_GZIP (state = bits and bytes of the member's DEFLATE payload; the header is
 * parsed again per call). */
int32_t spng_inflate_resume_batch(spng_ctx *ctx, const spng_stream_desc *descs, const uint64_t *h_state, uint32_t count,
                                  spng_result *d_results, spng_result *h_results);

/* replaces the row walker PNG.Decoder.push (PNG.Decoder.swift:59-148), PNG.Decoder.defilter
 * (:152-196) and PNG.Image.assign (PNG.Image.swift:186-285).  d_rows_len: device array of the
 * number of valid bytes in each d_rows (NULL = U for every image). */
int32_t spng_unfilter_batch(spng_ctx *ctx, const spng_image_desc *descs, uint32_t count,
                            const uint64_t *d_rows_len,
                            spng_result *d_results, spng_result *h_results);

/* replaces PNG.Context.push(data:) end to end (PNG.Context.swift:88-102): concatenated IDAT ->
 * storage.  The north-star entry point. */
int32_t spng_decode_batch(spng_ctx *ctx, const spng_image_desc *descs, uint32_t count,
                          spng_result *d_results, spng_result *h_results);

/* ---- decode: host-pointer convenience (copies in/out, synchronous) ------------------------- */
int32_t spng_inflate(spng_ctx *ctx, const void *src, uint64_t n, int32_t format,
                     void *dst, uint64_t cap, spng_result *result);
int32_t spng_unfilter(spng_ctx *ctx, const void *rows, uint64_t rows_len,
                      uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                      void *storage, spng_result *result);
int32_t spng_decode(spng_ctx *ctx, const void *idat, uint64_t n, int32_t format,
                    uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                    void *storage, spng_result *result);
/* Adler-32 of a host buffer computed on the device (LZ77.MRC32, Wrappers/LZ77.MRC32.swift:26-50) */
int32_t spng_adler32(spng_ctx *ctx, const void *data, uint64_t n, uint32_t *out);

/* ---- files: the step in front of the decode path, and behind the encode path ------------------------ */
/* One PNG file resident in HBM, and where its concatenated IDAT payloads go (capacity: len is always enough). */
typedef struct spng_file_desc {
    const void *d_png;  uint64_t len;
    void       *d_idat; uint64_t idat_cap;
} spng_file_desc;
typedef struct spng_lexed {
    int32_t  status;                            /* SPNG_DONE: lexed through IEND with every checksum right */
    uint32_t chunks;                            /* chunks lexed */
    uint64_t aux[2];                            /* error payload */
    uint32_t width, height;                     /* IHDR */
    uint8_t  depth, color, compression, filter, interlace;
    uint8_t  ios;                               /* a CgBI chunk was seen (PNG.Standard.ios) */
    uint8_t  pad[2];
    uint64_t idat_len;                          /* bytes written to d_idat */
    uint64_t plte_off, trns_off;                /* offsets of the PLTE / tRNS payloads in the file (0: absent) */
    uint32_t plte_len, trns_len;
    uint64_t consumed;                          /* bytes lexed */
} spng_lexed;
/* replaces the lexing half of PNG.Image.decompress(stream:): signature(), chunk() with its CRC-32 check and
 * chunk-type validation (Sources/PNG/Lexing/PNG.BytestreamSource.swift:44-108, PNG.Chunk.swift:69-88), the
 * IHDR layout, and the IDAT loop (PNG.Image.swift:385-389) -- for whole files already in HBM. */
int32_t spng_lex_batch(spng_ctx *ctx, const spng_file_desc *files, uint32_t count,
                       spng_lexed *d_infos, spng_lexed *h_infos);
/* One zlib stream to cut into IDAT chunks of chunk_bytes payload bytes (the last one shorter). */
typedef struct spng_chunking_desc {
    const void *d_stream; uint64_t len;
    void       *d_out;    uint64_t out_cap;     /* len + 12 * ceil(len / chunk_bytes) bytes are written */
    uint64_t    chunk_bytes;
} spng_chunking_desc;
/* replaces PNG.BytestreamDestination.format(type: .IDAT, data:) for every chunk PNG.Encoder.pull returns
 * (Sources/PNG/Lexing/PNG.BytestreamDestination.swift:66-88, PNG.Image.swift:658-665). */
int32_t spng_write_idat_batch(spng_ctx *ctx, const spng_chunking_desc *descs, uint32_t count,
                              spng_result *d_results, spng_result *h_results);
/* CRC-32 of a host buffer computed on the device (swift-hash CRC32 as used by chunk()) */
int32_t spng_crc32(spng_ctx *ctx, const void *data, uint64_t n, uint32_t *out);

/* A stream that arrives in pieces (PNG.Context.push(data:) per IDAT chunk): defilters and assigns the
 * scanlines that became complete between h_prev_len[i] and h_now_len[i] inflated bytes -- what PNG.Decoder.row / pass keep
 * track of (Sources/PNG/Decoding/PNG.Decoder.swift:20-21, 88-94, 121-135) -- each row once, with the defiltered row above it
 * as an earlier call left it (in d_storage for 8 / 16-bit non-interlaced images; in d_work[i], a buffer of the size and layout
 * of d_rows, for interlaced and sub-byte ones: d_rows itself stays as inflated, it is the LZ77 window of the next push).
 * results: written = scanline bytes defiltered by this call, consumed = inflated bytes that are whole rows by now.
 * descs[i].reserved & SPNG_IMAGE_OVERDRAW: push(data:overdraw: true) -- see SPNG_IMAGE_OVERDRAW. */
int32_t spng_unfilter_resume_batch(spng_ctx *ctx, const spng_image_desc *descs, void *const *d_work,
                                   const uint64_t *h_prev_len, const uint64_t *h_now_len, uint32_t count,
                                   spng_result *d_results, spng_result *h_results);

/* ---- pixels: the step behind the decode path ----------------------------------------------------- */
/* One image to unpack.  palette: indexed formats only, palette_count x 4 bytes (r, g, b, a): PLTE with the
 * tRNS alphas folded in, as PNG.Format keeps it.  key: the tRNS chroma key of a v / rgb / bgr format. */
typedef struct spng_unpack_desc {
    const void *d_storage;                      /* PNG.Image.storage, S bytes                        */
    void       *d_out;                          /* width * height RGBA quadruplets of `target` bits  */
    const void *d_palette;
    uint32_t    width, height;
    uint32_t    palette_count;
    uint16_t    key[3];
    uint8_t     depth, channels;                /* as in spng_image_desc                             */
    uint8_t     indexed;                        /* PNG.Format.indexed1/2/4/8                         */
    uint8_t     bgr;                            /* PNG.Format.bgr8 / bgra8 (CgBI)                    */
    uint8_t     has_key;
    uint8_t     target;                         /* 8 or 16: T = UInt8 / UInt16                        */
    uint8_t     layout;                         /* SPNG_TARGET_RGBA: PNG.RGBA<T>, SPNG_TARGET_VA: PNG.VA<T> (v = the grey value or
                                                   the red channel, a) -- d_out holds width * height (v, a) pairs then;
                                                   SPNG_TARGET_SCALAR: T, the scalar unpack of PNG.Image.swift:682-760, 1030-1040
                                                   (grey value / red channel / palette[i].r; keys ignored, no premultiply) */
    uint8_t     premultiply;                    /* 0: straight;  SPNG_PREMULTIPLY: .premultiplied (PNG.RGBA.swift:121-127,
                                                   PNG.VA.swift:57-60);  SPNG_PREMULTIPLY_AS_U8 (target 16 only):
                                                   .premultiplied(as: UInt8.self) (PNG.RGBA.swift:146-158), the form the
                                                   reference's iOS goldens are compared in (Roundtripping.swift:206-215) */
    uint8_t     reserved[6];
} spng_unpack_desc;
enum { SPNG_TARGET_RGBA = 0, SPNG_TARGET_VA = 1, SPNG_TARGET_SCALAR = 2 };
enum { SPNG_PREMULTIPLY = 1, SPNG_PREMULTIPLY_AS_U8 = 2 };
/* replaces PNG.RGBA<T>.unpack(_:of:deindexer:) / PNG.VA<T>.unpack(_:of:deindexer:) with the default deindexers, T = UInt8 / UInt16
 * (Sources/PNG/ColorTargets/PNG.RGBA.swift:259-365, depth rescaling Sources/PNG/PNG.swift:255-312,
 * 495-524; PNG.VA.swift:184-290; premultiplication PNG.swift:55-66); what PNG.Image.unpack(as:) returns.  All descs
 * of a call share `target`.  Output: r, g, b, a (or v, a) per pixel in host byte order. */
int32_t spng_unpack_batch(spng_ctx *ctx, const spng_unpack_desc *descs, uint32_t count);
int32_t spng_unpack(spng_ctx *ctx, const void *storage, uint32_t w, uint32_t h, int depth, int channels,
                    int indexed, int bgr, int target, const void *palette, uint32_t palette_count,
                    const uint16_t *key, void *out);
/* the same with a colour-target layout (SPNG_TARGET_*) and premultiplication (0 / SPNG_PREMULTIPLY*) */
int32_t spng_unpack_as(spng_ctx *ctx, const void *storage, uint32_t w, uint32_t h, int depth, int channels,
                       int indexed, int bgr, int target, int layout, int premultiply, const void *palette,
                       uint32_t palette_count, const uint16_t *key, void *out);

/* ---- pixels: the step in front of the encode path ------------------------------------------------ */
/* One image to pack: the inverse of spng_unpack_desc.  d_pixels: width * height RGBA<T> quadruplets (r, g, b, a), VA<T> pairs
 * (v, a) or scalars T in host byte order, T = UInt8 / UInt16 (`source` bits); d_storage: PNG.Image.storage, S bytes. */
typedef struct spng_pack_desc {
    const void *d_pixels;
    void       *d_storage;
    const void *d_palette;                      /* indexed formats: palette_count x (r, g, b, a), as in spng_unpack_desc */
    uint32_t    width, height;
    uint32_t    palette_count;
    uint8_t     depth, channels;                /* as in spng_image_desc                             */
    uint8_t     indexed;                        /* PNG.Format.indexed1/2/4/8                         */
    uint8_t     bgr;                            /* PNG.Format.bgr8 / bgra8 (CgBI)                    */
    uint8_t     source;                         /* 8 or 16: T = UInt8 / UInt16                        */
    uint8_t     layout;                         /* SPNG_TARGET_RGBA / _VA / _SCALAR                   */
    uint8_t     reserved[6];
} spng_pack_desc;
/* replaces PNG.RGBA<T>.pack(_:as:indexer:) / PNG.VA<T>.pack(_:as:indexer:) / the scalar PNG.Image.pack<T> with the default
 * indexers, T = UInt8 / UInt16 -- what PNG.Image.init(packing:size:layout:) stores (Sources/PNG/ColorTargets/PNG.RGBA.swift:409-478,
 * PNG.VA.swift:334-403, PNG.Image.swift:767-834, 935-1010, 1043-1062; depth rescaling PNG.deconvolve, Sources/PNG/PNG.swift:699-1285;
 * default indexers PNG.Color.swift:158-226).  Components of T are narrowed to the format's depth by a right shift or widened by
 * the quantum, stored big-endian; colour formats without alpha drop it; chroma keys play no part.  Indexed formats: the palette
 * entry equal to the pixel reduced to 8 bits, entry 0 when there is none (the reference traps on a palette that holds a colour
 * twice; here the lowest index wins).  All descs of a call share `source`.  The storage can go straight to spng_encode_batch. */
int32_t spng_pack_batch(spng_ctx *ctx, const spng_pack_desc *descs, uint32_t count);
/* host-pointer convenience (copies in / out, synchronous) */
int32_t spng_pack_as(spng_ctx *ctx, const void *pixels, uint32_t w, uint32_t h, int depth, int channels,
                     int indexed, int bgr, int source, int layout, const void *palette, uint32_t palette_count, void *storage);

/* ---- encode -------------------------------------------------------------------------------- */
/* replaces PNG.Encoder.filter (PNG.Encoder.swift:132-204) + PNG.Image.collect
 * (PNG.Image.swift:431-544): storage -> filtered rows with the reference's filter choice. */
int32_t spng_filter_batch(spng_ctx *ctx, const spng_image_desc *descs, uint32_t count,
                          spng_result *d_results, spng_result *h_results);
int32_t spng_filter(spng_ctx *ctx, const void *storage,
                    uint32_t w, uint32_t h, int depth, int channels, int interlaced,
                    void *rows, spng_result *result);

/* replaces LZ77.Deflator(format:level:exponent:hint:) push(all, last: true) + concatenated pull() output:
 * Sources/LZ77/Deflator/LZ77.Deflator.swift:8-44, LZ77.DeflatorBuffers.swift:46-93,
 * LZ77.DeflatorBuffers.Stream.swift:30-709, LZ77.DeflatorMatches.swift:225-379 (levels >= 8).  Every level of
 * LZ77.DeflatorSearch (:13-35) runs on the device: greedy 0-3, lazy 4-7, shortest path 8 and up (>= 13: the
 * last row).  `hint` only sizes the reference's output chunks (platform dependent there): the host re-chunks
 * the concatenated stream into IDATs of any size.  d_dst capacity: spng_deflate_bound(src_len).  The window
 * exponent travels in spng_stream_desc.reserved (batch form) / the `exponent` argument; PNG always uses 15,
 * and LZ77.Format.ios ignores it (LZ77.DeflatorBuffers.swift:52-55).  SPNG_FORMAT_GZIP: Gzip.Deflator (header, CRC-32 and
 * byte count of the input appended on the device).  Levels >= 8: the call waits for a small probe kernel (how
 * repetitive is each input: compressible streams get helper waves) before it enqueues the compression, so it is
 * asynchronous only from there on. */
uint64_t spng_deflate_bound(uint64_t n);
int32_t spng_deflate_batch(spng_ctx *ctx, const spng_stream_desc *descs, const int32_t *levels, uint32_t count,
                           spng_result *d_results, spng_result *h_results);
int32_t spng_deflate(spng_ctx *ctx, const void *src, uint64_t n, int32_t format, int32_t level,
                     void *dst, uint64_t cap, spng_result *result);
int32_t spng_deflate_window(spng_ctx *ctx, const void *src, uint64_t n, int32_t format, int32_t level, int32_t exponent,
                            void *dst, uint64_t cap, spng_result *result);
/* LZ77.Deflator.push(_:last:) for streams that arrive in pieces (Sources/LZ77/Deflator/LZ77.Deflator.swift:14-30; the reference
 * compresses whenever more than 4096 bytes are buffered, LZ77.DeflatorBuffers.swift:68-93, and its output does not depend on how
 * the input was pushed): d_src / src_len = ALL input bytes received so far (the caller appends to its device buffer), d_dst = the
 * stream so far, kept between calls.  d_states[i]: spng_deflate_state_bytes() bytes of device memory per stream, zero-filled
 * before the first push and left alone afterwards -- the device-side counterpart of LZ77.DeflatorBuffers.Stream (parse position,
 * queued terms, symbol costs, block limit, bit writer, checksum).  last[i] != 0: this is all the input.  A call emits what the
 * bytes so far determine: levels 0-7 every term whose positions see their whole 258-byte look-ahead, levels 8 and up every whole
 * block (2047, 4095, ... vertices); the rest waits for the next push.  Results: SPNG_NEED_MORE_INPUT with written = stream bytes
 * so far, consumed = input bytes parsed, aux = the host-side part of the state to hand to the next call as h_state[2 i], [2 i + 1]
 * (levels >= 8: it sizes the call's launches; NULL / {0, 0} at the first push); SPNG_DONE with the final counts when last. */
uint64_t spng_deflate_state_bytes(void);
int32_t spng_deflate_resume_batch(spng_ctx *ctx, const spng_stream_desc *descs, const int32_t *levels, void *const *d_states,
                                  const uint8_t *last, const uint64_t *h_state, uint32_t count,
                                  spng_result *d_results, spng_result *h_results);
/* replaces PNG.Encoder.pull end to end (PNG.Encoder.swift:33-129): storage -> zlib stream; d_rows is
 * scratch for the filtered scanlines (>= U bytes), d_idat receives the stream (capacity idat_len). */
int32_t spng_encode_batch(spng_ctx *ctx, const spng_image_desc *descs, int32_t level, uint32_t count,
                          spng_result *d_results, spng_result *h_results);

/* ---- several devices ------------------------------------------------------- */
/* Images are independent units: a batch is cut into contiguous blocks of ceil(count / parts) -- block `index` is
 * [*first, *first + *n) -- one per device (128 per GPU for 1024 images on 8). */
int32_t spng_shard(uint32_t count, uint32_t parts, uint32_t index, uint32_t *first, uint32_t *n);
/* spng_decode_batch (PNG.Context.push end to end) over n_ctx contexts, one per device of the node: descs[i]'s device
 * pointers live on the device of the context its block belongs to (spng_shard); every context decodes its block without
 * talking to the others.  d_gather (NULL: leave the rasters where they are): per image, where on the FIRST context's device its
 * raster is wanted -- the one exchange step of the path; each goes as a peer-to-peer copy on the context's own copy stream behind
 * the decode of its group (a shard goes in SPNG_CFG_MULTI_GROUPS groups, so a group's rasters travel while the next decodes;
 * xGMI is point to point: the seven peers' copies ride their own links at once).  Peer access to the first context's device is
 * switched on once per pair (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess); where the devices refuse, the copies are staged
 * by the runtime and spng_last_error_string says so.  The caller's current device is left as it was.  Returns when everything
 * has arrived; results in h_results. */
int32_t spng_decode_batch_multi(spng_ctx *const *ctxs, uint32_t n_ctx, const spng_image_desc *descs, uint32_t count,
                                void *const *d_gather, spng_result *h_results);

/* ---- measurement and housekeeping ------------------------------------------------------------------ */
/* The copy ceiling: milliseconds per copy of `bytes` from d_src to d_dst by a kernel of this library with nothing to do in
 * between (HIP events on the context's stream, `repeats` copies after a first touch).  pattern 0: 16 bytes per lane,
 * grid-stride -- the denominator next to the 8 TB/s spec peak.  pattern 1: 64 rows of 16384 bytes per wave in 256-byte tiles,
 * row r trailing row r - 1 by 4 bytes on both sides -- the access pattern of the scanline kernel before its stores were made
 * line-aligned: what that cost, and a known byte count in that pattern to calibrate FETCH_SIZE / WRITE_SIZE against. */
int32_t spng_copy_ceiling(spng_ctx *ctx, void *d_dst, const void *d_src, uint64_t bytes, int32_t pattern, int32_t repeats,
                          double *ms_per_copy);
/* Gives the context's scratch back to the device (token pool, symbol scratch, deflate slab and rings): the next call that
 * needs one allocates it again.  Waits for the context's work first. */
int32_t spng_trim(spng_ctx *ctx);
/* 1: this device's LDS serves the lanes of an atomic exchange on one address in ascending lane order (probed when the context was
 * created) and the deflater's match search inserts a batch of positions with ONE exchange per lane; 0: it keeps the read-back form.
 * No reference counterpart (which of two bit-identical code paths runs: bench and tests report it). */
int32_t spng_lds_exchange_ordered(spng_ctx *ctx, int32_t *ordered);

#ifdef __cplusplus
}
#endif
#endif
