/*
 * spng_oracle.h -- CPU ORACLE for the swift-png hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the *semantics* of tayloraswift/swift-png's
 * pure-Swift LZ77 + PNG scanline code (reference snapshot 2025-02-26).  It is the
 * parity checker for the HIP kernels.  Only `tests/`, `__graft_entry__.smoke()` and
 * `bench.py`'s `cpu_baseline` leg may load it; the product (swift_png_amd/) never does.
 *
 * Pinning (see tests/test_oracle_*.py, DESIGN.md "Oracle"):
 *   decode  -- pinned: all 161+32 PngSuite inputs vs the reference's own RGBA goldens
 *              (Sources/PNGIntegrationTests/RGBA/ *.rgba), differential vs zlib/Pillow.
 *   encode  -- level 9 pinned bit-exactly against the 28 committed swift-png outputs
 *              (Tests/Outputs/ *.png); other levels: "parity unpinned" beyond round trips
 *              (the reference holds no golden stream for them).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference root).
 */
#ifndef SPNG_ORACLE_H
#define SPNG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status vocabulary.  Numeric values are shared *by convention* with
 * include/spng_mi355.h (the product header); tests assert they agree. */
enum {
    ORC_DONE = 0,                 /* Swift `nil`: stream complete (LZ77.Inflator.swift:39-40)        */
    ORC_NEED_MORE_INPUT = 1,      /* Swift `()`: wants more input (LZ77.Inflator.swift:46-47)        */
    /* LZ77.StreamHeaderError (Inflator/LZ77.StreamHeaderError.swift:6-28) */
    ORC_E_COMPRESSION_METHOD = 16, /* aux0 = method code */
    ORC_E_WINDOW_SIZE = 17,        /* aux0 = exponent    */
    ORC_E_CHECK_BITS = 18,
    ORC_E_DICTIONARY = 19,
    /* LZ77.DecompressionError (Inflator/LZ77.DecompressionError.swift:19-59) */
    ORC_E_STREAM_CHECKSUM = 32,    /* aux0 = declared, aux1 = computed */
    ORC_E_BLOCK_TYPE = 33,         /* aux0 = code */
    ORC_E_BLOCK_COUNT_PARITY = 34, /* aux0 = LEN, aux1 = NLEN */
    ORC_E_RUNLITERAL_COUNT = 35,   /* aux0 = count */
    ORC_E_CODELENGTH_TABLE = 36,
    ORC_E_CODELENGTH_SEQUENCE = 37,
    ORC_E_HUFFMAN_TABLE = 38,
    ORC_E_STRING_REFERENCE = 39,
    /* PNG.DecodingError (Decoding/PNG.DecodingError.swift) */
    ORC_E_EXTRANEOUS_IMAGE_DATA = 48,
    ORC_E_EXTRANEOUS_COMPRESSED_DATA = 49,
    ORC_E_INCOMPLETE_DATASTREAM = 50,
    /* boundary-level (no reference counterpart) */
    ORC_E_OUTPUT_CAPACITY = 64,
    ORC_E_ARGUMENT = 65,
    ORC_E_REFERENCE_UNDEFINED = 67 /* input drives the reference into reading uninitialised memory */
};

enum { ORC_FORMAT_ZLIB = 0, ORC_FORMAT_IOS = 1 };

/* ---------------------------------------------------------------- decode side */

/* Adler-32 ("MRC32").  Wrappers/LZ77.MRC32.swift:26-50.  Pass adler=1 to start. */
uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n);

/* Whole-stream inflate == one LZ77.Inflator.push(all bytes) followed by pull().
 * LZ77.Inflator.swift:30-61, LZ77.InflatorBuffers.swift:25-137,
 * LZ77.InflatorBuffers.Stream.swift:59-429.
 * consumed (optional) = bytes of src read through the end of the stream. */
int orc_inflate(const uint8_t *src, size_t n, int format,
                uint8_t *dst, size_t cap, size_t *written, size_t *consumed,
                uint64_t aux[2]);

/* PNG.paeth, PNG.swift:124-147 */
uint8_t orc_paeth(uint8_t a, uint8_t b, uint8_t c);

/* PNG.Decoder.defilter, Decoding/PNG.Decoder.swift:152-196.
 * line/last have n = pitch+1 bytes (index 0 = filter byte).  In place. */
void orc_defilter(uint8_t *line, const uint8_t *last, size_t n, int delay);

/* Geometry helpers (PNG.Decoder.swift:59-84, PNG.Image.swift:73-74). */
size_t orc_inflated_size(int w, int h, int depth, int channels, int interlaced);
size_t orc_storage_size(int w, int h, int depth, int channels);

/* rows (filter byte + pitch per row, pass after pass) -> PNG.Image.storage.
 * PNG.Decoder.push row walker (PNG.Decoder.swift:59-148) + PNG.Image.assign
 * (PNG.Image.swift:186-285).  Only complete rows within rows_len are processed
 * (a short stream silently yields an incomplete image, PNG.Decoder.swift:88-94).
 * Returns ORC_DONE, or ORC_E_EXTRANEOUS_IMAGE_DATA when rows_len > U. */
int orc_unfilter(const uint8_t *rows, size_t rows_len,
                 int w, int h, int depth, int channels, int interlaced,
                 uint8_t *storage);

/* End to end: concatenated IDAT payload -> storage (PNG.Context.push, PNG.Context.swift:88). */
int orc_decode(const uint8_t *idat, size_t n, int format,
               int w, int h, int depth, int channels, int interlaced,
               uint8_t *storage, uint64_t aux[2]);

/* ---------------------------------------------------------------- encode side */

/* PNG.Encoder.filter, Encoding/PNG.Encoder.swift:132-204 (+ score :229-234).
 * line/last: n = pitch+1 raw bytes (line[0] == 0); out receives n bytes. Returns filter id. */
int orc_filter_row(const uint8_t *line, const uint8_t *last, size_t n, int delay, uint8_t *out);

/* storage -> filtered rows.  PNG.Encoder.pull row walker (PNG.Encoder.swift:33-129) +
 * PNG.Image.collect (PNG.Image.swift:431-544).  rows must hold orc_inflated_size bytes. */
int orc_filter(const uint8_t *storage, int w, int h, int depth, int channels,
               int interlaced, uint8_t *rows);

/* Whole-stream deflate == LZ77.Deflator(format:level:exponent:hint:) push(all,last:true)
 * + concatenated pop()/pull() output.  Deflator/ *.swift.  Returns ORC_DONE or
 * ORC_E_OUTPUT_CAPACITY. */
int orc_deflate(const uint8_t *src, size_t n, int format, int level, int exponent,
                uint8_t *dst, size_t cap, size_t *written);

/* Upper bound on orc_deflate output for n input bytes. */
size_t orc_deflate_bound(size_t n);

/* storage -> concatenated zlib stream (PNG.Encoder.pull end to end). */
int orc_encode(const uint8_t *storage, int w, int h, int depth, int channels,
               int interlaced, int format, int level,
               uint8_t *dst, size_t cap, size_t *written);

#ifdef __cplusplus
}
#endif
#endif
