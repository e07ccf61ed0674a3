/*
 * oracle/png_rows.c -- CPU ORACLE (test infrastructure, never shipped).
 * Restates swift-png's scanline layer: defilter / filter-select, the Adam7 row walker and the
 * scanline <-> PNG.Image.storage scatter/gather.
 *
 *   PNG.adam7                 Sources/PNG/Decoding/PNG.Decoder.swift:6-15
 *   PNG.Decoder.push          Sources/PNG/Decoding/PNG.Decoder.swift:47-149
 *   PNG.Decoder.defilter      Sources/PNG/Decoding/PNG.Decoder.swift:152-196
 *   PNG.paeth                 Sources/PNG/PNG.swift:124-147
 *   PNG.Image.assign          Sources/PNG/PNG.Image.swift:186-285
 *   PNG.Image.collect         Sources/PNG/PNG.Image.swift:431-544
 *   PNG.Encoder.pull/filter   Sources/PNG/Encoding/PNG.Encoder.swift:33-204,229-234
 */
#include "spng_oracle.h"
#include <stdlib.h>
#include <string.h>

/* PNG.adam7 (PNG.Decoder.swift:6-15): (base.x, base.y, exponent.x, exponent.y) */
static const int ADAM7[7][4] = {
    {0, 0, 3, 3}, {4, 0, 3, 3}, {0, 4, 2, 3}, {2, 0, 2, 2}, {0, 2, 1, 2}, {1, 0, 1, 1}, {0, 1, 0, 1},
};

typedef struct { int bx, by, sx, sy, w, h; size_t pitch; } subimage;

/* sub-image geometry, PNG.Decoder.swift:63-82.  Non-interlaced = one pass with unit strides. */
static int passes(int w, int h, int volume, int interlaced, subimage out[7])
{
    int n = 0;
    if (!interlaced) {
        if (w > 0 && h > 0) {
            out[0] = (subimage){0, 0, 1, 1, w, h, ((size_t)w * volume + 7) >> 3};
            n = 1;
        }
        return n;
    }
    for (int z = 0; z < 7; ++z) {
        int bx = ADAM7[z][0], by = ADAM7[z][1], ex = ADAM7[z][2], ey = ADAM7[z][3];
        int sx = 1 << ex, sy = 1 << ey;
        int sw = (w + sx - bx - 1) >> ex, sh = (h + sy - by - 1) >> ey;
        if (sw <= 0 || sh <= 0) continue;                   /* :76-80 */
        out[n++] = (subimage){bx, by, sx, sy, sw, sh, ((size_t)sw * volume + 7) >> 3};
    }
    return n;
}

size_t orc_inflated_size(int w, int h, int depth, int channels, int interlaced)
{
    subimage p[7];
    int n = passes(w, h, depth * channels, interlaced, p);
    size_t u = 0;
    for (int i = 0; i < n; ++i) u += (p[i].pitch + 1) * (size_t)p[i].h;
    return u;
}

size_t orc_storage_size(int w, int h, int depth, int channels)
{
    /* PNG.Image.swift:73-74: count * ((volume + 7) >> 3) */
    return (size_t)w * (size_t)h * (size_t)((depth * channels + 7) >> 3);
}

uint8_t orc_paeth(uint8_t a, uint8_t b, uint8_t c)
{
    /* PNG.swift:124-147, branch-free form restated with the same tie order */
    int16_t d0 = (int16_t)b - c, d1 = (int16_t)a - c;
    int16_t f0 = d0 < 0 ? -d0 : d0, f1 = d1 < 0 ? -d1 : d1, s = d0 + d1, f2 = s < 0 ? -s : s;
    if (!(f1 < f0) && !(f2 < f0)) return a;
    return f2 < f1 ? c : b;
}

void orc_defilter(uint8_t *line, const uint8_t *last, size_t n, int delay)
{
    size_t d = (size_t)delay;
    switch (line[0]) {
    case 1:
        for (size_t i = 1 + d; i < n; ++i) line[i] = (uint8_t)(line[i] + line[i - d]);
        break;
    case 2:
        for (size_t i = 1; i < n; ++i) line[i] = (uint8_t)(line[i] + last[i]);
        break;
    case 3:
        for (size_t i = 1; i < n && i < 1 + d; ++i) line[i] = (uint8_t)(line[i] + (last[i] >> 1));
        for (size_t i = 1 + d; i < n; ++i)
            line[i] = (uint8_t)(line[i] + (((uint16_t)line[i - d] + (uint16_t)last[i]) >> 1));
        break;
    case 4:
        for (size_t i = 1; i < n && i < 1 + d; ++i) line[i] = (uint8_t)(line[i] + orc_paeth(0, last[i], 0));
        for (size_t i = 1 + d; i < n; ++i)
            line[i] = (uint8_t)(line[i] + orc_paeth(line[i - d], last[i], last[i - d]));
        break;
    default:                                                /* 0, and any invalid byte (:193-194) */
        break;
    }
}

/* PNG.Image.assign (PNG.Image.swift:186-285): pixel i of the scanline -> storage pixel
 * (bx + i*sx, y).  Sub-byte samples are expanded MSB-first to one byte each, unscaled. */
static void assign(uint8_t *storage, int w, const uint8_t *scan, int depth, int channels,
                   int bx, int y, int sx)
{
    int volume = depth * channels;
    if (volume < 8) {
        int per = 8 / depth, mask = (1 << depth) - 1;
        int i = 0;
        for (int x = bx; x < w; x += sx, ++i) {
            int a = i / per, sh = (~i & (per - 1)) * depth;
            storage[(size_t)y * w + x] = (uint8_t)((scan[a] >> sh) & mask);
        }
    } else {
        int bpp = volume >> 3, i = 0;
        for (int x = bx; x < w; x += sx, ++i)
            memcpy(storage + ((size_t)y * w + x) * bpp, scan + (size_t)i * bpp, (size_t)bpp);
    }
}

/* PNG.Image.collect (PNG.Image.swift:431-544) */
static void collect(const uint8_t *storage, int w, uint8_t *scan, size_t pitch, int depth, int channels,
                    int bx, int y, int sx)
{
    int volume = depth * channels;
    if (volume < 8) {
        int per = 8 / depth, mask = (1 << depth) - 1;
        memset(scan, 0, pitch);
        int i = 0;
        for (int x = bx; x < w; x += sx, ++i) {
            int a = i / per, sh = (~i & (per - 1)) * depth;
            scan[a] |= (uint8_t)((storage[(size_t)y * w + x] & mask) << sh);
        }
    } else {
        int bpp = volume >> 3, i = 0;
        for (int x = bx; x < w; x += sx, ++i)
            memcpy(scan + (size_t)i * bpp, storage + ((size_t)y * w + x) * bpp, (size_t)bpp);
    }
}

int orc_unfilter(const uint8_t *rows, size_t rows_len,
                 int w, int h, int depth, int channels, int interlaced,
                 uint8_t *storage)
{
    int volume = depth * channels, delay = (volume + 7) >> 3;
    subimage p[7];
    int n = passes(w, h, volume, interlaced, p);
    size_t off = 0, maxline = 1;
    for (int z = 0; z < n; ++z) if (p[z].pitch + 1 > maxline) maxline = p[z].pitch + 1;
    uint8_t *line = (uint8_t *)malloc(maxline), *last = (uint8_t *)malloc(maxline);
    if (!line || !last) { free(line); free(last); return ORC_E_ARGUMENT; }
    for (int z = 0; z < n; ++z) {
        size_t len = p[z].pitch + 1;
        memset(last, 0, len);                               /* :83-84 / :116-117 */
        for (int y = 0; y < p[z].h; ++y) {
            if (off + len > rows_len) goto done;            /* pull() == nil (:88-94): silent stop */
            memcpy(line, rows + off, len);
            off += len;
            orc_defilter(line, last, len, delay);
            assign(storage, w, line + 1, depth, channels, p[z].bx, p[z].by + y * p[z].sy, p[z].sx);
            uint8_t *t = last; last = line; line = t;
        }
    }
done:
    free(line); free(last);
    /* :142-147: anything left in the inflator after the last row */
    return rows_len > orc_inflated_size(w, h, depth, channels, interlaced)
        ? ORC_E_EXTRANEOUS_IMAGE_DATA : ORC_DONE;
}

int orc_decode(const uint8_t *idat, size_t n, int format,
               int w, int h, int depth, int channels, int interlaced,
               uint8_t *storage, uint64_t aux[2])
{
    size_t u = orc_inflated_size(w, h, depth, channels, interlaced);
    /* headroom so that surplus inflated bytes are seen (-> extraneousImageData) */
    size_t cap = u + 65536, written = 0;
    uint8_t *rows = (uint8_t *)malloc(cap ? cap : 1);
    if (!rows) return ORC_E_ARGUMENT;
    int st = orc_inflate(idat, n, format, rows, cap, &written, NULL, aux);
    if (st == ORC_E_OUTPUT_CAPACITY) st = ORC_E_EXTRANEOUS_IMAGE_DATA;
    if (st == ORC_DONE || st == ORC_NEED_MORE_INPUT) {
        /* inflator.push threw nothing: rows decoded so far are assigned (PNG.Decoder.swift:57-140) */
        int ust = orc_unfilter(rows, written, w, h, depth, channels, interlaced, storage);
        if (ust != ORC_DONE) st = ust;
    }
    free(rows);
    return st;
}

/* ------------------------------------------------------------------ encode side */

static long score(const uint8_t *p, size_t n)              /* PNG.Encoder.swift:229-234 */
{
    long s = 0;
    for (size_t i = 0; i < n; ++i) { int v = (int8_t)p[i]; s += v < 0 ? -v : v; }
    return s;
}

int orc_filter_row(const uint8_t *line, const uint8_t *last, size_t n, int delay, uint8_t *out)
{
    /* PNG.Encoder.swift:132-204: all five candidates, first strict minimum of sum|int8| wins */
    size_t d = (size_t)delay, pitch = n - 1;
    uint8_t *cand = (uint8_t *)malloc(5 * n);
    if (!cand) return -1;
    uint8_t *c0 = cand, *c1 = cand + n, *c2 = cand + 2 * n, *c3 = cand + 3 * n, *c4 = cand + 4 * n;
    memcpy(c0, line, n);
    c1[0] = 1; c2[0] = 2; c3[0] = 3; c4[0] = 4;
    for (size_t i = 1; i < n; ++i) {
        uint8_t x = line[i], b = last[i];
        uint8_t a = i > d ? line[i - d] : 0, c = i > d ? last[i - d] : 0;
        int lead = i <= d;                                  /* first `delay` bytes */
        c1[i] = lead ? x : (uint8_t)(x - a);
        c2[i] = (uint8_t)(x - b);
        c3[i] = lead ? (uint8_t)(x - (b >> 1)) : (uint8_t)(x - (uint8_t)(((uint16_t)a + (uint16_t)b) >> 1));
        c4[i] = lead ? (uint8_t)(x - orc_paeth(0, b, 0)) : (uint8_t)(x - orc_paeth(a, b, c));
    }
    int best = 0; long min = -1;
    for (int f = 0; f < 5; ++f) {
        long s = score(cand + f * n + 1, pitch);
        if (min < 0 || s < min) { min = s; best = f; }
    }
    memcpy(out, cand + best * n, n);
    free(cand);
    return best;
}

int orc_filter(const uint8_t *storage, int w, int h, int depth, int channels,
               int interlaced, uint8_t *rows)
{
    int volume = depth * channels, delay = (volume + 7) >> 3;
    subimage p[7];
    int n = passes(w, h, volume, interlaced, p);
    size_t off = 0, maxline = 1;
    for (int z = 0; z < n; ++z) if (p[z].pitch + 1 > maxline) maxline = p[z].pitch + 1;
    uint8_t *line = (uint8_t *)malloc(maxline), *last = (uint8_t *)malloc(maxline);
    if (!line || !last) { free(line); free(last); return ORC_E_ARGUMENT; }
    for (int z = 0; z < n; ++z) {
        size_t len = p[z].pitch + 1;
        memset(last, 0, len);                               /* PNG.Encoder.swift:65-66 / :97-98 */
        for (int y = 0; y < p[z].h; ++y) {
            line[0] = 0;
            collect(storage, w, line + 1, p[z].pitch, depth, channels, p[z].bx, p[z].by + y * p[z].sy, p[z].sx);
            orc_filter_row(line, last, len, delay, rows + off);
            off += len;
            uint8_t *t = last; last = line; line = t;       /* last = raw scanline (:89, :118) */
        }
    }
    free(line); free(last);
    return ORC_DONE;
}
