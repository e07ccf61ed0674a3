/*
 * oracle/inflate.c -- CPU ORACLE (test infrastructure, never shipped).
 * Restates swift-png's LZ77.Inflator for a whole stream handed over in one push.
 *
 *   state machine      Sources/LZ77/Inflator/LZ77.InflatorBuffers.swift:25-137
 *   block readers      Sources/LZ77/Inflator/LZ77.InflatorBuffers.Stream.swift:59-429
 *   zlib header        Sources/LZ77/Inflator/LZ77.StreamHeader.swift:16-54
 *   tree validation    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:80-201
 *   LUT indexing       Sources/LZ77/Inflator/LZ77.InflatorTables.swift:101-119
 *   length/distance    Sources/LZ77/LZ77.Composites.swift:19-111
 *   Adler-32           Sources/LZ77/Wrappers/LZ77.MRC32.swift:26-50
 */
#include "spng_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- Adler-32 (LZ77.MRC32.swift:26-50: s1=1, s2=0, reduce every 5552 bytes) ---- */
uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
    uint32_t s1 = adler & 0xffff, s2 = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        for (size_t i = 0; i < k; ++i) { s1 += p[i]; s2 += s1; }
        s1 %= 65521; s2 %= 65521;
        p += k; n -= k;
    }
    return s2 << 16 | s1;
}

/* ---- bit input: LSB-first, 48 zero padding bits past the end
 *      (LZ77.InflatorIn.swift:130-133,156-198) ---- */
typedef struct { const uint8_t *p; size_t n; size_t count; /* bits */ } bitin;

static inline uint64_t peek(const bitin *in, size_t b)
{
    size_t i = b >> 3;
    uint64_t v = 0;
    if (i + 8 <= in->n) memcpy(&v, in->p + i, 8);           /* little-endian host */
    else for (size_t k = 0; k < 8 && i + k < in->n; ++k) v |= (uint64_t)in->p[i + k] << (8 * k);
    return v >> (b & 7);                                     /* >= 56 valid bits */
}
static inline uint32_t getbits(const bitin *in, size_t b, int count)
{
    return (uint32_t)(peek(in, b) & ((1ull << count) - 1));
}

/* ---- static tables ---- */
static uint8_t REV8[256];                                    /* LZ77.Reversed.swift:17-51 */
/* LZ77.Composites.swift:25-110: (extra, base); index 0 and 30,31 are zero padding rows */
static const uint16_t RUN_EXTRA[32] = {0, 0,0,0,0,0, 0,0,0,1,1, 1,1,2,2,2, 2,3,3,3,3, 4,4,4,4,5, 5,5,5,0, 0,0};
static const uint16_t RUN_BASE [32] = {0, 3,4,5,6,7, 8,9,10,11,13, 15,17,19,23,27, 31,35,43,51,59,
                                       67,83,99,115,131, 163,195,227,258, 0,0};
static const uint16_t DIST_EXTRA[32] = {0,0,0,0,1, 1,2,2,3,3, 4,4,5,5,6, 6,7,7,8,8, 9,9,10,10,11,
                                        11,12,12,13,13, 0,0};
static const uint16_t DIST_BASE [32] = {1,2,3,4,5, 7,9,13,17,25, 33,49,65,97,129, 193,257,385,513,769,
                                        1025,1537,2049,3073,4097, 6145,8193,12289,16385,24577, 0,0};

static void init_static(void)
{
    static int done = 0;
    if (done) return;
    for (int i = 0; i < 256; ++i) {
        int r = 0;
        for (int k = 0; k < 8; ++k) if (i >> k & 1) r |= 0x80 >> k;
        REV8[i] = (uint8_t)r;
    }
    done = 1;
}

/* ---- Huffman LUT, same two-level shape as the reference:
 *      256 first-level entries indexed by the bit-reversed first byte, then one 128-entry
 *      sub-table per 8-bit prefix that is still interior (HuffmanTree.swift:176-201,
 *      InflatorTables.swift:113-119).  len == 0 marks an entry the reference leaves
 *      uninitialised (stub trees, HuffmanTree.swift:52-65). ---- */
typedef struct { uint16_t sym; uint8_t len; } hent;
typedef struct { hent *e; int fence; int z; } htable;

#define HT_MAX (256 + 256 * 128)

static inline hent ht_lookup(const htable *t, uint16_t codeword)
{
    int first = REV8[codeword & 0xff];
    int idx = first < t->fence ? first
            : (((first - t->fence + 2) << 8) | REV8[codeword >> 8]) >> 1;
    return t->e[idx];
}

/* HuffmanTree.size (:80-108) + validate (:138-174) + table (:176-201).
 * lengths[0..count) for symbols 0..count-1.  Returns 0 if the tree is not complete. */
static int ht_build(htable *t, const uint8_t *lengths, int count)
{
    int counts[16] = {0};
    for (int i = 0; i < count; ++i) counts[lengths[i]]++;
    int interior = 1;
    for (int l = 1; l <= 8; ++l) interior = 2 * interior - counts[l];
    int n = 256 - interior, z = 256;
    for (int l = 9; l <= 15; ++l) { z += counts[l] << (15 - l); interior = 2 * interior - counts[l]; }
    if (interior != 0) return 0;
    /* canonical order: by length, then by symbol value (validate's `packed`, :164-172) */
    int base[17]; base[1] = 0;
    for (int l = 1; l <= 15; ++l) base[l + 1] = base[l] + counts[l];
    uint16_t packed[288];
    int fill[16];
    for (int l = 1; l <= 15; ++l) fill[l] = base[l];
    for (int i = 0; i < count; ++i) if (lengths[i]) packed[fill[lengths[i]]++] = (uint16_t)i;
    hent *cur = t->e;
    for (int l = 1; l <= 8; ++l) {
        int clones = 256 >> l;
        for (int k = base[l]; k < base[l + 1]; ++k)
            for (int c = 0; c < clones; ++c) { cur->sym = packed[k]; cur->len = (uint8_t)l; ++cur; }
    }
    cur = t->e + 256;
    for (int l = 9; l <= 15; ++l) {
        int clones = 32768 >> l;
        for (int k = base[l]; k < base[l + 1]; ++k)
            for (int c = 0; c < clones; ++c) { cur->sym = packed[k]; cur->len = (uint8_t)l; ++cur; }
    }
    t->fence = n; t->z = z;
    return 1;
}

/* HuffmanTree.init(stub:) (:52-65): size (256,256); table() writes 128 clones of the single
 * 1-bit symbol, or nothing.  The rest of the table is uninitialised in the reference. */
static void ht_stub(htable *t, int symbol)
{
    for (int i = 0; i < 256; ++i) { t->e[i].sym = 0; t->e[i].len = 0; }
    if (symbol >= 0) for (int i = 0; i < 128; ++i) { t->e[i].sym = (uint16_t)symbol; t->e[i].len = 1; }
    t->fence = 256; t->z = 256;
}

/* HuffmanTree.validate(symbols:normalizing:) (:112-135) */
static int ht_build_normalizing(htable *t, const uint8_t *lengths, int count)
{
    int first = -1;
    for (int i = 0; i < count; ++i) {
        if (!lengths[i]) continue;
        if (first < 0 && lengths[i] == 1) first = i;
        else return ht_build(t, lengths, count);
    }
    ht_stub(t, first);
    return 1;
}

/* fixed trees, HuffmanTree.swift:24-47 */
static void fixed_lengths(uint8_t lit[288], uint8_t dist[32])
{
    for (int i = 0; i < 144; ++i) lit[i] = 8;
    for (int i = 144; i < 256; ++i) lit[i] = 9;
    for (int i = 256; i < 280; ++i) lit[i] = 7;
    for (int i = 280; i < 288; ++i) lit[i] = 8;
    for (int i = 0; i < 32; ++i) dist[i] = 5;
}

typedef struct {
    htable lit, dist, meta;
    hent lit_e[HT_MAX], dist_e[HT_MAX], meta_e[256 + 128];
} tables;

#define FAIL(code, a0, a1) do { if (aux) { aux[0] = (uint64_t)(a0); aux[1] = (uint64_t)(a1); } status = (code); goto out; } while (0)

int orc_inflate(const uint8_t *src, size_t n, int format,
                uint8_t *dst, size_t cap, size_t *written, size_t *consumed,
                uint64_t aux[2])
{
    init_static();
    int status = ORC_NEED_MORE_INPUT;
    bitin in = { src, n, n * 8 };
    size_t b = 0, end = 0;
    tables *T = (tables *)malloc(sizeof(tables));
    if (!T) return ORC_E_ARGUMENT;
    T->lit.e = T->lit_e; T->dist.e = T->dist_e; T->meta.e = T->meta_e;
    if (aux) aux[0] = aux[1] = 0;

    /* .initial (InflatorBuffers.swift:92-104; StreamHeader.swift:16-54) */
    if (format != ORC_FORMAT_IOS) {
        if (b + 16 > in.count) goto out;
        uint32_t cm = getbits(&in, b, 4);
        if (cm != 8) FAIL(ORC_E_COMPRESSION_METHOD, cm, 0);
        uint32_t e = getbits(&in, b + 4, 4);
        if (e >= 8) FAIL(ORC_E_WINDOW_SIZE, e + 8, 0);
        uint32_t flags = getbits(&in, b + 8, 8);
        if (((e << 12 | 8 << 8) + flags) % 31 != 0) FAIL(ORC_E_CHECK_BITS, 0, 0);
        if (flags & 0x20) FAIL(ORC_E_DICTIONARY, 0, 0);
        b += 16;
    }

    for (;;) {
        /* .metadata: readBlockMetadata (InflatorBuffers.Stream.swift:59-141) */
        if (b + 3 > in.count) goto out;
        int final = (int)getbits(&in, b, 1);
        uint32_t type = getbits(&in, b + 1, 2);
        if (type == 0) {
            size_t boundary = (b + 3 + 7) & ~(size_t)7;
            if (boundary + 32 > in.count) goto out;
            uint32_t l = getbits(&in, boundary, 16), m = getbits(&in, boundary + 16, 16);
            if (l != (~m & 0xffff)) FAIL(ORC_E_BLOCK_COUNT_PARITY, l, m);
            b = boundary + 32;
            /* readBlock(upTo:) (:384-399): byte at a time until input runs dry */
            size_t stop = end + l;
            while (end < stop) {
                if (b + 8 > in.count) goto out;
                if (end >= cap) FAIL(ORC_E_OUTPUT_CAPACITY, 0, 0);
                dst[end++] = src[b >> 3];
                b += 8;
            }
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                b += 3;
                uint8_t ll[288], dl[32];
                fixed_lengths(ll, dl);
                ht_build(&T->lit, ll, 288);
                ht_build(&T->dist, dl, 32);
            } else {
                if (b + 17 > in.count) goto out;
                int codelengths = 4 + (int)getbits(&in, b + 13, 4);
                if (b + 17 + 3 * (size_t)codelengths > in.count) goto out;
                int literals = 257 + (int)getbits(&in, b + 3, 5);
                int distances = 1 + (int)getbits(&in, b + 8, 5);
                if (literals > 286) FAIL(ORC_E_RUNLITERAL_COUNT, literals, 0);
                static const uint8_t ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t ml[19] = {0};
                for (int i = 0; i < codelengths; ++i)
                    ml[ORDER[i]] = (uint8_t)getbits(&in, b + 17 + 3 * (size_t)i, 3);
                if (!ht_build(&T->meta, ml, 19)) FAIL(ORC_E_CODELENGTH_TABLE, 0, 0);
                b += 17 + 3 * (size_t)codelengths;

                /* .tables: readBlockTables (:144-263) */
                int total = literals + distances, have = 0;
                uint8_t lengths[286 + 32 + 138];
                while (have < total) {
                    if (b >= in.count) goto out;
                    hent mw = ht_lookup(&T->meta, (uint16_t)(peek(&in, b) & 0xff));
                    if (b + mw.len > in.count) goto out;
                    if (mw.sym < 16) { lengths[have++] = (uint8_t)mw.sym; b += mw.len; continue; }
                    int element, extra, base;
                    if (mw.sym == 16) {
                        if (!have) FAIL(ORC_E_CODELENGTH_SEQUENCE, 0, 0);
                        element = lengths[have - 1]; extra = 2; base = 3;
                    } else if (mw.sym == 17) { element = 0; extra = 3; base = 3; }
                    else                      { element = 0; extra = 7; base = 11; }
                    if (b + mw.len + extra > in.count) goto out;
                    int reps = base + (int)getbits(&in, b + mw.len, extra);
                    for (int r = 0; r < reps; ++r) lengths[have++] = (uint8_t)element;
                    b += mw.len + extra;
                }
                if (have != total) FAIL(ORC_E_CODELENGTH_SEQUENCE, 0, 0);
                if (!ht_build(&T->lit, lengths, literals) ||
                    !ht_build_normalizing(&T->dist, lengths + literals, distances))
                    FAIL(ORC_E_HUFFMAN_TABLE, 0, 0);
            }
            /* .compressed: readBlock(with:) (:266-381) */
            for (;;) {
                if (b >= in.count) goto out;
                uint64_t slug = peek(&in, b);
                hent rl = ht_lookup(&T->lit, (uint16_t)slug);
                if (rl.sym < 256) {
                    if (b + rl.len > in.count) goto out;
                    if (end >= cap) FAIL(ORC_E_OUTPUT_CAPACITY, 0, 0);
                    b += rl.len;
                    dst[end++] = (uint8_t)rl.sym;
                } else if (rl.sym == 256) {
                    if (b + rl.len > in.count) goto out;
                    b += rl.len;
                    break;
                } else {
                    slug >>= rl.len;
                    int decade = rl.sym & 0xff;
                    int cx = RUN_EXTRA[decade & 31];
                    size_t count = RUN_BASE[decade & 31] + (size_t)(slug & ((1ull << cx) - 1));
                    slug >>= cx;
                    hent d = ht_lookup(&T->dist, (uint16_t)slug);
                    if (d.len == 0) FAIL(ORC_E_REFERENCE_UNDEFINED, 0, 0);
                    slug >>= d.len;
                    int ox = DIST_EXTRA[d.sym & 31];
                    size_t offset = DIST_BASE[d.sym & 31] + (size_t)(slug & ((1ull << ox) - 1));
                    size_t nb = b + rl.len + cx + d.len + ox;
                    if (nb > in.count) goto out;
                    if (offset > end) FAIL(ORC_E_STRING_REFERENCE, 0, 0);
                    if (count && !offset) FAIL(ORC_E_REFERENCE_UNDEFINED, 0, 0);
                    if (end + count > cap) FAIL(ORC_E_OUTPUT_CAPACITY, 0, 0);
                    /* InflatorOut.expand (:124-139): forward byte copy, overlap replicates */
                    for (size_t i = 0; i < count; ++i) dst[end + i] = dst[end + i - offset];
                    end += count;
                    b = nb;
                }
            }
        } else {
            FAIL(ORC_E_BLOCK_TYPE, type, 0);
        }
        if (final) break;
    }

    /* .checksum (InflatorBuffers.swift:112-130; Stream.swift:402-429) */
    if (format != ORC_FORMAT_IOS) {
        size_t boundary = (b + 7) & ~(size_t)7;
        if (boundary + 32 > in.count) goto out;
        b = boundary + 32;
        const uint8_t *q = src + (boundary >> 3);
        uint32_t declared = (uint32_t)q[0] << 24 | (uint32_t)q[1] << 16 | (uint32_t)q[2] << 8 | q[3];
        uint32_t computed = orc_adler32(1, dst, end);
        if (declared != computed) FAIL(ORC_E_STREAM_CHECKSUM, declared, computed);
    }
    status = ORC_DONE;
out:
    if (written) *written = end;
    if (consumed) *consumed = (b + 7) >> 3;
    free(T);
    return status;
}
