/*
 * oracle/deflate.c -- CPU ORACLE (test infrastructure, never shipped).
 * Restates swift-png's LZ77.Deflator so that its DEFLATE bitstream can be reproduced bit for bit.
 *
 *   level table          Sources/LZ77/Deflator/LZ77.DeflatorSearch.swift:13-35
 *   push / final tail    Sources/LZ77/Deflator/LZ77.DeflatorBuffers.swift:46-93
 *   compress loops       Sources/LZ77/Deflator/LZ77.DeflatorBuffers.Stream.swift:30-404
 *   block writer         Sources/LZ77/Deflator/LZ77.DeflatorBuffers.Stream.swift:406-709
 *   window / hash chain  Sources/LZ77/Deflator/LZ77.DeflatorWindow.swift:27-212
 *   terms / graph        Sources/LZ77/Deflator/LZ77.DeflatorMatches.swift:55-380
 *   cost model           Sources/LZ77/Deflator/LZ77.DeflatorMatches.Depths.swift:4-112
 *   term packing         Sources/LZ77/Deflator/LZ77.DeflatorTerm.swift:10-56, LZ77.DeflatorTerm.Meta.swift
 *   decades              Sources/LZ77/Deflator/LZ77.Decades.swift
 *   tree construction    Sources/LZ77/HuffmanCoding/LZ77.HuffmanTree.swift:204-404, LZ77.Heap.swift
 *   codewords            Sources/LZ77/HuffmanCoding/LZ77.Codeword.swift:19-33
 *   stream header        Sources/LZ77/Inflator/LZ77.StreamHeader.swift:56-62
 *
 * F14.HashTable (Sources/LZ77/F14) is an exact UInt32 -> UInt16 map (LZ77Tests/HardwareAcceleration.swift
 * asserts Dictionary equivalence); any exact map reproduces it, here open addressing with
 * backward-shift deletion.
 *
 * Pin: level 9 reproduces the 28 streams swift-png itself committed under Tests/Outputs
 * (tests/test_oracle_encode.py).  Other levels have no golden stream in the reference: parity
 * unpinned beyond round trips through the (pinned) inflate oracle and zlib.
 */
#include "spng_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---- LZ77.Composites.swift:25-110 ---- */
static const uint16_t RUN_EXTRA[32] = {0, 0,0,0,0,0, 0,0,0,1,1, 1,1,2,2,2, 2,3,3,3,3, 4,4,4,4,5, 5,5,5,0, 0,0};
static const uint16_t RUN_BASE [32] = {0, 3,4,5,6,7, 8,9,10,11,13, 15,17,19,23,27, 31,35,43,51,59,
                                       67,83,99,115,131, 163,195,227,258, 0,0};
static const uint16_t DIST_EXTRA[32] = {0,0,0,0,1, 1,2,2,3,3, 4,4,5,5,6, 6,7,7,8,8, 9,9,10,10,11,
                                        11,12,12,13,13, 0,0};
static const uint16_t DIST_BASE [32] = {1,2,3,4,5, 7,9,13,17,25, 33,49,65,97,129, 193,257,385,513,769,
                                        1025,1537,2049,3073,4097, 6145,8193,12289,16385,24577, 0,0};

/* LZ77.Decades (LZ77.Decades.swift): run 3...258 -> 1...29, distance 1...32768 -> 0...29 */
static uint8_t RUN_DECADE[259], DIST_DECADE_LO[257], DIST_DECADE_HI[257];
static void init_decades(void)
{
    static int done = 0;
    if (done) return;
    for (int dec = 1; dec <= 29; ++dec) {
        int lo = RUN_BASE[dec], hi = dec == 29 ? 258 : RUN_BASE[dec] + (1 << RUN_EXTRA[dec]) - 1;
        for (int r = lo; r <= hi; ++r) RUN_DECADE[r] = (uint8_t)dec;
    }
    RUN_DECADE[258] = 29;                      /* table row "28 x31, 29": 258 belongs to 285 */
    for (int dec = 0; dec <= 29; ++dec) {
        int lo = DIST_BASE[dec], hi = DIST_BASE[dec] + (1 << DIST_EXTRA[dec]) - 1;
        for (int d = lo; d <= hi; ++d) {
            if (d <= 256) DIST_DECADE_LO[d] = (uint8_t)dec;
            else DIST_DECADE_HI[(d - 1) >> 7] = (uint8_t)dec;
        }
    }
    done = 1;
}
static inline int run_decade(int run) { return RUN_DECADE[run]; }
static inline int dist_decade(int d) { return d <= 256 ? DIST_DECADE_LO[d] : DIST_DECADE_HI[(d - 1) >> 7]; }

/* ---- exact map UInt32 -> UInt16 (F14.HashTable semantics) ---- */
typedef struct { uint32_t *key; int32_t *val; uint32_t cap, mask; } hmap;
static inline uint32_t hmix(uint32_t k) { k *= 0x9E3779B1u; return k ^ (k >> 15); }
static void hmap_init(hmap *h, uint32_t cap)
{
    h->cap = cap; h->mask = cap - 1;
    h->key = (uint32_t *)calloc(cap, 4);
    h->val = (int32_t *)malloc(cap * 4);
    for (uint32_t i = 0; i < cap; ++i) h->val[i] = -1;
}
static void hmap_free(hmap *h) { free(h->key); free(h->val); }
/* F14.HashTable.update (F14.HashTable.swift:173): returns the previous value or -1 */
static int32_t hmap_update(hmap *h, uint32_t key, int32_t value)
{
    uint32_t i = hmix(key) & h->mask;
    while (h->val[i] >= 0) {
        if (h->key[i] == key) { int32_t old = h->val[i]; h->val[i] = value; return old; }
        i = (i + 1) & h->mask;
    }
    h->key[i] = key; h->val[i] = value;
    return -1;
}
/* F14.HashTable.remove(key:value:) (:113): only if the entry still holds `value` */
static void hmap_remove(hmap *h, uint32_t key, int32_t value)
{
    uint32_t i = hmix(key) & h->mask;
    while (h->val[i] >= 0) {
        if (h->key[i] == key) {
            if (h->val[i] != value) return;
            uint32_t j = i;                              /* backward-shift deletion */
            for (;;) {
                j = (j + 1) & h->mask;
                if (h->val[j] < 0) break;
                uint32_t home = hmix(h->key[j]) & h->mask;
                if (((j - home) & h->mask) >= ((j - i) & h->mask)) {
                    h->key[i] = h->key[j]; h->val[i] = h->val[j]; i = j;
                }
            }
            h->val[i] = -1;
            return;
        }
        i = (i + 1) & h->mask;
    }
}

/* ---- the deflator ---- */
enum { GREEDY, LAZY, FULL };

typedef struct {
    int format, kind, goal, iterations;
    long attempts;
    /* LZ77.DeflatorIn */
    uint8_t *in; int64_t in_len, in_cap, start;
    /* LZ77.DeflatorWindow */
    int64_t end_index; uint32_t w, v; int mask;
    int32_t *next; uint8_t *val; hmap head;
    /* LZ77.DeflatorMatches */
    uint32_t *store; int64_t store_words;
    int count, limit, capacity;
    uint8_t depths[542]; int generic;
    /* LZ77.DeflatorOut (only the concatenated bitstream matters) */
    uint8_t *out; size_t out_len, out_cap; uint64_t acc; int nacc; int overflow;
} deflator;

/* bits LSB-first (LZ77.DeflatorOut.append, :104-145) */
static void put(deflator *d, uint32_t bits, int count)
{
    d->acc |= (uint64_t)(bits & ((1u << count) - 1)) << d->nacc;
    d->nacc += count;
    while (d->nacc >= 8) {
        if (d->out_len < d->out_cap) d->out[d->out_len] = (uint8_t)d->acc; else d->overflow = 1;
        d->out_len++; d->acc >>= 8; d->nacc -= 8;
    }
}
static void pad_to_byte(deflator *d) { if (d->nacc) put(d, 0, 8 - d->nacc); }

static uint8_t DEPTH_DEFAULT[542];
static void init_depths(void)
{
    /* Depths.default (Depths.swift:31-44): 1/4-bit units; literal 8.25 b, run 7.5 b + extra, distance 4.75 b + extra */
    for (int i = 0; i < 256; ++i) DEPTH_DEFAULT[i] = 33;
    for (int run = 3; run <= 258; ++run) DEPTH_DEFAULT[253 + run] = (uint8_t)(30 + (RUN_EXTRA[run_decade(run)] << 2));
    for (int dec = 0; dec < 30; ++dec) DEPTH_DEFAULT[512 + dec] = (uint8_t)(19 + (DIST_EXTRA[dec] << 2));
}

static inline int64_t in_count(const deflator *d) { return d->in_len - d->start; }
static inline uint8_t dequeue(deflator *d)
{
    /* reads past the end return whatever the Swift buffer holds; the value never reaches the output */
    uint8_t v = d->start < d->in_len ? d->in[d->start] : 0;
    d->start++;
    return v;
}
static inline int unfilled(const deflator *d) { return d->limit - 1 - d->count; }
static inline uint8_t win_literal(const deflator *d) { return (uint8_t)(d->v >> 24); }

static void win_initialize(deflator *d, uint8_t v) { d->v = d->v << 8 | v; d->end_index++; }   /* :59-74 */

/* LZ77.DeflatorWindow.update (:78-113) */
static void win_update(deflator *d, uint8_t byte, int *index, int32_t *nxt)
{
    int a = (int)(d->end_index & d->mask), b = (int)((d->end_index + 3) & d->mask);
    d->w = d->w << 8 | d->val[b];
    d->v = d->v << 8 | byte;
    if (d->end_index > d->mask) hmap_remove(&d->head, d->w, a);
    int32_t n = hmap_update(&d->head, d->v, a);
    d->next[a] = n; d->val[a] = win_literal(d);
    d->end_index++;
    if (index) { *index = a; *nxt = n; }
}

/* graph helpers (DeflatorMatches.swift:162-223) */
static void graph_reserve(deflator *d, int64_t vertices)
{
    int64_t need = vertices * 32;
    if (need <= d->store_words) return;
    int64_t cap = d->store_words ? d->store_words : 2048 * 32;
    while (cap < need) cap *= 2;
    d->store = (uint32_t *)realloc(d->store, (size_t)cap * 4);
    memset(d->store + d->store_words, 0, (size_t)(cap - d->store_words) * 4);
    d->store_words = cap;
}
static int store_vertex(deflator *d, uint8_t literal)
{
    graph_reserve(d, (int64_t)d->count + 2);
    int base = d->count << 5;
    d->store[base] = literal;
    d->store[base | 1] = 0xffffffffu;
    for (int o = 2; o < 32; ++o) d->store[base | o] = 0;
    d->count++;
    return base | 2;
}
static void set_edge(deflator *d, int run, int distance, int base)
{
    int pos = base + dist_decade(distance);
    if ((uint32_t)run > (d->store[pos] & 0xffff)) d->store[pos] = (uint32_t)distance << 16 | (uint32_t)run;
}
static void store_literal(deflator *d, uint8_t lit) { d->store[d->count++] = 0xf8000000u | lit; }
static void store_match(deflator *d, int run, int distance)
{
    /* LZ77.DeflatorTerm.init(run:distance:) (DeflatorTerm.swift:34-56) */
    int rd = run_decade(run), dd = dist_decade(distance);
    d->store[d->count++] = (uint32_t)dd << 27 | 0x100u | (uint32_t)rd |
        (uint32_t)(distance - DIST_BASE[dd]) << 14 | (uint32_t)(run - RUN_BASE[rd]) << 9;
}

/* LZ77.DeflatorWindow.match (:132-212).  mode 0: best candidate (first strictly longest, run > 5,
 * :115-130) into brun and bdist, returns 1 if found.  mode 1: every candidate becomes a graph edge,
 * returns the longest run seen (>= 1). */
static int win_match(deflator *d, int head_index, int32_t head_next, long attempts, int goal,
                     int mode, int edge_base, int *brun, int *bdist)
{
    int best_run = 5, best_dist = 1, extent = 1;
    if (head_next >= 0) {
        const uint8_t *v = d->in + d->start - 4;             /* lookahead pointer at offset -4 (DeflatorIn.swift:213-221) */
        int64_t la = in_count(d);
        int limit = (int)(la + 4 < 258 ? la + 4 : 258);
        int mask = d->mask;
        int current = head_next;
        int distance = (head_index - current) & mask;
        long remaining = attempts;
        for (;;) {
            int run = 4;
            int a = distance < limit ? distance : limit;
            int broke = 0;
            while (run < a) {
                if (d->val[(current + run) & mask] != v[run]) { broke = 1; break; }
                run++;
            }
            if (!broke) {
                int i = 4 - distance > 0 ? 4 - distance : 0;
                while (run < limit && v[i] == v[run]) { i++; run++; }
            }
            if (mode == 0) { if (best_run < run) { best_run = run; best_dist = distance; } }
            else { if (run > extent) extent = run; set_edge(d, run, distance, edge_base); }
            remaining--;
            if (!(remaining > 0 && goal > run)) break;
            int32_t nx = d->next[current];
            if (nx < 0) break;
            int previous = current;
            current = nx;
            distance += (previous - current) & mask;
            if (!(distance < mask)) break;
        }
    }
    if (mode == 0) { if (best_run > 5) { *brun = best_run; *bdist = best_dist; return 1; } return 0; }
    return extent;
}

/* ---- Huffman tree construction (HuffmanTree.swift:247-404, Heap.swift) ---- */
typedef struct { long key; int n; int *lv; } hnode;          /* lv[n-1] is the root level */

static int heap_lowest(hnode *h, int count, int parent)      /* Heap.lowest(below:) (:94-111), 1-based */
{
    int r = (parent << 1) + 1, l = parent << 1;
    if (l >= count + 1) return 0;
    if (r >= count + 1) return h[l - 1].key < h[parent - 1].key ? l : 0;
    int c = h[r - 1].key < h[l - 1].key ? r : l;
    return h[c - 1].key < h[parent - 1].key ? c : 0;
}
static void heap_sift_down(hnode *h, int count, int i)
{
    for (;;) {
        int c = heap_lowest(h, count, i);
        if (!c) return;
        hnode t = h[i - 1]; h[i - 1] = h[c - 1]; h[c - 1] = t;
        i = c;
    }
}
static void heap_sift_up(hnode *h, int i)
{
    for (;;) {
        int p = i >> 1;
        if (p < 1 || !(h[i - 1].key < h[p - 1].key)) return;
        hnode t = h[i - 1]; h[i - 1] = h[p - 1]; h[p - 1] = t;
        i = p;
    }
}
static hnode heap_dequeue(hnode *h, int *count)              /* Heap.dequeue (:149-164) */
{
    if (*count == 1) { (*count)--; return h[0]; }
    hnode t = h[0]; h[0] = h[*count - 1]; h[*count - 1] = t;
    hnode out = h[*count - 1];
    (*count)--;
    heap_sift_down(h, *count, 1);
    return out;
}

/* HuffmanTree.limitHeight (:348-404) */
static int limit_height(int *levels, int n, int height)
{
    if (n <= height) return n;
    int unhoused = 0;
    for (int l = n - 1; l >= height; --l) {
        int pairs = levels[l] >> 1;
        unhoused += pairs;
        levels[l - 1] += pairs;
    }
    n = height;
    int split = height - 2;
    while (unhoused > 0) {
        if (!(levels[split] > 0)) { split--; continue; }
        int resettled = levels[split] < unhoused ? levels[split] : unhoused;
        unhoused -= resettled;
        levels[split] -= resettled;
        levels[split + 1] += 2 * resettled;
        if (split < height - 2) split++;
    }
    return n;
}

/* HuffmanTree.init(frequencies:limit:) -> code length per symbol (0 = unused) */
static void build_tree(const long *freq, int n, int limit, uint8_t *lengths)
{
    int symbols[320], m = 0;
    memset(lengths, 0, (size_t)n);
    for (int i = 0; i < n; ++i) if (freq[i] > 0) symbols[m++] = i;
    /* descending frequency, stable (ties keep ascending symbol order) */
    for (int i = 1; i < m; ++i) {
        int s = symbols[i], j = i;
        while (j > 0 && freq[symbols[j - 1]] < freq[s]) { symbols[j] = symbols[j - 1]; --j; }
        symbols[j] = s;
    }
    if (m <= 1) { if (m == 1) lengths[symbols[0]] = 1; return; }   /* stub tree (:52-65) */

    int stride = m + 2;
    int *pool = (int *)malloc(sizeof(int) * (size_t)stride * (size_t)(2 * m));
    hnode *heap = (hnode *)malloc(sizeof(hnode) * (size_t)m);
    int used = 0, count = m;
    for (int i = 0; i < m; ++i) {                            /* symbols.reversed(): ascending frequency */
        int s = symbols[m - 1 - i];
        heap[i].key = freq[s]; heap[i].n = 1; heap[i].lv = pool + stride * used++; heap[i].lv[0] = 1;
    }
    for (int i = ((count) >> 1); i >= 1; --i) heap_sift_down(heap, count, i);   /* heapify (:166-175) */

    int levels[320], nl = 0;
    for (;;) {
        hnode first = heap_dequeue(heap, &count);
        if (count == 0) {
            /* drop the root level, reverse: levels[0] = leaves at depth 1 */
            nl = first.n - 1;
            for (int i = 0; i < nl; ++i) levels[i] = first.lv[nl - 1 - i];
            break;
        }
        hnode second = heap_dequeue(heap, &count);
        hnode *big = first.n > second.n ? &first : &second, *small = first.n > second.n ? &second : &first;
        int *lv = pool + stride * used++;
        memcpy(lv, big->lv, sizeof(int) * (size_t)big->n);
        for (int k = 0; k < small->n; ++k) lv[big->n - 1 - k] += small->lv[small->n - 1 - k];
        lv[big->n] = 0;
        hnode merged = { first.key + second.key, big->n + 1, lv };
        heap[count++] = merged;
        heap_sift_up(heap, count);
    }
    nl = limit_height(levels, nl, limit);
    int at = 0;
    for (int l = 0; l < nl; ++l)
        for (int k = 0; k < levels[l]; ++k) lengths[symbols[at++]] = (uint8_t)(l + 1);
    free(pool); free(heap);
}

/* canonical codewords, bit-reversed for LSB-first emission (HuffmanTree.codewords :206-230, Codeword.swift:19-33) */
static void codewords(const uint8_t *lengths, int n, uint16_t *bits)
{
    uint32_t counter = 0;
    for (int len = 1; len <= 15; ++len) {
        for (int s = 0; s < n; ++s) if (lengths[s] == len) {
            uint32_t r = 0;
            for (int k = 0; k < len; ++k) if (counter >> k & 1) r |= 1u << (len - 1 - k);
            bits[s] = (uint16_t)r;
            counter++;
        }
        counter <<= 1;
    }
}

/* ---- full search: shortest path over the match graph (DeflatorMatches.swift:225-379) ---- */
static void explore(deflator *d, int index)
{
    uint32_t *g = d->store;
    uint32_t cur_up = g[index << 5], cur_depth = g[index << 5 | 1];
    uint32_t next_up = g[(index + 1) << 5], next_depth = g[(index + 1) << 5 | 1];
    uint8_t lit = (uint8_t)cur_up;
    uint32_t ldepth = cur_depth + d->depths[lit];
    if (ldepth < next_depth) {
        g[(index + 1) << 5] = 0x0001ff00u | (next_up & 0xff);
        g[(index + 1) << 5 | 1] = ldepth;
    }
    int remaining = d->count - index;
    if (remaining < 3) return;
    for (int decade = 0; decade < 30; ++decade) {
        int maxlen = (int)(g[index << 5 | (decade + 2)] & 0xffff);
        if (maxlen > remaining) maxlen = remaining;
        if (maxlen <= 0) continue;
        uint32_t base = cur_depth + d->depths[512 + decade];
        for (int length = 3; length <= maxlen; ++length) {
            uint32_t depth = base + d->depths[253 + length];
            uint32_t *t = g + ((index + length) << 5);
            if (!(depth < t[1])) continue;
            t[0] = (uint32_t)length << 16 | (uint32_t)decade << 8 | (t[0] & 0xff);
            t[1] = depth;
        }
    }
}

static void minimize(deflator *d, long *freq /* 318 */)
{
    uint32_t *g = d->store;
    g[0 << 5 | 1] = 0;
    g[d->count << 5 | 1] = 0xffffffffu;
    for (int node = 0; node < d->count; ++node) explore(d, node);
    memset(freq, 0, sizeof(long) * 318);
    int cur = d->count;
    uint32_t cur_up = g[cur << 5];
    do {
        int length = (int)(cur_up >> 16);
        int nxt = cur - length;
        uint32_t nxt_up = g[nxt << 5];
        g[nxt << 5] = (cur_up & 0xffffff00u) | (nxt_up & 0xff);
        if (length == 1) freq[nxt_up & 0xff]++;
        else { freq[256 | run_decade(length)]++; freq[288 + ((cur_up >> 8) & 0xff)]++; }
        cur = nxt; cur_up = nxt_up;
    } while (cur > 0);
    freq[256] = 1;
}

/* Depths.update (Depths.swift:53-86) */
static void depths_update(deflator *d, const uint8_t *ll, const uint8_t *dl)
{
    for (int len = 1; len <= 15; ++len)
        for (int s = 0; s < 286; ++s) if (ll[s] == len) {
            if (s < 256) d->depths[s] = (uint8_t)(len << 2);
            else if (s > 256) {
                int dec = s & 0xff;
                int l2 = len + RUN_EXTRA[dec], base = 253 + RUN_BASE[dec], cnt = 1 << RUN_EXTRA[dec];
                for (int l = base; l < base + cnt; ++l) d->depths[l] = (uint8_t)(l2 << 2);
            }
        }
    for (int len = 1; len <= 15; ++len)
        for (int s = 0; s < 30; ++s) if (dl[s] == len) d->depths[512 + s] = (uint8_t)((len + DIST_EXTRA[s]) << 2);
    d->generic = 0;
}
static void depths_generalize(deflator *d)
{
    for (int i = 0; i < 542; ++i) {
        uint8_t s = d->depths[i], g = DEPTH_DEFAULT[i];
        d->depths[i] = (uint8_t)((s & g) + ((s ^ g) >> 1));
    }
}

/* ---- block writer (DeflatorBuffers.Stream.swift:440-709) ---- */
static void write_block(deflator *d, int final)
{
    uint8_t ll[288], dl[32], ml[19];
    long freq[320];
    memset(ll, 0, sizeof ll); memset(dl, 0, sizeof dl);
    if (d->kind != FULL) {
        /* DeflatorMatches.trees() (:138-159) */
        memset(freq, 0, sizeof freq);
        for (int i = 0; i < d->count; ++i) {
            uint32_t t = d->store[i];
            freq[t & 0x1ff]++; freq[288 + (t >> 27)]++;
        }
        freq[256] = 1;
        build_tree(freq, 286, 15, ll);
        build_tree(freq + 288, 30, 15, dl);
    } else {
        /* DeflatorMatches.trees(iterations:) (:225-260) */
        d->limit = 2 * d->limit < d->capacity ? 2 * d->limit : d->capacity;
        int i = d->generic ? -d->iterations : 0;
        for (;;) {
            minimize(d, freq);
            build_tree(freq, 286, 15, ll);
            build_tree(freq + 288, 30, 15, dl);
            i++;
            if (!(i < d->iterations)) break;
            depths_update(d, ll, dl);
            for (int k = 0; k < d->count; ++k) d->store[k << 5 | 1] = 0xffffffffu;
        }
    }

    uint8_t lengths[318];
    memset(lengths, 0, sizeof lengths);
    memcpy(lengths, ll, 286);
    int r = 286; while (r > 0 && lengths[r - 1] == 0) --r;
    if (r < 257) r = 257;
    for (int s = 0; s < 30; ++s) if (dl[s]) lengths[r + s] = dl[s];
    int dn = 32; while (dn > 0 && (r + dn - 1 >= 318 || lengths[r + dn - 1] == 0)) --dn;
    if (dn < 1) dn = 1;

    /* code-length RLE (:483-543) */
    uint8_t msym[320], mbits[320]; int nm = 0;
    {
        int reps = 1; uint8_t last = lengths[0];
        for (int idx = 1; ; ++idx) {
            int have = idx < r + dn;
            uint8_t cur = have ? lengths[idx] : 0;
            if (have && cur == last) { reps++; continue; }
            if (last == 0) {
                while (reps > 138) { msym[nm] = 18; mbits[nm++] = 138 - 11; reps -= 138; }
                if (reps > 2) { if (reps < 11) { msym[nm] = 17; mbits[nm++] = (uint8_t)(reps - 3); }
                                else { msym[nm] = 18; mbits[nm++] = (uint8_t)(reps - 11); } }
                else for (int k = 0; k < reps; ++k) { msym[nm] = 0; mbits[nm++] = 0; }
            } else {
                msym[nm] = last; mbits[nm++] = 0; reps -= 1;
                while (reps > 6) { msym[nm] = 16; mbits[nm++] = 6 - 3; reps -= 6; }
                if (reps > 2) { msym[nm] = 16; mbits[nm++] = (uint8_t)(reps - 3); }
                else for (int k = 0; k < reps; ++k) { msym[nm] = last; mbits[nm++] = 0; }
            }
            if (!have) break;
            last = cur; reps = 1;
        }
    }
    long mfreq[19] = {0};
    for (int k = 0; k < nm; ++k) mfreq[msym[k]]++;
    build_tree(mfreq, 19, 7, ml);

    /* writeBlockMetadata (:577-612) */
    static const int ZPOS[19] = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2};
    uint8_t cl[19] = {0};
    for (int s = 0; s < 19; ++s) if (ml[s]) cl[ZPOS[s]] = ml[s];
    int ncl = 19; while (ncl > 0 && cl[ncl - 1] == 0) --ncl;
    if (ncl < 4) ncl = 4;
    put(d, final ? 5 : 4, 3);
    put(d, (uint32_t)(r - 257), 5);
    put(d, (uint32_t)(dn - 1), 5);
    put(d, (uint32_t)(ncl - 4), 4);
    for (int k = 0; k < ncl; ++k) put(d, cl[k], 3);

    uint16_t lbits[288], dbits[32], mcode[19];
    memset(lbits, 0, sizeof lbits); memset(dbits, 0, sizeof dbits); memset(mcode, 0, sizeof mcode);
    codewords(ll, 288, lbits); codewords(dl, 32, dbits); codewords(ml, 19, mcode);

    /* writeBlockTables (:615-623) */
    for (int k = 0; k < nm; ++k) {
        put(d, mcode[msym[k]], ml[msym[k]]);
        int extra = msym[k] == 18 ? 7 : msym[k] == 17 ? 3 : msym[k] == 16 ? 2 : 0;
        put(d, mbits[k], extra);
    }

    /* writeBlock(with:) (:626-709) */
    if (d->kind != FULL) {
        for (int i = 0; i < d->count; ++i) {
            uint32_t t = d->store[i];
            int sym = (int)(t & 0x1ff), dsym = (int)(t >> 27);
            put(d, lbits[sym], ll[sym]);
            if (sym > 256) {
                put(d, (t >> 9) & 0x1f, RUN_EXTRA[sym & 0xff]);
                put(d, dbits[dsym], dl[dsym]);
                put(d, (t >> 14) & 0x1fff, DIST_EXTRA[dsym]);
            }
        }
        put(d, lbits[256], ll[256]);
        d->count = 0;                                        /* resetTerms */
    } else {
        int index = 0;
        while (index < d->count) {
            uint32_t up = d->store[index << 5];
            int count = (int)(up >> 16);
            if (count == 1) {
                int lit = (int)(up & 0xff);
                put(d, lbits[lit], ll[lit]);
            } else {
                int rd = run_decade(count), dd = (int)((up >> 8) & 0xff);
                int offset = (int)(d->store[index << 5 | (2 + dd)] >> 16);
                put(d, lbits[256 | rd], ll[256 | rd]);
                put(d, (uint32_t)(count - RUN_BASE[rd]), RUN_EXTRA[rd]);
                put(d, dbits[dd], dl[dd]);
                put(d, (uint32_t)(offset - DIST_BASE[dd]), DIST_EXTRA[dd]);
            }
            index += count;
        }
        put(d, lbits[256], ll[256]);
        d->count = 0;                                        /* resetGraph */
        depths_generalize(d);
    }
}

/* Stream.compress (:64-404): 1 = "()" (term buffer full, write a block), 0 = nil */
static int compress(deflator *d, int all)
{
    int index; int32_t nxt;
    if (d->kind == GREEDY) {
        int64_t lookahead = all ? 0 : 258;
        while (d->end_index < 0 && in_count(d) > lookahead) win_initialize(d, dequeue(d));
        while (in_count(d) > lookahead) {
            if (!(unfilled(d) > 0)) return 1;
            win_update(d, dequeue(d), &index, &nxt);
            int run, dist;
            if (win_match(d, index, nxt, d->attempts, d->goal, 0, 0, &run, &dist)) {
                for (int k = 1; k < run; ++k) win_update(d, dequeue(d), NULL, NULL);
                store_match(d, run, dist);
            } else store_literal(d, win_literal(d));
        }
    } else if (d->kind == LAZY) {
        int64_t lookahead = all ? 0 : 259;
        while (d->end_index < 0 && in_count(d) > lookahead) win_initialize(d, dequeue(d));
        while (in_count(d) > lookahead) {
            if (!(unfilled(d) > 1)) return 1;
            win_update(d, dequeue(d), &index, &nxt);
            uint8_t first = win_literal(d);
            int erun, edist;
            if (win_match(d, index, nxt, d->attempts, d->goal, 0, 0, &erun, &edist)) {
                win_update(d, dequeue(d), &index, &nxt);
                int lrun, ldist;
                if (win_match(d, index, nxt, d->attempts, d->goal, 0, 0, &lrun, &ldist) && erun < lrun) {
                    store_literal(d, first);
                    store_match(d, lrun, ldist);
                    for (int k = 1; k < lrun; ++k) win_update(d, dequeue(d), NULL, NULL);
                } else {
                    store_match(d, erun, edist);
                    for (int k = 2; k < erun; ++k) win_update(d, dequeue(d), NULL, NULL);
                }
            } else store_literal(d, first);
        }
    } else {
        int64_t lookahead = all ? 0 : 258;
        while (d->end_index < 0 && in_count(d) > lookahead) win_initialize(d, dequeue(d));
        while (in_count(d) > lookahead) {
            if (!(unfilled(d) > 0)) return 1;
            win_update(d, dequeue(d), &index, &nxt);
            int base = store_vertex(d, win_literal(d));
            int extent = win_match(d, index, nxt, d->attempts, d->goal, 1, base, NULL, NULL);
            int skip = extent - 100 < unfilled(d) ? extent - 100 : unfilled(d);
            for (int k = 0; k < skip; ++k) {
                win_update(d, dequeue(d), NULL, NULL);
                store_vertex(d, win_literal(d));
            }
        }
    }
    if (!all) return 0;
    /* epilogue: the three positions still in the window pipeline (:254-265) */
    int64_t epilogue = -3 - (d->end_index < 0 ? d->end_index : 0);
    while (in_count(d) > epilogue) {
        if (!(unfilled(d) > 0)) return 1;
        win_update(d, dequeue(d), NULL, NULL);
        if (d->kind == FULL) store_vertex(d, win_literal(d)); else store_literal(d, win_literal(d));
    }
    return 0;
}

/* Stream.compressBlocks (:30-61) */
static void compress_blocks(deflator *d, int final)
{
    if (!final) { while (compress(d, 0)) write_block(d, 0); return; }
    int64_t count = in_count(d);
    if (count >= 3) {
        while (compress(d, 1)) write_block(d, 0);
        write_block(d, 1);
    } else {
        /* stored tail (:417-434).  NB: this also fires when a non-final compress() left fewer than
         * three bytes queued, dropping the pending terms -- the reference's latent bug, kept. */
        put(d, 1, 3);
        pad_to_byte(d);
        uint32_t l = (uint32_t)count & 0xffff;
        put(d, l, 16); put(d, ~l & 0xffff, 16);
        for (int64_t k = 0; k < count; ++k) put(d, dequeue(d), 8);
    }
}

static deflator *deflator_new(int format, int level, int exponent, uint8_t *out, size_t cap)
{
    init_decades(); init_depths();
    deflator *d = (deflator *)calloc(1, sizeof *d);
    d->format = format;
    /* DeflatorSearch.init(level:) (:13-35) */
    static const struct { int kind; long attempts; int goal, iterations; } T[14] = {
        {GREEDY, 1, 6, 0}, {GREEDY, 2, 8, 0}, {GREEDY, 4, 10, 0}, {GREEDY, 40, 24, 0},
        {LAZY, 20, 32, 0}, {LAZY, 40, 54, 0}, {LAZY, 64, 80, 0}, {LAZY, 100, 160, 0},
        {FULL, 14, 20, 1}, {FULL, 20, 32, 2}, {FULL, 30, 50, 3}, {FULL, 60, 80, 4}, {FULL, 100, 133, 5},
        {FULL, 0x7fffffffffffffffL, 258, 6}};
    int li = level < 0 ? 0 : level > 13 ? 13 : level;
    d->kind = T[li].kind; d->attempts = T[li].attempts; d->goal = T[li].goal; d->iterations = T[li].iterations;
    if (format == ORC_FORMAT_IOS) exponent = 15;             /* DeflatorBuffers.swift:52-55 */
    d->end_index = -3; d->mask = (1 << exponent) - 1;
    d->next = (int32_t *)malloc(sizeof(int32_t) << exponent);
    d->val = (uint8_t *)calloc(1u << exponent, 1);
    hmap_init(&d->head, 1u << 17);
    /* DeflatorMatches.init ignores its `limit` argument (:66-70): 2048 for both shapes */
    d->limit = 2048;
    if (d->kind == FULL) { d->capacity = 1 << 21; graph_reserve(d, 2048); }
    else { d->capacity = 1 << 15; d->store_words = 2048; d->store = (uint32_t *)calloc(2048, 4); }
    memcpy(d->depths, DEPTH_DEFAULT, 542); d->generic = 1;
    d->out = out; d->out_cap = cap;
    if (format == ORC_FORMAT_ZLIB) {
        /* StreamHeader.write (StreamHeader.swift:56-62) */
        uint32_t unpaired = (uint32_t)(exponent - 8) << 4 | 0x08;
        uint32_t check = ~(((unpaired << 8 | unpaired >> 8) & 0xffff) % 31) & 31;
        put(d, check << 8 | unpaired, 16);
    }
    return d;
}

/* DeflatorBuffers.push (:68-93) */
static void deflator_push(deflator *d, const uint8_t *data, size_t n, int last)
{
    if (n) {
        if (d->in_len + (int64_t)n + 8 > d->in_cap) {
            d->in_cap = (d->in_len + (int64_t)n + 8) * 2;
            d->in = (uint8_t *)realloc(d->in, (size_t)d->in_cap);
        }
        memcpy(d->in + d->in_len, data, n);
        d->in_len += (int64_t)n;
    }
    if (!(in_count(d) > 4096 || last)) return;
    compress_blocks(d, last);
    if (last && d->format == ORC_FORMAT_ZLIB) {
        uint32_t sum = orc_adler32(1, d->in, (size_t)d->in_len);
        pad_to_byte(d);
        put(d, sum >> 24, 8); put(d, (sum >> 16) & 0xff, 8); put(d, (sum >> 8) & 0xff, 8); put(d, sum & 0xff, 8);
    }
}

static size_t deflator_finish(deflator *d, int *overflow)
{
    pad_to_byte(d);                                          /* DeflatorOut.pull flushes padding bits */
    size_t n = d->out_len;
    *overflow = d->overflow;
    free(d->in); free(d->next); free(d->val); hmap_free(&d->head); free(d->store); free(d);
    return n;
}

size_t orc_deflate_bound(size_t n) { return n + n / 4 + 4096; }

int orc_deflate(const uint8_t *src, size_t n, int format, int level, int exponent,
                uint8_t *dst, size_t cap, size_t *written)
{
    if (exponent < 8 || exponent > 15) return ORC_E_ARGUMENT;
    deflator *d = deflator_new(format, level, exponent, dst, cap);
    d->in_cap = (int64_t)n + 16; d->in = (uint8_t *)malloc((size_t)d->in_cap);
    deflator_push(d, src, n, 1);
    int overflow;
    size_t w = deflator_finish(d, &overflow);
    if (written) *written = w;
    return overflow ? ORC_E_OUTPUT_CAPACITY : ORC_DONE;
}

/* PNG.Encoder.pull end to end (PNG.Encoder.swift:33-129): one push per filtered scanline, then
 * push([], last: true) -- the push pattern matters only for the stored-tail quirk above. */
int orc_encode(const uint8_t *storage, int w, int h, int depth, int channels,
               int interlaced, int format, int level,
               uint8_t *dst, size_t cap, size_t *written)
{
    size_t u = orc_inflated_size(w, h, depth, channels, interlaced);
    uint8_t *rows = (uint8_t *)malloc(u ? u : 1);
    if (!rows) return ORC_E_ARGUMENT;
    orc_filter(storage, w, h, depth, channels, interlaced, rows);
    deflator *d = deflator_new(format, level, 15, dst, cap);
    /* walk the rows pass by pass exactly as orc_filter laid them out */
    size_t off = 0;
    int volume = depth * channels;
    static const int A7[7][4] = {{0,0,3,3},{4,0,3,3},{0,4,2,3},{2,0,2,2},{0,2,1,2},{1,0,1,1},{0,1,0,1}};
    for (int z = 0; z < (interlaced ? 7 : 1); ++z) {
        int sw = w, sh = h;
        if (interlaced) {
            int sx = 1 << A7[z][2], sy = 1 << A7[z][3];
            sw = (w + sx - A7[z][0] - 1) >> A7[z][2]; sh = (h + sy - A7[z][1] - 1) >> A7[z][3];
            if (sw <= 0 || sh <= 0) continue;
        }
        size_t len = (((size_t)sw * volume + 7) >> 3) + 1;
        for (int y = 0; y < sh; ++y) { deflator_push(d, rows + off, len, 0); off += len; }
    }
    deflator_push(d, NULL, 0, 1);
    int overflow;
    size_t wr = deflator_finish(d, &overflow);
    free(rows);
    if (written) *written = wr;
    return overflow ? ORC_E_OUTPUT_CAPACITY : ORC_DONE;
}

/* Test hook: the reference's own match-finder known-answer test (LZ77Tests/Bitstreams.swift:97-185)
 * drives DeflatorWindow directly with a 16-entry window.  Emits each output run as
 * (length byte, bytes...).  Returns the number of bytes written to `out`. */
int orc_kat_matching(const uint8_t *data, const int *seg_lens, int nseg, int exponent, int lookahead_nonfinal,
                     uint8_t *out, int out_cap)
{
    uint8_t sink[64];
    deflator *d = deflator_new(ORC_FORMAT_IOS, 0, 15, sink, sizeof sink);
    free(d->next); free(d->val);
    d->mask = (1 << exponent) - 1;
    d->next = (int32_t *)malloc(sizeof(int32_t) << exponent);
    d->val = (uint8_t *)calloc(1u << exponent, 1);
    int w = 0, index; int32_t nxt;
    size_t off = 0;
    for (int s = 0; s < nseg; ++s) {
        size_t n = (size_t)seg_lens[s];
        if (d->in_len + (int64_t)n + 8 > d->in_cap) {
            d->in_cap = (d->in_len + (int64_t)n + 8) * 2;
            d->in = (uint8_t *)realloc(d->in, (size_t)d->in_cap);
        }
        memcpy(d->in + d->in_len, data + off, n); d->in_len += (int64_t)n; off += n;
        int64_t lookahead = s == nseg - 1 ? 0 : lookahead_nonfinal;
        while (d->end_index < 0 && in_count(d) > lookahead) win_initialize(d, dequeue(d));
        while (in_count(d) > lookahead) {
            win_update(d, dequeue(d), &index, &nxt);
            int run, dist;
            if (win_match(d, index, nxt, 0x7fffffffffffffffL, 0x7fffffff, 0, 0, &run, &dist)) {
                if (w + 1 + run > out_cap) return -1;
                out[w++] = (uint8_t)run; out[w++] = win_literal(d);
                for (int k = 1; k < run; ++k) { win_update(d, dequeue(d), NULL, NULL); out[w++] = win_literal(d); }
            } else { if (w + 2 > out_cap) return -1; out[w++] = 1; out[w++] = win_literal(d); }
        }
        if (s != nseg - 1) continue;
        int64_t epilogue = -3 - (d->end_index < 0 ? d->end_index : 0);
        while (in_count(d) > epilogue) {
            win_update(d, dequeue(d), NULL, NULL);
            if (w + 2 > out_cap) return -1;
            out[w++] = 1; out[w++] = win_literal(d);
        }
    }
    int ov; deflator_finish(d, &ov);
    return w;
}
