/* CPU oracle -- see miekki_oracle.h.  TEST INFRASTRUCTURE, never linked into or
 * called from the product path.  All citations are into /root/reference. */
#include "miekki_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NUMBER_HASH 5u          /* Miekki.h:79, Miekki.cpp:703 */
#define NUMBER_BIT_MANTIS 5u    /* main.cpp:196 */

struct mko_index {
    uint32_t k, h, fp_bits, bloom_log2, threshold;
    uint32_t nmin;              /* number_minimizer = 2^h */
    uint32_t W;                 /* bytes per stored fingerprint */
    uint32_t empty;             /* maximal_minimizer: 255 or 65535 */
    uint64_t kmask;             /* offsetUpdatekmer - 1 (Miekki.h:76-77) */
    uint64_t bloom_bits;        /* bloom_size */
    uint8_t *bloom;             /* bloom_size/8 bytes */
    uint32_t G, capG;
    uint8_t **col;              /* one growable byte string per partition (Miekki.h:54) */
    uint32_t *sketch_size;
    uint64_t *genome_size;
};

/* ------------------------------------------------------------------ utils */

uint64_t mko_nuc2int(char c)            /* utils.cpp:31-49: anything but C,G,T is 0 */
{
    switch (c) { case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; }
}

uint64_t mko_nuc2intrc(char c)          /* utils.cpp:107-125: anything but A,C,G is 0 */
{
    switch (c) { case 'A': return 3; case 'C': return 2; case 'G': return 1; default: return 0; }
}

uint64_t mko_str2numstrand(const char *s, size_t n)  /* utils.cpp:252-272 */
{
    uint64_t res = 0;
    for (size_t i = 0; i < n; ++i) {
        res <<= 2;
        switch (s[i]) {
        case 'A': case 'a': break;
        case 'C': case 'c': res += 1; break;
        case 'G': case 'g': res += 2; break;
        case 'T': case 't': res += 3; break;
        default: return 0;       /* any other character zeroes the whole word */
        }
    }
    return res;
}

static char revcomp_char(char c)        /* utils.cpp:203-215 */
{
    switch (c) {
    case 'C': case 'c': return 'G';
    case 'G': case 'g': return 'C';
    case 'T': case 't': return 'A';
    }
    return 'T';
}

uint64_t mko_str2num(const char *s, size_t n)        /* utils.cpp:276-278 (+219-224) */
{
    char rc[64];
    if (n > sizeof rc) n = sizeof rc;
    for (size_t i = 0; i < n; ++i) rc[i] = revcomp_char(s[n - 1 - i]);
    uint64_t a = mko_str2numstrand(s, n), b = mko_str2numstrand(rc, n);
    return a < b ? a : b;
}

uint64_t mko_revhash64(uint64_t x)      /* utils.cpp:179-184 */
{
    x = ((x >> 32) ^ x) * 0xD6E8FEB86659FD93ULL;
    x = ((x >> 32) ^ x) * 0xD6E8FEB86659FD93ULL;
    x = ((x >> 32) ^ x);
    return x;
}

uint64_t mko_unrevhash64(uint64_t x)    /* utils.cpp:188-193 */
{
    x = ((x >> 32) ^ x) * 0xCFEE444D8B59A89BULL;
    x = ((x >> 32) ^ x) * 0xCFEE444D8B59A89BULL;
    x = ((x >> 32) ^ x);
    return x;
}

uint64_t mko_universal_hash(uint64_t x, uint32_t i)  /* utils.cpp:197-199 */
{
    uint32_t m = i * 69u;                            /* u32 product, then widened */
    return mko_unrevhash64(x) + (((uint64_t)m * mko_revhash64(x)) % 1024u);
}

static int floor_log2_u64(uint64_t x)   /* utils.cpp:84-91 (x86 bsr), x != 0 */
{
    int p = 0;
    while (x >>= 1) ++p;
    return p;
}

/* ------------------------------------------------------------------ index */

mko_index *mko_create(uint32_t k, uint32_t h, uint32_t fp_bits, uint32_t bloom_log2,
                      uint32_t threshold)
{
    if (fp_bits != 8 && fp_bits != 16) return NULL;  /* Miekki.cpp:235-237 "not implemented" */
    if (k < 2 || k > 31 || h < 1 || h > 30) return NULL;
    mko_index *ix = (mko_index *)calloc(1, sizeof *ix);
    ix->k = k; ix->h = h; ix->fp_bits = fp_bits; ix->bloom_log2 = bloom_log2;
    ix->threshold = threshold;
    ix->nmin = 1u << h;
    ix->W = fp_bits / 8;
    ix->empty = fp_bits == 8 ? 255u : 65535u;
    ix->kmask = ((uint64_t)1 << (2 * k)) - 1;
    ix->bloom_bits = bloom_log2 ? (uint64_t)1 << bloom_log2 : 0;   /* Miekki.h:85-89 */
    if (ix->bloom_bits) ix->bloom = (uint8_t *)calloc(ix->bloom_bits / 8, 1);
    ix->col = (uint8_t **)calloc(ix->nmin, sizeof(uint8_t *));
    return ix;
}

void mko_destroy(mko_index *ix)
{
    if (!ix) return;
    for (uint32_t p = 0; p < ix->nmin; ++p) free(ix->col[p]);
    free(ix->col); free(ix->bloom); free(ix->sketch_size); free(ix->genome_size);
    free(ix);
}

uint32_t mko_k(const mko_index *ix) { return ix->k; }
uint32_t mko_h(const mko_index *ix) { return ix->h; }
uint32_t mko_fp_bits(const mko_index *ix) { return ix->fp_bits; }
uint32_t mko_index_size(const mko_index *ix) { return ix->G; }
uint32_t mko_threshold(const mko_index *ix) { return ix->threshold; }
uint64_t mko_bloom_bytes(const mko_index *ix) { return ix->bloom_bits / 8; }
const uint8_t *mko_bloom(const mko_index *ix) { return ix->bloom; }
const uint32_t *mko_sketch_size(const mko_index *ix) { return ix->sketch_size; }
const uint64_t *mko_genome_size(const mko_index *ix) { return ix->genome_size; }
const uint8_t *mko_column(const mko_index *ix, uint32_t p) { return ix->col[p]; }

static void grow(mko_index *ix, uint32_t need)
{
    if (need <= ix->capG) return;
    uint32_t cap = ix->capG ? ix->capG : 16;
    while (cap < need) cap *= 2;
    for (uint32_t p = 0; p < ix->nmin; ++p)
        ix->col[p] = (uint8_t *)realloc(ix->col[p], (size_t)cap * ix->W);
    ix->sketch_size = (uint32_t *)realloc(ix->sketch_size, (size_t)cap * sizeof(uint32_t));
    ix->genome_size = (uint64_t *)realloc(ix->genome_size, (size_t)cap * sizeof(uint64_t));
    ix->capG = cap;
}

/* Miekki.cpp:91-113 */
uint32_t mko_mantis(const mko_index *ix, uint64_t n)
{
    if (n == 0) return ix->empty;
    int64_t prefix = floor_log2_u64(n);
    int64_t exp = prefix - 32 + (int64_t)ix->h;
    if (exp < 0) exp = 0;
    int offset = (int)(prefix - (int64_t)(ix->fp_bits - NUMBER_BIT_MANTIS));
    if (offset < 0) offset = 0;
    uint64_t suffix = n - ((uint64_t)1 << prefix);
    suffix >>= offset;
    uint64_t res = suffix + ((uint64_t)exp << (ix->fp_bits - NUMBER_BIT_MANTIS));
    return (uint32_t)(res & ix->empty);              /* truncation to `minimizer` */
}

static uint64_t rcb(const mko_index *ix, uint64_t m)  /* Miekki.cpp:66-76 */
{
    uint64_t res = 0, offset = (uint64_t)1 << (2 * ix->k - 2);
    for (uint32_t i = 0; i < ix->k; ++i) {
        res += (3 - (m % 4)) * offset;
        m >>= 2;
        offset >>= 2;
    }
    return res;
}

/* Miekki.cpp:150-197 */
uint32_t mko_sketch(const mko_index *ix, const char *seq, uint64_t len,
                    uint16_t *fp_out, uint64_t *hash_out)
{
    const uint32_t k = ix->k;
    const uint64_t vmod = (uint64_t)1 << (64 - ix->h);           /* line 151 `mask` */
    for (uint32_t i = 0; i < ix->nmin; ++i) { fp_out[i] = (uint16_t)ix->empty; hash_out[i] = ~0ULL; }
    uint32_t active = 0;
    size_t npre = len < k - 1 ? (size_t)len : k - 1;             /* substr(0,k-1) */
    uint64_t S = mko_str2numstrand(seq, npre);                   /* line 158 */
    uint64_t RC = rcb(ix, S);                                    /* line 160 */
    for (uint64_t i = 0; i + k < len; ++i) {                     /* line 162: last k-mer skipped */
        char c = seq[i + k - 1];
        S = ((S << 2) + mko_nuc2int(c)) & ix->kmask;             /* update_kmer 51-55 */
        RC = (RC >> 2) + (mko_nuc2intrc(c) << (2 * k - 2));      /* update_kmer_RC 59-62 */
        uint64_t anc = mko_revhash64(S < RC ? S : RC);           /* 167-168 */
        uint64_t bucket = anc / vmod, value = anc % vmod;        /* 169-170 */
        uint32_t fp = mko_mantis(ix, value);                     /* 171 */
        if (fp < fp_out[bucket]) {                               /* 172: strict < */
            if (fp_out[bucket] == ix->empty) ++active;
            fp_out[bucket] = (uint16_t)fp;
            hash_out[bucket] = anc;
        }
    }
    return active;
}

int mko_check_bloom(const mko_index *ix, uint64_t num)           /* Miekki.cpp:135-146 */
{
    for (uint32_t i = 0; i < NUMBER_HASH; ++i) {
        uint64_t hash = mko_universal_hash(num, i) >> ix->bloom_log2;
        /* (cell && mask[hit]) is a LOGICAL and: the test is "byte != 0" */
        if (ix->bloom[hash / 8] == 0) return 0;
    }
    return 1;
}

void mko_insert_bloom(mko_index *ix, uint64_t num)               /* Miekki.cpp:121-131 */
{
    for (uint32_t i = 0; i < NUMBER_HASH; ++i) {
        uint64_t hash = mko_universal_hash(num, i) >> ix->bloom_log2;
        uint8_t hit = (uint8_t)(hash % 8);
        if (ix->bloom[hash / 8] == 0) ix->bloom[hash / 8] += (uint8_t)(1u << hit);
    }
}

void mko_sketch_solid(const mko_index *ix, const char *seq, uint64_t len, uint16_t *fp_out)
{                                                                /* Miekki.cpp:214-224 */
    uint64_t *hash = (uint64_t *)malloc((size_t)ix->nmin * sizeof(uint64_t));
    mko_sketch(ix, seq, len, fp_out, hash);
    for (uint32_t i = 0; i < ix->nmin; ++i)
        if (!mko_check_bloom(ix, hash[i])) fp_out[i] = (uint16_t)ix->empty;
    free(hash);
}

/* Miekki.cpp:277-314 */
void mko_insert_sequences(mko_index *ix, const char *const *seqs, const uint64_t *lens, uint32_t n)
{
    uint16_t *fp = (uint16_t *)malloc((size_t)ix->nmin * sizeof(uint16_t));
    uint64_t *hs = (uint64_t *)malloc((size_t)ix->nmin * sizeof(uint64_t));
    const uint32_t shift = ix->fp_bits - NUMBER_BIT_MANTIS;
    for (uint32_t g = 0; g < n; ++g) {
        mko_sketch(ix, seqs[g], lens[g], fp, hs);
        grow(ix, ix->G + 1);
        double card = 0;
        uint32_t active = 0;                                     /* line 289: u32 */
        for (uint32_t i = 0; i < ix->nmin; ++i) {
            /* add_index 228-239: 16-bit values are stored big-endian */
            if (ix->W == 2) {
                ix->col[i][2 * (size_t)ix->G] = (uint8_t)(fp[i] / 256);
                ix->col[i][2 * (size_t)ix->G + 1] = (uint8_t)(fp[i] % 256);
            } else {
                ix->col[i][ix->G] = (uint8_t)fp[i];
            }
            if (fp[i] != ix->empty) {
                card += 1.0 / (double)((uint64_t)1 << (fp[i] >> shift));   /* 1/pow(2,e), exact */
                ++active;
                if (!mko_check_bloom(ix, hs[i])) mko_insert_bloom(ix, hs[i]);
            }
        }
        ix->sketch_size[ix->G] = active;
        uint32_t sq = active * active;                           /* line 306: wraps in u32 */
        card = 0.72134 * (double)sq / card;
        if (card > (double)lens[g]) ix->genome_size[ix->G] = lens[g];
        else ix->genome_size[ix->G] = (uint64_t)card;
        ix->G++;
    }
    free(fp); free(hs);
}

/* Miekki.cpp:243-273 (behind index_file 518-536): insert_sequences for one genome, except that active_minimizer is a
 * double here (line 246), so active*active at line 265 does not wrap */
void mko_insert_sequence(mko_index *ix, const char *seq, uint64_t len)
{
    const uint64_t g = ix->G;
    mko_insert_sequences(ix, &seq, &len, 1);                     /* sketch, add_index, Bloom: the same statements */
    const uint32_t shift = ix->fp_bits - NUMBER_BIT_MANTIS;
    double card = 0, active = 0;
    for (uint32_t i = 0; i < ix->nmin; ++i) {
        uint32_t fp = ix->W == 2 ? (uint32_t)ix->col[i][2 * g] * 256u + ix->col[i][2 * g + 1] : ix->col[i][g];
        if (fp != ix->empty) { card += 1.0 / (double)((uint64_t)1 << (fp >> shift)); active += 1; }
    }
    card = 0.72134 * (active * active) / card;                   /* line 265 */
    if (card > (double)len) ix->genome_size[g] = len;            /* 266-270 */
    else ix->genome_size[g] = (uint64_t)card;
}

static uint32_t col_value(const mko_index *ix, uint32_t p, uint32_t g)   /* get_minimizers 881-898 */
{
    const uint8_t *c = ix->col[p];
    return ix->W == 2 ? (uint32_t)c[2 * (size_t)g] * 256u + c[2 * (size_t)g + 1] : c[g];
}

/* Miekki.cpp:344-372 */
void mko_query_sequences(const mko_index *ix, const char *const *seqs, const uint64_t *lens,
                         uint32_t nq, uint32_t *scores)
{
    const uint32_t G = ix->G;
    uint16_t *sk = (uint16_t *)malloc((size_t)nq * ix->nmin * sizeof(uint16_t));
    for (uint32_t q = 0; q < nq; ++q) mko_sketch_solid(ix, seqs[q], lens[q], sk + (size_t)q * ix->nmin);
    memset(scores, 0, (size_t)nq * G * sizeof(uint32_t));
    for (uint32_t p = 0; p < ix->nmin; ++p)
        for (uint32_t q = 0; q < nq; ++q) {
            uint32_t mini = sk[(size_t)q * ix->nmin + p];
            if (mini == ix->empty) continue;
            uint32_t *row = scores + (size_t)q * G;
            if (ix->W == 1) {
                const uint8_t *c = ix->col[p];
                for (uint32_t g = 0; g < G; ++g) row[g] += (c[g] == mini);
            } else {
                for (uint32_t g = 0; g < G; ++g) row[g] += (col_value(ix, p, g) == mini);
            }
        }
    free(sk);
}

/* Miekki.cpp:318-340 */
uint32_t mko_query_sequence(const mko_index *ix, const char *seq, uint64_t len, uint32_t *scores)
{
    const uint32_t G = ix->G;
    uint16_t *fp = (uint16_t *)malloc((size_t)ix->nmin * sizeof(uint16_t));
    uint64_t *hs = (uint64_t *)malloc((size_t)ix->nmin * sizeof(uint64_t));
    mko_sketch(ix, seq, len, fp, hs);
    memset(scores, 0, (size_t)G * sizeof(uint32_t));
    uint32_t active = 0;
    for (uint32_t p = 0; p < ix->nmin; ++p) {
        if (fp[p] == ix->empty) continue;
        if (!mko_check_bloom(ix, hs[p])) continue;
        ++active;
        for (uint32_t g = 0; g < G; ++g) scores[g] += (col_value(ix, p, g) == fp[p]);
    }
    free(fp); free(hs);
    return active;
}

/* ---- libstdc++ heap primitives, restated (bits/stl_heap.h, GCC 11: __push_heap,
 * __adjust_heap, __pop_heap, sort_heap) with the comparator of Miekki.cpp:377:
 * comp(a,b) = a.intersection > b.intersection (=> min-heap on intersection). */
static int hcomp(const mko_hit *a, const mko_hit *b) { return a->intersection > b->intersection; }

static void h_push(mko_hit *first, long hole, long top, mko_hit value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && hcomp(&first[parent], &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void h_adjust(mko_hit *first, long hole, long len, mko_hit value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (hcomp(&first[child], &first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    h_push(first, hole, top, value);
}

/* std::pop_heap(first, first+n): moves the front to first[n-1] */
static void h_pop(mko_hit *first, long n)
{
    if (n > 1) {
        mko_hit value = first[n - 1];
        first[n - 1] = first[0];
        h_adjust(first, 0, n - 1, value);
    }
}

/* Miekki.cpp:376-397 */
uint32_t mko_filter_results(const mko_index *ix, const uint32_t *scores, uint32_t nresults,
                            uint32_t min_score, double min_intersection, mko_hit *out)
{
    mko_hit *heap = (mko_hit *)malloc(((size_t)nresults + 1) * sizeof(mko_hit));
    long n = 0;
    for (uint32_t g = 0; g < ix->G; ++g) {
        uint32_t score = scores[g];
        if (score < min_score) continue;
        double jaccard = (double)score / ix->sketch_size[g];
        double intersection = jaccard * ix->genome_size[g];
        if (intersection < min_intersection) continue;
        if ((size_t)n >= nresults) {
            if (n == 0) continue;                                /* nresults == 0: UB in the reference */
            if (heap[0].intersection > intersection) continue;   /* line 387: ties replace */
            h_pop(heap, n);
            --n;
        }
        mko_hit v = { g, score, jaccard, intersection };
        heap[n++] = v;
        h_push(heap, n - 1, 0, heap[n - 1]);                     /* std::push_heap */
    }
    for (long m = n; m > 1; --m) h_pop(heap, m);                 /* std::sort_heap */
    memcpy(out, heap, (size_t)n * sizeof(mko_hit));
    free(heap);
    return (uint32_t)n;
}

/* Miekki.cpp:440-444; std::to_string(double) is "%f", uint(x) truncates */
size_t mko_format_query_line(const char *name, const mko_hit *hits, uint32_t nhits, char *buf)
{
    char *p = buf;
    p += sprintf(p, "%s:", name);
    for (uint32_t i = 0; i < nhits; ++i)
        p += sprintf(p, "%u\t%u\t%u\t%f;", hits[i].genome, hits[i].matches,
                     (unsigned)hits[i].intersection, hits[i].jaccard);
    *p++ = '\n';
    *p = 0;
    return (size_t)(p - buf);
}

/* ------------------------------------------------------------ persistence */

#define HDR_BYTES 39u

uint64_t mko_serial_size(const mko_index *ix)
{
    return HDR_BYTES + (uint64_t)ix->nmin * ix->G * ix->W + 8ull * ix->G + ix->bloom_bits / 8
           + 4ull * ix->G;
}

static uint8_t *put32(uint8_t *b, uint32_t v) { memcpy(b, &v, 4); return b + 4; }
static uint8_t *put64(uint8_t *b, uint64_t v) { memcpy(b, &v, 8); return b + 8; }

/* Miekki.cpp:649-678 */
void mko_serialize(const mko_index *ix, uint8_t *b)
{
    b = put32(b, ix->k);
    b = put32(b, ix->h);
    b = put32(b, ix->fp_bits);
    b = put32(b, NUMBER_BIT_MANTIS);
    b = put32(b, ix->G);
    b = put32(b, ix->bloom_log2);
    b = put64(b, ix->bloom_bits);
    *b++ = 0;                      /* jaccard_estimation: uninitialised in the reference */
    *b++ = 0;                      /* containment_estimation = false */
    b = put32(b, ix->threshold);
    *b++ = 1;                      /* compressed: true after main.cpp:198 on the -l path */
    for (uint32_t p = 0; p < ix->nmin; ++p) {
        memcpy(b, ix->col[p], (size_t)ix->G * ix->W);
        b += (size_t)ix->G * ix->W;
    }
    memcpy(b, ix->genome_size, 8ull * ix->G); b += 8ull * ix->G;
    if (ix->bloom_bits) { memcpy(b, ix->bloom, ix->bloom_bits / 8); b += ix->bloom_bits / 8; }
    memcpy(b, ix->sketch_size, 4ull * ix->G);
}

/* Miekki.cpp:682-719 */
mko_index *mko_deserialize(const uint8_t *b, uint64_t n)
{
    if (n < HDR_BYTES) return NULL;
    uint32_t k, h, fpb, nbm, G, bl2, thr;
    uint64_t bbits;
    memcpy(&k, b, 4); memcpy(&h, b + 4, 4); memcpy(&fpb, b + 8, 4); memcpy(&nbm, b + 12, 4);
    memcpy(&G, b + 16, 4); memcpy(&bl2, b + 20, 4); memcpy(&bbits, b + 24, 8);
    memcpy(&thr, b + 34, 4);
    (void)nbm;
    mko_index *ix = mko_create(k, h, fpb, 0, thr);
    if (!ix) return NULL;
    ix->bloom_log2 = bl2;
    ix->bloom_bits = bbits;
    grow(ix, G ? G : 1);
    ix->G = G;
    if (mko_serial_size(ix) != n) { mko_destroy(ix); return NULL; }
    b += HDR_BYTES;
    for (uint32_t p = 0; p < ix->nmin; ++p) {
        memcpy(ix->col[p], b, (size_t)G * ix->W);
        b += (size_t)G * ix->W;
    }
    memcpy(ix->genome_size, b, 8ull * G); b += 8ull * G;
    if (bbits) {
        ix->bloom = (uint8_t *)malloc(bbits / 8);
        memcpy(ix->bloom, b, bbits / 8); b += bbits / 8;
    }
    memcpy(ix->sketch_size, b, 4ull * G);
    return ix;
}

/* -------------------------------------------------------------- exact mode */

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static uint64_t sort_unique(uint64_t *v, uint64_t n)
{
    if (!n) return 0;
    qsort(v, n, sizeof(uint64_t), cmp_u64);
    uint64_t m = 1;
    for (uint64_t i = 1; i < n; ++i) if (v[i] != v[m - 1]) v[m++] = v[i];
    return m;
}

/* Miekki.cpp:800-823: getline loop; a '>' line flushes the accumulated contig
 * only if it is at least k long (otherwise it keeps growing into the next one) */
uint64_t mko_exact_genome_set(const char *fasta, uint64_t n, uint32_t k, uint64_t **set)
{
    char *ref = (char *)malloc(n + 1);
    uint64_t rl = 0, cap = 1024, cnt = 0;
    uint64_t *v = (uint64_t *)malloc(cap * sizeof(uint64_t));
    uint64_t pos = 0;
    int done = 0;
    while (!done) {
        uint64_t e = pos;
        while (e < n && fasta[e] != '\n') ++e;
        const char *line = fasta + pos;
        uint64_t ll = e - pos;
        int flush = 0;
        if (ll > 0 && line[0] == '>') flush = 1;
        else { memcpy(ref + rl, line, ll); rl += ll; }
        if (e >= n) { done = 1; flush = 2; }             /* eof: lines 817-822 */
        if (flush && rl >= k) {
            for (uint64_t i = 0; i + k - 1 < rl; ++i) {
                if (cnt == cap) { cap *= 2; v = (uint64_t *)realloc(v, cap * sizeof(uint64_t)); }
                v[cnt++] = mko_str2num(ref + i, k);
            }
            rl = 0;
        }
        pos = e + 1;
    }
    free(ref);
    *set = v;
    return sort_unique(v, cnt);
}

/* Miekki.cpp:826-842 */
void mko_exact_query(const uint64_t *set, uint64_t nset, const char *seq, uint64_t len,
                     uint32_t k, uint64_t *inter, uint64_t *uni)
{
    uint64_t na = len >= k ? len - k + 1 : 0;
    uint64_t *a = (uint64_t *)malloc((na ? na : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < na; ++i) a[i] = mko_str2num(seq + i, k);
    na = sort_unique(a, na);
    uint64_t in = 0, un = nset;
    for (uint64_t i = 0; i < na; ++i) {
        if (bsearch(&a[i], set, nset, sizeof(uint64_t), cmp_u64)) ++in; else ++un;
    }
    free(a);
    *inter = in; *uni = un;
}

void mko_free(void *p) { free(p); }

/* test hook: set G and the two per-genome size vectors without sketching
 * (column contents are left undefined) -- for synthetic filter_results cases */
void mko_poke_sizes(mko_index *ix, uint32_t G, const uint32_t *ss, const uint64_t *gs)
{
    grow(ix, G ? G : 1);
    ix->G = G;
    memcpy(ix->sketch_size, ss, 4ull * G);
    memcpy(ix->genome_size, gs, 8ull * G);
}
