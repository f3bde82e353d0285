/* CPU oracle for Miekki's sketch-build + fingerprint-intersection path.
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  It is a plain-C
 * restatement of the reference algorithm, structured like the reference (dense
 * 2^h query sketches, per-column byte compare) for clarity, not speed.  Every
 * function cites the reference file:line (under /root/reference) it follows.
 *
 * Parity pin: tests/golden/ holds outputs of the real reference compiled in the
 * authoring container (oracle/Makefile target `ref`, generator
 * tests/golden/make_golden.py); tests/test_oracle_golden.py checks this
 * restatement against every one of them.
 */
#ifndef MIEKKI_ORACLE_H
#define MIEKKI_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mko_index mko_index;

/* Miekki.h:27-31 */
typedef struct {
    uint32_t genome;
    uint32_t matches;
    double jaccard;
    double intersection;
} mko_hit;

/* ---- leaf arithmetic (utils.cpp) -------------------------------------- */
uint64_t mko_nuc2int(char c);                       /* utils.cpp:31-49   */
uint64_t mko_nuc2intrc(char c);                     /* utils.cpp:107-125 */
uint64_t mko_str2numstrand(const char *s, size_t n);/* utils.cpp:252-272 */
uint64_t mko_str2num(const char *s, size_t n);      /* utils.cpp:276-278 */
uint64_t mko_revhash64(uint64_t x);                 /* utils.cpp:179-184 */
uint64_t mko_unrevhash64(uint64_t x);               /* utils.cpp:188-193 */
uint64_t mko_universal_hash(uint64_t x, uint32_t i);/* utils.cpp:197-199 */

/* ---- index object (Miekki.h:66-90 constructor) ------------------------ */
/* fp_bits = number_bit_minimizer = 5 + f: 8 (stock build) or 16 (reference
 * compiled with `minimizer` = uint16_t). */
mko_index *mko_create(uint32_t k, uint32_t h, uint32_t fp_bits, uint32_t bloom_log2,
                      uint32_t threshold);
void mko_destroy(mko_index *ix);

uint32_t mko_k(const mko_index *ix);
uint32_t mko_h(const mko_index *ix);
uint32_t mko_fp_bits(const mko_index *ix);
uint32_t mko_index_size(const mko_index *ix);
uint32_t mko_threshold(const mko_index *ix);
uint64_t mko_bloom_bytes(const mko_index *ix);
const uint8_t *mko_bloom(const mko_index *ix);
const uint32_t *mko_sketch_size(const mko_index *ix);
const uint64_t *mko_genome_size(const mko_index *ix);
/* column p of the fingerprint matrix exactly as the reference stores it:
 * G * (fp_bits/8) bytes, 16-bit values big-endian (Miekki.cpp:228-239) */
const uint8_t *mko_column(const mko_index *ix, uint32_t p);

/* Miekki.cpp:91-113 (mantis) incl. the truncation to `minimizer` */
uint32_t mko_mantis(const mko_index *ix, uint64_t n);

/* Miekki.cpp:150-197.  fp_out[2^h] (empty = 255 / 65535), hash_out[2^h]
 * (empty = 2^64-1); returns active_minimizer. */
uint32_t mko_sketch(const mko_index *ix, const char *seq, uint64_t len,
                    uint16_t *fp_out, uint64_t *hash_out);

/* Miekki.cpp:201-224: sketch, then drop every bucket whose winning k-mer
 * fails the Bloom filter.  fp_out[2^h]. */
void mko_sketch_solid(const mko_index *ix, const char *seq, uint64_t len, uint16_t *fp_out);

int mko_check_bloom(const mko_index *ix, uint64_t num);   /* Miekki.cpp:135-146 */
void mko_insert_bloom(mko_index *ix, uint64_t num);       /* Miekki.cpp:121-131 */

/* Miekki.cpp:277-314 (the `-l` path).  Genome ids = call order. */
void mko_insert_sequences(mko_index *ix, const char *const *seqs, const uint64_t *lens,
                          uint32_t n);
void mko_insert_sequence(mko_index *ix, const char *seq, uint64_t len);   /* Miekki.cpp:243-273 */

/* Miekki.cpp:344-372: scores[nq][G] row-major, zero-initialised here. */
void mko_query_sequences(const mko_index *ix, const char *const *seqs, const uint64_t *lens,
                         uint32_t nq, uint32_t *scores);

/* Miekki.cpp:318-340: scores[G]; returns active_minimizer (Bloom-passing
 * non-empty partitions). */
uint32_t mko_query_sequence(const mko_index *ix, const char *seq, uint64_t len,
                            uint32_t *scores);

/* Miekki.cpp:376-397: returns number of hits written to out[nresults],
 * descending by intersection with the reference's heap tie behaviour. */
uint32_t mko_filter_results(const mko_index *ix, const uint32_t *scores, uint32_t nresults,
                            uint32_t min_score, double min_intersection, mko_hit *out);

/* Miekki.cpp:440-444: "name:" + hits + "\n" appended to buf; returns bytes
 * written (buf must be large enough: strlen(name) + 2 + 80 * nhits). */
size_t mko_format_query_line(const char *name, const mko_hit *hits, uint32_t nhits, char *buf);

/* ---- persistence, SURVEY.md row P (Miekki.cpp:649-719) ---------------- */
/* size in bytes of the uncompressed stream */
uint64_t mko_serial_size(const mko_index *ix);
/* write the uncompressed stream into buf (byte 32, the reference's
 * uninitialised jaccard_estimation, is written as 0) */
void mko_serialize(const mko_index *ix, uint8_t *buf);
/* rebuild from an uncompressed stream; NULL on malformed input */
mko_index *mko_deserialize(const uint8_t *buf, uint64_t n);

/* ---- exact mode, Miekki.cpp:792-859 ----------------------------------- */
/* Parse FASTA text the way ground_truth_batch does (per-contig k-mers, short
 * contigs leak into the next one) and return the sorted distinct canonical
 * k-mer set; caller frees *set with mko_free. */
uint64_t mko_exact_genome_set(const char *fasta, uint64_t n, uint32_t k, uint64_t **set);
/* distinct canonical k-mers of seq vs the set: |A∩B| and |B| + |A\B| */
void mko_exact_query(const uint64_t *set, uint64_t nset, const char *seq, uint64_t len,
                     uint32_t k, uint64_t *inter, uint64_t *uni);
void mko_free(void *p);
/* test hook: set G, sketch_size[], genome_size[] directly (columns undefined) */
void mko_poke_sizes(mko_index *ix, uint32_t G, const uint32_t *ss, const uint64_t *gs);

#ifdef __cplusplus
}
#endif
#endif
