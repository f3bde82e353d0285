/* miekki_hip.h -- C ABI of libmiekki_hip.so, the MI355X (gfx950) implementation of
 * Miekki's sketch-build + fingerprint-intersection hot path.
 *
 * The reference has no FFI of its own: main.cpp calls Miekki's members directly.
 * Each entry point below replaces one of those L3 -> L2 calls (citations are into
 * the reference tree) and is what a binding in the
 * reference would bind; INTEGRATION.md shows that binding.
 *
 * Conventions: plain C, opaque handle, every call returns 0 on success or a
 * negative mk_status and leaves a message for mk_last_error() (thread-local).
 * Host pointers unless a parameter name starts with `d_` (device memory of the
 * context's GPU).  One context owns one GPU; calls on one context must not
 * overlap -- with one exception: mk_dev_copy may run with a context as its
 * DESTINATION while another thread uses that context (it reads nothing of the
 * destination but its device ordinal; the multi-GPU driver gathers the shards'
 * rows into the first GPU that way while that GPU still scans).  There is no
 * CPU fallback: without a usable HIP device mk_create fails.
 */
#ifndef MIEKKI_HIP_H
#define MIEKKI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_ABI_VERSION 5

typedef enum {
    MK_OK = 0,
    MK_ERR_ARG = -1,          /* bad argument */
    MK_ERR_UNSUPPORTED = -2,  /* the reference's "not implemented" (Miekki.cpp:235-237, 893-895) */
    MK_ERR_DEVICE = -3,       /* HIP runtime error / no GPU */
    MK_ERR_NOMEM = -4,
    MK_ERR_STATE = -5         /* call not valid in the context's current state */
} mk_status;

typedef struct mk_ctx mk_ctx;

/* Constructor arguments of Miekki(k, h, bit_per_min, 5, 0, out, b, threshold, t)
 * (Miekki.h:66-90, main.cpp:196). */
typedef struct {
    uint32_t k;               /* -k, 2..31 (offsetUpdatekmer = 1 << 2k, Miekki.h:76-77) */
    uint32_t h;               /* -h, log2 of the number of partitions, 1..28 */
    uint32_t fp_bits;         /* number_bit_minimizer = 5 + f: 8 (-f 3) or 16 (-f 11) */
    uint32_t bloom_log2;      /* -b, >= 32 as in the reference */
    uint32_t threshold;       /* -s truncated to u32 (main.cpp:196) */
    int32_t device;           /* HIP device ordinal */
    uint32_t genome_id_base;  /* id of this context's first genome (multi-GPU genome sharding) */
    uint32_t reserved;
} mk_params;

/* Same layout as the reference's similarity_score (Miekki.h:27-31). */
typedef struct {
    uint32_t genome;
    uint32_t matches;
    double jaccard;
    double intersection;
} mk_hit;

/* Device-time accounting of the last query call(s), from HIP events recorded on
 * the context's stream around each kernel. */
typedef struct {
    double sketch_ms;         /* query sketch + Bloom gate */
    double scan_ms;           /* fingerprint scan launches */
    double filter_ms;         /* top-hit selection over the score rows */
    uint64_t scan_launches;
    uint64_t comparisons;     /* G * sum of active partitions */
    uint64_t active_partitions;
    uint64_t scan_algo_bytes; /* comparisons * W + 4 * queries * G */
    double build_sketch_ms;   /* index build: k-mer hashing + minimizer selection */
    double build_finalize_ms; /* transposed matrix write, sizes, Bloom insert */
    uint64_t build_kmers;
    uint64_t build_genomes;
    uint64_t scan_slab_launches; /* of scan_launches: launches of the slab schedule (scan_slab_kernel) */
    /* mk_dev_copy calls with this context as the SOURCE, by the path they took: a direct GPU-to-GPU
     * copy (same GPU, or peer access over xGMI), or through host memory because the two GPUs have
     * no peer access -- a staged exchange would otherwise just look like a slow xGMI */
    uint64_t peer_copies, staged_copies;
    uint64_t peer_copy_bytes, staged_copy_bytes;
} mk_stats;

const char *mk_last_error(void);
uint32_t mk_abi_version(void);

/* new Miekki(...)  (main.cpp:196) */
int mk_create(const mk_params *params, mk_ctx **out);
void mk_destroy(mk_ctx *ctx);

/* Pre-size the fingerprint matrix for n_genomes rows per partition.  Optional;
 * appending past the reservation re-lays the matrix out (needs 2x memory).
 * A matrix larger than its HBM budget (environment MIEKKI_HBM_MATRIX_MIB, or simply
 * more than the free device memory) keeps its first partition rows in HBM and the rest
 * in page-locked host memory; every entry point works unchanged, queries over the cold
 * partition ranges are streamed at PCIe speed.  (What compress_index / decompress_index,
 * Miekki.cpp:863-877, were for in the reference: a collection beyond fast memory.) */
int mk_reserve(mk_ctx *ctx, uint32_t n_genomes);

/* Miekki::compress_index / decompress_index (Miekki.cpp:863-877; main.cpp:198 compresses after -l) for the rows that live
 * in host memory because the matrix exceeds its HBM budget (mk_reserve): per piece of 1,024 genomes of a row, one bit per
 * genome "differs from the genome before" + the differing fingerprints -- what related strains next to each other in the
 * list leave of a column (README.md:136-138); a row that would not shrink is kept as it is, and a collection that does
 * not shrink by 5 % is left alone.  Queries then move the PACKED bytes of a cold range over PCIe and expand them in HBM;
 * appends, exports and imports unpack first by themselves (the reference's dump decompresses first too, 662-664).
 * raw_bytes / packed_bytes (may be NULL): the cold rows before and after (equal: nothing was packed). */
int mk_index_compress(mk_ctx *ctx, uint64_t *raw_bytes, uint64_t *packed_bytes);
int mk_index_decompress(mk_ctx *ctx);
uint32_t mk_index_size(const mk_ctx *ctx);                 /* Miekki::index_size */
int mk_get_params(const mk_ctx *ctx, mk_params *out);
int mk_get_stats(const mk_ctx *ctx, mk_stats *out);
int mk_reset_stats(mk_ctx *ctx);
/* Measurement aid (bench.py): best of `rounds` read-only streaming passes (16-byte loads,
 * eight in flight per lane) over the resident fingerprint matrix -> GB/s and the bytes one
 * pass read.  The box's own HBM read ceiling, to put beside the 8 TB/s spec figure. */
int mk_probe_stream_read(mk_ctx *ctx, uint32_t rounds, double *gbps, uint64_t *bytes);
/* Measurement aid (bench.py's PCIe-fed build sample): the synthetic genomes of SURVEY.md 8d
 * (ids first_id .., `length` bases each, concatenated) written to host memory dst. */
int mk_probe_synth_genomes(mk_ctx *ctx, uint64_t first_id, uint32_t n, uint64_t length, char *dst);

/* ---- index build --------------------------------------------------------- */

/* Miekki::insert_sequences (Miekki.cpp:277-314; called from index_file_of_file,
 * Miekki.cpp:572,580).  Genome ids follow call order.  Sequences shorter than k
 * are rejected with MK_ERR_ARG (the driver filters them, Miekki.cpp:569).
 * Pipelined: the sequences are copied to the GPU while the kernels of the previous
 * call still run; the call returns when the copy is done (seqs may be reused) and
 * leaves this batch's kernels in flight.  Every other entry point first settles
 * the batch in flight, so results are always those of a synchronous build. */
int mk_index_append(mk_ctx *ctx, const char *const *seqs, const uint64_t *lens, uint32_t n);

/* Miekki::insert_sequence (Miekki.cpp:243-273; behind index_file, 518-536 -- neither is reached from the reference's
 * CLI, both are public members).  One genome; sketch, column and Bloom inserts are insert_sequences', the size estimate
 * is this member's own: the count of active partitions is a double there, so its square does not wrap at 2^32 the way
 * insert_sequences' u32 does (Miekki.cpp:289, 306) -- genome_size differs from mk_index_append's above 65,535 active
 * partitions, i.e. from -h 17 on. */
int mk_index_insert_sequence(mk_ctx *ctx, const char *seq, uint64_t len);

/* ---- packed ingest (2-bit packing overlapped with the copy to the GPU) ----
 * A sequence as 2 bits per base and, only when it holds characters other than A, C, G, T, one
 * exception bit per base -- a quarter (three eighths) of the bytes mk_index_append moves:
 *   codes   base i at bits 2 * (i % 32) of 64-bit word i / 32: A 0, C 1, G 2, T 3, anything else 0
 *           (nuc2int, utils.cpp:31-49)
 *   except  bit i % 64 of word i / 64 is set where the character is not one of "ACGT" (upper case):
 *           for those the reverse strand's digit is 0 as well, not 3 - code (nuc2intrc,
 *           utils.cpp:107-125); NULL when the sequence has none
 *   head    the first min(len, 32) characters as they came: the k-1 characters of the seed go
 *           through str2numstrand (utils.cpp:252-272: case-insensitive, any other character zeroes
 *           the whole seed), which codes and exception bits cannot express
 * Results are those of mk_index_append on the characters, bit for bit. */
typedef struct {
    const uint64_t *codes;    /* (len + 31) / 32 words */
    const uint64_t *except;   /* (len + 63) / 64 words, or NULL */
    uint64_t len;             /* bases */
    char head[32];
} mk_packed_seq;

/* Host-side packer (no GPU involved; thread-safe, AVX2 when the CPU has it): appends n characters at
 * base position `at` of a sequence being packed, e.g. FASTA line by FASTA line straight into
 * page-locked buffers (mk_host_alloc).  The words that hold position `at` keep their lower positions,
 * everything above at + n in the words written becomes zero: pack in ascending order, nothing needs
 * clearing first.  Both arrays must have room for mk_pack_code_words / mk_pack_except_words of the
 * final length.  Returns 1 when any of the n characters was an exception (hand `except` over then),
 * 0 when none was, -1 on a null argument. */
uint64_t mk_pack_code_words(uint64_t len);
uint64_t mk_pack_except_words(uint64_t len);
int mk_pack_append(uint64_t *codes, uint64_t *except, uint64_t at, const char *chars, uint64_t n);

/* Miekki::insert_sequences (Miekki.cpp:277-314) for packed sequences; pipelined like
 * mk_index_append (the arrays may be reused when the call returns). */
int mk_index_append_packed(mk_ctx *ctx, const mk_packed_seq *seqs, uint32_t n);

/* Page-locked host memory for sequences handed to mk_index_append / mk_qset_upload:
 * from such buffers the copy to the GPU is a direct DMA (no staging copy on the
 * host) and runs beside the kernels.  Ordinary memory works too, only slower.
 * Thread-safe. */
int mk_host_alloc(mk_ctx *ctx, uint64_t bytes, void **out);
void mk_host_free(mk_ctx *ctx, void *p);

/* mk_index_append for the synthetic genomes of SURVEY.md 8d, generated on the device
 * (ids first_id .. first_id+n-1, `length` bases each): no PCIe traffic. */
int mk_index_append_synthetic(mk_ctx *ctx, uint64_t first_id, uint32_t n, uint64_t length);
/* The same for a collection of RELATED genomes (measurement aid for the column codec, mk_index_compress): genome g is
 * strain g % strains of species g / strains; strain 0 is the synthetic genome `species` of SURVEY.md 8d itself, the
 * others are independent descendants of it with about rate_ppm substitutions per million bases (tests/synth.py:
 * strain_device is the same generator on the host). */
int mk_index_append_synthetic_strains(mk_ctx *ctx, uint64_t first_id, uint32_t n, uint64_t length, uint32_t strains,
                                      uint32_t rate_ppm);

/* ---- gzip'd input on the device.  The reference reads every genome file through zstr::ifstream (Miekki.cpp:559-567;
 * zstr.hpp:78: inflateInit2(15 + 32), i.e. gzip members or plain text); these calls take the FILES' bytes as they are.
 * A file is decoded only if it is nothing but well-formed gzip members from its first byte to its last, every member's
 * CRC-32 and ISIZE included; anything else gets a status and is left to the caller's own inflater. */
typedef enum {
    MK_GZ_OK = 0,
    MK_GZ_EMPTY = 1,          /* no stream */
    MK_GZ_NOT_GZIP = 2,       /* no gzip header where one must be (magic, method, reserved flags) */
    MK_GZ_TRUNCATED = 3,      /* the input ends inside a member */
    MK_GZ_BAD_BLOCK = 4,      /* block type 3 */
    MK_GZ_BAD_STORED = 5,     /* a stored block's length and its complement disagree */
    MK_GZ_BAD_LENGTHS = 6,    /* a block's code lengths: over-subscribed, incomplete, bad repeat, no end-of-block code */
    MK_GZ_BAD_CODE = 7,       /* bits that are no code of the block's code */
    MK_GZ_BAD_DISTANCE = 8,   /* a match that reaches back beyond the member's first byte */
    MK_GZ_TOKEN_ROOM = 9,     /* (no longer reported: segments that outgrow their token slots are decoded again with exact rooms) */
    MK_GZ_OUTPUT_ROOM = 10,   /* more text than the room given */
    MK_GZ_TRAILING = 11,      /* bytes behind the last member that are not another member */
    MK_GZ_BAD_CRC = 12,
    MK_GZ_BAD_SIZE = 13,
    MK_GZ_INTERNAL = 14       /* the decoder disagrees with itself (never expected; the file goes to the host like any other failure) */
} mk_gz_status;

/* n whole files -> their text in out[i] (room out_room[i] bytes): out_bytes[i] and status[i] (mk_gz_status) per file. */
int mk_gz_inflate(mk_ctx *ctx, const uint8_t *const *gz, const uint64_t *gz_bytes, uint32_t n, uint8_t *const *out,
                  const uint64_t *out_room, uint64_t *out_bytes, int32_t *status);

/* n genome FILES (gzip'd FASTA) inflated and kept as text ON THE DEVICE, their sequences as index_file_of_file reads them
 * (Miekki.cpp:559-567: every line that does not start with '>' appended, line feeds dropped) measured: mk_gz_sequence says
 * how long file i's is, mk_index_append_gz appends files of the batch -- their sequences are stripped out of the text straight
 * into the build's buffers, nothing crosses PCIe again.  A file whose status is not MK_GZ_OK has no sequence here: inflate it
 * on the host.  The batch's memory (the files, their text and about as much again) is the context's until mk_gz_free -- the
 * appends that read it must have returned -- and stays with the context for the next batch until mk_gz_trim or mk_destroy.
 * mk_gz_unpack works on a stream of its own and touches nothing of the context but its device and its list of spare memory:
 * it may run on a thread of its own while other calls use the context (the driver inflates the next batch while it appends
 * the one before). */
typedef struct mk_gz_batch mk_gz_batch;
int mk_gz_unpack(mk_ctx *ctx, const uint8_t *const *gz, const uint64_t *gz_bytes, uint32_t n, mk_gz_batch **out);
/* mk_gz_unpack in steps, for a caller whose reader threads fetch the files themselves (zstr's read loop, zstr.hpp:186-190, is
 * what they replace): mk_gz_open lays the batch out from the files' sizes; mk_gz_put hands over bytes [at, at + bytes) of file i
 * -- staged != 0: `data` is a page-locked piece lent by mk_gz_stage (*cap bytes; NULL when none is to be had), the copy is a
 * DMA nobody waits for and the piece is the library's again; staged == 0: any memory, copied before the call returns --;
 * mk_gz_run, when every file has been put, inflates and measures the sequences (mk_gz_sequence, mk_index_append_gz).  mk_gz_stage
 * and mk_gz_put may be called from any number of threads at once; a file never put reads as empty (MK_GZ_NOT_GZIP). */
int mk_gz_open(mk_ctx *ctx, const uint64_t *gz_bytes, uint32_t n, mk_gz_batch **out);
void *mk_gz_stage(mk_gz_batch *batch, uint64_t *cap);
int mk_gz_put(mk_gz_batch *batch, uint32_t i, uint64_t at, const void *data, uint64_t bytes, int staged);
/* the files' places in the batch's input: offsets[0 .. n] (file i's bytes at offsets[i], zeros from its end to offsets[i + 1]);
 * files that follow each other, laid out like that in one piece, go with one call (and one copy): mk_gz_put_span */
int mk_gz_layout(const mk_gz_batch *batch, uint64_t *offsets);
int mk_gz_put_span(mk_gz_batch *batch, uint32_t first, uint32_t count, const void *data, uint64_t bytes, int staged);
int mk_gz_run(mk_gz_batch *batch);
int mk_gz_sequence(const mk_gz_batch *batch, uint32_t i, uint64_t *len, int32_t *status);
/* insert_sequences (Miekki.cpp:277-314) for files which[0 .. n) of the batch, in that order (each must have status MK_GZ_OK and
 * at least k characters); pipelined like mk_index_append. */
int mk_index_append_gz(mk_ctx *ctx, const mk_gz_batch *batch, const uint32_t *which, uint32_t n);
void mk_gz_free(mk_gz_batch *batch);
void mk_gz_trim(mk_ctx *ctx);                /* give the device memory kept between batches back (after a build) */

/* ---- persistence: the payload of dump_disk / the loading constructor
 * (Miekki.cpp:649-719), streamed in ranges so that the host never
 * needs the whole matrix at once. --------------------------------------------- */

/* columns [p_begin, p_end): (p_end-p_begin) * G * (fp_bits/8) bytes, partition-
 * major, 16-bit values big-endian -- byte-for-byte what dump_disk writes. */
int mk_index_export_columns(mk_ctx *ctx, uint32_t p_begin, uint32_t p_end, uint8_t *dst);
/* The columns of n chosen genomes (ids as the context reports them), every partition: dst[p][j] = the
 * fingerprint of genome ids[j] in partition p, 2^h * n * (fp_bits/8) bytes in dump_disk's byte order --
 * what dump_disk (Miekki.cpp:665-668) would write as the column block of an index holding just those genomes.
 * A sample of a 100,000-genome collection costs megabytes this way, not an export of the whole matrix. */
int mk_index_export_genomes(mk_ctx *ctx, const uint32_t *ids, uint32_t n, uint8_t *dst);
int mk_index_export_sizes(mk_ctx *ctx, uint64_t *genome_size, uint32_t *sketch_size);
/* Bloom bytes [begin, end) of the 2^(b-3)-byte table */
int mk_index_export_bloom(mk_ctx *ctx, uint64_t begin, uint64_t end, uint8_t *dst);

int mk_index_import_begin(mk_ctx *ctx, uint32_t n_genomes);    /* empties the index first */
int mk_index_import_columns(mk_ctx *ctx, uint32_t p_begin, uint32_t p_end, const uint8_t *src);

/* mk_index_import_columns for rows that are still Huffman-coded -- the form this program's own index files keep the
 * fingerprint columns in (host/fastz.cpp, deflate_huffman_only: deflate blocks of literals only, one Huffman code of at
 * most 12 bits per MiB, a block per 16 KiB; each gzip member's extra field lists where every block's first symbol lies).
 * The device inflates them, one lane per block (huff.hip): the reference's `in->read` of the columns (Miekki.cpp:708-712)
 * without an inflater on the host.  `payload`: the members' deflate streams as they lie in the file, one after the other
 * (page-locked memory makes the copy a DMA).  `blocks`: n_blocks of them, a multiple of 64 -- every 64 consecutive entries
 * use ONE code (entry 64 i's `code`; slots with out_len 0 fill a group up) -- that together produce exactly the bytes of
 * rows [p_begin, p_end) in the dump's byte order.  `lens`: n_codes x 257 code lengths (literals 0 .. 255, end-of-block).
 * Out: crc_out[i] = the CRC-32 remainder of block i's bytes (start value 0, no final complement: the caller folds them
 * with the blocks' lengths and compares with the members' trailers); *bad_out = blocks whose code is not a complete code
 * of at most 12 bits, that point outside the payload or the rows, hold an end-of-block code early or lack it at their
 * end -- the index is not usable unless it is 0 and the CRCs hold. */
typedef struct {
    uint64_t bit;       /* where the block's first symbol starts: bits from payload[0] */
    uint64_t out;       /* where its bytes go: bytes from the first byte of row p_begin */
    uint32_t out_len;   /* how many bytes it holds */
    uint32_t code;      /* which of the n_codes length tables it is coded with */
} mk_huff_block;
int mk_index_import_columns_huffman(mk_ctx *ctx, uint32_t p_begin, uint32_t p_end, const uint8_t *payload, uint64_t payload_bytes,
                                    const mk_huff_block *blocks, uint32_t n_blocks, const uint8_t *lens, uint32_t n_codes,
                                    uint32_t *crc_out, uint32_t *bad_out);
int mk_index_import_sizes(mk_ctx *ctx, const uint64_t *genome_size, const uint32_t *sketch_size);
int mk_index_import_bloom(mk_ctx *ctx, uint64_t begin, uint64_t end, const uint8_t *src);

/* ---- queries --------------------------------------------------------------- */

/* Miekki::query_sequences (Miekki.cpp:344-372): scores[nq][G], row-major u32. */
int mk_query_scores(mk_ctx *ctx, const char *const *seqs, const uint64_t *lens, uint32_t nq,
                    uint32_t *scores);

/* filter_results(query_sequences(batch), nresults, min_score, min_intersection)
 * (Miekki.cpp:437; 376-422): hits[nq][nresults], nhits[nq]; each row descending by
 * intersection with the reference's heap tie behaviour.  active[nq] (may be NULL)
 * receives query_sequence's active_minimizer (Miekki.cpp:318-340). */
int mk_query(mk_ctx *ctx, const char *const *seqs, const uint64_t *lens, uint32_t nq,
             uint32_t nresults, uint32_t min_score, double min_intersection,
             mk_hit *hits, uint32_t *nhits, uint32_t *active);

/* Pure host function: the heap of Miekki::filter_results (Miekki.cpp:386-396) over
 * candidates that already passed both thresholds, given in ascending genome
 * order (all of them, or just the heap entrants mk_qset_run emits).  Used by
 * mk_query and by the multi-GPU merge on rank 0. */
uint32_t mk_filter_candidates(const mk_hit *cand, uint32_t ncand, uint32_t nresults, mk_hit *out);

/* The same heap on the device, for whole batches: d_count[world][nq] and
 * d_cand[world][nq][cap] are the entrant rows of `world` genome shards in shard
 * order (world = 1: what mk_qset_run wrote; world > 1: the gathered rows on rank
 * 0).  Writes d_hits[nq][nresults] and d_nhits[nq]; a query for which some shard's
 * count exceeds `cap` gets d_nhits = MK_MERGE_OVERFLOW and no hits (answer it
 * again from a dense score row).  nresults <= 64.  All pointers are device
 * memory; asynchronous on the context's stream. */
#define MK_MERGE_OVERFLOW 0xffffffffu
int mk_merge_entrants(mk_ctx *ctx, const uint32_t *d_count, const mk_hit *d_cand, uint32_t world,
                      uint32_t nq, uint32_t cap, uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits);

/* filter_results' heap over gathered exchange rows: d_rows[world][nq][1 + cap] (what
 * mk_qset_run_compact wrote on `world` genome shards, in shard order).  Sizes of the
 * genomes the rows name come from mk_merge_set_sizes (all shards' sketch_size /
 * genome_size concatenated in shard order; id_base = id of entry 0); without it, and
 * with world == 1, the context's own index is used.  Output as mk_merge_entrants. */
int mk_merge_set_sizes(mk_ctx *ctx, const uint64_t *genome_size, const uint32_t *sketch_size,
                       uint32_t n_genomes, uint32_t id_base);
/* the sizes mk_merge_set_sizes (or mk_comm_share_sizes) left on the context, back on the host (n = how many were set) */
int mk_merge_get_sizes(mk_ctx *ctx, uint64_t *genome_size, uint32_t *sketch_size, uint32_t n);
int mk_merge_compact(mk_ctx *ctx, const uint64_t *d_rows, uint32_t world, uint32_t nq, uint32_t cap,
                     uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits);

/* ---- device-resident query sets (bench / multi-GPU path) ------------------- */

typedef struct mk_qset mk_qset;

/* Upload sequences once; they stay in HBM until mk_qset_free. */
int mk_qset_upload(mk_ctx *ctx, const char *const *seqs, const uint64_t *lens, uint32_t nq,
                   mk_qset **out);
/* SURVEY 8d synthetic queries first_id.. cut from synthetic genomes
 * (id mod n_genomes_total, length genome_len), generated on the device. */
int mk_qset_synthetic(mk_ctx *ctx, uint64_t first_id, uint32_t nq, uint64_t n_genomes_total,
                      uint64_t genome_len, uint64_t query_len, mk_qset **out);
void mk_qset_free(mk_ctx *ctx, mk_qset *qs);

/* One pass of the hot path over the set: sketch + Bloom gate + scan + top-hit
 * selection.  For every query the device emits, in ascending genome id, the
 * ENTRANTS of filter_results' heap (Miekki.cpp:376-397): the genomes that pass
 * min_score / min_intersection and are not skipped by the heap-minimum test
 * (Miekki.cpp:387) for a heap of `nresults` (<= 64).  Only entrants shape the
 * reference's result, so mk_filter_candidates over a row reproduces it exactly;
 * rows of several genome shards concatenated in shard order do too (multi-GPU
 * merge).  Output goes to device memory owned by the caller: d_count[nq] (u32; a
 * value above `cap` marks a row that overflowed) and d_cand[nq][cap] (mk_hit).
 * Asynchronous on the context's stream; mk_sync waits.
 * min_score 0 over an index that holds a genome with sketch_size 0 (a sequence
 * exactly k long) would make NaN intersections (0 / 0, Miekki.cpp:381-383), whose
 * fate in the reference's heap follows no order: MK_ERR_UNSUPPORTED -- mk_query
 * answers such calls, through the host's own heap calls. */
int mk_qset_run(mk_ctx *ctx, mk_qset *qs, uint32_t nresults, uint32_t min_score,
                double min_intersection, uint32_t cap, uint32_t *d_count, mk_hit *d_cand);
/* The same pass with the output in the 8-byte EXCHANGE form of the multi-GPU path
 * ("(genome_id u32, score u32)" per entrant): d_rows[nq][1 + cap] 64-bit
 * words per query -- word 0 = number of entrants (a value above `cap` marks a row that
 * overflowed), then genome | matches << 32 in ascending genome id.  jaccard and
 * intersection are not shipped: the merging side recomputes them from the two sizes
 * of the genome (mk_merge_set_sizes) with the reference's operations
 * (Miekki.cpp:382-383), bit-identically.  One buffer, so one collective. */
int mk_qset_run_compact(mk_ctx *ctx, mk_qset *qs, uint32_t nresults, uint32_t min_score,
                        double min_intersection, uint32_t cap, uint64_t *d_rows);
/* A set keeps its sketch, Bloom gate result and schedule tables until the index changes
 * (genomes appended / imported, Bloom cells written); this forces the next run to redo
 * them anyway (bench.py: a timed step is a complete pass). */
int mk_qset_invalidate(mk_ctx *ctx, mk_qset *qs);
/* Raw scores of queries [q_begin, q_end) of the set into d_scores[(q_end-q_begin)][G]. */
int mk_qset_scores(mk_ctx *ctx, mk_qset *qs, uint32_t q_begin, uint32_t q_end, uint32_t *d_scores);
/* active partitions per query after the last run/scores call */
int mk_qset_active(mk_ctx *ctx, mk_qset *qs, uint32_t *active);
int mk_sync(mk_ctx *ctx);

/* ---- several GPUs in one process (the `miekki` binary; Miekki.cpp:546-581, 430-480 scale
 * inside one executable with -t).  One context per GPU, genomes sharded in list order. ---- */

int mk_device_count(void);                                  /* visible HIP devices (0 without a GPU) */
/* ids this context reports = local index + base; may be set after the build, once the
 * number of genomes the shards before it kept is known. */
int mk_set_genome_id_base(mk_ctx *ctx, uint32_t base);
/* Device memory on the context's GPU for callers without a GPU runtime of their own. */
int mk_dev_alloc(mk_ctx *ctx, uint64_t bytes, void **d_out);
void mk_dev_free(mk_ctx *ctx, void *d_ptr);
int mk_dev_upload(mk_ctx *ctx, void *d_dst, const void *src, uint64_t bytes);
int mk_dev_download(mk_ctx *ctx, void *dst, const void *d_src, uint64_t bytes);
/* d_src on src's GPU -> d_dst on dst's GPU: a peer DMA over xGMI, ordered behind src's
 * queued work; complete on return. */
int mk_dev_copy(mk_ctx *dst, void *d_dst, mk_ctx *src, const void *d_src, uint64_t bytes);
/* Bloom cells [begin, end) to / from device memory of the context's GPU. */
int mk_index_export_bloom_device(mk_ctx *ctx, uint64_t begin, uint64_t end, uint8_t *d_dst);
int mk_index_import_bloom_device(mk_ctx *ctx, uint64_t begin, uint64_t end, const uint8_t *d_src);
/* First-writer fold of a genome-sharded build (the reference's single filter keeps the byte of a
 * cell's first inserter in genome order, Miekki.cpp:125-129): cells [begin, end) of this context
 * keep their byte where it is non-zero and take d_later's otherwise.  Called on the first shard
 * with the filters of the later shards in shard order; begin must be a multiple of 16. */
int mk_index_merge_bloom_device(mk_ctx *ctx, uint64_t begin, uint64_t end, const uint8_t *d_later);
/* bytes of the Bloom table a 2k-bit k-mer can reach (everything above stays zero) */
uint64_t mk_bloom_reachable_bytes(const mk_ctx *ctx);

/* ---- one process per GPU: the collectives of the genome-sharded index, on RCCL over xGMI --------------
 * The reference scales inside one executable (-t threads, main.cpp:190-196; Miekki.cpp:546-581, 430-480) and has
 * no distributed backend; SURVEY.md 8e defines the multi-GPU form: rank r owns a contiguous genome range, the
 * Bloom filter is made global once after the build, and a query batch needs ONE exchange step -- ncclGather
 * (rccl.h:745) of the per-query heap-entrant rows (mk_qset_run_compact) to the merging rank.  librccl is bound at
 * run time (a process that already holds a copy, e.g. PyTorch's, keeps using it; MIEKKI_RCCL_LIB names one);
 * single-GPU users never load it.  One communicator per context; every call below is COLLECTIVE: all ranks make
 * it, in the same order.  Device pointers (d_*) are memory of the context's GPU; the collectives are queued on
 * the context's stream like its kernels (mk_sync waits), so a gather needs no host wait after the scan. */
typedef struct mk_comm mk_comm;
#define MK_COMM_ID_BYTES 128
/* rank 0 draws the id (ncclGetUniqueId) and hands it to the other ranks by any means (a file, the launcher's
 * store, an environment variable); then every rank creates its communicator (ncclCommInitRank). */
int mk_comm_unique_id(uint8_t *id /* MK_COMM_ID_BYTES */);
/* (This RCCL build prints a version banner on stdout from rank 0's first communicator.  With MIEKKI_COMM_BANNER_TO_STDERR=1
 * in the environment mk_comm_create points descriptor 1 at descriptor 2 while the communicator initialises -- process-wide,
 * for as long as the slowest rank takes to arrive: only for programs that own their stdout, like the `miekki` binary and
 * bench.py, which set it themselves.  Without it nothing is redirected.) */
int mk_comm_create(mk_ctx *ctx, int rank, int world, const uint8_t *id, mk_comm **out);
void mk_comm_destroy(mk_comm *comm);
int mk_comm_rank(const mk_comm *comm);
int mk_comm_world(const mk_comm *comm);
/* ncclGather / ncclAllGather / ncclBroadcast of plain bytes; d_recv holds world * bytes (gather: on root only). */
int mk_comm_gather(mk_comm *comm, const void *d_send, uint64_t bytes, void *d_recv, int root);
int mk_comm_allgather(mk_comm *comm, const void *d_send, uint64_t bytes, void *d_recv);
int mk_comm_broadcast(mk_comm *comm, void *d_buf, uint64_t bytes, int root);
int mk_comm_allreduce_max_f64(mk_comm *comm, double *d_values, uint32_t n);
int mk_comm_barrier(mk_comm *comm);                                /* waits on the host, too */
/* The single exchange step: every rank's exchange rows (`words` 64-bit words, what mk_qset_run_compact wrote) ->
 * d_recv[world][words] on root, rank-major = shard order = genome order: the layout mk_merge_compact takes. */
int mk_comm_gather_rows(mk_comm *comm, const uint64_t *d_rows, uint64_t words, uint64_t *d_recv, int root);
/* Once after the build.  The reference has ONE Bloom filter and a cell keeps the byte of its first inserter in
 * genome order (Miekki.cpp:125-129); with contiguous shards in rank order that is the lowest rank whose cell is
 * non-zero: a MIN all-reduce over (rank << 8 | byte), in place on the context's filter, byte-exact. */
int mk_comm_sync_bloom(mk_comm *comm);
/* Once after the build: all-gathers how many genomes every rank holds and their sketch_size / genome_size, sets
 * this context's genome id base to the genomes of the ranks before it (ids = those of a single-process run) and
 * hands all sizes to the merge (mk_merge_set_sizes).  id_base / total may be NULL. */
int mk_comm_share_sizes(mk_comm *comm, uint32_t *id_base, uint32_t *total);
/* mk_qset_run_compact with its exchange step folded in: the rows of queries that are finished leave for `root` on
 * the communicator's own stream WHILE the next chunk of queries is scanned (sets of >= 4096 queries go in four
 * blocks, MIEKKI_EXCHANGE_BLOCKS; a smaller set is one ncclGather).  d_rows[nq][1 + cap] on every rank,
 * d_recv[world][nq][1 + cap] on root (NULL elsewhere).  On return everything is queued and the context's stream
 * waits for the exchange: mk_merge_compact(ctx, d_recv, world, ...) may follow at once. */
int mk_qset_run_compact_gather(mk_ctx *ctx, mk_comm *comm, mk_qset *qs, uint32_t nresults, uint32_t min_score,
                               double min_intersection, uint32_t cap, uint64_t *d_rows, uint64_t *d_recv, int root);

/* ---- exact mode (ground_truth_batch, Miekki.cpp:792-859) ------------------- */

/* genome: contig sequences as ground_truth_batch delimits them (Miekki.cpp:803-822);
 * for each query: |A n B| and |B| + |A \ B| over distinct canonical k-mers. */
int mk_exact(mk_ctx *ctx, const char *const *contigs, const uint64_t *contig_lens,
             uint32_t n_contigs, const char *const *queries, const uint64_t *query_lens,
             uint32_t nq, uint64_t *inter, uint64_t *uni);
/* The same in two steps, for callers that verify several batches of queries against one
 * genome file (the reference rebuilds its unordered_set at every flush of 100 pending
 * queries, Miekki.cpp:745-748, ~7 s each): the genome's k-mer set stays resident on the
 * context until the next load.  Any number of contigs and queries per call. */
int mk_exact_load_genome(mk_ctx *ctx, const char *const *contigs, const uint64_t *contig_lens,
                         uint32_t n_contigs);
int mk_exact_query(mk_ctx *ctx, const char *const *queries, const uint64_t *query_lens, uint32_t nq,
                   uint64_t *inter, uint64_t *uni);

#ifdef __cplusplus
}
#endif
#endif
