// Internal declarations shared by the translation units of libmiekki_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/miekki_hip.h"
#include "mk_device.hpp"

namespace mk {

void set_error(const char *fmt, ...);
#define MK_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            mk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                          __LINE__);                                                          \
            return e_ == hipErrorOutOfMemory ? MK_ERR_NOMEM : MK_ERR_DEVICE;                  \
        }                                                                                     \
    } while (0)
#define MK_TRY(expr)                                                                          \
    do {                                                                                      \
        int s_ = (expr);                                                                      \
        if (s_ != MK_OK) return s_;                                                           \
    } while (0)

constexpr uint32_t kShortMax = 4096;     // k-mers a query may have for the in-LDS sketch
}  // namespace mk
// ONE definition of "more k-mers than the short path takes" (len - k of them are sketched: Miekki.cpp:162 skips the last), used
// by a query set's own classes and by the callers that split a mixed set
struct mk_ctx;
namespace mk {
inline bool beyond_short_len(uint32_t k, uint64_t len) { return len > (uint64_t)k + kShortMax; }
constexpr uint32_t kBuildBatch = 64;     // most genomes sketched per build batch
constexpr uint32_t kTileBytes = 1024;    // bytes of one matrix row a wave scans (64 lanes x 16 B)
constexpr uint64_t kEmptyKey = ~0ULL;
constexpr int kPosBits = 40;                      // a minimum key in a table: fingerprint << 40 | position
constexpr int kCopyStreams = 2;                   // copy streams of a packed append (see mk_ctx::copy_extra; three: 30.9k sketches/s, four: 37.8k, two: 39.0k)
constexpr uint32_t kBloomRegionLog2 = 12;          // cells per region of the Bloom sweep (bloom_sweep_kernel): one pass of a workgroup

// packed query-sketch entry: partition in the low word, fingerprint in the high word
__host__ __device__ inline uint64_t make_entry(uint32_t p, uint32_t fp) { return (uint64_t)p | ((uint64_t)fp << 32); }

struct Timer {
    hipEvent_t a, b;
    int kind;            // 0 sketch, 1 scan, 2 filter, 3 build sketch, 4 build finalize
};

}  // namespace mk

// Where the rows of the fingerprint matrix live (SURVEY 8f row N4 -- what compress_index /
// decompress_index were for in the reference, Miekki.cpp:863-877: a collection larger than fast
// memory).  Rows [0, P_hot) are in HBM; when the matrix exceeds its HBM budget the remaining COLD
// rows sit in page-locked host memory (device-visible).  Kernels address row p through mat_row;
// the slab schedule streams whole cold partition ranges through a staging buffer in HBM instead
// of reading them piecemeal over PCIe (api.hip: qset_scan_slab).
struct MatRef {
    uint8_t *hot;                  // row p < P_hot at hot + p * ld
    uint8_t *cold_m;               // row p >= P_hot at cold_m + p * ld (= cold rows' base - P_hot * ld); null when all rows are hot
    uint32_t P_hot;
};
__device__ __forceinline__ uint8_t *mat_row(const MatRef &m, uint32_t p, uint64_t ld)
{
    return (p < m.P_hot ? m.hot : m.cold_m) + (uint64_t)p * ld;
}

struct mk_ctx {
    mk_params p;
    uint32_t P, W, f, empty;
    hipStream_t stream;
    // fingerprint matrix, partition-major: row p at d_M + p * ld (bytes); 16-bit values native LE
    uint8_t *d_M;
    uint64_t ld;
    uint8_t *h_M;                  // cold rows [P_hot, P) in page-locked host memory, same pitch (null: all rows in HBM)
    uint32_t P_hot;                // rows kept in HBM (= P when the matrix fits its budget)
    uint64_t hbm_matrix_budget;    // bytes of HBM the matrix may take (MIEKKI_HBM_MATRIX_MIB; 0 = whatever is free)
    // the cold rows PACKED (cold.hip: mk_index_compress): h_M is null then, h_Z holds them, row i of the cold rows at
    // h_Z + h_zoff[i]; d_zoff = the same offsets on the device; d_zstage = packed staging beside d_cold_stage's halves
    uint8_t *h_Z;
    uint64_t z_bytes;
    uint64_t *d_zoff;
    std::vector<uint64_t> h_zoff;
    uint8_t *d_zstage[2];
    uint64_t zstage_cap;
    uint8_t *d_cold_stage;         // HBM staging for cold partition ranges (slab schedule)
    uint64_t cold_stage_rows;      // rows of ONE of its two halves
    hipEvent_t ev_cold[5];         // copy done [2], scan done [2], entry
    uint32_t capG, G;
    uint32_t *d_sketch_size;
    uint64_t *d_genome_size;
    float *d_ratio;                // genome_size / sketch_size (select.hip), capG entries, current as of ratio_gen
    uint64_t ratio_gen, ratio_cap;
    uint8_t *d_colstage;           // staging of mk_index_export_columns / _import_columns (kept between calls)
    uint64_t colstage_cap;
    uint8_t *d_huff = nullptr;     // mk_index_import_columns_huffman: coded bytes, block list, code lengths, remainders (kept between calls)
    uint64_t huff_cap = 0;
    void *exact_buf[10];           // exact mode (K7) scratch, grown on demand, freed with the context
    uint64_t exact_cap[10];
    bool exact_have_B;             // set B of the genome loaded last (mk_exact_load_genome) is resident
    uint64_t exact_nB;             // its number of distinct k-mers
    uint32_t exact_log2B;
    bool has_empty_sketch;         // some genome has sketch_size 0 (see nan_candidates_possible in api.hip)
    bool next_single = false;      // the batch being queued comes from mk_index_insert_sequence
    // sizes of ALL genomes of a sharded index (mk_merge_set_sizes), for the compact merge on the
    // context that receives the gathered rows
    uint32_t *d_all_ss;
    uint64_t *d_all_gs;
    uint32_t all_n, all_base;
    uint64_t gen;                  // index generation: bumped whenever genomes or Bloom cells change
    std::vector<uint32_t> h_sketch_size;
    std::vector<uint64_t> h_genome_size;
    // Bloom filter: only the cells a 2k-bit k-mer can reach live on the device
    uint8_t *d_bloom;
    uint64_t bloom_dev_bytes, bloom_bytes;
    uint32_t *d_bloom_order;       // first-writer arbitration keys (build only), lazily allocated: one per reachable cell,
                                   // (genome in batch << h | partition) << 3 | bit of the first k-mer (in that order) that asked for the cell
    // build scratch (lazily allocated)
    uint32_t build_batch;          // genomes per build batch (<= kBuildBatch, bounded by table memory)
    uint64_t *d_tables;            // build_batch x P min-keys
    uint32_t *d_active;            // per batch genome: non-empty partitions
    uint64_t *d_cardsum;           // per batch genome: sum of 2^(31-exp)
    uint32_t *d_seed_valid;
    // batch sequences, concatenated.  Two buffers: while the kernels of one batch run out
    // of one, the next batch is copied into the other on copy_stream (mk_index_append)
    char *d_seq[2];
    uint64_t seq_cap[2];
    int seq_cur;                   // buffer of the batch enqueued last
    uint64_t *d_seq_off;           // kBuildBatch + 1
    uint32_t *d_dirty;             // per batch genome: some character is not A, C, G or T (alias of the side's counters)
    // one bit per 8 Bloom cells: all eight are non-zero (so none of them can change any more).
    // 1 MiB for the 64 MiB of reachable cells at -b 33: L2-resident, and once the filter has
    // filled up it answers almost every probe of the build without touching the cells
    uint32_t *d_bloom_full;
    uint64_t *d_bloom_full2;       // coarse level: one bit per 2048 cells = all of them are taken (the build's scatter kernel asks it)
    bool bloom_full_stale;         // the cells were written behind the summary's back (import)
    hipStream_t copy_stream;
    hipEvent_t ev_copy;
    // further copy streams for batches that arrive as many separate buffers (mk_index_append_packed: one per sequence,
    // ~1 MB each): a DMA engine spends as long on starting such a copy as on moving it, several engines overlap that
    static constexpr int kCopyExtra = mk::kCopyStreams - 1;
    hipStream_t copy_extra[kCopyExtra];
    hipEvent_t ev_extra[kCopyExtra];
    int n_copy_extra;
    // The batch whose kernels are enqueued but whose results (active counts, cardinality
    // sums, overflow mark) the host has not folded into the index yet.  Every entry point
    // other than the appends settles it first (settle_build in api.hip).
    struct BuildInFlight {
        bool on, binned;
        bool have_chars;           // the batch's characters are in d_seq[buf] (else only its packed form exists, in d_pk[buf])
        bool have_heads;           // d_heads[buf] holds the sequences' first 32 characters (packed input)
        bool single;               // mk_index_insert_sequence: the size estimate of Miekki.cpp:243-273 (active count in a double)
        uint32_t n;
        uint32_t g0;               // its first column of the matrix (set when the back stage is queued)
        int buf;
        uint64_t off[mk::kBuildBatch + 1];
    } build, older, front;         // build: the batch whose back stage was queued last; older: the one before it, not folded
                                   // in yet (its back stage is running or done); front: front stage queued, back stage not
    uint32_t G_back;               // genomes whose back stage has been queued: G + those of `older` and `build`
    // The batch in packed form (build.hip): d_pk[b] = [codes: pk_cap[b] bytes][exception bits: pk_cap[b] / 2 bytes],
    // sequence g at byte offset pk_off[g] (16-byte aligned) of the codes and pk_off[g] / 2 of the exception bits.
    // Two of everything: the next batch is generated or copied into one while the kernels of the batch in flight
    // read the other.
    uint8_t *d_pk[2];
    uint64_t pk_cap[2];
    uint64_t *d_pk_off[2];         // kBuildBatch + 1 offsets
    char *d_heads[2];              // kBuildBatch x 32 characters
    // host images of the small per-batch arrays: they are copied asynchronously, and an append returns with its
    // batch still in flight, so they live here and not on a caller's stack
    // (page-locked: a copy from pageable memory makes the host wait until the stream has reached it -- for the
    // front stream that is the end of the batch's PCIe copy)
    struct HostImages {
        uint64_t pk_off[2][mk::kBuildBatch + 1];
        uint64_t off[2][mk::kBuildBatch + 1];
        uint32_t dirty[2][mk::kBuildBatch];
        char heads[2][mk::kBuildBatch * 32];
    } *h_img;
    // Per-batch counters of the build live in ONE device block (one memset before a batch, one copy
    // back after it): d_ovf_count, d_dirty, d_active, d_cardsum point into d_counters.  BuildCounters
    // is its layout, and that of the pinned read-back block.
    struct BuildCounters {
        uint32_t ovf, pad;             // overflow mark of the batch (> the fold limit: the host redoes the batch)
        uint32_t dirty[mk::kBuildBatch];
        uint32_t act[mk::kBuildBatch];
        uint64_t card[mk::kBuildBatch];
    };
    BuildCounters *d_counters;
    BuildCounters *h_back;         // pinned, so that the copy back does not block the host
    // The build runs as a two-stage pipeline over two SIDES (build.hip): the FRONT stage of a batch -- packing,
    // seed digits, scatter: instruction-bound -- is queued on front_stream while the BACK stage of the batch before
    // it -- reduce, matrix rows, Bloom passes: bound by memory latency -- still runs on `stream`; the two kinds of
    // kernels share the CUs.  Everything a front stage writes is per side.  d_counters, h_back, d_seq_off,
    // d_seed_valid, d_ovf (and d_ovf_count, d_dirty, d_active, d_cardsum) above alias the side whose back stage was
    // queued last; the long-query sketches, which run only when no batch is in flight, borrow them.
    struct BuildSide {
        BuildCounters *d_counters, *h_back;
        uint64_t *d_seq_off;
        uint32_t *d_seed_valid;
        uint64_t *d_ovf;
        hipEvent_t ev_back;        // the batch's back stage (and its counters' copy back) is done
        void *d_slots;
        uint64_t slots_bytes;
        hipEvent_t ev_front;       // front stage done
        uint32_t shape[8];         // BuildShape of the batch (build.hip)
        bool fits, key32;
    } side[2];
    hipStream_t front_stream;
    // sizes of the batch just settled on their way to the device (pinned, one per buffer parity: the copy
    // is queued, not waited for)
    struct SizeUpload { uint32_t ss[mk::kBuildBatch]; uint64_t gs[mk::kBuildBatch]; } *h_sizes;
    int size_parity;
    // query scratch
    uint32_t *d_scores;            // score matrix of the query chunk in flight
    uint64_t scores_cap;
    uint8_t *d_partials;           // slab schedule: per-range mismatch counts of the chunk in flight
    uint64_t partials_cap;         // bytes
    uint32_t *d_flag;              // one word for device-side eligibility checks
    uint32_t *d_count;
    mk_hit *d_cand;
    uint64_t cand_cap_q;           // queries d_count / d_cand are sized for
    mk_hit *d_hits;                // [queries per chunk][nresults]: K6b output of mk_query
    uint32_t *d_nhits;
    uint64_t hits_cap;             // records d_hits is sized for
    uint64_t nhits_cap;            // queries d_nhits is sized for
    // small calls (mk_query with a handful of queries) are round trips, not kernels: one cached
    // device arena for the call's transient query set, one pinned block for its upload image and
    // one for its results, so that a call is one copy in, the kernels, one sync, one copy out
    uint8_t *d_qarena;
    uint64_t qarena_cap;
    bool qarena_busy;
    uint8_t *h_stage;              // pinned
    uint64_t stage_cap;
    uint8_t *h_res;                // pinned
    uint64_t res_cap;
    uint64_t *d_long_table;        // P keys, long-query path
    uint8_t *d_fpT;                // build: fingerprints of the batch, genome-major [build_batch][P] (fused build kernel)
    uint8_t *d_bloom_touched;      // build: per region of 2^kBloomRegionLog2 cells "a first-writer key was posted here" (allocated with d_bloom_order),
                                   // and behind those flags (bloom_regions(c) of them) "swept since the last summary": what bloom_summary_kernel redoes
    uint64_t *d_ovf;               // (genome << 32 | bucket, item) pairs that missed their slot
    uint32_t *d_ovf_count;
    // stats
    mk_stats stats;
    std::vector<mk::Timer> pending;
    std::vector<mk::Timer> free_timers;
    // device blocks of finished mk_gz_unpack batches, kept for the next ones (allocating and freeing tens of gigabytes per
    // batch cost half a second each): (pointer, bytes); mk_gz_trim and mk_destroy free them
    struct GzBlock { void *first; uint64_t second; int role; };   // role: 0 a batch's input + token block, 1 its text block
    std::vector<GzBlock> gz_blocks;
    uint32_t gz_making = 0;              // block makers at work (threads of their own: gunzip.hip, gz_open)
    bool gz_closing = false;
    std::condition_variable gz_cv;
    std::vector<std::pair<void *, uint64_t>> gz_pins;   // page-locked staging of the inflater's small copies, kept likewise
    std::mutex gz_m;
    // page-locked pieces the callers read their files into (mk_gz_stage): free ones, and those whose copy to the device is
    // still queued (each with the event that says when it is done), oldest first; the streams those copies take turns on
    struct GzStage { void *p; hipEvent_t ev; };
    std::vector<GzStage> gz_stage_free;
    std::deque<GzStage> gz_stage_busy;
    std::vector<GzStage> gz_stage_held;  // ... and those a caller is reading a file into
    uint32_t gz_stage_made = 0;
    hipStream_t gz_up[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t gz_up_next = 0;
};

namespace mk { struct DenseLut; struct mk_gz_stream; struct mk_gz_seg; }
struct mk_qset {
    uint32_t nq;
    mk_ctx *owner;
    bool arena_borrowed;           // d_arena is the context's cached arena (transient sets of mk_query)
    uint64_t head_bytes;           // arena bytes [0, head_bytes) = sequences, offsets, entry offsets: the upload image
    uint64_t o_off, o_ent_off;     // byte offsets of d_off / d_ent_off in the arena (d_seq is at 0)
    uint8_t *d_arena;              // the set's one device allocation; the arrays below point into it
    bool split_in_arena;           // d_split too (room for split_room ranges), else it is its own allocation
    uint32_t split_room;
    char *d_seq;
    uint64_t total_len;
    uint64_t *d_off;               // nq + 1 offsets into d_seq
    uint64_t *d_ent_off;           // nq + 1 offsets into d_entries (capacity max(len-k,0) each)
    uint64_t *d_entries;
    uint32_t *d_nent;              // active partitions per query
    std::vector<uint64_t> h_off, h_ent_off;
    std::vector<uint32_t> long_q;  // queries with more than kShortMax k-mers (sparse long path)
    // dense long path: queries with >= P/4 k-mers (whole genomes) keep their full 2^h
    // fingerprint vector, four queries interleaved per group: dense[group][p][4] (W bytes each)
    std::vector<uint32_t> dense_q; // set indices, group-major; 0xffffffff pads the last group
    uint8_t *d_dense;
    uint32_t *d_dense_q;
    mk::DenseLut *d_dense_lut = nullptr;   // one-byte fingerprints: the dense queries' field tables, [octet][P] (scan_kernel.hpp)
    uint32_t *d_scan_n;            // entries the sparse scan walks: nent for sparse queries, 0 for dense ones
    uint32_t short_max_nk;         // longest short query (k-mers)
    uint32_t *d_split;             // [nq][S + 1] entry index of each partition-range boundary
    uint32_t S;                    // ranges of the slab schedule (0 = not prepared)
    uint32_t chunk;                // small sets: entries per range, ranges cut by count (0 = by partition, d_split)
    bool slab_ok;                  // every (query, range) fits the packed counters
    bool sketched;
    uint64_t gen;                  // index generation the sketch / range table were made against
    // A MIXED set (short queries next to long reads / contigs / whole genomes; query_file batches whatever the file holds,
    // Miekki.cpp:465-471) is a shell over two sets of its own -- the short queries, which keep the slab schedule, and the
    // others, which take the plain / dense kernels: part[i] runs as a set, part_q[i] = its queries' places in this set
    // (ascending), d_part_q[i] the same on the device, and the parts' rows are put in their places behind the parts' runs.
    mk_qset *part[2] = {nullptr, nullptr};
    std::vector<uint32_t> part_q[2];
    uint32_t *d_part_q[2] = {nullptr, nullptr};
    uint8_t *d_part_out = nullptr;  // a part's output rows before they go to their places
    uint64_t part_out_bytes = 0;
};

namespace mk {

inline MatRef mat_ref(const mk_ctx *c)
{
    MatRef m;
    m.hot = c->d_M;
    m.P_hot = c->P_hot;
    m.cold_m = c->h_M ? c->h_M - (uint64_t)c->P_hot * c->ld : nullptr;
    return m;
}

// parameters every sketch kernel takes
struct SketchParams {
    uint32_t k, h, f, empty, bloom_log2, P;
    uint64_t kmask;
};
inline SketchParams make_sp(const mk_ctx *c)
{
    SketchParams s;
    s.k = c->p.k; s.h = c->p.h; s.f = c->f; s.empty = c->empty; s.bloom_log2 = c->p.bloom_log2;
    s.P = c->P;
    s.kmask = (c->p.k < 32) ? ((1ULL << (2 * c->p.k)) - 1) : ~0ULL;
    return s;
}

// timing helpers (api.hip)
int timer_begin(mk_ctx *c, int kind, Timer &t, hipStream_t stream = nullptr);
int timer_end(mk_ctx *c, Timer &t, hipStream_t stream = nullptr);
int drain_timers(mk_ctx *c);
struct ScopedTimer {
    mk_ctx *c; Timer t; bool on; hipStream_t stream;
    ScopedTimer(mk_ctx *ctx, int kind, hipStream_t st = nullptr) : c(ctx), on(false), stream(st) { on = timer_begin(c, kind, t, stream) == MK_OK; }
    ~ScopedTimer() { if (on) (void)timer_end(c, t, stream); }
};

// ---- api.hip, for its sister files (api_query.hip, api_multi.hip)
// gunzip.hip: the inflater keeps its batches' device blocks for the next batches (tens of gigabytes after an ingest of gzip'd
// genomes); whoever finds device memory short gives the idle ones back -- every context's -- and tries again
uint64_t gz_release_idle_blocks();
template <typename T>
inline int dev_alloc(T **p, uint64_t count)
{
    *p = nullptr;
    if (!count) return MK_OK;
    if (hipMalloc((void **)p, count * sizeof(T)) == hipSuccess) return MK_OK;
    (void)hipGetLastError();
    *p = nullptr;
    if (gz_release_idle_blocks() == 0) { set_error("HIP error: out of memory (%llu bytes)", (unsigned long long)(count * sizeof(T))); return MK_ERR_DEVICE; }
    MK_HIP(hipMalloc((void **)p, count * sizeof(T)));
    return MK_OK;
}
template <typename T>
inline void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}
// Every entry point starts here: bind the device and, unless the caller is an append that wants to overlap with it, fold
// the build batches still in flight into the index.
int use_device(const mk_ctx *c, bool settle = true);
int settle_build(mk_ctx *c);
// for_append = false: the long-query sketches borrow the tables / slots only
int ensure_build_scratch(mk_ctx *c, uint64_t seq_bytes, int buf = 0, bool for_append = true);

// ---- sketch.hip
int launch_genome_sketch(mk_ctx *c, const char *d_seq, const uint64_t *d_off, const uint64_t *h_off,
                         const uint32_t *d_valid, uint32_t n, uint64_t *d_tables);
// d_abort (may be null): the binned sketch's overflow counter; the kernels do nothing if it ran over
int launch_finalize(mk_ctx *c, const uint64_t *d_tables, uint32_t n, uint32_t g0, const uint32_t *d_abort);
int launch_bloom_insert(mk_ctx *c, uint64_t *d_tables, const char *d_seq, const uint64_t *d_off,
                        const uint32_t *d_valid, uint32_t n, const uint32_t *d_abort);
// Page-locked host memory for the ingest's buffers (api.hip): ordinary memory in transparent huge pages, registered with the
// runtime -- a 16 MiB piece is 8 pages to pin instead of 4,096, and the page faults are the calling thread's own, where sixteen
// reader threads asking hipHostMalloc for their first pieces stood in one queue for 70 ms (profiles/r6_cli_startup.txt).
// hipHostMalloc when that does not work.  pinned_free takes either kind.
void *pinned_alloc(uint64_t bytes);
void pinned_free(void *p);
int launch_bloom_summary(mk_ctx *c, bool after_sweep = false);   // after_sweep: only the regions the sweep has just changed, when the rest is current
uint64_t bloom_regions(const mk_ctx *c);                          // flags per array of d_bloom_touched
int launch_build_tail(mk_ctx *c, uint32_t n, uint32_t g0);     // matrix rows, Bloom pass B, summary (after a fused reduce kernel)
// ---- build.hip: the index build from packed sequences (2-bit codes + exception bits)
// (these four run on c->front_stream, on side b's arrays; launch_unpack on c->stream)
int launch_pack(mk_ctx *c, int b, const char *d_seq, const uint64_t *h_off, uint32_t n, uint8_t *d_codes, uint8_t *d_except,
                const uint64_t *d_code_off, hipStream_t st = nullptr);       // st: the front stream unless given
int launch_seed_fix(mk_ctx *c, int b, const char *d_seq, const char *d_heads, uint32_t n, uint8_t *d_codes, uint8_t *d_except,
                    const uint64_t *d_code_off, hipStream_t st = nullptr);
int launch_synth_packed(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t len, uint8_t *d_codes, const uint64_t *d_code_off,
                        uint32_t strains = 0, uint32_t rate_ppm = 0);      // strains != 0: related genomes (mk_device.hpp: strain_word)
int launch_unpack(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off, const char *d_heads,
                  const uint64_t *h_off, uint32_t n, char *d_seq);
// front stage of a batch on c->front_stream: the scatter kernel into side `b` (*used = false: the shape does not suit
// the bins, nothing was launched); back stage on c->stream: reduce + fingerprints + sizes + Bloom pass A, matrix rows,
// Bloom pass B, summary
int launch_build_front(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off,
                       const uint64_t *h_off, uint32_t n, bool *used, hipStream_t st = nullptr, bool for_queries = false);
// K1 of a run of long queries through the build's packed kernels (side 0, c->stream; no build is in flight when a query
// runs): d_tables[q][p] = fingerprint << kPosBits | position of partition p's minimum, kEmptyKey where there is none; the
// queries' packed form stays in c->d_pk[0] (offsets c->d_pk_off[0], "has exceptions" flags in side 0's counters) for the
// kernels that gate the winners.  *used = false: the shape does not fit the bins, nothing was written.
int launch_query_tables(mk_ctx *c, const char *d_seq, const uint64_t *d_off, const uint64_t *h_off, uint32_t n, uint64_t *d_tables,
                        bool *used);
uint64_t packed_offsets(const uint64_t *h_off, uint32_t n, uint64_t *pk_off);
int ensure_packed(mk_ctx *c, int buf, uint64_t code_bytes);
int launch_build_back(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off, uint32_t n,
                      uint32_t g0);
int ensure_build_side(mk_ctx *c, int b);
void use_build_side(mk_ctx *c, int b);       // point the aliases (d_counters, ...) at side b
int launch_bloom_merge(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_later);
// api.hip: one pass of the hot path over a query set (mk_qset_run / mk_qset_run_compact; comm.hip hooks the chunks)
int qset_run(mk_ctx *c, mk_qset *qs, uint32_t nresults, uint32_t min_score, double min_inter, uint32_t cap, uint32_t *d_count,
             mk_hit *d_cand, uint64_t *d_rows, const std::function<int(uint32_t, uint32_t)> *after_chunk, uint32_t min_chunks);
int forget_bloom_summary(mk_ctx *c);         // the Bloom cells were replaced: the summaries may claim nothing until recomputed
// api.hip: scratch shared by the build and the long-query sketches
int ensure_build_counters(mk_ctx *c);
int ensure_bloom_summary(mk_ctx *c);
int ensure_bloom_summary_arrays(mk_ctx *c);           // allocated and zeroed ("nothing is full"), not computed
uint64_t bloom_summary_bytes(const mk_ctx *c);         // of the coarse level: one bit per 2048 cells
int launch_query_sketch_short(mk_ctx *c, mk_qset *qs);
int launch_query_sketch_long(mk_ctx *c, mk_qset *qs, uint32_t q);
int launch_query_sketch_long_batch(mk_ctx *c, mk_qset *qs, uint32_t q0, uint32_t n, bool *done);
// O(length) sketches of long reads / contigs (per-query hash tables in d_scratch), see sketch.hip
bool query_is_mid_length(const mk_ctx *c, uint64_t nk);
int launch_query_sketch_mid(mk_ctx *c, mk_qset *qs, const std::vector<uint32_t> &which, unsigned long long *d_scratch, uint64_t scratch_slots);
int launch_query_sketch_dense(mk_ctx *c, mk_qset *qs, uint32_t slot);   // slot = index into qs->dense_q
int launch_query_sketch_dense_batch(mk_ctx *c, mk_qset *qs, uint32_t slot, uint32_t n, bool *done);
int launch_scan_counts(mk_ctx *c, mk_qset *qs);                          // fills qs->d_scan_n
// range boundaries of every query's (sorted) entry list; *d_flag |= 1 when a
// (query, range) holds more than `limit` entries
int launch_query_split(mk_ctx *c, mk_qset *qs, uint32_t S, uint32_t limit, uint32_t *d_flag);
int launch_synth_genomes(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t len, char *d_out);
int launch_synth_queries(mk_ctx *c, uint64_t first_id, uint32_t nq, uint64_t G, uint64_t L, uint64_t qlen,
                         char *d_out);
int launch_convert_columns(mk_ctx *c, bool to_device, uint32_t p_begin, uint32_t p_end, uint8_t *d_staging);
// huff.hip: one lane per deflate block of literals (mk_index_import_columns_huffman)
int launch_huff_decode(mk_ctx *c, const uint8_t *d_payload, uint64_t payload_bytes, const mk_huff_block *d_blocks, uint32_t n_blocks,
                       const uint8_t *d_lens, uint32_t n_codes, uint8_t *d_out, uint64_t out_bytes, uint32_t *d_crc, uint32_t *d_bad);
int launch_export_genomes(mk_ctx *c, const uint32_t *d_ids, uint32_t n, uint8_t *d_dst);   // d_dst[P][n] (W bytes each, dump byte order)
inline MatRef mat_ref(const mk_ctx *c);

// ---- gunzip.hip: gzip streams inflated on the device
struct mk_gz_stream {              // a stream (a file) of a batch
    uint64_t in_off;               // its bytes at gz + in_off (16-byte aligned, >= 16 zero bytes behind them)
    uint64_t out_off;              // its text at text + out_off (16-byte aligned)
    uint32_t in_len, n_tok, out_len, status, members;      // status: mk_gz_status
    uint32_t cand_lo, cand_hi;     // its candidate block starts: sorted[cand_lo, cand_hi) (ascending bit offsets)
    uint32_t n_chain;              // segments on its chain: chain[cand_lo + its index ...] (a stream has one more segment than candidates)
    uint32_t rewrite;              // a segment on its chain had more tokens than its slots: those segments are decoded once more, with exact rooms
    uint32_t pad[3];
};
struct mk_gz_seg {                 // what a lane of gz_tokens_kernel found: a stretch of a stream from a block's first bit on
    uint64_t tok_ptr;              // where its tokens lie (a device address, 16-byte aligned)
    uint32_t tok_cap;              // room there (a multiple of four); n_tok > tok_cap: the tokens beyond were counted, not kept
    uint32_t n_tok, out_len, status, members;
    uint32_t link;                 // the segment it arrived at between two blocks (all ones: the stream's end)
};
struct mk_gz_job { uint64_t tok_ptr; uint32_t seg, tok_cap; };   // a segment to decode again, with its exact room
int launch_fasta_count(mk_ctx *c, const uint8_t *d_text, const mk_gz_stream *d_jobs, uint32_t n, const uint32_t *d_chunk_first, uint32_t n_chunks,
                       void *d_scratch, uint64_t *d_seq_len, hipStream_t st);
int launch_fasta_strip(mk_ctx *c, const uint8_t *d_text, const mk_gz_stream *d_jobs, const uint32_t *which, uint32_t m, const uint32_t *h_chunk_first,
                       uint32_t n_chunks, const void *d_scratch, uint8_t *d_dst, const uint64_t *dst_off, hipStream_t st);
uint64_t fasta_scratch_bytes(uint32_t n_chunks);
void gz_release_staging(mk_ctx *c);     // the inflater's page-locked pieces, upload streams and symbol lists (mk_destroy, mk_gz_trim)
}  // namespace mk
struct mk_gz_batch;
namespace mk {
int gz_batch_strip(mk_ctx *c, const mk_gz_batch *b, const uint32_t *which, uint32_t m, uint8_t *d_dst, const uint64_t *dst_off, hipStream_t st);
int gz_batch_used(const mk_gz_batch *b, hipStream_t st);       // a kernel that reads the batch's text has been queued on st
uint32_t gz_batch_size(const mk_gz_batch *b);
bool gz_batch_ok(const mk_gz_batch *b, uint32_t i);
uint64_t gz_batch_len(const mk_gz_batch *b, uint32_t i);
const mk_ctx *gz_batch_owner(const mk_gz_batch *b);

// ---- cold.hip: the cold rows packed (delta vs the genome before + bit packing)
int pack_cold(mk_ctx *c, uint64_t *raw_bytes, uint64_t *packed_bytes);
int need_raw_cold(mk_ctx *c);                 // unpack if packed: for everything that needs the rows as they are
int stage_cold_rows(mk_ctx *c, uint64_t r_lo, uint64_t r_hi, uint8_t *d_dst, int b, hipStream_t st);
int ensure_zstage(mk_ctx *c, uint64_t rows);
inline bool has_cold(const mk_ctx *c) { return c->h_M != nullptr || c->h_Z != nullptr; }

// ---- scan.hip
struct ScanArgs {
    const uint8_t *M;
    const uint8_t *Mc;             // cold rows (MatRef::cold_m) or null
    uint32_t P_hot;
    uint64_t ld;
    uint32_t G, ntiles, nq, q_begin;
    const uint64_t *entries;
    const uint64_t *ent_off;
    const uint32_t *nent;
    uint32_t *scores;              // see scan_kernel.hpp: two-stride addressing
    uint64_t score_tile_stride, score_q_stride;
    uint32_t score_vec;            // 16-byte stores allowed (strides and padding permit it)
    // window launches (a matrix with cold rows, scan_windows in api.hip): only entries of partitions
    // [row_lo, row_hi) count; accumulate: add to the scores instead of storing them
    uint32_t windowed, row_lo, row_hi, accumulate;
};
int launch_scan(mk_ctx *c, const ScanArgs &a);
// slab schedule (scan_kernel.hpp: scan_slab_kernel)
struct SlabArgs {
    const uint8_t *M;
    const uint8_t *Mc;             // cold rows (MatRef::cold_m) or null
    uint32_t P_hot;
    uint32_t r_begin, r_count;     // ranges [r_begin, r_begin + r_count) of the S this launch walks
    uint64_t ld;
    uint32_t G, ntiles, nq, q_begin, S;
    const uint64_t *entries;
    const uint64_t *ent_off;
    const uint32_t *split;         // [query][S + 1] entry indices of the range boundaries
    uint8_t *partials;             // [tile][range][query][1 KiB]
    uint32_t chunk;                // != 0: range r of a query = its entries [r * chunk, (r + 1) * chunk) instead of the table
    const uint32_t *nent;          // entries per query (chunk mode)
};
int launch_scan_slab(mk_ctx *c, const SlabArgs &a);
// the two layouts the pipeline uses
struct ScoreLayout { uint64_t tile_stride, q_stride; uint32_t vec; };
inline ScoreLayout score_layout_rows(uint32_t W, uint64_t pitch, uint32_t G)      // [query][pitch]
{
    const uint64_t tg = kTileBytes / W;
    return {tg, pitch, (uint32_t)((pitch % 4 == 0) && ((uint64_t)((G + tg - 1) / tg) * tg <= pitch))};
}
inline ScoreLayout score_layout_tiles(uint32_t W, uint32_t nq)                     // [tile][query][genomes per tile]
{
    const uint64_t tg = kTileBytes / W;
    return {(uint64_t)nq * tg, tg, 1u};
}

// dense long queries (scan_kernel.hpp: scan_dense_kernel): adds into the score rows
struct DenseLut;
struct DenseArgs {
    const uint8_t *M;
    const uint8_t *Mc;             // cold rows (MatRef::cold_m) or null
    uint32_t P_hot;
    uint64_t ld;
    uint32_t G, ntiles, P, rows_per_item, nchunks, ngroups;
    uint32_t row_lo, row_hi;       // rows this launch walks (nchunks covers them in pieces of rows_per_item)
    const uint8_t *dense;          // [group][P][4] fingerprints (W bytes each), empty = inactive
    const uint32_t *dense_q;       // [group][4] set index of each slot or 0xffffffff
    const DenseLut *lut;    // one-byte fingerprints: [octet = two groups][P] field tables (scan_kernel.hpp), or null
    uint32_t noctets;
    uint32_t q0, q1;               // set range the score buffer covers
    uint32_t *scores;
    uint64_t score_tile_stride, score_q_stride;
    uint32_t empty;
};
int launch_scan_dense(mk_ctx *c, const DenseArgs &a);
int launch_dense_lut(mk_ctx *c, const uint8_t *d_dense, uint32_t ngroups, DenseLut *d_lut);
// how the table kernel shares rows between its sets of queries (scan.hip, scan_kernel.hpp): sets of sixteen (eight) queries,
// sharing groups of 1, 2 or 4 sets -- and the rows a chunk may hold for the counters' bit planes
inline uint32_t dense_share(uint32_t noctets) { const uint32_t no = noctets >= 2 ? 2 : 1, nsets = (noctets + no - 1) / no; return nsets >= 3 ? 4u : nsets == 2 ? 2u : 1u; }
inline uint32_t dense_chunk_rows(uint32_t noctets) { (void)noctets; return 16368u; }
int probe_stream_read(mk_ctx *c, uint32_t rounds, double *gbps, uint64_t *bytes);

// ---- select.hip
struct SelectArgs {
    const uint32_t *scores;        // tile-major [tile][nq][tile_genomes] (plain schedule) or null
    const uint8_t *partials;       // [tile][range][nq][1 KiB] mismatch counts (slab schedule) or null
    const uint32_t *nent;          // active partitions per query, offset to this chunk (slab schedule)
    uint32_t S, W;
    uint32_t tile_genomes, G, nq;
    uint32_t nresults, min_score;
    double min_inter;
    const uint32_t *sketch_size;
    const uint64_t *genome_size;
    uint32_t genome_id_base, cap;
    const float *ratio;            // genome_size / sketch_size per genome (slab schedule: the screen's one load), or null
    uint32_t *count;               // [nq]
    mk_hit *cand;                  // [nq][cap]
    uint64_t *rows;                // compact form instead of count/cand: [nq][1 + cap], see mk_qset_run_compact
};
constexpr uint32_t kSelectMaxResults = 64;   // top-N sizes the device selection supports
int launch_select(mk_ctx *c, const SelectArgs &a);
int launch_ratio(mk_ctx *c, float *d_ratio, uint32_t padded);

// ---- merge.hip (K6b): filter_results' heap over the entrant rows of `world` shards
constexpr uint32_t kMergeOverflow = 0xffffffffu;   // nhits value of a query some shard's row overflowed for
struct MergeArgs {
    const uint32_t *count;         // [world][nq]
    const mk_hit *cand;            // [world][nq][cap]
    uint32_t world, nq, cap, nresults;
    mk_hit *hits;                  // [nq][nresults]
    uint32_t *nhits;               // [nq]
    // compact exchange form instead of count / cand (null otherwise): [world][nq][1 + cap] words
    const uint64_t *rows;
    const uint32_t *ss;            // sketch_size of every genome a row may name, indexed by id - id_base
    const uint64_t *gs;            // genome_size likewise
    uint32_t id_base;
};
int launch_merge(mk_ctx *c, const MergeArgs &a);

// ---- exact.hip
int exact_load_genome(mk_ctx *c, const char *const *contigs, const uint64_t *contig_lens, uint32_t n_contigs);
int exact_queries(mk_ctx *c, const char *const *queries, const uint64_t *query_lens, uint32_t nq, uint64_t *inter,
                  uint64_t *uni);

}  // namespace mk
