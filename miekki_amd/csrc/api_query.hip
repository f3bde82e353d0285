// C ABI of libmiekki_hip.so, the query side: query sets (sketch, Bloom gate, range tables), the scan schedules (slab, plain,
// dense, windows over rows in host memory), selection, and mk_query / mk_query_scores / mk_qset_* / mk_exact* above them.
// Host-side orchestration only: the kernels are in sketch.hip, scan.hip, select.hip, merge.hip, exact.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>

#include "mk_internal.hpp"

namespace mk {

// A genome without a single stored fingerprint (its sequence is exactly k long, Miekki.cpp:162,
// 569) has sketch_size 0: with min_score 0 its score 0 passes and jaccard = 0 / 0 is NaN
// (Miekki.cpp:381-383).  What the reference's heap does with NaNs is whatever its comparison
// sequence happens to yield; the device selection assumes ordered values, so such calls take
// the host replay (dense score rows + the same std:: heap calls), which reproduces it.
static bool nan_candidates_possible(const mk_ctx *c, uint32_t min_score)
{
    return min_score == 0 && c->has_empty_sketch;
}

// ---- query sets ------------------------------------------------------------------
// rows of a part of a mixed set to their places in the whole set's output: row i -> dst[idx[i]] (row_bytes a multiple of 8),
// and, when there are separate counts, cnt[i] -> dst_cnt[idx[i]]
__global__ __launch_bounds__(256) void place_rows_kernel(const uint8_t *__restrict__ rows, uint64_t row_bytes, const uint32_t *__restrict__ idx, uint32_t n,
                                                         uint8_t *__restrict__ dst, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ dst_cnt)
{
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t q = idx[i];
    const uint64_t *__restrict__ s = reinterpret_cast<const uint64_t *>(rows + (uint64_t)i * row_bytes);
    uint64_t *__restrict__ d = reinterpret_cast<uint64_t *>(dst + (uint64_t)q * row_bytes);
    // (a count form's row is only as full as its count says -- 24 bytes per hit -- but copying the slots is cheaper than asking)
    for (uint64_t w = threadIdx.x; w < row_bytes / 8; w += blockDim.x) d[w] = s[w];
    if (cnt && threadIdx.x == 0) dst_cnt[q] = cnt[i];
}

static int launch_place_rows(mk_ctx *c, const uint8_t *rows, uint64_t row_bytes, const uint32_t *d_idx, uint32_t n, uint8_t *dst, const uint32_t *cnt,
                             uint32_t *dst_cnt)
{
    if (!n) return MK_OK;
    hipLaunchKernelGGL(place_rows_kernel, dim3(n), dim3(256), 0, c->stream, rows, row_bytes, d_idx, n, dst, cnt, dst_cnt);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

static void qset_release(mk_qset *qs)
{
    if (!qs) return;
    for (int i = 0; i < 2; ++i) { qset_release(qs->part[i]); dev_free(qs->d_part_q[i]); }
    dev_free(qs->d_part_out);
    if (!qs->split_in_arena) dev_free(qs->d_split);
    if (qs->arena_borrowed) qs->owner->qarena_busy = false;      // the context keeps its arena for the next call
    else dev_free(qs->d_arena);                  // every other device array of the set lives in it
    delete qs;
}

static int qset_alloc(mk_ctx *c, const uint64_t *lens, uint32_t nq, mk_qset **out, bool transient = false)
{
    std::unique_ptr<mk_qset, void (*)(mk_qset *)> qs(new mk_qset(), qset_release);
    qs->owner = c; qs->arena_borrowed = false; qs->head_bytes = 0; qs->o_off = qs->o_ent_off = 0;
    qs->nq = nq; qs->d_seq = nullptr; qs->d_off = nullptr; qs->d_ent_off = nullptr; qs->d_entries = nullptr;
    qs->d_nent = nullptr; qs->sketched = false; qs->gen = 0; qs->short_max_nk = 0;
    qs->d_split = nullptr; qs->S = 0; qs->chunk = 0; qs->slab_ok = false;
    qs->d_dense = nullptr; qs->d_dense_q = nullptr; qs->d_scan_n = nullptr;
    qs->d_arena = nullptr; qs->split_in_arena = false; qs->split_room = 0;
    qs->h_off.assign(nq + 1, 0); qs->h_ent_off.assign(nq + 1, 0);
    // Long queries that activate a large share of the partitions (whole genomes, -A) keep a dense fingerprint vector
    // instead of an entry list and are scored by passes over ALL rows, sixteen queries per pass (scan_dense_lut_kernel).  From
    // which share on that is cheaper depends on how many there are to share a pass: a pass costs what 16 x 0.115 P entries cost
    // the sparse scan (27.7 ms per 105 GB against 7.1 TB/s of entries), i.e. a query with more than P / 8 k-mers (0.118 P
    // active partitions) is better off dense when a pass is full, and one with P / 4 (0.22 P) even when it has a pass nearly
    // to itself.  (The vectors and tables of the dense queries stay below 8 GiB.)
    uint64_t dense_div = 4;
    {
        uint64_t n8 = 0;
        for (uint32_t q = 0; q < nq; ++q) {
            const uint64_t nk = lens[q] > c->p.k ? lens[q] - c->p.k : 0;
            n8 += beyond_short_len(c->p.k, lens[q]) && nk >= c->P / 8 ? 1 : 0;
        }
        if (n8 >= 16 && n8 * c->P * c->W * 3 <= (8ull << 30)) dense_div = 8;       // (vector: P W bytes per query; tables: 2 P W per query)
    }
    for (uint32_t q = 0; q < nq; ++q) {
        const uint64_t nk = lens[q] > c->p.k ? lens[q] - c->p.k : 0;
        if (lens[q] >= (1ull << 40)) { set_error("query too long"); return MK_ERR_ARG; }
        qs->h_off[q + 1] = qs->h_off[q] + lens[q];
        const bool dense = beyond_short_len(c->p.k, lens[q]) && nk >= c->P / dense_div;
        qs->h_ent_off[q + 1] = qs->h_ent_off[q] + (dense ? 0 : std::min<uint64_t>(nk, c->P));
        if (dense) qs->dense_q.push_back(q);
        else if (beyond_short_len(c->p.k, lens[q])) qs->long_q.push_back(q);
        else qs->short_max_nk = std::max<uint32_t>(qs->short_max_nk, (uint32_t)nk);
    }
    qs->total_len = qs->h_off[nq];
    // One device allocation per set (a small call is dominated by allocator round trips, not by
    // kernels): the arrays are carved out of it at 256-byte boundaries.
    uint64_t dense_bytes = 0;
    if (!qs->dense_q.empty()) {
        while (qs->dense_q.size() % 4) qs->dense_q.push_back(0xffffffffu);          // pad the last group
        dense_bytes = (uint64_t)(qs->dense_q.size() / 4) * c->P * 4 * c->W;
    }
    constexpr uint32_t kSplitS = 32;                                // room for the slab schedule's range table up to S = 32
    qs->split_room = (uint64_t)nq * (kSplitS + 1) * 4 <= (64ull << 20) ? kSplitS : 0;
    uint64_t at = 0;
    auto carve = [&at](uint64_t bytes) { const uint64_t o = at; at += (bytes + 255) / 256 * 256; return o; };
    const uint64_t o_seq = carve(qs->total_len + 64), o_off = carve(((uint64_t)nq + 1) * 8),
                   o_ent_off = carve(((uint64_t)nq + 1) * 8), o_entries = carve((qs->h_ent_off[nq] + 1) * 8),
                   o_nent = carve(((uint64_t)nq + 1) * 4), o_scan_n = carve(((uint64_t)nq + 1) * 4),
                   o_dense = carve(dense_bytes), o_dense_q = carve(qs->dense_q.size() * 4),
                   o_lut = carve((uint64_t)((qs->dense_q.size() / 4 + 1) / 2) * c->P * 32),     // (32 bytes of tables per octet of queries and row, either width)
                   o_split = carve(qs->split_room ? (uint64_t)nq * (qs->split_room + 1) * 4 : 0);
    if (transient && !c->qarena_busy) {
        if (at > c->qarena_cap) {
            MK_HIP(hipStreamSynchronize(c->stream));
            dev_free(c->d_qarena);
            c->qarena_cap = 0;
            const uint64_t cap = std::max<uint64_t>(at + at / 2, 4ull << 20);
            MK_TRY(dev_alloc(&c->d_qarena, cap));
            c->qarena_cap = cap;
        }
        qs->d_arena = c->d_qarena; qs->arena_borrowed = true; c->qarena_busy = true;
    } else {
        MK_TRY(dev_alloc(&qs->d_arena, at));
    }
    qs->o_off = o_off; qs->o_ent_off = o_ent_off;
    qs->head_bytes = o_ent_off + ((uint64_t)nq + 1) * 8;          // o_seq == 0: sequences, offsets, entry offsets in a row
    qs->d_seq = reinterpret_cast<char *>(qs->d_arena + o_seq);
    qs->d_off = reinterpret_cast<uint64_t *>(qs->d_arena + o_off);
    qs->d_ent_off = reinterpret_cast<uint64_t *>(qs->d_arena + o_ent_off);
    qs->d_entries = reinterpret_cast<uint64_t *>(qs->d_arena + o_entries);
    qs->d_nent = reinterpret_cast<uint32_t *>(qs->d_arena + o_nent);
    qs->d_scan_n = reinterpret_cast<uint32_t *>(qs->d_arena + o_scan_n);
    if (qs->split_room) { qs->d_split = reinterpret_cast<uint32_t *>(qs->d_arena + o_split); qs->split_in_arena = true; }
    if (!qs->dense_q.empty()) {
        qs->d_dense = qs->d_arena + o_dense;
        qs->d_dense_q = reinterpret_cast<uint32_t *>(qs->d_arena + o_dense_q);
        qs->d_dense_lut = reinterpret_cast<DenseLut *>(qs->d_arena + o_lut);
        MK_HIP(hipMemsetAsync(qs->d_dense, 0xFF, dense_bytes, c->stream));           // every slot starts empty
        MK_HIP(hipMemcpyAsync(qs->d_dense_q, qs->dense_q.data(), qs->dense_q.size() * 4, hipMemcpyHostToDevice,
                              c->stream));
    }
    *out = qs.release();                                         // offsets travel with the sequences (qset_upload)
    return MK_OK;
}

static int qset_copy_offsets(mk_ctx *c, mk_qset *qs)
{
    MK_HIP(hipMemcpyAsync(qs->d_off, qs->h_off.data(), (size_t)(qs->nq + 1) * 8, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipMemcpyAsync(qs->d_ent_off, qs->h_ent_off.data(), (size_t)(qs->nq + 1) * 8, hipMemcpyHostToDevice,
                          c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));                     // the host vectors are pageable
    return MK_OK;
}

static int ensure_pinned(uint8_t *&p, uint64_t &cap, uint64_t need)
{
    if (need <= cap) return MK_OK;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const uint64_t want = std::max<uint64_t>(need + need / 2, 1ull << 20);
    MK_HIP(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
    cap = want;
    return MK_OK;
}

static int qset_prepare_slab(mk_ctx *c, mk_qset *qs);

static int qset_sketch_only(mk_ctx *c, mk_qset *qs)
{
    MK_TRY(ensure_bloom_summary(c));
    ScopedTimer t(c, 0);
    MK_TRY(launch_query_sketch_short(c, qs));
    if (!qs->long_q.empty()) {
        if (!c->d_long_table) MK_TRY(dev_alloc(&c->d_long_table, (uint64_t)c->P));
        if (!c->d_seed_valid) MK_TRY(dev_alloc(&c->d_seed_valid, kBuildBatch));
        // neighbours in the set share one run of the build's packed kernels and one gate-and-append launch; shapes those
        // kernels do not take (h > 22) go one by one through the atomic kernel
        MK_TRY(ensure_build_scratch(c, 0, 0, false));
        // long reads and contigs (up to 2^18 k-mers): per-query hash tables, O(length) -- no 2^h table is touched
        std::vector<uint32_t> mid, rest;
        for (uint32_t q : qs->long_q) {
            const uint64_t len = qs->h_off[q + 1] - qs->h_off[q];
            (query_is_mid_length(c, len - c->p.k) ? mid : rest).push_back(q);
        }
        // (MIEKKI_MID_SLOTS: fewer slots per round than the scratch holds -- the tests make small sets take several rounds)
        static const uint64_t slot_cap = [] { const char *e = getenv("MIEKKI_MID_SLOTS"); return e ? (uint64_t)std::max(1L, atol(e)) : ~0ull; }();
        MK_TRY(launch_query_sketch_mid(c, qs, mid, reinterpret_cast<unsigned long long *>(c->d_tables),
                                       std::min<uint64_t>((uint64_t)c->build_batch * c->P, slot_cap)));
        const std::vector<uint32_t> &long_q = rest;
        for (size_t i = 0; i < long_q.size();) {
            uint32_t n = 1;
            while (i + n < long_q.size() && n < c->build_batch && long_q[i + n] == long_q[i] + n) ++n;
            bool done = false;
            MK_TRY(launch_query_sketch_long_batch(c, qs, long_q[i], n, &done));     // (a loner too: a run of one)
            if (!done)
                for (uint32_t j = 0; j < n; ++j) MK_TRY(launch_query_sketch_long(c, qs, long_q[i + j]));
            i += n;
        }
    }
    if (!qs->dense_q.empty()) {
        if (!c->d_long_table) MK_TRY(dev_alloc(&c->d_long_table, (uint64_t)c->P));
        if (!c->d_seed_valid) MK_TRY(dev_alloc(&c->d_seed_valid, kBuildBatch));
        // the same for whole-genome (dense) queries, up to a build batch at a time
        MK_TRY(ensure_build_scratch(c, 0, 0, false));
        for (uint32_t slot = 0; slot < qs->dense_q.size();) {
            if (qs->dense_q[slot] == 0xffffffffu) { ++slot; continue; }
            uint32_t n = 1;
            while (slot + n < qs->dense_q.size() && n < c->build_batch && qs->dense_q[slot + n] == qs->dense_q[slot] + n) ++n;
            bool done = false;
            MK_TRY(launch_query_sketch_dense_batch(c, qs, slot, n, &done));
            if (!done)
                for (uint32_t j = 0; j < n; ++j) MK_TRY(launch_query_sketch_dense(c, qs, slot + j));
            slot += n;
        }
        // the field tables the dense scan looks bytes up in (scan_dense_lut_kernel)
        if (qs->d_dense_lut) MK_TRY(launch_dense_lut(c, qs->d_dense, (uint32_t)(qs->dense_q.size() / 4), qs->d_dense_lut));
    }
    MK_TRY(launch_scan_counts(c, qs));
    qs->sketched = true;
    return MK_OK;
}

// Sketch, Bloom gate and slab range table of a set are functions of the set and of the index
// (Bloom cells, slab shape): they are kept until either changes (the index generation stamp)
// or the caller asks for a fresh pass (mk_qset_invalidate).
static int qset_sketch(mk_ctx *c, mk_qset *qs)
{
    if (qs->sketched && qs->gen == c->gen) return MK_OK;
    qs->sketched = false;
    MK_TRY(qset_sketch_only(c, qs));
    MK_TRY(qset_prepare_slab(c, qs));
    qs->gen = c->gen;
    return MK_OK;
}

static uint32_t ntiles_of(const mk_ctx *c)
{
    return (uint32_t)(((uint64_t)c->G * c->W + kTileBytes - 1) / kTileBytes);
}

// Ranges of the slab schedule: the (2^h / S) x 1 KiB column slab the waves in flight
// share should fit the 256 MiB Infinity Cache with room to spare (target 128 MiB).
static uint32_t slab_ranges(const mk_ctx *c)
{
    uint64_t target = 128ull << 20;
    if (const char *e = getenv("MIEKKI_SLAB_MIB")) {             // tuning knob
        const long v = atol(e);
        if (v >= 1 && v <= 4096) target = (uint64_t)v << 20;
    }
    const uint64_t slab = (uint64_t)c->P * kTileBytes;
    uint32_t S = 1;
    while (S < 32 && slab / S > target) S <<= 1;
    return S;
}

// Prepare the slab schedule for a sketched set: range boundaries per query, and the
// check that every (query, range) fits the packed 8/16-bit counters.  Sets with long
// (unsorted) queries, or that fail the check, use the plain schedule.
static int qset_prepare_slab(mk_ctx *c, mk_qset *qs)
{
    uint32_t S = slab_ranges(c);
    qs->slab_ok = false;
    qs->chunk = 0;
    if (!qs->long_q.empty() || !qs->dense_q.empty() || !qs->nq) { qs->S = S; return MK_OK; }
    const uint32_t limit = c->W == 1 ? 255u : 65535u;
    // A handful of queries has no reuse to schedule -- what it needs is parallelism: one wave per
    // (query, tile) would walk ~900 entries in ~110 dependent steps (a single query: 13 waves on
    // 256 CUs).  So small sets cut every entry list into S pieces BY COUNT: S x as many waves, each
    // a few steps long, no range table, and no eligibility check (a piece holds at most `chunk`
    // <= 255 entries by construction), i.e. no host round trip either.
    uint32_t small_below = 512;
    if (const char *e = getenv("MIEKKI_SLAB_MIN_QUERIES")) small_below = (uint32_t)std::max(0L, atol(e));   // tests force the range-table path
    // (with cold rows the ranges are cut by partition whatever the set's size: whole cold ranges are then staged
    // through HBM once per chunk, where pieces cut by count would have every wave read its rows over PCIe)
    if (qs->nq < small_below && !has_cold(c)) {
        const uint64_t waves = (uint64_t)ntiles_of(c) * qs->nq;
        // (up to eight pieces: that is what select_kernel sums with its words prefetched; more only
        // when the packed counters ask for it)
        uint32_t Sc = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, (4096 + waves - 1) / std::max<uint64_t>(waves, 1)));
        Sc = std::max<uint32_t>(Sc, (qs->short_max_nk + limit - 1) / limit);
        Sc = std::max<uint32_t>(Sc, 1);
        qs->S = Sc;
        qs->chunk = std::max<uint32_t>(1, (qs->short_max_nk + Sc - 1) / Sc);
        qs->slab_ok = true;
        return MK_OK;
    }
    if (S < 2) { qs->S = S; return MK_OK; }
    // longer queries need more (smaller) ranges to keep every (query, range) within the
    // packed counters: aim at <= 180 entries per range on average, the device check
    // below still decides
    while (S < 64 && (uint64_t)qs->short_max_nk > (uint64_t)S * (limit * 7 / 10)) S <<= 1;
    if (qs->S != S || !qs->d_split) {
        if (!(qs->split_in_arena && S <= qs->split_room)) {        // more ranges than the set reserved room for
            if (!qs->split_in_arena) dev_free(qs->d_split);
            qs->split_in_arena = false; qs->d_split = nullptr;
            MK_TRY(dev_alloc(&qs->d_split, (uint64_t)qs->nq * (S + 1)));
        }
        qs->S = S;
    }
    if (!c->d_flag) MK_TRY(dev_alloc(&c->d_flag, 1));
    MK_HIP(hipMemsetAsync(c->d_flag, 0, 4, c->stream));
    MK_TRY(launch_query_split(c, qs, S, limit, c->d_flag));
    uint32_t flag = 1;
    MK_HIP(hipMemcpyAsync(&flag, c->d_flag, 4, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    qs->slab_ok = flag == 0;
    return MK_OK;
}

static uint32_t tile_genomes(const mk_ctx *c) { return kTileBytes / c->W; }
// entries of one query's scores in the tile-major matrix (whole tiles)
static uint64_t score_row_entries(const mk_ctx *c) { return (uint64_t)ntiles_of(c) * tile_genomes(c); }

// queries per chunk so that the chunk's score matrix stays within ~4 GiB
// Bytes a query chunk's score / partial buffer may take: `want`, but never more than what is
// already allocated or a third of the free device memory (a nearly full GPU scans in smaller
// chunks instead of failing).
static uint64_t chunk_budget(uint64_t want, uint64_t have)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return want;
    // (what an ingest of gzip'd genomes left with the inflater counts as free: given back when the chunk would shrink for it)
    if (free_b / 3 < want && want > have && gz_release_idle_blocks() && hipMemGetInfo(&free_b, &total_b) != hipSuccess) return want;
    return std::max<uint64_t>(std::min<uint64_t>(want, std::max<uint64_t>(have, free_b / 3)), 64ull << 20);
}

static uint32_t chunk_queries(const mk_ctx *c, uint32_t nq)
{
    const uint64_t budget = chunk_budget(4ull << 30, c->scores_cap * 4) / 4;
    const uint64_t per = std::max<uint64_t>(1, budget / std::max<uint64_t>(score_row_entries(c), 1));
    return (uint32_t)std::min<uint64_t>(per, std::max<uint32_t>(nq, 1));
}

static int ensure_scores(mk_ctx *c, uint64_t rows)
{
    const uint64_t need = rows * score_row_entries(c);
    if (need > c->scores_cap) {
        dev_free(c->d_scores);
        c->scores_cap = 0;
        MK_TRY(dev_alloc(&c->d_scores, need));
        c->scores_cap = need;
    }
    return MK_OK;
}

// two staging buffers in HBM for cold rows (the copy of one piece runs beside the scan of the previous one), each
// `unit` rows or a multiple of it: as many as fit a sixteenth of the hot part, at least `unit`, at most the cold rows
static int ensure_cold_stage(mk_ctx *c, uint64_t unit)
{
    unit = std::max<uint64_t>(unit, 1);
    if (c->d_cold_stage && c->cold_stage_rows >= unit && c->cold_stage_rows % unit == 0) return MK_OK;
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_HIP(hipStreamSynchronize(c->copy_stream));
    dev_free(c->d_cold_stage);
    c->cold_stage_rows = 0;
    uint64_t rows = std::max<uint64_t>(unit, (uint64_t)c->P_hot / 16 / unit * unit);
    rows = std::min<uint64_t>(rows, ((uint64_t)c->P - c->P_hot + unit - 1) / unit * unit + unit);
    MK_TRY(dev_alloc(&c->d_cold_stage, 2 * rows * c->ld));
    c->cold_stage_rows = rows;
    for (int i = 0; i < 5; ++i)
        if (!c->ev_cold[i]) MK_HIP(hipEventCreateWithFlags(&c->ev_cold[i], hipEventDisableTiming));
    return MK_OK;
}

// Row windows of a matrix with cold rows, for the kernels that walk whole entry lists (plain schedule) or whole
// row ranges (dense queries): first the rows in HBM, where they lie; then the cold rows, a staging buffer's
// worth at a time -- copied from host memory on the copy stream beside the launch over the previous window, and
// presented to the kernel as "the matrix" by a shifted base.  launch(M, Mc, P_hot, row_lo, row_hi, first).
template <typename Launch>
static int scan_windows(mk_ctx *c, Launch launch)
{
    if (!has_cold(c)) return launch(c->d_M, (const uint8_t *)nullptr, c->P, 0u, c->P, true);
    MK_TRY(ensure_cold_stage(c, std::max<uint64_t>(1, c->P / 64)));
    MK_TRY(ensure_zstage(c, c->cold_stage_rows));
    hipEvent_t ev_enter = c->ev_cold[4];
    hipEvent_t *ev_copy = c->ev_cold, *ev_scan = c->ev_cold + 2;
    MK_HIP(hipEventRecord(ev_enter, c->stream));
    MK_HIP(hipStreamWaitEvent(c->copy_stream, ev_enter, 0));
    bool first = true;
    if (c->P_hot) { MK_TRY(launch(c->d_M, (const uint8_t *)nullptr, c->P, 0u, c->P_hot, true)); first = false; }
    uint32_t i = 0;
    for (uint64_t r = c->P_hot; r < c->P; r += c->cold_stage_rows, ++i) {
        const uint64_t nr = std::min<uint64_t>(c->cold_stage_rows, c->P - r);
        const int b = (int)(i & 1u);
        uint8_t *stage = c->d_cold_stage + (uint64_t)b * c->cold_stage_rows * c->ld;
        if (i >= 2) MK_HIP(hipStreamWaitEvent(c->copy_stream, ev_scan[b], 0));     // the launch that read this buffer last
        MK_TRY(stage_cold_rows(c, r, r + nr, stage, b, c->copy_stream));      // (packed rows: their packed bytes cross PCIe, cold.hip)
        MK_HIP(hipEventRecord(ev_copy[b], c->copy_stream));
        MK_HIP(hipStreamWaitEvent(c->stream, ev_copy[b], 0));
        MK_TRY(launch(stage - r * c->ld, (const uint8_t *)nullptr, c->P, (uint32_t)r, (uint32_t)(r + nr), first));
        first = false;
        MK_HIP(hipEventRecord(ev_scan[b], c->stream));
    }
    return MK_OK;
}

// scan queries [q0, q1) of the set into d_scores laid out as `lay` describes (the
// tile-major layout is per call: its tile stride is (q1 - q0) * genomes per tile)
static int qset_scan(mk_ctx *c, mk_qset *qs, uint32_t q0, uint32_t q1, uint32_t *d_scores, const ScoreLayout &lay)
{
    if (q1 <= q0 || c->G == 0) return MK_OK;
    const uint32_t nt = ntiles_of(c);
    const uint32_t per_launch = std::max<uint32_t>(1, 0x7ffffff0u / nt);
    const bool windowed = has_cold(c);                             // cold rows: one launch per window of rows
    // One pass over the row windows for both kernels (a cold window is copied to HBM once): the sparse kernel
    // first -- in the first window it also writes the zero rows of the dense queries (scan_n = 0) -- then the dense
    // kernel, which adds the whole-genome queries' scores, up to eight queries per pass over the rows.
    return scan_windows(c, [&](const uint8_t *M, const uint8_t *Mc, uint32_t P_hot, uint32_t row_lo, uint32_t row_hi, bool first) {
        for (uint32_t q = q0; q < q1; q += per_launch) {
            const uint32_t n = std::min(per_launch, q1 - q);
            ScanArgs a;
            a.M = M; a.Mc = Mc; a.P_hot = P_hot; a.ld = c->ld; a.G = c->G; a.ntiles = nt; a.nq = n; a.q_begin = q;
            a.entries = qs->d_entries; a.ent_off = qs->d_ent_off; a.nent = qs->d_scan_n;
            a.scores = d_scores + (uint64_t)(q - q0) * lay.q_stride;
            a.score_tile_stride = lay.tile_stride; a.score_q_stride = lay.q_stride; a.score_vec = lay.vec;
            a.windowed = windowed ? 1u : 0u; a.row_lo = row_lo; a.row_hi = row_hi; a.accumulate = first ? 0u : 1u;
            ScopedTimer t(c, 1);
            MK_TRY(launch_scan(c, a));
        }
        if (qs->dense_q.empty()) return (int)MK_OK;
        DenseArgs d;
        d.M = M; d.Mc = Mc; d.P_hot = P_hot; d.ld = c->ld; d.G = c->G; d.ntiles = nt; d.P = c->P;
        d.row_lo = row_lo; d.row_hi = row_hi;
        d.ngroups = (uint32_t)(qs->dense_q.size() / 4);
        d.rows_per_item = std::min<uint32_t>(row_hi - row_lo, dense_chunk_rows((d.ngroups + 1) / 2));   // (the table kernel counts in 14 or 13 bit planes)
        d.nchunks = (row_hi - row_lo + d.rows_per_item - 1) / d.rows_per_item;
        d.dense = qs->d_dense; d.dense_q = qs->d_dense_q; d.q0 = q0; d.q1 = q1; d.scores = d_scores;
        d.lut = qs->d_dense_lut; d.noctets = (d.ngroups + 1) / 2;
        d.score_tile_stride = lay.tile_stride; d.score_q_stride = lay.q_stride; d.empty = c->empty;
        ScopedTimer t(c, 1);
        return launch_scan_dense(c, d);
    });
}

// ---- slab schedule: per-range partial counts instead of a u32 score matrix
static uint64_t partial_bytes_per_query(const mk_ctx *c, uint32_t S) { return (uint64_t)ntiles_of(c) * S * kTileBytes; }

static uint32_t chunk_queries_slab(const mk_ctx *c, uint32_t nq, uint32_t S)
{
    const uint64_t budget = chunk_budget(16ull << 30, c->partials_cap);
    uint64_t per = std::max<uint64_t>(1, budget / std::max<uint64_t>(partial_bytes_per_query(c, S), 1));
    per = std::min<uint64_t>(per, 0x7ffffff0ull / std::max<uint64_t>((uint64_t)ntiles_of(c) * S, 1));   // one launch
    return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(per, 1), std::max<uint32_t>(nq, 1));
}

static int ensure_partials(mk_ctx *c, uint64_t bytes)
{
    if (bytes > c->partials_cap) {
        dev_free(c->d_partials);
        c->partials_cap = 0;
        MK_TRY(dev_alloc(&c->d_partials, bytes));
        c->partials_cap = bytes;
    }
    return MK_OK;
}

static int qset_scan_slab(mk_ctx *c, mk_qset *qs, uint32_t q0, uint32_t q1)
{
    const uint32_t rows_per_range = qs->S ? c->P / qs->S : c->P;
    // (ranges cut by count, or no ranges at all: cold rows are read in place below -- as they are, so unpack them BEFORE
    // the matrix's addresses are taken: need_raw_cold gives the cold rows a new home)
    if (has_cold(c) && (qs->chunk || qs->S < 2 || rows_per_range == 0)) MK_TRY(need_raw_cold(c));
    SlabArgs a;
    a.M = c->d_M; a.Mc = mat_ref(c).cold_m; a.P_hot = c->P_hot; a.ld = c->ld; a.G = c->G; a.ntiles = ntiles_of(c);
    a.nq = q1 - q0; a.q_begin = q0; a.S = qs->S; a.r_begin = 0; a.r_count = qs->S;
    a.entries = qs->d_entries; a.ent_off = qs->d_ent_off; a.split = qs->d_split; a.partials = c->d_partials;
    a.chunk = qs->chunk; a.nent = qs->d_scan_n;
    c->stats.scan_slab_launches++;
    if (!has_cold(c) || qs->chunk || qs->S < 2 || rows_per_range == 0) {
        // everything in HBM -- or ranges cut by count (small sets), which do not map to partition
        // ranges: cold rows, if any, are then read in place over PCIe
        ScopedTimer t(c, 1);
        return launch_scan_slab(c, a);
    }
    // Cold partition ranges are STREAMED: a range's rows are copied once into a staging buffer in
    // HBM and every query of the chunk scans them there, instead of each wave fetching its 1 KiB
    // pieces over PCIe.  The range the hot / cold boundary falls into is staged as a whole (its hot
    // rows by a device copy, the rest from host memory).
    const uint32_t S_hot = c->P_hot / rows_per_range;               // ranges that lie in HBM completely
    // two staging buffers (the copy of one group of ranges runs beside the scan of the previous one),
    // each as many ranges as fit a sixteenth of the hot part -- at least one range
    MK_TRY(ensure_cold_stage(c, rows_per_range));
    MK_TRY(ensure_zstage(c, c->cold_stage_rows));
    hipEvent_t ev_enter = c->ev_cold[4];
    hipEvent_t *ev_copy = c->ev_cold, *ev_scan = c->ev_cold + 2;
    // the copies may start as soon as everything queued so far (earlier scans out of the stage) is done
    MK_HIP(hipEventRecord(ev_enter, c->stream));
    MK_HIP(hipStreamWaitEvent(c->copy_stream, ev_enter, 0));
    if (S_hot) {                                                    // ... i.e. beside the launch over the hot ranges
        a.r_begin = 0; a.r_count = S_hot;
        ScopedTimer t(c, 1);
        MK_TRY(launch_scan_slab(c, a));
    }
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, c->cold_stage_rows / rows_per_range);
    uint32_t i = 0;
    for (uint32_t r = S_hot; r < qs->S; r += per, ++i) {
        const uint32_t nr = std::min(per, qs->S - r);
        const uint64_t first = (uint64_t)r * rows_per_range;
        if ((uint64_t)nr * rows_per_range > c->cold_stage_rows) {   // a range larger than a stage (a set with few, huge ranges): in place
            MK_TRY(need_raw_cold(c));
            a.M = c->d_M; a.Mc = mat_ref(c).cold_m; a.P_hot = c->P_hot;
        } else {
            const int b = (int)(i & 1u);
            uint8_t *stage = c->d_cold_stage + (uint64_t)b * c->cold_stage_rows * c->ld;
            const uint64_t last = first + (uint64_t)nr * rows_per_range;          // rows [first, last)
            const uint64_t hot_rows = first < c->P_hot ? std::min<uint64_t>(last, c->P_hot) - first : 0;
            if (i >= 2) MK_HIP(hipStreamWaitEvent(c->copy_stream, ev_scan[b], 0));   // the scan that read this buffer last
            if (hot_rows)
                MK_HIP(hipMemcpyAsync(stage, c->d_M + first * c->ld, hot_rows * c->ld, hipMemcpyDeviceToDevice, c->copy_stream));
            if (first + hot_rows < last) MK_TRY(stage_cold_rows(c, first + hot_rows, last, stage + hot_rows * c->ld, b, c->copy_stream));
            MK_HIP(hipEventRecord(ev_copy[b], c->copy_stream));
            MK_HIP(hipStreamWaitEvent(c->stream, ev_copy[b], 0));
            // row p of these ranges now lives at stage + (p - first) * ld: present the stage as "the matrix"
            a.M = stage - first * c->ld; a.Mc = nullptr; a.P_hot = c->P;
        }
        a.r_begin = r; a.r_count = nr;
        {
            ScopedTimer t(c, 1);
            MK_TRY(launch_scan_slab(c, a));
        }
        MK_HIP(hipEventRecord(ev_scan[i & 1u], c->stream));
    }
    return MK_OK;
}

// entrants of filter_results' heap for the rows in d_scores (see select.hip)
static int qset_select(mk_ctx *c, uint32_t n, const uint32_t *d_scores, const uint8_t *d_partials, uint32_t S,
                       const uint32_t *d_nent, uint32_t nresults, uint32_t min_score, double min_inter, uint32_t cap,
                       uint32_t *d_count, mk_hit *d_cand, uint64_t *d_rows = nullptr)
{
    SelectArgs a;
    a.scores = d_scores; a.partials = d_partials; a.nent = d_nent; a.S = S; a.W = c->W;
    a.tile_genomes = tile_genomes(c); a.G = c->G; a.nq = n; a.nresults = nresults;
    a.min_score = min_score; a.min_inter = min_inter; a.sketch_size = c->d_sketch_size;
    a.genome_size = c->d_genome_size; a.genome_id_base = c->p.genome_id_base; a.cap = cap;
    a.ratio = nullptr;
    if (d_partials) {                                              // the slab schedule's selection screens with one float per genome
        if (c->ratio_cap < c->capG) {
            dev_free(c->d_ratio);
            c->ratio_cap = 0;
            MK_TRY(dev_alloc(&c->d_ratio, (uint64_t)c->capG));
            c->ratio_cap = c->capG; c->ratio_gen = 0;
        }
        if (c->ratio_gen != c->gen) { MK_TRY(launch_ratio(c, c->d_ratio, c->capG)); c->ratio_gen = c->gen; }
        a.ratio = c->d_ratio;
    }
    a.count = d_count; a.cand = d_cand; a.rows = d_rows;
    ScopedTimer t(c, 2);
    return launch_select(c, a);
}

}  // namespace mk

using namespace mk;


// transient: the set lives for one mk_query call -- borrowed arena, and no wait for the copy
// (the caller's buffers have been copied into the pinned image; the call's own final wait covers it)
static int qset_upload(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t nq, mk_qset **out, bool transient)
{
    mk_qset *qs = nullptr;
    MK_TRY(qset_alloc(c, lens, nq, &qs, transient));
    std::unique_ptr<mk_qset, void (*)(mk_qset *)> guard(qs, qset_release);
    constexpr uint64_t kImageMax = 8ull << 20;
    if (qs->head_bytes <= kImageMax) {
        // small batch: ONE copy of a pinned image of (sequences, offsets, entry offsets)
        MK_TRY(ensure_pinned(c->h_stage, c->stage_cap, qs->head_bytes));
        for (uint32_t q = 0; q < nq; ++q) memcpy(c->h_stage + qs->h_off[q], seqs[q], lens[q]);
        memcpy(c->h_stage + qs->o_off, qs->h_off.data(), ((size_t)nq + 1) * 8);
        memcpy(c->h_stage + qs->o_ent_off, qs->h_ent_off.data(), ((size_t)nq + 1) * 8);
        MK_HIP(hipMemcpyAsync(qs->d_arena, c->h_stage, qs->head_bytes, hipMemcpyHostToDevice, c->stream));
        if (!transient) MK_HIP(hipStreamSynchronize(c->stream));    // the image is reused by the next upload
        *out = guard.release();
        return MK_OK;
    }
    MK_TRY(qset_copy_offsets(c, qs));
    // Short sequences are gathered so that a run of them is ONE copy (100,000 reads must not be
    // 100,000 copies); a long one (a contig, a whole genome) goes straight from the caller's
    // buffer -- a DMA when that buffer is pinned (mk_host_alloc), and no extra pass over it.
    constexpr uint64_t kDirect = 256u << 10;
    std::vector<char> host;
    bool ok = true;
    for (uint32_t q = 0; q < nq && ok;) {
        if (lens[q] >= kDirect) {
            ok = hipMemcpyAsync(qs->d_seq + qs->h_off[q], seqs[q], lens[q], hipMemcpyHostToDevice, c->stream) == hipSuccess;
            ++q;
            continue;
        }
        uint32_t e = q;
        while (e < nq && lens[e] < kDirect && qs->h_off[e + 1] - qs->h_off[q] <= (1ull << 30)) ++e;
        if (e == q) e = q + 1;
        const uint64_t bytes = qs->h_off[e] - qs->h_off[q];
        host.resize(bytes);
        for (uint32_t i = q; i < e; ++i) memcpy(host.data() + (qs->h_off[i] - qs->h_off[q]), seqs[i], lens[i]);
        // synchronous: `host` is reused for the next run
        if (bytes) ok = hipMemcpy(qs->d_seq + qs->h_off[q], host.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
        q = e;
    }
    if (ok) ok = hipStreamSynchronize(c->stream) == hipSuccess;    // the caller's buffers are free again
    if (!ok) { set_error("query upload failed"); return MK_ERR_DEVICE; }
    *out = guard.release();
    return MK_OK;
}

extern "C" {

int mk_qset_upload(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t nq, mk_qset **out)
{
    if (!c || !out || (nq && (!seqs || !lens))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    // a set that mixes short queries with longer ones: a shell over two sets (mk_internal.hpp), so that the short ones keep
    // the slab schedule inside ONE mk_qset_run
    std::vector<uint32_t> idx[2];
    for (uint32_t q = 0; q < nq; ++q) idx[beyond_short_len(c->p.k, lens[q]) ? 1 : 0].push_back(q);
    if (idx[0].empty() || idx[1].empty()) return qset_upload(c, seqs, lens, nq, out, false);
    std::unique_ptr<mk_qset, void (*)(mk_qset *)> shell(new mk_qset(), qset_release);
    mk_qset *qs = shell.get();
    qs->owner = c; qs->nq = nq; qs->arena_borrowed = false; qs->split_in_arena = false; qs->d_split = nullptr; qs->d_arena = nullptr;
    qs->d_seq = nullptr; qs->d_off = nullptr; qs->d_ent_off = nullptr; qs->d_entries = nullptr; qs->d_nent = nullptr;
    qs->d_dense = nullptr; qs->d_dense_q = nullptr; qs->d_scan_n = nullptr; qs->S = 0; qs->chunk = 0; qs->slab_ok = false;
    qs->sketched = false; qs->gen = 0; qs->short_max_nk = 0; qs->head_bytes = 0; qs->total_len = 0;
    for (int i = 0; i < 2; ++i) {
        const uint32_t n = (uint32_t)idx[i].size();
        std::vector<const char *> s(n);
        std::vector<uint64_t> l(n);
        for (uint32_t j = 0; j < n; ++j) { s[j] = seqs[idx[i][j]]; l[j] = lens[idx[i][j]]; }
        MK_TRY(qset_upload(c, s.data(), l.data(), n, &qs->part[i], false));
        MK_TRY(dev_alloc(&qs->d_part_q[i], n));
        MK_HIP(hipMemcpy(qs->d_part_q[i], idx[i].data(), (size_t)n * 4, hipMemcpyHostToDevice));
        qs->part_q[i] = std::move(idx[i]);
    }
    *out = shell.release();
    return MK_OK;
}

int mk_qset_synthetic(mk_ctx *c, uint64_t first_id, uint32_t nq, uint64_t G, uint64_t L, uint64_t qlen,
                      mk_qset **out)
{
    if (!c || !out) { set_error("null argument"); return MK_ERR_ARG; }
    if (!G || qlen == 0 || L <= qlen) { set_error("bad synthetic query shape (need genome_len > query_len)"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    std::vector<uint64_t> lens(nq, qlen);
    mk_qset *qs = nullptr;
    MK_TRY(qset_alloc(c, lens.data(), nq, &qs));
    int rc = qset_copy_offsets(c, qs);
    if (rc == MK_OK) rc = launch_synth_queries(c, first_id, nq, G, L, qlen, qs->d_seq);
    if (rc != MK_OK) { qset_release(qs); return rc; }
    *out = qs;
    return MK_OK;
}

int mk_qset_invalidate(mk_ctx *c, mk_qset *qs)
{
    if (!c || !qs) { set_error("null argument"); return MK_ERR_ARG; }
    qs->sketched = false;
    for (int i = 0; i < 2; ++i) if (qs->part[i]) qs->part[i]->sketched = false;
    return MK_OK;
}

void mk_qset_free(mk_ctx *c, mk_qset *qs)
{
    if (c) { (void)hipSetDevice(c->p.device); (void)hipStreamSynchronize(c->stream); }
    qset_release(qs);
}

// mk_qset_run / mk_qset_run_compact: the output is either (d_count, d_cand) or d_rows
// after_chunk (may be null): called once the scan + selection of queries [q0, q1) have been QUEUED on the context's
// stream (comm.hip: the chunk's exchange rows go out on the communicator's stream while the next chunk scans);
// min_chunks: cut the set into at least that many chunks (so that there is a next chunk to overlap with)
}  // extern "C"
namespace mk {
int qset_run(mk_ctx *c, mk_qset *qs, uint32_t nresults, uint32_t min_score, double min_inter, uint32_t cap,
             uint32_t *d_count, mk_hit *d_cand, uint64_t *d_rows, const std::function<int(uint32_t, uint32_t)> *after_chunk,
             uint32_t min_chunks)
{
    if (nresults > kSelectMaxResults) { set_error("device selection supports nresults <= 64"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (nan_candidates_possible(c, min_score)) {
        set_error("min_score 0 over an index with empty sketches yields NaN intersections: use mk_query");
        return MK_ERR_UNSUPPORTED;
    }
    if (qs->part[0]) {
        // a mixed set: each part runs as a set with its own schedule into a buffer of the shell, and a copy kernel puts its
        // rows in their places; the exchange of a sharded run follows the whole set (its blocks are ranges of queries)
        const uint64_t row_bytes = d_rows ? ((uint64_t)cap + 1) * 8 : (uint64_t)cap * sizeof(mk_hit);
        const uint32_t most = (uint32_t)std::max(qs->part_q[0].size(), qs->part_q[1].size());
        const uint64_t need = (uint64_t)most * (row_bytes + 4) + 256;
        if (need > qs->part_out_bytes) {
            dev_free(qs->d_part_out);
            qs->part_out_bytes = 0;
            MK_TRY(dev_alloc(&qs->d_part_out, need));
            qs->part_out_bytes = need;
        }
        for (int i = 0; i < 2; ++i) {
            mk_qset *p = qs->part[i];
            uint8_t *rows = qs->d_part_out;                            // [n][row_bytes], then (count form) [n] counts
            uint32_t *cnt = reinterpret_cast<uint32_t *>(qs->d_part_out + (((uint64_t)p->nq * row_bytes + 255) & ~255ull));
            MK_TRY(qset_run(c, p, nresults, min_score, min_inter, cap, d_rows ? nullptr : cnt, d_rows ? nullptr : reinterpret_cast<mk_hit *>(rows),
                            d_rows ? reinterpret_cast<uint64_t *>(rows) : nullptr, nullptr, 1));
            MK_TRY(launch_place_rows(c, rows, row_bytes, qs->d_part_q[i], p->nq, d_rows ? reinterpret_cast<uint8_t *>(d_rows) : reinterpret_cast<uint8_t *>(d_cand),
                                     d_rows ? nullptr : cnt, d_count));
        }
        if (after_chunk) MK_TRY((*after_chunk)(0, qs->nq));
        return MK_OK;
    }
    MK_TRY(qset_sketch(c, qs));
    const uint64_t rstride = (uint64_t)cap + 1;
    if (c->G == 0) {                                             // an empty shard still takes part in the exchange
        if (d_rows) MK_HIP(hipMemsetAsync(d_rows, 0, (size_t)qs->nq * rstride * 8, c->stream));
        else MK_HIP(hipMemsetAsync(d_count, 0, (size_t)qs->nq * 4, c->stream));
        if (after_chunk) MK_TRY((*after_chunk)(0, qs->nq));
        return MK_OK;
    }
    const bool slab = qs->slab_ok;
    uint32_t per = slab ? chunk_queries_slab(c, qs->nq, qs->S) : chunk_queries(c, qs->nq);
    if (min_chunks > 1) per = std::max<uint32_t>(1, std::min<uint32_t>(per, (qs->nq + min_chunks - 1) / min_chunks));
    if (slab) MK_TRY(ensure_partials(c, (uint64_t)per * partial_bytes_per_query(c, qs->S)));
    else MK_TRY(ensure_scores(c, per));
    for (uint32_t q0 = 0; q0 < qs->nq; q0 += per) {
        const uint32_t q1 = std::min(qs->nq, q0 + per);
        uint32_t *cnt = d_rows ? nullptr : d_count + q0;
        mk_hit *cand = d_rows ? nullptr : d_cand + (uint64_t)q0 * cap;
        uint64_t *rows = d_rows ? d_rows + (uint64_t)q0 * rstride : nullptr;
        if (slab) {
            MK_TRY(qset_scan_slab(c, qs, q0, q1));
            MK_TRY(qset_select(c, q1 - q0, nullptr, c->d_partials, qs->S, qs->d_nent + q0, nresults, min_score,
                               min_inter, cap, cnt, cand, rows));
        } else {
            MK_TRY(qset_scan(c, qs, q0, q1, c->d_scores, score_layout_tiles(c->W, q1 - q0)));
            MK_TRY(qset_select(c, q1 - q0, c->d_scores, nullptr, 0, nullptr, nresults, min_score, min_inter, cap,
                               cnt, cand, rows));
        }
        if (after_chunk) MK_TRY((*after_chunk)(q0, q1));
    }
    return MK_OK;
}
}  // namespace mk
extern "C" {

int mk_qset_run(mk_ctx *c, mk_qset *qs, uint32_t nresults, uint32_t min_score, double min_inter, uint32_t cap,
                uint32_t *d_count, mk_hit *d_cand)
{
    if (!c || !qs || !d_count || !d_cand || !cap) { set_error("null argument"); return MK_ERR_ARG; }
    return qset_run(c, qs, nresults, min_score, min_inter, cap, d_count, d_cand, nullptr, nullptr, 1);
}

int mk_qset_run_compact(mk_ctx *c, mk_qset *qs, uint32_t nresults, uint32_t min_score, double min_inter,
                        uint32_t cap, uint64_t *d_rows)
{
    if (!c || !qs || !d_rows || !cap) { set_error("null argument"); return MK_ERR_ARG; }
    return qset_run(c, qs, nresults, min_score, min_inter, cap, nullptr, nullptr, d_rows, nullptr, 1);
}

int mk_qset_scores(mk_ctx *c, mk_qset *qs, uint32_t q0, uint32_t q1, uint32_t *d_scores)
{
    if (!c || !qs || !d_scores) { set_error("null argument"); return MK_ERR_ARG; }
    if (q0 > q1 || q1 > qs->nq) { set_error("query range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (qs->part[0]) {                                           // a mixed set: query by query from the part that holds it
        for (uint32_t q = q0; q < q1; ++q)
            for (int i = 0; i < 2; ++i) {
                const auto it = std::lower_bound(qs->part_q[i].begin(), qs->part_q[i].end(), q);
                if (it == qs->part_q[i].end() || *it != q) continue;
                const uint32_t at = (uint32_t)(it - qs->part_q[i].begin());
                MK_TRY(mk_qset_scores(c, qs->part[i], at, at + 1, d_scores + (uint64_t)(q - q0) * c->G));
            }
        return MK_OK;
    }
    MK_TRY(qset_sketch(c, qs));
    return qset_scan(c, qs, q0, q1, d_scores, score_layout_rows(c->W, c->G, c->G));   // dense rows for the caller
}

int mk_qset_active(mk_ctx *c, mk_qset *qs, uint32_t *active)
{
    if (!c || !qs || !active) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (qs->part[0]) {
        for (int i = 0; i < 2; ++i) {
            std::vector<uint32_t> a(qs->part[i]->nq);
            MK_TRY(mk_qset_active(c, qs->part[i], a.data()));
            for (uint32_t j = 0; j < qs->part[i]->nq; ++j) active[qs->part_q[i][j]] = a[j];
        }
        return MK_OK;
    }
    MK_TRY(qset_sketch(c, qs));
    MK_HIP(hipStreamSynchronize(c->stream));
    if (qs->nq) MK_HIP(hipMemcpy(active, qs->d_nent, (size_t)qs->nq * 4, hipMemcpyDeviceToHost));
    return MK_OK;
}

static int account(mk_ctx *c, mk_qset *qs, std::vector<uint32_t> &act)
{
    act.resize(qs->nq);
    MK_TRY(mk_qset_active(c, qs, act.data()));
    uint64_t a = 0;
    for (uint32_t v : act) a += v;
    c->stats.active_partitions += a;
    c->stats.comparisons += a * c->G;
    c->stats.scan_algo_bytes += a * c->G * c->W + 4ull * qs->nq * c->G;
    return MK_OK;
}

int mk_query_scores(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t nq, uint32_t *scores)
{
    if (!c || (nq && (!seqs || !lens || !scores))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (!nq || !c->G) return MK_OK;
    mk_qset *qs = nullptr;
    MK_TRY(qset_upload(c, seqs, lens, nq, &qs, true));
    std::unique_ptr<mk_qset, void (*)(mk_qset *)> guard(qs, qset_release);
    MK_TRY(qset_sketch(c, qs));
    const uint32_t per = chunk_queries(c, nq);
    const uint64_t ld = score_row_entries(c);                 // whole tiles per row: 16-byte stores everywhere
    MK_TRY(ensure_scores(c, per));
    for (uint32_t q0 = 0; q0 < nq; q0 += per) {
        const uint32_t q1 = std::min(nq, q0 + per);
        MK_TRY(qset_scan(c, qs, q0, q1, c->d_scores, score_layout_rows(c->W, ld, c->G)));
        MK_HIP(hipMemcpy2DAsync(scores + (uint64_t)q0 * c->G, (size_t)c->G * 4, c->d_scores, (size_t)ld * 4,
                                (size_t)c->G * 4, q1 - q0, hipMemcpyDeviceToHost, c->stream));
        MK_HIP(hipStreamSynchronize(c->stream));
    }
    std::vector<uint32_t> act;
    MK_TRY(account(c, qs, act));
    return MK_OK;
}


int mk_query(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t nq, uint32_t nresults,
             uint32_t min_score, double min_inter, mk_hit *hits, uint32_t *nhits, uint32_t *active)
{
    if (!c || (nq && (!seqs || !lens || !hits || !nhits))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (!nq) return MK_OK;
    if (!c->G) {
        memset(nhits, 0, (size_t)nq * 4);
        if (active) memset(active, 0, (size_t)nq * 4);     // no column is ever compared
        return MK_OK;
    }
    // Very large calls are answered in slices: the device-side query set (sequences, entry
    // lists) grows with the number of queries, the result does not depend on the slicing.
    constexpr uint32_t kMaxCall = 1u << 18;
    if (nq > kMaxCall) {
        for (uint32_t q0 = 0; q0 < nq; q0 += kMaxCall) {
            const uint32_t n = std::min(kMaxCall, nq - q0);
            MK_TRY(mk_query(c, seqs + q0, lens + q0, n, nresults, min_score, min_inter, hits + (size_t)q0 * nresults,
                            nhits + q0, active ? active + q0 : nullptr));
        }
        return MK_OK;
    }
    // A batch that mixes short queries with long ones is answered as two batches, so
    // that the short ones keep the slab schedule (long ones need the plain / dense kernels)
    {
        std::vector<uint32_t> idx_short, idx_long;
        for (uint32_t q = 0; q < nq; ++q)
            (beyond_short_len(c->p.k, lens[q]) ? idx_long : idx_short).push_back(q);
        if (!idx_short.empty() && !idx_long.empty() && nresults > 0) {
            for (const std::vector<uint32_t> *part : {&idx_short, &idx_long}) {
                const uint32_t n = (uint32_t)part->size();
                std::vector<const char *> s(n);
                std::vector<uint64_t> l(n);
                std::vector<mk_hit> h((size_t)n * nresults);
                std::vector<uint32_t> nh(n), act(n);
                for (uint32_t i = 0; i < n; ++i) { s[i] = seqs[(*part)[i]]; l[i] = lens[(*part)[i]]; }
                MK_TRY(mk_query(c, s.data(), l.data(), n, nresults, min_score, min_inter, h.data(), nh.data(), act.data()));
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t q = (*part)[i];
                    nhits[q] = nh[i];
                    if (active) active[q] = act[i];
                    std::copy(h.begin() + (size_t)i * nresults, h.begin() + (size_t)i * nresults + nh[i],
                              hits + (size_t)q * nresults);
                }
            }
            return MK_OK;
        }
    }
    mk_qset *qs = nullptr;
    MK_TRY(qset_upload(c, seqs, lens, nq, &qs, true));
    std::unique_ptr<mk_qset, void (*)(mk_qset *)> guard(qs, qset_release);
    MK_TRY(qset_sketch(c, qs));
    const uint32_t cap = 256;
    const bool on_device = nresults <= kSelectMaxResults && !nan_candidates_possible(c, min_score);
    const bool slab = on_device && qs->slab_ok;
    const uint32_t per = slab ? chunk_queries_slab(c, nq, qs->S) : chunk_queries(c, nq);
    if (slab) {
        MK_TRY(ensure_partials(c, (uint64_t)per * partial_bytes_per_query(c, qs->S)));
        MK_TRY(ensure_scores(c, 1));
    } else {
        MK_TRY(ensure_scores(c, per + 1));                    // + one row-major row for replays
    }
    uint32_t *const d_replay_row = c->d_scores + (slab ? 0 : (uint64_t)per * score_row_entries(c));
    if (on_device && (uint64_t)per > c->cand_cap_q) {
        dev_free(c->d_count); dev_free(c->d_cand);
        c->cand_cap_q = 0;
        MK_TRY(dev_alloc(&c->d_count, (uint64_t)per));
        MK_TRY(dev_alloc(&c->d_cand, (uint64_t)per * cap));
        c->cand_cap_q = per;
    }
    if (on_device && (uint64_t)per * std::max(nresults, 1u) > c->hits_cap) {
        dev_free(c->d_hits);
        c->hits_cap = 0;
        MK_TRY(dev_alloc(&c->d_hits, (uint64_t)per * std::max(nresults, 1u)));
        c->hits_cap = (uint64_t)per * std::max(nresults, 1u);
    }
    if (on_device && (uint64_t)per > c->nhits_cap) {              // sized on its own: nresults differs from call to call
        dev_free(c->d_nhits);
        c->nhits_cap = 0;
        MK_TRY(dev_alloc(&c->d_nhits, (uint64_t)per));
        c->nhits_cap = per;
    }
    std::vector<uint32_t> row;
    std::vector<mk_hit> full;
    std::vector<uint32_t> act(nq);
    for (uint32_t q0 = 0; q0 < nq; q0 += per) {
        const uint32_t q1 = std::min(nq, q0 + per), n = q1 - q0;
        // results of a small chunk come back through one pinned block (counts, active partitions,
        // hits): three queued copies and ONE wait, instead of a blocking copy per array
        const uint64_t res_bytes = (uint64_t)n * (8 + (uint64_t)nresults * sizeof(mk_hit));
        const bool pinned = on_device && res_bytes <= (1ull << 20);
        uint32_t *p_nh = nullptr, *p_act = nullptr;
        mk_hit *p_hits = nullptr;
        if (on_device) {
            if (slab) {
                MK_TRY(qset_scan_slab(c, qs, q0, q1));
                MK_TRY(qset_select(c, n, nullptr, c->d_partials, qs->S, qs->d_nent + q0, nresults, min_score,
                                   min_inter, cap, c->d_count, c->d_cand));
            } else {
                MK_TRY(qset_scan(c, qs, q0, q1, c->d_scores, score_layout_tiles(c->W, n)));
                MK_TRY(qset_select(c, n, c->d_scores, nullptr, 0, nullptr, nresults, min_score, min_inter, cap,
                                   c->d_count, c->d_cand));
            }
            // the heap over the entrants runs on the device too (K6b): only the hits come back
            MergeArgs ma{c->d_count, c->d_cand, 1, n, cap, nresults, c->d_hits, c->d_nhits};
            MK_TRY(launch_merge(c, ma));
            if (pinned) {
                MK_TRY(ensure_pinned(c->h_res, c->res_cap, res_bytes + 64));
                p_hits = reinterpret_cast<mk_hit *>(c->h_res);
                p_nh = reinterpret_cast<uint32_t *>(c->h_res + (uint64_t)n * nresults * sizeof(mk_hit));
                p_act = p_nh + n;
            } else {
                p_hits = hits + (size_t)q0 * nresults; p_nh = nhits + q0; p_act = act.data() + q0;
            }
            MK_HIP(hipMemcpyAsync(p_nh, c->d_nhits, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            MK_HIP(hipMemcpyAsync(p_act, qs->d_nent + q0, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            if (nresults)
                MK_HIP(hipMemcpyAsync(p_hits, c->d_hits, (size_t)n * nresults * sizeof(mk_hit), hipMemcpyDeviceToHost,
                                      c->stream));
        }
        MK_HIP(hipStreamSynchronize(c->stream));
        if (pinned) {
            memcpy(nhits + q0, p_nh, (size_t)n * 4);
            memcpy(act.data() + q0, p_act, (size_t)n * 4);
            if (nresults) memcpy(hits + (size_t)q0 * nresults, p_hits, (size_t)n * nresults * sizeof(mk_hit));
        }
        for (uint32_t i = 0; i < n; ++i) {
            mk_hit *out = hits + (size_t)(q0 + i) * nresults;
            if (on_device && nhits[q0 + i] != kMergeOverflow) continue;
            // more heap entrants than the device row holds (or a top-N beyond the device
            // selection): replay this query over a dense score row of its own
            uint32_t *d_row = d_replay_row;
            row.resize(c->G);
            MK_TRY(qset_scan(c, qs, q0 + i, q0 + i + 1, d_row, score_layout_rows(c->W, score_row_entries(c), c->G)));
            MK_HIP(hipMemcpyAsync(row.data(), d_row, (size_t)c->G * 4, hipMemcpyDeviceToHost, c->stream));
            MK_HIP(hipStreamSynchronize(c->stream));
            full.clear();
            for (uint32_t g = 0; g < c->G; ++g) {
                if (row[g] < min_score) continue;
                const double jac = (double)row[g] / c->h_sketch_size[g];
                const double inter = jac * c->h_genome_size[g];
                if (inter < min_inter) continue;
                full.push_back(mk_hit{g + c->p.genome_id_base, row[g], jac, inter});
            }
            nhits[q0 + i] = mk_filter_candidates(full.data(), (uint32_t)full.size(), nresults, out);
        }
    }
    if (!on_device) MK_HIP(hipMemcpy(act.data(), qs->d_nent, (size_t)nq * 4, hipMemcpyDeviceToHost));
    {
        uint64_t a = 0;
        for (uint32_t v : act) a += v;
        c->stats.active_partitions += a;
        c->stats.comparisons += a * c->G;
        c->stats.scan_algo_bytes += a * c->G * c->W + 4ull * nq * c->G;
    }
    if (active) memcpy(active, act.data(), (size_t)nq * 4);
    return drain_timers(c);                                      // every event has fired: fold them in, keep the list short
}

int mk_exact(mk_ctx *c, const char *const *contigs, const uint64_t *contig_lens, uint32_t n_contigs,
             const char *const *queries, const uint64_t *query_lens, uint32_t nq, uint64_t *inter, uint64_t *uni)
{
    if (!c || (n_contigs && (!contigs || !contig_lens)) || (nq && (!queries || !query_lens || !inter || !uni))) {
        set_error("null argument");
        return MK_ERR_ARG;
    }
    MK_TRY(use_device(c));
    MK_TRY(exact_load_genome(c, contigs, contig_lens, n_contigs));
    return exact_queries(c, queries, query_lens, nq, inter, uni);
}

int mk_exact_load_genome(mk_ctx *c, const char *const *contigs, const uint64_t *contig_lens, uint32_t n_contigs)
{
    if (!c || (n_contigs && (!contigs || !contig_lens))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    return exact_load_genome(c, contigs, contig_lens, n_contigs);
}

int mk_exact_query(mk_ctx *c, const char *const *queries, const uint64_t *query_lens, uint32_t nq, uint64_t *inter,
                   uint64_t *uni)
{
    if (!c || (nq && (!queries || !query_lens || !inter || !uni))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    return exact_queries(c, queries, query_lens, nq, inter, uni);
}

}  // extern "C"
