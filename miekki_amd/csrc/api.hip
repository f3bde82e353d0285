// C ABI of libmiekki_hip.so: context, index build / import / export, query
// pipeline.  Host-side orchestration only; the arithmetic lives in sketch.hip and
// scan.hip.  There is deliberately no CPU fallback anywhere in this file.
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

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

template <typename T>
static int dev_alloc(T **p, uint64_t count)
{
    *p = nullptr;
    if (!count) return MK_OK;
    MK_HIP(hipMalloc((void **)p, count * sizeof(T)));
    return MK_OK;
}

template <typename T>
static void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// ---- device timers -----------------------------------------------------------
int timer_begin(mk_ctx *c, int kind, Timer &t, hipStream_t stream)
{
    if (!c->free_timers.empty()) { t = c->free_timers.back(); c->free_timers.pop_back(); }
    else { MK_HIP(hipEventCreate(&t.a)); MK_HIP(hipEventCreate(&t.b)); }
    t.kind = kind;
    MK_HIP(hipEventRecord(t.a, stream ? stream : c->stream));
    return MK_OK;
}

int timer_end(mk_ctx *c, Timer &t, hipStream_t stream)
{
    MK_HIP(hipEventRecord(t.b, stream ? stream : c->stream));
    c->pending.push_back(t);
    return MK_OK;
}

int drain_timers(mk_ctx *c)
{
    for (Timer &t : c->pending) {
        MK_HIP(hipEventSynchronize(t.b));
        float ms = 0;
        MK_HIP(hipEventElapsedTime(&ms, t.a, t.b));
        switch (t.kind) {
        case 0: c->stats.sketch_ms += ms; break;
        case 1: c->stats.scan_ms += ms; c->stats.scan_launches++; break;
        case 2: c->stats.filter_ms += ms; break;
        case 3: c->stats.build_sketch_ms += ms; break;
        case 4: c->stats.build_finalize_ms += ms; break;
        }
        c->free_timers.push_back(t);
    }
    c->pending.clear();
    return MK_OK;
}

struct ScopedTimer {
    mk_ctx *c; Timer t; bool on; hipStream_t stream;
    ScopedTimer(mk_ctx *ctx, int kind, hipStream_t st = nullptr) : c(ctx), on(false), stream(st) { on = timer_begin(c, kind, t, stream) == MK_OK; }
    ~ScopedTimer() { if (on) (void)timer_end(c, t, stream); }
};

static int settle_build(mk_ctx *c);
static int build_from_characters(mk_ctx *c);

// Every entry point starts here: bind the device and, unless the caller is an append
// that wants to overlap with it, fold the build batch still in flight into the index.
static int use_device(const mk_ctx *cc, bool settle = true)
{
    MK_HIP(hipSetDevice(cc->p.device));
    mk_ctx *c = const_cast<mk_ctx *>(cc);
    if (settle && (c->build.on || c->older.on)) MK_TRY(settle_build(c));
    return MK_OK;
}

// ---- matrix capacity -----------------------------------------------------------
// rows that fit the matrix's HBM budget at pitch ld: all P, or a multiple of P / 64 (the boundary then falls
// between two of the slab schedule's finest partition ranges, S = 64; with coarser ranges the one it falls
// into is staged as a whole, qset_scan_slab)
static uint32_t hot_rows_for(const mk_ctx *c, uint64_t ld, uint64_t budget)
{
    if (!budget || (uint64_t)c->P * ld <= budget) return c->P;
    const uint32_t unit = std::max<uint32_t>(1, c->P / 64);
    return (uint32_t)std::min<uint64_t>(c->P, budget / ld / unit * unit);
}

static int ensure_capacity(mk_ctx *c, uint32_t need)
{
    if (need <= c->capG) return MK_OK;
    MK_TRY(need_raw_cold(c));                                      // (the re-layout copies rows as they are)
    uint64_t cap = std::max<uint64_t>(need, std::max<uint64_t>(256, (uint64_t)c->capG * 2));
    cap = std::min<uint64_t>(cap, 0xffffff00ull);
    uint64_t ld = (cap * c->W + kTileBytes - 1) / kTileBytes * kTileBytes;
    cap = ld / c->W;
    uint8_t *nM = nullptr, *nH = nullptr;
    uint32_t *nss = nullptr;
    uint64_t *ngs = nullptr;
    uint32_t P_hot = hot_rows_for(c, ld, c->hbm_matrix_budget);
    if (P_hot < c->P) {
        // over budget at the doubled size: grow by a quarter instead (still geometric -- a build that appends
        // batch after batch past its reservation must not re-lay the cold rows out every 64 genomes: each
        // re-layout is a second page-locked copy of all of them)
        const uint64_t want = std::min<uint64_t>(0xffffff00ull, std::max<uint64_t>(need, (uint64_t)c->capG + c->capG / 4));
        ld = (want * c->W + kTileBytes - 1) / kTileBytes * kTileBytes;
        cap = ld / c->W;
        P_hot = hot_rows_for(c, ld, c->hbm_matrix_budget);
    }
    if (hipMalloc((void **)&nM, std::max<uint64_t>((uint64_t)P_hot * ld, 16)) != hipSuccess) {
        // the doubled matrix does not fit beside the old one: take exactly what is needed, and if
        // even that does not fit the free memory, keep the rows that do and put the rest in host memory
        (void)hipGetLastError();
        ld = ((uint64_t)need * c->W + kTileBytes - 1) / kTileBytes * kTileBytes;
        cap = ld / c->W;
        P_hot = hot_rows_for(c, ld, c->hbm_matrix_budget);
        if (hipMalloc((void **)&nM, std::max<uint64_t>((uint64_t)P_hot * ld, 16)) != hipSuccess) {
            (void)hipGetLastError();
            size_t free_b = 0, total_b = 0;
            MK_HIP(hipMemGetInfo(&free_b, &total_b));
            P_hot = hot_rows_for(c, ld, (uint64_t)free_b * 3 / 4);
            MK_TRY(dev_alloc(&nM, std::max<uint64_t>((uint64_t)P_hot * ld, 16)));
        }
    }
    if (P_hot < c->P)                                              // cold rows: page-locked host memory the GPU can address
        if (hipHostMalloc((void **)&nH, (uint64_t)(c->P - P_hot) * ld, hipHostMallocDefault) != hipSuccess) {
            // (the old cold rows are still page-locked beside the new ones: an index this close to the host's
            // limit has to be reserved up front, mk_reserve)
            (void)hipGetLastError();
            dev_free(nM);
            set_error("matrix of %llu bytes exceeds its HBM budget and the cold rows (%llu bytes) do not fit page-locked host memory",
                      (unsigned long long)c->P * ld, (unsigned long long)(c->P - P_hot) * ld);
            return MK_ERR_NOMEM;
        }
    if (dev_alloc(&nss, cap) != MK_OK || dev_alloc(&ngs, cap) != MK_OK) {
        dev_free(nM); dev_free(nss); dev_free(ngs);
        if (nH) (void)hipHostFree(nH);
        return MK_ERR_NOMEM;
    }
    MK_HIP(hipMemsetAsync(nM, 0, std::max<uint64_t>((uint64_t)P_hot * ld, 16), c->stream));
    if (nH) memset(nH, 0, (uint64_t)(c->P - P_hot) * ld);
    if (c->G) {
        // old row p -> new row p, whichever side either lives on
        const size_t w = (size_t)c->G * c->W;
        auto old_row = [&](uint32_t p) { return p < c->P_hot ? c->d_M + (uint64_t)p * c->ld : c->h_M + (uint64_t)(p - c->P_hot) * c->ld; };
        auto new_row = [&](uint32_t p) { return p < P_hot ? nM + (uint64_t)p * ld : nH + (uint64_t)(p - P_hot) * ld; };
        const uint32_t cuts[4] = {0, std::min(c->P_hot, P_hot), std::max(c->P_hot, P_hot), c->P};
        for (int i = 0; i < 3; ++i)
            if (cuts[i + 1] > cuts[i])
                MK_HIP(hipMemcpy2DAsync(new_row(cuts[i]), ld, old_row(cuts[i]), c->ld, w, cuts[i + 1] - cuts[i], hipMemcpyDefault,
                                        c->stream));
        MK_HIP(hipMemcpyAsync(nss, c->d_sketch_size, (size_t)c->G * 4, hipMemcpyDeviceToDevice, c->stream));
        MK_HIP(hipMemcpyAsync(ngs, c->d_genome_size, (size_t)c->G * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    MK_HIP(hipStreamSynchronize(c->stream));
    dev_free(c->d_M); dev_free(c->d_sketch_size); dev_free(c->d_genome_size);
    if (c->h_M) (void)hipHostFree(c->h_M);
    dev_free(c->d_cold_stage);
    c->cold_stage_rows = 0;                                        // the stage was sized for the old pitch; its events do not depend on it and stay
    c->d_M = nM; c->h_M = nH; c->P_hot = P_hot; c->d_sketch_size = nss; c->d_genome_size = ngs;
    c->ld = ld; c->capG = (uint32_t)cap;
    return MK_OK;
}

// ---- index build ---------------------------------------------------------------
// the per-batch counters, offsets, overflow list ... of the build (mk_ctx::BuildSide); until a build has run the
// aliases point at side 0
int ensure_build_counters(mk_ctx *c)
{
    if (c->d_counters) return MK_OK;
    MK_TRY(ensure_build_side(c, 0));
    use_build_side(c, 0);
    return MK_OK;
}

// bytes of the coarse summary level: one bit per 2048 cells, written 16 bits per wave of bloom_summary_kernel
uint64_t bloom_summary_bytes(const mk_ctx *c)
{
    const uint64_t nwords = (c->bloom_dev_bytes / 8 + 31) / 32 + 1, threads = (nwords + 1) / 2;
    return (threads + 63) / 64 * 2;
}

int ensure_bloom_summary_arrays(mk_ctx *c)
{
    if (!c->d_bloom || c->d_bloom_full) return MK_OK;
    const uint64_t nwords = (c->bloom_dev_bytes / 8 + 31) / 32 + 1, words2 = (bloom_summary_bytes(c) + 7) / 8 + 8;
    MK_TRY(dev_alloc(&c->d_bloom_full, nwords));
    MK_TRY(dev_alloc(&c->d_bloom_full2, words2));
    MK_HIP(hipMemset(c->d_bloom_full, 0, nwords * 4));
    MK_HIP(hipMemset(c->d_bloom_full2, 0, words2 * 8));
    MK_HIP(hipStreamSynchronize(nullptr));                        // (the front stream does not wait for the null stream by itself)
    c->bloom_full_stale = true;
    return MK_OK;
}

// Cells are about to be replaced (an import): the summaries may not claim anything until they have been recomputed.  The
// build's front stage reads the coarse level before the back stage refreshes it, so "stale" alone is not enough here.
int forget_bloom_summary(mk_ctx *c)
{
    c->bloom_full_stale = true;
    // (the build's first-writer keys are never reset cell by cell -- a posted key means the cell was set in the same batch
    // and is never looked at again -- but cells that are REPLACED may be empty again under an old key)
    if (c->d_bloom_order) MK_HIP(hipMemsetAsync(c->d_bloom_order, 0xFF, c->bloom_dev_bytes * 4, c->stream));
    if (!c->d_bloom_full) return MK_OK;
    const uint64_t nwords = (c->bloom_dev_bytes / 8 + 31) / 32 + 1, words2 = (bloom_summary_bytes(c) + 7) / 8 + 8;
    MK_HIP(hipMemsetAsync(c->d_bloom_full, 0, nwords * 4, c->stream));
    MK_HIP(hipMemsetAsync(c->d_bloom_full2, 0, words2 * 8, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

// the "all eight cells set" summary of the Bloom filter, current
int ensure_bloom_summary(mk_ctx *c)
{
    if (!c->d_bloom) return MK_OK;
    MK_TRY(ensure_bloom_summary_arrays(c));
    if (c->bloom_full_stale) {
        MK_TRY(launch_bloom_summary(c));
        c->bloom_full_stale = false;
    }
    return MK_OK;
}

// for_append = false: the long-query sketches borrow the tables / slots only; the Bloom
// first-writer keys (8 bytes per reachable cell: 512 MiB at -b 33) are build-only.
// seq_bytes: characters the batch brings (0: it comes packed, or is generated on the device).
static int ensure_build_scratch(mk_ctx *c, uint64_t seq_bytes, int buf = 0, bool for_append = true)
{
    if (!c->d_tables) {
        const uint64_t budget = 1ull << 30;                       // table bytes per batch
        c->build_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(kBuildBatch, budget / ((uint64_t)c->P * 8)));
        MK_TRY(dev_alloc(&c->d_tables, (uint64_t)c->build_batch * c->P));
    }
    MK_TRY(ensure_build_counters(c));
    if (for_append && c->d_bloom && !c->d_bloom_order) {
        MK_TRY(dev_alloc(&c->d_bloom_order, c->bloom_dev_bytes));
        MK_HIP(hipMemsetAsync(c->d_bloom_order, 0xFF, c->bloom_dev_bytes * 4, c->stream));
        const uint64_t regions = (c->bloom_dev_bytes >> kBloomRegionLog2) + 1;
        MK_TRY(dev_alloc(&c->d_bloom_touched, regions));
        MK_HIP(hipMemsetAsync(c->d_bloom_touched, 0, regions, c->stream));
    }
    if (!c->h_sizes) MK_HIP(hipHostMalloc((void **)&c->h_sizes, 2 * sizeof *c->h_sizes, hipHostMallocDefault));
    if (!c->h_img) MK_HIP(hipHostMalloc((void **)&c->h_img, sizeof *c->h_img, hipHostMallocDefault));
    if (for_append)
        for (int b = 0; b < 2; ++b) {
            MK_TRY(ensure_build_side(c, b));
            if (!c->d_pk_off[b]) MK_TRY(dev_alloc(&c->d_pk_off[b], kBuildBatch + 1));
            if (!c->d_heads[b]) MK_TRY(dev_alloc(&c->d_heads[b], (uint64_t)kBuildBatch * 32));
        }
    if (seq_bytes > c->seq_cap[buf]) {                             // never the buffer of the batch in flight
        const uint64_t old_cap = c->seq_cap[buf];
        dev_free(c->d_seq[buf]);
        c->seq_cap[buf] = 0;
        const uint64_t cap = std::max<uint64_t>(seq_bytes, std::max(c->seq_cap[buf ^ 1], old_cap * 2));
        MK_TRY(dev_alloc(&c->d_seq[buf], cap + 64));
        c->seq_cap[buf] = cap;
    }
    return MK_OK;
}

// Offsets of a batch's sequences in its packed form (mk_ctx::d_pk): 8 bytes of codes per 32 positions plus
// 32 bytes of slack each (the kernels read whole words past a sequence's end), 16-byte aligned so that half an
// offset -- the sequence's place among the exception bits -- is 8-byte aligned.  Returns the bytes of codes.
uint64_t packed_offsets(const uint64_t *h_off, uint32_t n, uint64_t *pk_off)
{
    pk_off[0] = 0;
    for (uint32_t g = 0; g < n; ++g)
        pk_off[g + 1] = (pk_off[g] + 8 * ((h_off[g + 1] - h_off[g] + 31) / 32) + 32 + 15) / 16 * 16;
    return pk_off[n];
}

// room for a batch of `code_bytes` of codes (+ half as many bytes of exception bits) in packed buffer `buf`
// (never the buffer of the batch in flight)
int ensure_packed(mk_ctx *c, int buf, uint64_t code_bytes)
{
    if (code_bytes <= c->pk_cap[buf]) return MK_OK;
    const uint64_t old_cap = c->pk_cap[buf];
    dev_free(c->d_pk[buf]);
    c->pk_cap[buf] = 0;
    const uint64_t cap = (std::max<uint64_t>(code_bytes, std::max(c->pk_cap[buf ^ 1], old_cap * 2)) + 255) / 256 * 256;
    MK_TRY(dev_alloc(&c->d_pk[buf], cap + cap / 2 + 256));
    c->pk_cap[buf] = cap;
    return MK_OK;
}

// genome_size / sketch_size of Miekki.cpp:303-311 from the device's exact sums
// (`single`: insert_sequence, Miekki.cpp:245-270, counts the active partitions in a double -- its square does not wrap)
static uint64_t estimate_genome_size(uint32_t active, uint64_t cardsum, uint64_t len, bool single)
{
    const double card = (double)cardsum / 2147483648.0;          // sum of 2^-exp, exact
    const uint32_t sq = active * active;                         // u32 wrap-around, Miekki.cpp:306
    const double est = single ? 0.72134 * ((double)active * (double)active) / card : 0.72134 * (double)sq / card;
    if (est > (double)len) return len;
    if (std::isnan(est)) return 0x8000000000000000ull;           // what the x86 conversion yields
    return (uint64_t)est;
}

// One batch goes through two stages.  Its sequences are already on their way to the device, in one of three forms:
//   kChars    characters in c->d_seq[buf] at offsets h_off[0..n] (packed by the front stage, pack_kernel)
//   kPacked   codes / exception bits in c->d_pk[buf] at pk_off, first characters in c->d_heads[buf], the
//             "has exceptions" flags in h_dirty (mk_index_append_packed)
//   kSynth    codes generated into c->d_pk[buf] on the front stream (no exceptions anywhere)
// FRONT (enqueue_front, on c->front_stream, behind `after` -- the copy): offsets, counters, packing, seed digits,
// the scatter kernel into side `buf`.  It is queued while the batch before it is still in flight: its kernels are
// bound by instruction issue, that batch's back stage by memory latency, and the two share the CUs.
// BACK (enqueue_back, on c->stream, once that batch has been settled): reduce, matrix rows, Bloom passes, the
// counters' copy back.  Nothing here waits for the device; settle_build folds the results in.
enum BatchForm { kChars, kPacked, kSynth };

static int enqueue_front(mk_ctx *c, const uint64_t *h_off, uint32_t n, int buf, BatchForm form, hipEvent_t after,
                         const uint32_t *h_dirty = nullptr)
{
    mk_ctx::BuildSide &sd = c->side[buf];
    mk_ctx::BuildInFlight &f = c->front;
    f.n = n; f.buf = buf; f.binned = false; f.single = c->next_single;
    f.have_chars = form == kChars; f.have_heads = form == kPacked;
    memcpy(f.off, h_off, (size_t)(n + 1) * 8);
    memcpy(c->h_img->off[buf], h_off, (size_t)(n + 1) * 8);
    hipStream_t fs = c->front_stream;
    if (after) MK_HIP(hipStreamWaitEvent(fs, after, 0));
    MK_HIP(hipMemcpyAsync(sd.d_seq_off, c->h_img->off[buf], (size_t)(n + 1) * 8, hipMemcpyHostToDevice, fs));
    MK_HIP(hipMemsetAsync(sd.d_counters, 0, sizeof *sd.d_counters, fs));      // overflow marks, exception flags, sums: one block
    uint8_t *codes = c->d_pk[buf], *except = c->d_pk[buf] + c->pk_cap[buf];
    if (form == kChars) {
        uint64_t *pk_off = c->h_img->pk_off[buf];
        MK_TRY(ensure_packed(c, buf, packed_offsets(f.off, n, pk_off)));
        codes = c->d_pk[buf]; except = c->d_pk[buf] + c->pk_cap[buf];
        MK_HIP(hipMemcpyAsync(c->d_pk_off[buf], pk_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, fs));
        MK_TRY(launch_pack(c, buf, c->d_seq[buf], f.off, n, codes, except, c->d_pk_off[buf]));
    } else if (form == kPacked && h_dirty) {
        MK_HIP(hipMemcpyAsync(sd.d_counters->dirty, h_dirty, (size_t)n * 4, hipMemcpyHostToDevice, fs));
    }
    // the k-1 seed digits of every sequence (str2numstrand / rcb), and the seeds' validity for the fallback
    // (generated sequences are plain ACGT: neither characters nor heads, nothing to rewrite)
    MK_TRY(launch_seed_fix(c, buf, form == kChars ? c->d_seq[buf] : nullptr, form == kPacked ? c->d_heads[buf] : nullptr, n, codes,
                           except, c->d_pk_off[buf]));
    {
        ScopedTimer t(c, 3, fs);
        MK_TRY(launch_build_front(c, buf, codes, except, c->d_pk_off[buf], f.off, n, &f.binned));
    }
    MK_HIP(hipEventRecord(sd.ev_front, fs));
    f.on = true;
    return MK_OK;
}

static int settle_one(mk_ctx *c, mk_ctx::BuildInFlight &b);

// The back stage of the batch whose front stage was queued last goes onto c->stream right behind the back stage of the
// batch before it -- the device starts it the moment that one ends, no host round trip in between -- and only then does
// the host wait for that older batch and fold it in (the caller: settle_older).
static int enqueue_back(mk_ctx *c)
{
    if (c->older.on) MK_TRY(settle_one(c, c->older));            // (at most two batches are not folded in at any time)
    MK_TRY(need_raw_cold(c));                                    // new columns are written into the rows as they are
    const uint32_t n = c->front.n;
    if (c->G_back + n > c->capG) {
        // the matrix has to grow: that re-lays it out with the columns of c->G genomes -- fold everything in first
        if (c->build.on) MK_TRY(settle_one(c, c->build));
        MK_TRY(ensure_capacity(c, c->G + n));
    }
    if (c->build.on) c->older = c->build;
    mk_ctx::BuildInFlight &b = c->build;
    b = c->front;                                                // the batch whose front stage was queued last
    c->front.on = false;
    const int buf = b.buf;
    b.g0 = c->G_back;
    mk_ctx::BuildSide &sd = c->side[buf];
    use_build_side(c, buf);                                      // the aliases follow the batch whose kernels are being queued
    MK_HIP(hipStreamWaitEvent(c->stream, sd.ev_front, 0));
    MK_TRY(ensure_bloom_summary(c));
    if (b.binned) {
        ScopedTimer t(c, 4);
        MK_TRY(launch_build_back(c, buf, c->d_pk[buf], c->d_pk[buf] + c->pk_cap[buf], c->d_pk_off[buf], n, b.g0));
    } else {
        MK_TRY(build_from_characters(c));                        // shapes the bins do not fit
    }
    // one copy back of the batch's counters (active counts, cardinality sums)
    MK_HIP(hipMemcpyAsync(sd.h_back, sd.d_counters, sizeof *sd.h_back, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipEventRecord(sd.ev_back, c->stream));
    c->G_back += n;
    b.on = true;
    return MK_OK;
}

// The batch in flight once more, through the atomic kernel + separate passes, which work from characters:
// for shapes the bins do not fit.
// A batch that arrived packed is turned back into characters that mean the same to those kernels.
static int build_from_characters(mk_ctx *c)
{
    mk_ctx::BuildInFlight &b = c->build;
    const uint32_t n = b.n;
    if (!b.have_chars) {
        MK_TRY(ensure_build_scratch(c, b.off[n], b.buf));
        MK_TRY(launch_unpack(c, b.buf, c->d_pk[b.buf], c->d_pk[b.buf] + c->pk_cap[b.buf], c->d_pk_off[b.buf],
                             b.have_heads ? c->d_heads[b.buf] : nullptr, b.off, n, c->d_seq[b.buf]));
        b.have_chars = true;
    }
    const char *d_seq = c->d_seq[b.buf];
    { ScopedTimer t(c, 3); MK_TRY(launch_genome_sketch(c, d_seq, c->d_seq_off, b.off, c->d_seed_valid, n, c->d_tables)); }
    ScopedTimer t(c, 4);
    MK_TRY(launch_finalize(c, c->d_tables, n, b.g0, nullptr));
    return launch_bloom_insert(c, c->d_tables, d_seq, c->d_seq_off, c->d_seed_valid, n, nullptr);
}

// Wait for one batch whose back stage is queued and fold it into the index (Miekki.cpp:303-311); oldest first.
static int settle_one(mk_ctx *c, mk_ctx::BuildInFlight &b)
{
    if (!b.on) return MK_OK;
    b.on = false;
    mk_ctx::BuildSide &sd = c->side[b.buf];
    MK_HIP(hipEventSynchronize(sd.ev_back));
    const uint32_t n = b.n;
    // (the binned build cannot overflow: every scatter workgroup owns the places of its own 4096 k-mers)
    // the sizes go to the device from a pinned block of their own (two, alternating: the copies are queued
    // behind the kernels already in the stream and not waited for)
    mk_ctx::SizeUpload &up = c->h_sizes[c->size_parity ^= 1];
    for (uint32_t g = 0; g < n; ++g) {
        const uint64_t len = b.off[g + 1] - b.off[g];
        up.ss[g] = sd.h_back->act[g];
        up.gs[g] = estimate_genome_size(up.ss[g], sd.h_back->card[g], len, b.single);
        c->h_sketch_size.push_back(up.ss[g]);
        c->h_genome_size.push_back(up.gs[g]);
        if (up.ss[g] == 0) c->has_empty_sketch = true;
        c->stats.build_kmers += len > c->p.k ? len - c->p.k : 0;
    }
    MK_HIP(hipMemcpyAsync(c->d_sketch_size + b.g0, up.ss, n * 4, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipMemcpyAsync(c->d_genome_size + b.g0, up.gs, n * 8, hipMemcpyHostToDevice, c->stream));
    c->G += n;                                                   // (== b.g0 + n: batches are folded in in order)
    ++c->gen;
    c->stats.build_genomes += n;
    return MK_OK;
}

// after a back stage has been queued: the batch before it
static int settle_older(mk_ctx *c) { return settle_one(c, c->older); }

// everything in flight (every entry point other than the appends, and mk_sync)
static int settle_build(mk_ctx *c)
{
    MK_TRY(settle_one(c, c->older));
    MK_TRY(settle_one(c, c->build));
    MK_HIP(hipStreamSynchronize(c->stream));                     // (the sizes' copies included)
    return MK_OK;
}

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
static void qset_release(mk_qset *qs)
{
    if (!qs) return;
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
            n8 += nk > kShortMax && nk >= c->P / 8 ? 1 : 0;
        }
        if (n8 >= 16 && n8 * c->P * c->W * 3 <= (8ull << 30)) dense_div = 8;       // (vector: P W bytes per query; tables: 2 P W per query)
    }
    for (uint32_t q = 0; q < nq; ++q) {
        const uint64_t nk = lens[q] > c->p.k ? lens[q] - c->p.k : 0;
        if (lens[q] >= (1ull << 40)) { set_error("query too long"); return MK_ERR_ARG; }
        qs->h_off[q + 1] = qs->h_off[q] + lens[q];
        const bool dense = nk > kShortMax && nk >= c->P / dense_div;
        qs->h_ent_off[q + 1] = qs->h_ent_off[q] + (dense ? 0 : std::min<uint64_t>(nk, c->P));
        if (dense) qs->dense_q.push_back(q);
        else if (nk > kShortMax) qs->long_q.push_back(q);
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
                   o_lut = carve((uint64_t)((qs->dense_q.size() / 4 + 1) / 2) * c->P * 16 * c->W),
                   o_split = carve(qs->split_room ? (uint64_t)nq * (qs->split_room + 1) * 4 : 0);
    if (transient && !c->qarena_busy) {
        if (at > c->qarena_cap) {
            MK_HIP(hipStreamSynchronize(c->stream));
            dev_free(c->d_qarena);
    for (auto &blk : c->gz_blocks) (void)hipFree(blk.first);
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
        d.rows_per_item = std::min<uint32_t>(row_hi - row_lo, 8176);          // (below 2^13: the table kernel counts in 13 bit planes)
        d.nchunks = (row_hi - row_lo + d.rows_per_item - 1) / d.rows_per_item;
        d.ngroups = (uint32_t)(qs->dense_q.size() / 4);
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

extern "C" {

const char *mk_last_error(void) { return g_err; }
uint32_t mk_abi_version(void) { return MK_ABI_VERSION; }

int mk_create(const mk_params *p, mk_ctx **out)
{
    if (!p || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    if (p->fp_bits != 8 && p->fp_bits != 16) { set_error("not implemented"); return MK_ERR_UNSUPPORTED; }
    if (p->k < 2 || p->k > 31) { set_error("k must be in [2,31]"); return MK_ERR_ARG; }
    if (p->h < 1 || p->h > 28) { set_error("h must be in [1,28]"); return MK_ERR_ARG; }
    if (p->bloom_log2 != 0 && (p->bloom_log2 < 32 || p->bloom_log2 > 40)) {
        set_error("bloom_log2 must be 0 (no Bloom gate) or in [32,40]");
        return MK_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device: libmiekki_hip has no CPU path");
        return MK_ERR_DEVICE;
    }
    if (p->device < 0 || p->device >= ndev) { set_error("device %d out of range", p->device); return MK_ERR_ARG; }
    std::unique_ptr<mk_ctx> c(new mk_ctx());
    c->p = *p;
    c->P = 1u << p->h; c->W = p->fp_bits / 8; c->f = p->fp_bits - kMantisBits;
    c->empty = p->fp_bits == 8 ? 255u : 65535u;
    c->d_M = nullptr; c->ld = 0; c->capG = 0; c->G = 0; c->d_sketch_size = nullptr; c->d_genome_size = nullptr;
    c->d_ratio = nullptr; c->ratio_gen = 0; c->ratio_cap = 0;
    c->d_colstage = nullptr; c->colstage_cap = 0;
    c->h_Z = nullptr; c->z_bytes = 0; c->d_zoff = nullptr; c->d_zstage[0] = c->d_zstage[1] = nullptr; c->zstage_cap = 0;
    c->h_M = nullptr; c->P_hot = c->P; c->d_cold_stage = nullptr; c->cold_stage_rows = 0; c->hbm_matrix_budget = 0;
    for (int i = 0; i < 5; ++i) c->ev_cold[i] = nullptr;
    if (const char *e = getenv("MIEKKI_HBM_MATRIX_MIB")) { const long v = atol(e); if (v > 0) c->hbm_matrix_budget = (uint64_t)v << 20; }
    c->d_bloom = nullptr; c->d_bloom_order = nullptr; c->build_batch = 0; c->d_tables = nullptr;
    c->d_active = nullptr; c->d_cardsum = nullptr; c->d_seed_valid = nullptr; c->d_counters = nullptr; c->h_sizes = nullptr; c->size_parity = 0;
    c->d_seq[0] = c->d_seq[1] = nullptr; c->seq_cap[0] = c->seq_cap[1] = 0; c->seq_cur = 1;
    for (int b = 0; b < 2; ++b) { c->d_pk[b] = nullptr; c->pk_cap[b] = 0; c->d_pk_off[b] = nullptr; c->d_heads[b] = nullptr; }
    c->copy_stream = nullptr; c->ev_copy = nullptr; c->h_back = nullptr; c->front_stream = nullptr; c->n_copy_extra = 0;
    memset(c->side, 0, sizeof c->side);
    c->h_img = nullptr;
    memset(&c->front, 0, sizeof c->front);
    c->d_dirty = nullptr; c->d_bloom_full = nullptr; c->d_bloom_full2 = nullptr; c->bloom_full_stale = true;
    memset(&c->build, 0, sizeof c->build);
    memset(&c->older, 0, sizeof c->older);
    c->G_back = 0;
    c->d_seq_off = nullptr; c->d_scores = nullptr; c->scores_cap = 0; c->d_count = nullptr; c->d_cand = nullptr;
    c->d_partials = nullptr; c->partials_cap = 0; c->d_flag = nullptr;
    c->d_hits = nullptr; c->d_nhits = nullptr; c->hits_cap = 0; c->nhits_cap = 0;
    c->has_empty_sketch = false;
    c->d_all_ss = nullptr; c->d_all_gs = nullptr; c->all_n = 0; c->all_base = 0; c->gen = 1;
    for (int i = 0; i < 10; ++i) { c->exact_buf[i] = nullptr; c->exact_cap[i] = 0; }
    c->exact_have_B = false; c->exact_nB = 0; c->exact_log2B = 0;
    c->d_qarena = nullptr; c->qarena_cap = 0; c->qarena_busy = false;
    c->h_stage = nullptr; c->stage_cap = 0; c->h_res = nullptr; c->res_cap = 0;
    c->cand_cap_q = 0; c->d_long_table = nullptr; c->d_ovf = nullptr; c->d_ovf_count = nullptr;
    c->d_fpT = nullptr; c->d_bloom_touched = nullptr;
    memset(&c->stats, 0, sizeof c->stats);
    MK_HIP(hipSetDevice(p->device));
    MK_HIP(hipStreamCreate(&c->stream));
    MK_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    MK_HIP(hipStreamCreateWithFlags(&c->front_stream, hipStreamNonBlocking));
    MK_HIP(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    {
        c->n_copy_extra = kCopyStreams - 1;                           // all copy streams of a packed append, this one included
        for (int i = 0; i < c->n_copy_extra; ++i) {
            MK_HIP(hipStreamCreateWithFlags(&c->copy_extra[i], hipStreamNonBlocking));
            MK_HIP(hipEventCreateWithFlags(&c->ev_extra[i], hipEventDisableTiming));
        }
    }
    c->bloom_bytes = p->bloom_log2 ? (1ull << p->bloom_log2) / 8 : 0;
    c->bloom_dev_bytes = 0;
    if (p->bloom_log2) {
        // largest reachable cell: ((4^k - 1) + 1023) >> (b + 3)
        const uint64_t top = ((1ull << (2 * p->k)) - 1) + 1023;
        c->bloom_dev_bytes = std::min<uint64_t>(c->bloom_bytes, (top >> (p->bloom_log2 + 3)) + 1);
        mk_ctx *cc = c.get();
        MK_TRY(dev_alloc(&cc->d_bloom, c->bloom_dev_bytes));
        MK_HIP(hipMemsetAsync(c->d_bloom, 0, c->bloom_dev_bytes, c->stream));
        MK_HIP(hipStreamSynchronize(c->stream));
    }
    *out = c.release();
    return MK_OK;
}

void mk_destroy(mk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->front_stream) (void)hipStreamSynchronize(c->front_stream);
    (void)hipStreamSynchronize(c->stream);
    (void)drain_timers(c);
    for (Timer &t : c->free_timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    dev_free(c->d_M); dev_free(c->d_sketch_size); dev_free(c->d_genome_size); dev_free(c->d_bloom); dev_free(c->d_ratio); dev_free(c->d_colstage); dev_free(c->d_huff);
    if (c->h_M) (void)hipHostFree(c->h_M);
    if (c->h_Z) (void)hipHostFree(c->h_Z);
    dev_free(c->d_zoff); dev_free(c->d_zstage[0]); dev_free(c->d_zstage[1]);
    dev_free(c->d_cold_stage);
    for (int i = 0; i < 5; ++i) if (c->ev_cold[i]) (void)hipEventDestroy(c->ev_cold[i]);
    dev_free(c->d_hits); dev_free(c->d_nhits);
    for (int i = 0; i < 10; ++i) if (c->exact_buf[i]) (void)hipFree(c->exact_buf[i]);
    for (int b = 0; b < 2; ++b) {                                  // (d_counters, h_back, d_seq_off, d_seed_valid, d_ovf alias one of these)
        mk_ctx::BuildSide &sd = c->side[b];
        dev_free(sd.d_counters); dev_free(sd.d_seq_off); dev_free(sd.d_seed_valid); dev_free(sd.d_ovf);
        if (sd.d_slots) (void)hipFree(sd.d_slots);
        if (sd.h_back) (void)hipHostFree(sd.h_back);
        if (sd.ev_front) (void)hipEventDestroy(sd.ev_front);
        if (sd.ev_back) (void)hipEventDestroy(sd.ev_back);
    }
    dev_free(c->d_bloom_full); dev_free(c->d_bloom_full2);
    dev_free(c->d_bloom_order); dev_free(c->d_tables);
    for (int b = 0; b < 2; ++b) { dev_free(c->d_pk[b]); dev_free(c->d_pk_off[b]); dev_free(c->d_heads[b]); }
    dev_free(c->d_seq[0]); dev_free(c->d_seq[1]); dev_free(c->d_scores);
    dev_free(c->d_count); dev_free(c->d_cand); dev_free(c->d_long_table); dev_free(c->d_partials);
    dev_free(c->d_flag); dev_free(c->d_all_ss); dev_free(c->d_all_gs); dev_free(c->d_fpT); dev_free(c->d_bloom_touched);
    dev_free(c->d_qarena);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_sizes) (void)hipHostFree(c->h_sizes);
    if (c->h_img) (void)hipHostFree(c->h_img);
    for (int i = 0; i < c->n_copy_extra; ++i) { (void)hipEventDestroy(c->ev_extra[i]); (void)hipStreamDestroy(c->copy_extra[i]); }
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->front_stream) (void)hipStreamDestroy(c->front_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int mk_reserve(mk_ctx *c, uint32_t n)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    return ensure_capacity(c, n);
}

uint32_t mk_index_size(const mk_ctx *c)
{
    if (!c) return 0;
    if (c->build.on || c->older.on) (void)use_device(c);         // count the batches still in flight
    return c->G;
}

int mk_get_params(const mk_ctx *c, mk_params *out)
{
    if (!c || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = c->p;
    return MK_OK;
}

int mk_sync(mk_ctx *c)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    return drain_timers(c);
}

int mk_get_stats(const mk_ctx *cc, mk_stats *out)
{
    if (!cc || !out) { set_error("null argument"); return MK_ERR_ARG; }
    mk_ctx *c = const_cast<mk_ctx *>(cc);
    MK_TRY(mk_sync(c));
    *out = c->stats;
    return MK_OK;
}

int mk_probe_stream_read(mk_ctx *c, uint32_t rounds, double *gbps, uint64_t *bytes)
{
    if (!c || !gbps || !bytes) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    return probe_stream_read(c, rounds ? rounds : 3, gbps, bytes);
}

int mk_probe_synth_genomes(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t length, char *dst)
{
    if (!c || (n && !dst)) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    char *d = nullptr;
    MK_TRY(dev_alloc(&d, (uint64_t)n * length + 64));
    int rc = launch_synth_genomes(c, first_id, n, length, d);
    if (rc == MK_OK && hipMemcpyAsync(dst, d, (uint64_t)n * length, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_ERR_DEVICE) set_error("synthetic genome download failed");
    dev_free(d);
    return rc;
}

int mk_reset_stats(mk_ctx *c)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(mk_sync(c));
    memset(&c->stats, 0, sizeof c->stats);
    return MK_OK;
}

// ------------------------------------------------------------------ build
int mk_index_append(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t n)
{
    if (!c || (n && (!seqs || !lens))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    for (uint32_t g = 0; g < n; ++g)
        if (lens[g] < c->p.k) { set_error("sequence %u shorter than k", g); return MK_ERR_ARG; }
    MK_TRY(ensure_build_scratch(c, 0, c->seq_cur ^ 1));
    for (uint32_t g0 = 0; g0 < n;) {
        // a batch = up to build_batch genomes and (beyond the first) at most 2 GiB of sequence
        uint64_t off[kBuildBatch + 1];
        off[0] = 0;
        uint32_t nb = 0;
        while (nb < c->build_batch && g0 + nb < n && (nb == 0 || off[nb] + lens[g0 + nb] <= (2ull << 30))) {
            off[nb + 1] = off[nb] + lens[g0 + nb];
            ++nb;
        }
        // The copy goes into the buffer the batch in flight does NOT use, on its own stream,
        // so that it overlaps with that batch's kernels; then that batch is settled and this
        // one's kernels are queued behind the copy.  The call returns once the copy has
        // finished -- the caller's buffers are free again -- not when the kernels have.
        const int buf = c->seq_cur ^ 1;
        MK_TRY(ensure_build_scratch(c, off[nb], buf));
        for (uint32_t g = 0; g < nb; ++g)
            MK_HIP(hipMemcpyAsync(c->d_seq[buf] + off[g], seqs[g0 + g], lens[g0 + g], hipMemcpyHostToDevice, c->copy_stream));
        MK_HIP(hipEventRecord(c->ev_copy, c->copy_stream));
        MK_TRY(enqueue_front(c, off, nb, buf, kChars, c->ev_copy));
        MK_TRY(enqueue_back(c));
        MK_TRY(settle_older(c));
        c->seq_cur = buf;
        MK_HIP(hipEventSynchronize(c->ev_copy));
        g0 += nb;
    }
    return MK_OK;
}

// insert_sequences for files of an inflated batch (mk_gz_unpack): their sequences are stripped out of the batch's text
// straight into the build's sequence buffer -- no copy, nothing crosses PCIe -- and take the same kernels as characters.
int mk_index_append_gz(mk_ctx *c, const mk_gz_batch *batch, const uint32_t *which, uint32_t n)
{
    if (!c || !batch || (n && !which)) { set_error("null argument"); return MK_ERR_ARG; }
    if (gz_batch_owner(batch) != c) { set_error("the batch belongs to another context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    for (uint32_t g = 0; g < n; ++g) {
        if (which[g] >= gz_batch_size(batch) || !gz_batch_ok(batch, which[g])) { set_error("file %u of the batch has no sequence", which[g]); return MK_ERR_ARG; }
        if (gz_batch_len(batch, which[g]) < c->p.k) { set_error("sequence %u shorter than k", g); return MK_ERR_ARG; }
    }
    MK_TRY(ensure_build_scratch(c, 0, c->seq_cur ^ 1));
    for (uint32_t g0 = 0; g0 < n;) {
        uint64_t off[kBuildBatch + 1];
        off[0] = 0;
        uint32_t nb = 0;
        while (nb < c->build_batch && g0 + nb < n && (nb == 0 || off[nb] + gz_batch_len(batch, which[g0 + nb]) <= (2ull << 30))) {
            off[nb + 1] = off[nb] + gz_batch_len(batch, which[g0 + nb]);
            ++nb;
        }
        const int buf = c->seq_cur ^ 1;
        MK_TRY(ensure_build_scratch(c, off[nb], buf));
        MK_TRY(gz_batch_strip(c, batch, which + g0, nb, reinterpret_cast<uint8_t *>(c->d_seq[buf]), off, c->copy_stream));
        MK_HIP(hipEventRecord(c->ev_copy, c->copy_stream));
        MK_TRY(enqueue_front(c, off, nb, buf, kChars, c->ev_copy));
        MK_TRY(enqueue_back(c));
        MK_TRY(settle_older(c));
        c->seq_cur = buf;
        g0 += nb;
    }
    return MK_OK;
}

// insert_sequence (Miekki.cpp:243-273), the one-genome form behind index_file (518-536; the reference's CLI never calls
// either): sketch, column and Bloom inserts are insert_sequences'; only the size estimate differs -- it keeps the count of
// active partitions in a double, so its square does not wrap at 2^32 (more than 65,535 active partitions: -h 17 and up).
int mk_index_insert_sequence(mk_ctx *c, const char *seq, uint64_t len)
{
    if (!c || !seq) { set_error("null argument"); return MK_ERR_ARG; }
    c->next_single = true;
    const int rc = mk_index_append(c, &seq, &len, 1);
    c->next_single = false;
    return rc;
}

// insert_sequences for sequences that arrive packed: a quarter of the bytes cross
// PCIe -- 2 bits per base, plus 1 where a sequence has characters other than A, C, G, T -- and the
// device skips its own packing pass.  Pipelined exactly like mk_index_append.
int mk_index_append_packed(mk_ctx *c, const mk_packed_seq *seqs, uint32_t n)
{
    if (!c || (n && !seqs)) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    for (uint32_t g = 0; g < n; ++g) {
        if (seqs[g].len < c->p.k) { set_error("sequence %u shorter than k", g); return MK_ERR_ARG; }
        if (!seqs[g].codes) { set_error("sequence %u has no codes", g); return MK_ERR_ARG; }
    }
    MK_TRY(ensure_build_scratch(c, 0, c->seq_cur ^ 1));
    for (uint32_t g0 = 0; g0 < n;) {
        uint64_t off[kBuildBatch + 1];
        off[0] = 0;
        uint32_t nb = 0;
        while (nb < c->build_batch && g0 + nb < n && (nb == 0 || off[nb] + seqs[g0 + nb].len <= (2ull << 30))) {
            off[nb + 1] = off[nb] + seqs[g0 + nb].len;
            ++nb;
        }
        const int buf = c->seq_cur ^ 1;
        uint64_t *pk_off = c->h_img->pk_off[buf];
        uint32_t *dirty = c->h_img->dirty[buf];
        char *heads = c->h_img->heads[buf];
        MK_TRY(ensure_packed(c, buf, packed_offsets(off, nb, pk_off)));
        uint8_t *codes = c->d_pk[buf], *except = c->d_pk[buf] + c->pk_cap[buf];
        // small per-batch arrays first, then one copy per array of codes / exception bits -- a DMA each when the
        // caller's buffers are page-locked
        for (uint32_t g = 0; g < nb; ++g) memcpy(heads + 32 * g, seqs[g0 + g].head, 32);
        MK_HIP(hipMemcpyAsync(c->d_heads[buf], heads, (size_t)nb * 32, hipMemcpyHostToDevice, c->copy_stream));
        MK_HIP(hipMemcpyAsync(c->d_pk_off[buf], pk_off, (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, c->copy_stream));
        const int ways = 1 + c->n_copy_extra;                        // the sequences take turns on the copy streams
        for (uint32_t g = 0; g < nb; ++g) {
            const mk_packed_seq &q = seqs[g0 + g];
            hipStream_t cs = g % ways ? c->copy_extra[g % ways - 1] : c->copy_stream;
            MK_HIP(hipMemcpyAsync(codes + pk_off[g], q.codes, (size_t)((q.len + 31) / 32) * 8, hipMemcpyHostToDevice, cs));
            dirty[g] = q.except ? 1u : 0u;
            if (q.except)
                MK_HIP(hipMemcpyAsync(except + pk_off[g] / 2, q.except, (size_t)((q.len + 63) / 64) * 8, hipMemcpyHostToDevice, cs));
        }
        for (int i = 0; i < c->n_copy_extra; ++i) {                  // one event stands for all of them
            MK_HIP(hipEventRecord(c->ev_extra[i], c->copy_extra[i]));
            MK_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_extra[i], 0));
        }
        MK_HIP(hipEventRecord(c->ev_copy, c->copy_stream));
        MK_TRY(enqueue_front(c, off, nb, buf, kPacked, c->ev_copy, dirty));
        MK_TRY(enqueue_back(c));
        MK_TRY(settle_older(c));
        c->seq_cur = buf;
        MK_HIP(hipEventSynchronize(c->ev_copy));
        g0 += nb;
    }
    return MK_OK;
}

int mk_host_alloc(mk_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    MK_TRY(use_device(c, false));
    MK_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return MK_OK;
}

void mk_host_free(mk_ctx *c, void *p)
{
    if (!c || !p) return;
    (void)hipSetDevice(c->p.device);
    (void)hipHostFree(p);
}

static int append_synthetic(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t length, uint32_t strains, uint32_t rate_ppm);
int mk_index_append_synthetic(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t length) { return append_synthetic(c, first_id, n, length, 0, 0); }
int mk_index_append_synthetic_strains(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t length, uint32_t strains, uint32_t rate_ppm)
{
    if (!strains || rate_ppm > 30000) { set_error("strains per species must be positive, the rate at most 30,000 ppm"); return MK_ERR_ARG; }
    return append_synthetic(c, first_id, n, length, strains, rate_ppm);
}
static int append_synthetic(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t length, uint32_t strains, uint32_t rate_ppm)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    if (length < c->p.k) { set_error("sequence shorter than k"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    MK_TRY(ensure_build_scratch(c, 0, c->seq_cur ^ 1));
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c->build_batch, (2ull << 30) / length));
    for (uint32_t g0 = 0; g0 < n; g0 += per) {
        const uint32_t nb = std::min(per, n - g0);
        uint64_t off[kBuildBatch + 1];
        for (uint32_t g = 0; g <= nb; ++g) off[g] = (uint64_t)g * length;
        const int buf = c->seq_cur ^ 1;
        uint64_t *pk_off = c->h_img->pk_off[buf];
        MK_TRY(ensure_packed(c, buf, packed_offsets(off, nb, pk_off)));
        // the generator fills the buffer the batch in flight does NOT read -- in packed form, which is what the
        // build works from -- on the front stream, followed by this batch's front stage: the device is never idle
        // while the host settles the batch before
        MK_HIP(hipMemcpyAsync(c->d_pk_off[buf], pk_off, (size_t)(nb + 1) * 8, hipMemcpyHostToDevice, c->front_stream));
        MK_TRY(launch_synth_packed(c, first_id + g0, nb, length, c->d_pk[buf], c->d_pk_off[buf], strains, rate_ppm));
        MK_TRY(enqueue_front(c, off, nb, buf, kSynth, nullptr));
        MK_TRY(enqueue_back(c));
        MK_TRY(settle_older(c));
        c->seq_cur = buf;
    }
    return MK_OK;
}

// ------------------------------------------------------------------ persistence
static int staged_columns(mk_ctx *c, bool to_device, uint32_t pb, uint32_t pe, uint8_t *host)
{
    if (pb > pe || pe > c->P) { set_error("partition range out of bounds"); return MK_ERR_ARG; }
    if (pb == pe || c->G == 0) return MK_OK;
    MK_TRY(need_raw_cold(c));                                    // (dump_disk decompresses first too, Miekki.cpp:662-664)
    const uint64_t row = (uint64_t)c->G * c->W;
    const uint32_t rows_per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(pe - pb, (256ull << 20) / row));
    // (the staging buffer stays with the context: a dump or a load calls this a few thousand times)
    if ((uint64_t)rows_per * row > c->colstage_cap) {
        dev_free(c->d_colstage);
        c->colstage_cap = 0;
        MK_TRY(dev_alloc(&c->d_colstage, (uint64_t)rows_per * row));
        c->colstage_cap = (uint64_t)rows_per * row;
    }
    uint8_t *d_stage = c->d_colstage;
    int rc = MK_OK;
    for (uint32_t p = pb; p < pe && rc == MK_OK; p += rows_per) {
        const uint32_t r = std::min(rows_per, pe - p);
        uint8_t *h = host + (uint64_t)(p - pb) * row;
        if (to_device) {
            if (hipMemcpyAsync(d_stage, h, (size_t)r * row, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
            if (rc == MK_OK) rc = launch_convert_columns(c, true, p, p + r, d_stage);
        } else {
            rc = launch_convert_columns(c, false, p, p + r, d_stage);
            if (rc == MK_OK && hipMemcpyAsync(h, d_stage, (size_t)r * row, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
        }
        if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    }
    if (rc == MK_ERR_DEVICE) set_error("column transfer failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

int mk_index_export_columns(mk_ctx *c, uint32_t pb, uint32_t pe, uint8_t *dst)
{
    if (!c || !dst) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    return staged_columns(c, false, pb, pe, dst);
}

int mk_index_export_genomes(mk_ctx *c, const uint32_t *ids, uint32_t n, uint8_t *dst)
{
    if (!c || (n && (!ids || !dst))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    for (uint32_t j = 0; j < n; ++j)
        if (ids[j] < c->p.genome_id_base || ids[j] - c->p.genome_id_base >= c->G) { set_error("genome id %u is not in this index", ids[j]); return MK_ERR_ARG; }
    if (!n) return MK_OK;
    MK_TRY(need_raw_cold(c));
    // in pieces of at most 64 genomes: 2^h x 64 x W bytes of staging (128 MiB at -h 20, 2-byte fingerprints)
    uint32_t *d_ids = nullptr;
    uint8_t *d_stage = nullptr;
    const uint32_t per = std::min<uint32_t>(n, 64);
    const uint64_t col = (uint64_t)c->P * c->W;
    std::vector<uint32_t> local(per);
    std::vector<uint8_t> piece;
    int rc = dev_alloc(&d_ids, per);
    if (rc == MK_OK) rc = dev_alloc(&d_stage, col * per);
    for (uint32_t j0 = 0; j0 < n && rc == MK_OK; j0 += per) {
        const uint32_t m = std::min(per, n - j0);
        for (uint32_t j = 0; j < m; ++j) local[j] = ids[j0 + j] - c->p.genome_id_base;
        if (hipMemcpyAsync(d_ids, local.data(), (size_t)m * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
        if (rc == MK_OK) rc = launch_export_genomes(c, d_ids, m, d_stage);
        if (rc != MK_OK) break;
        if (m == n) {                                                // one piece: it IS the result
            if (hipMemcpyAsync(dst, d_stage, col * m, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
            if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
        } else {                                                     // dst[p][j0 .. j0 + m) of rows n genomes wide
            piece.resize(col * m);
            if (hipMemcpyAsync(piece.data(), d_stage, col * m, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
            if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
            if (rc == MK_OK)
                for (uint32_t p = 0; p < c->P; ++p)
                    memcpy(dst + ((uint64_t)p * n + j0) * c->W, piece.data() + (uint64_t)p * m * c->W, (size_t)m * c->W);
        }
    }
    if (rc == MK_ERR_DEVICE) set_error("genome column export failed: %s", hipGetErrorString(hipGetLastError()));
    dev_free(d_ids); dev_free(d_stage);
    return rc;
}

int mk_index_compress(mk_ctx *c, uint64_t *raw_bytes, uint64_t *packed_bytes)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_HIP(hipStreamSynchronize(c->copy_stream));
    if (c->h_Z) {                                                    // packed already: say what it came to
        if (raw_bytes) *raw_bytes = (uint64_t)(c->P - c->P_hot) * c->ld;
        if (packed_bytes) *packed_bytes = c->z_bytes;
        return MK_OK;
    }
    return pack_cold(c, raw_bytes, packed_bytes);
}

int mk_index_decompress(mk_ctx *c)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    return need_raw_cold(c);
}

int mk_index_export_sizes(mk_ctx *c, uint64_t *genome_size, uint32_t *sketch_size)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));                                       // counts the batch still in flight
    if (genome_size) memcpy(genome_size, c->h_genome_size.data(), (size_t)c->G * 8);
    if (sketch_size) memcpy(sketch_size, c->h_sketch_size.data(), (size_t)c->G * 4);
    return MK_OK;
}

int mk_index_export_bloom(mk_ctx *c, uint64_t begin, uint64_t end, uint8_t *dst)
{
    if (!c || !dst) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpy(dst, c->d_bloom + begin, dev_end - begin, hipMemcpyDeviceToHost));
    const uint64_t zfrom = std::max(begin, dev_end);
    if (zfrom < end) memset(dst + (zfrom - begin), 0, end - zfrom);   // cells no k-mer can reach
    return MK_OK;
}

int mk_index_import_begin(mk_ctx *c, uint32_t n)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_TRY(need_raw_cold(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    c->G = 0;
    c->h_sketch_size.clear(); c->h_genome_size.clear();
    MK_TRY(ensure_capacity(c, n));
    c->G = n;
    c->G_back = n;
    c->h_sketch_size.assign(n, 0); c->h_genome_size.assign(n, 0);
    c->has_empty_sketch = false;
    ++c->gen;
    if (c->d_bloom) MK_HIP(hipMemset(c->d_bloom, 0, c->bloom_dev_bytes));
    MK_TRY(forget_bloom_summary(c));
    return MK_OK;
}

int mk_index_import_columns(mk_ctx *c, uint32_t pb, uint32_t pe, const uint8_t *src)
{
    if (!c || !src) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    ++c->gen;
    return staged_columns(c, true, pb, pe, const_cast<uint8_t *>(src));
}

// the same for rows that arrive Huffman-coded: inflated on the device (huff.hip)
int mk_index_import_columns_huffman(mk_ctx *c, uint32_t pb, uint32_t pe, const uint8_t *payload, uint64_t payload_bytes,
                                               const mk_huff_block *blocks, uint32_t n_blocks, const uint8_t *lens, uint32_t n_codes,
                                               uint32_t *crc_out, uint32_t *bad_out)
{
    if (!c || !payload || !blocks || !lens || !crc_out || !bad_out) { set_error("null argument"); return MK_ERR_ARG; }
    if (pb > pe || pe > c->P) { set_error("partition range out of bounds"); return MK_ERR_ARG; }
    if (n_blocks % 64u) { set_error("blocks come in groups of 64 that share a code"); return MK_ERR_ARG; }
    *bad_out = 0;
    if (pb == pe || c->G == 0 || !n_blocks) return MK_OK;
    MK_TRY(use_device(c));
    MK_TRY(need_raw_cold(c));
    ++c->gen;
    const uint64_t row = (uint64_t)c->G * c->W, out_bytes = (uint64_t)(pe - pb) * row;
    if (out_bytes > (2ull << 30)) { set_error("at most 2 GiB of rows per call"); return MK_ERR_ARG; }
    if (out_bytes > c->colstage_cap) {
        dev_free(c->d_colstage);
        c->colstage_cap = 0;
        MK_TRY(dev_alloc(&c->d_colstage, out_bytes));
        c->colstage_cap = out_bytes;
    }
    // the coded bytes, the block list, the codes' lengths, the remainders and the count: one device buffer, kept
    const uint64_t o_blocks = (payload_bytes + 8 + 255) / 256 * 256, o_lens = o_blocks + ((uint64_t)n_blocks * sizeof(mk_huff_block) + 255) / 256 * 256;
    const uint64_t o_crc = o_lens + ((uint64_t)n_codes * 257u + 255) / 256 * 256, o_bad = o_crc + ((uint64_t)n_blocks * 4 + 255) / 256 * 256;
    const uint64_t need = o_bad + 256;
    if (need > c->huff_cap) {
        dev_free(c->d_huff);
        c->huff_cap = 0;
        MK_TRY(dev_alloc(&c->d_huff, need + need / 4));
        c->huff_cap = need + need / 4;
    }
    uint8_t *d = c->d_huff;
    hipStream_t st = c->stream;
    bool ok = hipMemcpyAsync(d, payload, payload_bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemsetAsync(d + payload_bytes, 0, 8, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(d + o_blocks, blocks, (size_t)n_blocks * sizeof(mk_huff_block), hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemcpyAsync(d + o_lens, lens, (size_t)n_codes * 257u, hipMemcpyHostToDevice, st) == hipSuccess;
    ok = ok && hipMemsetAsync(d + o_bad, 0, 4, st) == hipSuccess;
    if (!ok) { set_error("Huffman column upload failed: %s", hipGetErrorString(hipGetLastError())); return MK_ERR_DEVICE; }
    MK_TRY(launch_huff_decode(c, d, payload_bytes, reinterpret_cast<const mk_huff_block *>(d + o_blocks), n_blocks, d + o_lens, n_codes,
                              c->d_colstage, out_bytes, reinterpret_cast<uint32_t *>(d + o_crc), reinterpret_cast<uint32_t *>(d + o_bad)));
    MK_HIP(hipMemcpyAsync(crc_out, d + o_crc, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, st));
    MK_HIP(hipMemcpyAsync(bad_out, d + o_bad, 4, hipMemcpyDeviceToHost, st));
    // (the rows are laid out even when a block was bad: the caller fails the load on `bad_out` / the CRCs and drops the index)
    MK_TRY(launch_convert_columns(c, true, pb, pe, c->d_colstage));
    MK_HIP(hipStreamSynchronize(st));
    return MK_OK;
}

int mk_index_import_sizes(mk_ctx *c, const uint64_t *genome_size, const uint32_t *sketch_size)
{
    if (!c || !genome_size || !sketch_size) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    c->h_genome_size.assign(genome_size, genome_size + c->G);
    c->h_sketch_size.assign(sketch_size, sketch_size + c->G);
    c->has_empty_sketch = false;
    for (uint32_t g = 0; g < c->G; ++g)
        if (sketch_size[g] == 0) c->has_empty_sketch = true;
    if (c->G) {
        MK_HIP(hipMemcpy(c->d_genome_size, genome_size, (size_t)c->G * 8, hipMemcpyHostToDevice));
        MK_HIP(hipMemcpy(c->d_sketch_size, sketch_size, (size_t)c->G * 4, hipMemcpyHostToDevice));
    }
    return MK_OK;
}

int mk_index_import_bloom(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *src)
{
    if (!c || !src) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpy(c->d_bloom + begin, src, dev_end - begin, hipMemcpyHostToDevice));
    MK_TRY(forget_bloom_summary(c));
    ++c->gen;
    return MK_OK;
}

// ------------------------------------------------------------------ queries
}  // extern "C"

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
    return qset_upload(c, seqs, lens, nq, out, false);
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
    MK_TRY(qset_sketch(c, qs));
    return qset_scan(c, qs, q0, q1, d_scores, score_layout_rows(c->W, c->G, c->G));   // dense rows for the caller
}

int mk_qset_active(mk_ctx *c, mk_qset *qs, uint32_t *active)
{
    if (!c || !qs || !active) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
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

// Miekki::filter_results (Miekki.cpp:376-397) on pre-thresholded candidates in
// ascending genome order; same libstdc++ heap calls as the reference.
uint32_t mk_filter_candidates(const mk_hit *cand, uint32_t ncand, uint32_t nresults, mk_hit *out)
{
    const auto compare = [](const mk_hit &a, const mk_hit &b) { return a.intersection > b.intersection; };
    std::vector<mk_hit> heap;
    heap.reserve((size_t)nresults + 1);
    for (uint32_t i = 0; i < ncand; ++i) {
        if (heap.size() >= nresults) {
            if (heap.empty()) continue;
            if (heap.front().intersection > cand[i].intersection) continue;   // ties replace
            std::pop_heap(heap.begin(), heap.end(), compare);
            heap.pop_back();
        }
        heap.push_back(cand[i]);
        std::push_heap(heap.begin(), heap.end(), compare);
    }
    std::sort_heap(heap.begin(), heap.end(), compare);
    std::copy(heap.begin(), heap.end(), out);
    return (uint32_t)heap.size();
}

int mk_merge_entrants(mk_ctx *c, const uint32_t *d_count, const mk_hit *d_cand, uint32_t world, uint32_t nq,
                      uint32_t cap, uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits)
{
    if (!c || (nq && (!d_count || !d_cand || !d_nhits || (nresults && !d_hits)))) { set_error("null argument"); return MK_ERR_ARG; }
    if (!world || !cap) { set_error("world and cap must be positive"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MergeArgs ma{d_count, d_cand, world, nq, cap, nresults, d_hits, d_nhits};
    return launch_merge(c, ma);
}

int mk_merge_set_sizes(mk_ctx *c, const uint64_t *genome_size, const uint32_t *sketch_size, uint32_t n,
                       uint32_t id_base)
{
    if (!c || (n && (!genome_size || !sketch_size))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    dev_free(c->d_all_ss); dev_free(c->d_all_gs);
    c->all_n = 0; c->all_base = id_base;
    if (!n) return MK_OK;
    MK_TRY(dev_alloc(&c->d_all_ss, n));
    MK_TRY(dev_alloc(&c->d_all_gs, n));
    MK_HIP(hipMemcpy(c->d_all_ss, sketch_size, (size_t)n * 4, hipMemcpyHostToDevice));
    MK_HIP(hipMemcpy(c->d_all_gs, genome_size, (size_t)n * 8, hipMemcpyHostToDevice));
    c->all_n = n;
    return MK_OK;
}

int mk_merge_get_sizes(mk_ctx *c, uint64_t *genome_size, uint32_t *sketch_size, uint32_t n)
{
    if (!c || (n && (!genome_size || !sketch_size))) { set_error("null argument"); return MK_ERR_ARG; }
    if (n != c->all_n) { set_error("%u sizes were set (mk_merge_set_sizes), %u asked for", c->all_n, n); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (!n) return MK_OK;
    MK_HIP(hipMemcpy(sketch_size, c->d_all_ss, (size_t)n * 4, hipMemcpyDeviceToHost));
    MK_HIP(hipMemcpy(genome_size, c->d_all_gs, (size_t)n * 8, hipMemcpyDeviceToHost));
    return MK_OK;
}

int mk_merge_compact(mk_ctx *c, const uint64_t *d_rows, uint32_t world, uint32_t nq, uint32_t cap,
                     uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits)
{
    if (!c || (nq && (!d_rows || !d_nhits || (nresults && !d_hits)))) { set_error("null argument"); return MK_ERR_ARG; }
    if (!world || !cap) { set_error("world and cap must be positive"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MergeArgs ma{nullptr, nullptr, world, nq, cap, nresults, d_hits, d_nhits, d_rows, nullptr, nullptr, 0};
    if (c->all_n) { ma.ss = c->d_all_ss; ma.gs = c->d_all_gs; ma.id_base = c->all_base; }
    else if (world == 1) { ma.ss = c->d_sketch_size; ma.gs = c->d_genome_size; ma.id_base = c->p.genome_id_base; }
    else { set_error("mk_merge_compact over several shards needs mk_merge_set_sizes first"); return MK_ERR_STATE; }
    return launch_merge(c, ma);
}

int mk_set_genome_id_base(mk_ctx *c, uint32_t base)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    c->p.genome_id_base = base;
    return MK_OK;
}

// ---- device buffers for callers that have no GPU runtime of their own
int mk_dev_alloc(mk_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    MK_TRY(use_device(c, false));
    MK_HIP(hipMalloc(out, bytes ? bytes : 1));
    return MK_OK;
}

void mk_dev_free(mk_ctx *c, void *d)
{
    if (!c || !d) return;
    (void)hipSetDevice(c->p.device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
}

int mk_dev_upload(mk_ctx *c, void *d_dst, const void *src, uint64_t bytes)
{
    if (!c || (bytes && (!d_dst || !src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    if (bytes) MK_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

int mk_dev_download(mk_ctx *c, void *dst, const void *d_src, uint64_t bytes)
{
    if (!c || (bytes && (!dst || !d_src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    if (bytes) MK_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

// Device-to-device copy between two contexts' GPUs (a peer DMA over xGMI when they differ),
// queued on the SOURCE context's stream -- so behind the kernels that produced d_src -- and
// waited for before returning: afterwards d_dst is complete for any stream of dst.
// Peer access is asked for and enabled once per ordered pair of GPUs; a pair without it (a
// restricted container, say) copies through host memory instead, and every copy is counted by
// the path it took (mk_stats of the source context), so that a staged exchange is visible as
// such and not as a slow xGMI.  Reads nothing of dst but its device ordinal.
}  // extern "C"

namespace {
std::mutex g_peer_mutex;
int g_peer_state[64][64];                                        // [src][dst]: 0 not asked yet, 1 peer access on, 2 none

bool peer_access(int src, int dst)
{
    if (src == dst) return true;
    if (src < 0 || dst < 0 || src >= 64 || dst >= 64) return false;
    std::lock_guard<std::mutex> g(g_peer_mutex);
    if (g_peer_state[src][dst] == 0) {
        int can = 0;
        bool on = hipDeviceCanAccessPeer(&can, src, dst) == hipSuccess && can;
        if (on) {                                                // (the caller has bound `src`: the access is enabled FROM the current device)
            const hipError_t e = hipDeviceEnablePeerAccess(dst, 0);
            on = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();
        g_peer_state[src][dst] = on ? 1 : 2;
        if (getenv("MIEKKI_VERBOSE"))
            fprintf(stderr, "[miekki] GPU %d -> GPU %d: %s\n", src, dst, on ? "peer access (xGMI)" : "NO peer access: copies go through host memory");
    }
    return g_peer_state[src][dst] == 1;
}
}  // namespace

extern "C" {

int mk_dev_copy(mk_ctx *dst, void *d_dst, mk_ctx *src, const void *d_src, uint64_t bytes)
{
    if (!dst || !src || (bytes && (!d_dst || !d_src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(src, false));
    if (!bytes) { MK_HIP(hipStreamSynchronize(src->stream)); return MK_OK; }
    const int sd = src->p.device, dd = dst->p.device;
    bool direct = peer_access(sd, dd);
    if (direct) {
        const hipError_t e = sd == dd ? hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, src->stream)
                                      : hipMemcpyPeerAsync(d_dst, dd, d_src, sd, bytes, src->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (sd == dd) { set_error("device copy failed: %s", hipGetErrorString(e)); return MK_ERR_DEVICE; }
            { std::lock_guard<std::mutex> g(g_peer_mutex); g_peer_state[sd][dd] = 2; }     // the pair copies through the host from now on
            direct = false;
        }
    }
    if (direct) {
        MK_HIP(hipStreamSynchronize(src->stream));
        src->stats.peer_copies++; src->stats.peer_copy_bytes += bytes;
        return MK_OK;
    }
    // no peer path between the two GPUs: through (page-locked) host memory
    MK_HIP(hipStreamSynchronize(src->stream));
    void *tmp = nullptr;
    MK_HIP(hipHostMalloc(&tmp, bytes, hipHostMallocDefault));
    hipError_t e = hipMemcpy(tmp, d_src, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipSetDevice(dd);
    if (e == hipSuccess) e = hipMemcpy(d_dst, tmp, bytes, hipMemcpyHostToDevice);
    (void)hipSetDevice(sd);
    (void)hipHostFree(tmp);
    if (e != hipSuccess) { set_error("staged device copy failed: %s", hipGetErrorString(e)); return MK_ERR_DEVICE; }
    src->stats.staged_copies++; src->stats.staged_copy_bytes += bytes;
    return MK_OK;
}

int mk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mk_index_export_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, uint8_t *d_dst)
{
    if (!c || !d_dst) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpyAsync(d_dst, c->d_bloom + begin, dev_end - begin, hipMemcpyDeviceToDevice, c->stream));
    const uint64_t zfrom = std::max(begin, dev_end);
    if (zfrom < end) MK_HIP(hipMemsetAsync(d_dst + (zfrom - begin), 0, end - zfrom, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

int mk_index_import_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_src)
{
    if (!c || !d_src) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpyAsync(c->d_bloom + begin, d_src, dev_end - begin, hipMemcpyDeviceToDevice, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_TRY(forget_bloom_summary(c));
    ++c->gen;
    return MK_OK;
}

int mk_index_merge_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_later)
{
    if (!c || !d_later) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_TRY(launch_bloom_merge(c, begin, dev_end, d_later));
    MK_HIP(hipStreamSynchronize(c->stream));
    c->bloom_full_stale = true;
    ++c->gen;
    return MK_OK;
}

uint64_t mk_bloom_reachable_bytes(const mk_ctx *c) { return c ? c->bloom_dev_bytes : 0; }

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
            (lens[q] > (uint64_t)c->p.k + kShortMax ? idx_long : idx_short).push_back(q);
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
