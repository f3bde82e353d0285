// C ABI of libmiekki_hip.so: the context, the index build's pipeline (appends of every form), import / export.  The query
// side is api_query.hip, what several GPUs need of each other api_multi.hip.  Host-side orchestration only; the arithmetic
// lives in the kernels' files.  There is deliberately no CPU fallback anywhere in these files.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>

#include <sys/mman.h>

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

static int build_from_characters(mk_ctx *c);

// Every entry point starts here: bind the device and, unless the caller is an append
// that wants to overlap with it, fold the build batch still in flight into the index.
int use_device(const mk_ctx *cc, bool settle)
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
    if (hipMalloc((void **)&nM, std::max<uint64_t>((uint64_t)P_hot * ld, 16)) != hipSuccess && !(gz_release_idle_blocks() &&
        hipMalloc((void **)&nM, std::max<uint64_t>((uint64_t)P_hot * ld, 16)) == hipSuccess)) {
        // (the inflater's idle blocks have been given back and it still does not fit)
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

namespace {
std::mutex g_pinned_m;
std::vector<std::pair<void *, uint64_t>> g_registered;              // what pinned_alloc registered itself (pointer, bytes)
}
void *pinned_alloc(uint64_t bytes)
{
    constexpr uint64_t kHuge = 2ull << 20;
    void *p = nullptr;
    if (bytes >= kHuge) {
        const uint64_t size = (bytes + kHuge - 1) / kHuge * kHuge;
        if (posix_memalign(&p, kHuge, size) == 0 && p) {
            (void)madvise(p, size, MADV_HUGEPAGE);                    // (advice: ordinary pages do as well, more slowly)
            if (hipHostRegister(p, size, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> g(g_pinned_m);
                g_registered.emplace_back(p, size);
                return p;
            }
            (void)hipGetLastError();
            free(p);
            p = nullptr;
        }
    }
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pinned_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_m);
        for (size_t i = 0; i < g_registered.size(); ++i)
            if (g_registered[i].first == p) {
                g_registered.erase(g_registered.begin() + (long)i);
                (void)hipHostUnregister(p);
                free(p);
                return;
            }
    }
    (void)hipHostFree(p);
}

uint64_t bloom_regions(const mk_ctx *c)
{
    // (+ 2: the region of the last cell, and the one the summary's last thread may name behind it)
    return ((c->bloom_dev_bytes >> kBloomRegionLog2) + 2 + 15) / 16 * 16;
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
int ensure_build_scratch(mk_ctx *c, uint64_t seq_bytes, int buf, bool for_append)
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
        const uint64_t regions = bloom_regions(c);
        MK_TRY(dev_alloc(&c->d_bloom_touched, 2 * regions));          // "a key was posted", "swept since the last summary"
        MK_HIP(hipMemsetAsync(c->d_bloom_touched, 0, 2 * regions, c->stream));
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
int settle_build(mk_ctx *c)
{
    MK_TRY(settle_one(c, c->older));
    MK_TRY(settle_one(c, c->build));
    MK_HIP(hipStreamSynchronize(c->stream));                     // (the sizes' copies included)
    return MK_OK;
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
    gz_release_staging(c);                                         // (the inflater's staging first: block makers at work finish, gunzip.hip)
    for (auto &blk : c->gz_blocks) (void)hipFree(blk.first);
    c->gz_blocks.clear();
    for (auto &pin : c->gz_pins) (void)hipHostFree(pin.first);
    c->gz_pins.clear();
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
    {
        // (and no thread of the library's own is at work for the context: the inflater makes its next blocks on one -- a caller
        // that leaves by exit() after this must not meet it inside the runtime)
        std::unique_lock<std::mutex> g(c->gz_m);
        c->gz_cv.wait(g, [&] { return c->gz_making == 0; });
    }
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
        MK_TRY(gz_batch_used(batch, c->copy_stream));                // (the strip kernel is the batch's only reader: mk_gz_free waits for the last one)
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
    *out = pinned_alloc(bytes ? bytes : 1);
    if (!*out) { set_error("no page-locked host memory for %llu bytes", (unsigned long long)bytes); return MK_ERR_NOMEM; }
    return MK_OK;
}

void mk_host_free(mk_ctx *c, void *p)
{
    if (!c || !p) return;
    (void)hipSetDevice(c->p.device);
    pinned_free(p);
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


}  // extern "C"
