#include "multi_gpu.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace mkhost {

std::vector<int> device_list()
{
    std::vector<int> out;
    if (const char *e = getenv("MIEKKI_DEVICES")) {
        const char *p = e;
        while (*p) {
            char *end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            out.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
        }
        if (!out.empty()) return out;
    }
    if (const char *e = getenv("MIEKKI_DEVICE")) return {atoi(e)};          // one GPU, by ordinal
    const int n = mk_device_count();
    for (int d = 0; d < n; ++d) out.push_back(d);
    if (out.empty()) out.push_back(0);                                       // mk_create will say why it fails
    return out;
}

DeviceGroup::~DeviceGroup()
{
    if (!ctx_.empty()) {
        for (size_t d = 1; d < d_rows_.size(); ++d) mk_dev_free(ctx_[d], d_rows_[d]);
        mk_dev_free(ctx_[0], d_gather_);
        mk_dev_free(ctx_[0], d_hits_);
        mk_dev_free(ctx_[0], d_nhits_);
    }
    if (comm_) mk_comm_destroy(comm_);
    for (mk_ctx *c : ctx_) mk_destroy(c);
}

uint32_t DeviceGroup::total() const
{
    if (!base_.empty()) return base_.back();
    uint32_t n = 0;
    for (mk_ctx *c : ctx_) n += mk_index_size(c);
    return n;
}

size_t DeviceGroup::owner(uint32_t genome) const
{
    size_t d = 0;
    while (d + 1 < ctx_.size() && genome >= base_[d + 1]) ++d;
    return d;
}

int DeviceGroup::finish(bool merge_bloom, std::string &err)
{
    if (comm_) {
        // one process per GPU: the Bloom fold is a MIN all-reduce over rank-keyed cells, the id bases and the sizes of
        // all genomes come from two all-gathers (mk_comm_share_sizes hands them to the merge as well)
        uint32_t base = 0, total = 0;
        if ((merge_bloom && mk_comm_sync_bloom(comm_) != MK_OK) || mk_comm_share_sizes(comm_, &base, &total) != MK_OK) { err = mk_last_error(); return -1; }
        my_base_ = base;
        base_.assign({0u, total});
        gs_all_.assign(total, 0);
        ss_all_.assign(total, 0);
        if (total && mk_merge_get_sizes(ctx_[0], gs_all_.data(), ss_all_.data(), total) != MK_OK) { err = mk_last_error(); return -1; }
        any_empty_sketch_ = false;
        for (uint32_t v : ss_all_) any_empty_sketch_ |= v == 0;
        const uint32_t mine = mk_index_size(ctx_[0]);
        std::vector<uint8_t> all;
        if (all_gather_bytes(&mine, 4, all, err)) return -1;
        largest_shard_ = 0;
        for (int r = 0; r < world(); ++r) { uint32_t n; memcpy(&n, all.data() + 4 * r, 4); largest_shard_ = std::max(largest_shard_, n); }
        return 0;
    }
    const size_t D = ctx_.size();
    base_.assign(D + 1, 0);
    for (size_t d = 0; d < D; ++d) {
        base_[d + 1] = base_[d] + mk_index_size(ctx_[d]);
        if (mk_set_genome_id_base(ctx_[d], base_[d]) != MK_OK) { err = mk_last_error(); return -1; }
    }
    gs_all_.assign(base_[D], 0);
    ss_all_.assign(base_[D], 0);
    for (size_t d = 0; d < D; ++d)
        if (mk_index_export_sizes(ctx_[d], gs_all_.data() + base_[d], ss_all_.data() + base_[d]) != MK_OK) {
            err = mk_last_error();
            return -1;
        }
    if (D == 1) return 0;
    if (mk_merge_set_sizes(ctx_[0], gs_all_.data(), ss_all_.data(), base_[D], 0) != MK_OK) { err = mk_last_error(); return -1; }
    const uint64_t reach = mk_bloom_reachable_bytes(ctx_[0]);
    if (!merge_bloom || !reach) return 0;
    // fold the shards' filters in shard order on the first shard's GPU, then hand the result back
    void *stage = nullptr;
    std::vector<void *> buf(D, nullptr);
    int rc = 0;
    auto fail = [&]() { err = mk_last_error(); rc = -1; };
    if (mk_dev_alloc(ctx_[0], reach, &stage) != MK_OK) fail();
    for (size_t d = 1; d < D && !rc; ++d) {
        if (mk_dev_alloc(ctx_[d], reach, &buf[d]) != MK_OK) { fail(); break; }
        if (mk_index_export_bloom_device(ctx_[d], 0, reach, (uint8_t *)buf[d]) != MK_OK ||
            mk_dev_copy(ctx_[0], stage, ctx_[d], buf[d], reach) != MK_OK ||
            mk_index_merge_bloom_device(ctx_[0], 0, reach, (const uint8_t *)stage) != MK_OK)
            fail();
    }
    if (!rc && mk_index_export_bloom_device(ctx_[0], 0, reach, (uint8_t *)stage) != MK_OK) fail();
    for (size_t d = 1; d < D && !rc; ++d)
        if (mk_dev_copy(ctx_[d], buf[d], ctx_[0], stage, reach) != MK_OK ||
            mk_index_import_bloom_device(ctx_[d], 0, reach, (const uint8_t *)buf[d]) != MK_OK)
            fail();
    for (size_t d = 1; d < D; ++d) mk_dev_free(ctx_[d], buf[d]);
    mk_dev_free(ctx_[0], stage);
    return rc;
}

int DeviceGroup::ensure_buffers(uint32_t nq, uint32_t nresults, uint32_t cap, std::string &err)
{
    const size_t D = ctx_.size();
    const uint64_t words = (uint64_t)std::max<uint64_t>(nq, 1024) * (cap + 1);      // 8-byte row words per shard
    if (words > rows_cap_) {
        for (size_t d = 1; d < d_rows_.size(); ++d) mk_dev_free(ctx_[d], d_rows_[d]);
        mk_dev_free(ctx_[0], d_gather_);
        d_rows_.assign(D, nullptr);
        d_gather_ = nullptr;
        rows_cap_ = 0;
        if (mk_dev_alloc(ctx_[0], D * words * 8, &d_gather_) != MK_OK) { err = mk_last_error(); return -1; }
        for (size_t d = 1; d < D; ++d)
            if (mk_dev_alloc(ctx_[d], words * 8, &d_rows_[d]) != MK_OK) {
                // leave nothing half-made behind: the next call starts from empty buffers
                err = mk_last_error();
                for (size_t e = 1; e < D; ++e) { mk_dev_free(ctx_[e], d_rows_[e]); d_rows_[e] = nullptr; }
                mk_dev_free(ctx_[0], d_gather_);
                d_gather_ = nullptr;
                return -1;
            }
        rows_cap_ = words;
    }
    const uint64_t need_nh = std::max<uint64_t>(nq, 1024);
    if (need_nh > nhits_cap_) {
        mk_dev_free(ctx_[0], d_nhits_);
        d_nhits_ = nullptr; nhits_cap_ = 0;
        if (mk_dev_alloc(ctx_[0], need_nh * 4, &d_nhits_) != MK_OK) { err = mk_last_error(); return -1; }
        nhits_cap_ = need_nh;
    }
    const uint64_t need_hits = need_nh * std::max(nresults, 1u);
    if (need_hits > hits_cap_) {
        mk_dev_free(ctx_[0], d_hits_);
        d_hits_ = nullptr; hits_cap_ = 0;
        if (mk_dev_alloc(ctx_[0], need_hits * sizeof(mk_hit), &d_hits_) != MK_OK) { err = mk_last_error(); return -1; }
        hits_cap_ = need_hits;
    }
    return 0;
}

int DeviceGroup::query(const char *const *seqs, const uint64_t *lens, uint32_t nq, uint32_t nresults, uint32_t min_score,
                       double min_inter, mk_hit *hits, uint32_t *nhits, std::string &err)
{
    if (comm_) {
        if (!nq) return 0;
        mk_params p;
        mk_get_params(ctx_[0], &p);
        std::vector<uint32_t> idx_short, idx_long;
        for (uint32_t q = 0; q < nq; ++q) (lens[q] > (uint64_t)p.k + 4096 ? idx_long : idx_short).push_back(q);
        const uint32_t cap = std::min(entrant_cap(nresults, largest_shard_), kCapWide);
        for (const std::vector<uint32_t> *part : {&idx_short, &idx_long}) {
            if (part->empty()) continue;
            // NaN corner (min_score 0 over an index that holds an empty sketch ANYWHERE: every rank decides alike) and
            // top-N sizes beyond the device selection: dense score rows of every rank
            if (nresults > 64 || (min_score == 0 && any_empty_sketch_)) {
                if (replay_ranked(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, err)) return -1;
            } else if (query_ranked(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, cap, err)) {
                return -1;
            }
        }
        return 0;
    }
    if (ctx_.size() == 1) {
        if (mk_query(ctx_[0], seqs, lens, nq, nresults, min_score, min_inter, hits, nhits, nullptr) != MK_OK) {
            err = mk_last_error();
            return -1;
        }
        return 0;
    }
    if (!nq) return 0;
    mk_params p;
    mk_get_params(ctx_[0], &p);
    // short records and long ones are run as two sets, so that the short ones keep the slab
    // schedule (mk_query does the same for its batches)
    std::vector<uint32_t> idx_short, idx_long;
    for (uint32_t q = 0; q < nq; ++q) (lens[q] > (uint64_t)p.k + 4096 ? idx_long : idx_short).push_back(q);
    uint64_t largest = 0;                                             // the largest shard decides the row width
    for (size_t d = 0; d + 1 < base_.size(); ++d) largest = std::max<uint64_t>(largest, base_[d + 1] - base_[d]);
    const uint32_t cap = std::min(entrant_cap(nresults, largest), kCapWide);
    for (const std::vector<uint32_t> *part : {&idx_short, &idx_long}) {
        if (part->empty()) continue;
        if (nresults > 64) {
            if (replay(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, err)) return -1;
        } else if (query_part(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, cap, err)) {
            return -1;
        }
    }
    return 0;
}

int DeviceGroup::query_part(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens,
                            uint32_t nresults, uint32_t min_score, double min_inter, mk_hit *hits, uint32_t *nhits,
                            uint32_t cap, std::string &err)
{
    const size_t D = ctx_.size();
    const uint32_t n = (uint32_t)idx.size();
    if (ensure_buffers(n, nresults, cap, err)) return -1;
    std::vector<const char *> s(n);
    std::vector<uint64_t> l(n);
    for (uint32_t i = 0; i < n; ++i) { s[i] = seqs[idx[i]]; l[i] = lens[idx[i]]; }
    const uint64_t part_bytes = (uint64_t)n * (cap + 1) * 8;
    std::vector<int> rc(D, MK_OK);
    std::vector<std::string> msg(D);
    std::vector<std::thread> th;
    for (size_t d = 0; d < D; ++d)
        th.emplace_back([&, d] {
            // shard 0 writes straight into its slot of the gather buffer
            uint64_t *rows = d == 0 ? (uint64_t *)d_gather_ : (uint64_t *)d_rows_[d];
            mk_qset *qs = nullptr;
            int r = mk_qset_upload(ctx_[d], s.data(), l.data(), n, &qs);
            if (r == MK_OK) r = mk_qset_run_compact(ctx_[d], qs, nresults, min_score, min_inter, cap, rows);
            // the ONE exchange step: this shard's entrant rows -> the merging GPU (peer DMA over xGMI)
            if (r == MK_OK && d != 0)
                r = mk_dev_copy(ctx_[0], (uint8_t *)d_gather_ + d * part_bytes, ctx_[d], rows, part_bytes);
            if (r == MK_OK) r = mk_sync(ctx_[d]);
            if (r != MK_OK) msg[d] = mk_last_error();
            if (qs) mk_qset_free(ctx_[d], qs);
            rc[d] = r;
        });
    for (auto &t : th) t.join();
    bool unsupported = false;
    for (size_t d = 0; d < D; ++d) {
        if (rc[d] == MK_ERR_UNSUPPORTED) unsupported = true;      // NaN corner: answered from dense rows below
        else if (rc[d] != MK_OK) { err = msg[d]; return -1; }
    }
    if (unsupported) return replay(idx, seqs, lens, nresults, min_score, min_inter, hits, nhits, err);
    gather_bytes_ += (D - 1) * part_bytes;
    if (mk_merge_compact(ctx_[0], (const uint64_t *)d_gather_, (uint32_t)D, n, cap, nresults, (mk_hit *)d_hits_,
                         (uint32_t *)d_nhits_) != MK_OK) { err = mk_last_error(); return -1; }
    std::vector<uint32_t> nh(n);
    std::vector<mk_hit> hh((size_t)n * std::max(nresults, 1u));
    if (mk_dev_download(ctx_[0], nh.data(), d_nhits_, (uint64_t)n * 4) != MK_OK ||
        (nresults && mk_dev_download(ctx_[0], hh.data(), d_hits_, (uint64_t)n * nresults * sizeof(mk_hit)) != MK_OK)) {
        err = mk_last_error();
        return -1;
    }
    std::vector<uint32_t> over;
    for (uint32_t i = 0; i < n; ++i) {
        if (nh[i] == MK_MERGE_OVERFLOW) { over.push_back(idx[i]); continue; }
        nhits[idx[i]] = nh[i];
        std::copy(hh.begin() + (size_t)i * nresults, hh.begin() + (size_t)i * nresults + nh[i], hits + (size_t)idx[i] * nresults);
    }
    if (over.empty()) return 0;
    // More entrants than a row holds on some shard (tie-heavy collections: every copy of a genome is an
    // entrant).  Such queries run once more with wide rows -- still 8 bytes per entrant, still one exchange
    // step -- before anything falls back to dense score rows of every shard, which cost n x G words per
    // shard and the host's heap.  MIEKKI_SHARD_WIDE_ROWS=0 skips the second pass (the tests use it to keep
    // the dense replay covered).
    static const bool wide = [] { const char *e = getenv("MIEKKI_SHARD_WIDE_ROWS"); return !e || atoi(e) != 0; }();
    if (wide && cap < kCapWide) {
        rerun_queries_ += over.size();
        for (size_t i0 = 0; i0 < over.size(); i0 += 4096) {           // bounded buffers: 4096 x 4097 x 8 B = 134 MB per shard
            const std::vector<uint32_t> piece(over.begin() + i0, over.begin() + std::min(over.size(), i0 + 4096));
            if (query_part(piece, seqs, lens, nresults, min_score, min_inter, hits, nhits, kCapWide, err)) return -1;
        }
        return 0;
    }
    return replay(over, seqs, lens, nresults, min_score, min_inter, hits, nhits, err);
}

// filter_results over complete score rows of every shard (Miekki.cpp:376-397 as written): the
// fallback for overflowed rows, NaN intersections and top-N sizes beyond the device selection
int DeviceGroup::replay(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens,
                        uint32_t nresults, uint32_t min_score, double min_inter, mk_hit *hits, uint32_t *nhits,
                        std::string &err)
{
    const size_t D = ctx_.size();
    const uint32_t step = 64;
    replayed_queries_ += idx.size();
    for (size_t i0 = 0; i0 < idx.size(); i0 += step) {
        const uint32_t n = (uint32_t)std::min<size_t>(step, idx.size() - i0);
        std::vector<const char *> s(n);
        std::vector<uint64_t> l(n);
        for (uint32_t i = 0; i < n; ++i) { s[i] = seqs[idx[i0 + i]]; l[i] = lens[idx[i0 + i]]; }
        std::vector<std::vector<uint32_t>> sc(D);
        for (size_t d = 0; d < D; ++d) {
            const uint32_t Gd = base_[d + 1] - base_[d];
            sc[d].assign((size_t)n * Gd + 1, 0);
            if (Gd && mk_query_scores(ctx_[d], s.data(), l.data(), n, sc[d].data()) != MK_OK) { err = mk_last_error(); return -1; }
        }
        std::vector<mk_hit> full;
        for (uint32_t i = 0; i < n; ++i) {
            full.clear();
            for (size_t d = 0; d < D; ++d) {
                const uint32_t Gd = base_[d + 1] - base_[d];
                const uint32_t *row = sc[d].data() + (size_t)i * Gd;
                for (uint32_t g = 0; g < Gd; ++g) {
                    if (row[g] < min_score) continue;
                    const uint32_t id = base_[d] + g;
                    const double jac = (double)row[g] / ss_all_[id];
                    const double inter = jac * gs_all_[id];
                    if (inter < min_inter) continue;
                    full.push_back(mk_hit{id, row[g], jac, inter});
                }
            }
            const uint32_t q = idx[i0 + i];
            nhits[q] = mk_filter_candidates(full.data(), (uint32_t)full.size(), nresults, hits + (size_t)q * nresults);
        }
    }
    return 0;
}

// ---- the multi-process form: one shard here, the others behind the communicator -------------------------------

int DeviceGroup::all_gather_bytes(const void *mine, uint64_t bytes, std::vector<uint8_t> &all, std::string &err)
{
    const int W = world();
    all.assign(bytes * W, 0);
    if (!bytes) return 0;
    void *d = nullptr;
    if (mk_dev_alloc(ctx_[0], bytes * (W + 1), &d) != MK_OK) { err = mk_last_error(); return -1; }
    int rc = 0;
    if (mk_dev_upload(ctx_[0], (uint8_t *)d + bytes * W, mine, bytes) != MK_OK ||
        mk_comm_allgather(comm_, (uint8_t *)d + bytes * W, bytes, d) != MK_OK ||
        mk_dev_download(ctx_[0], all.data(), d, bytes * W) != MK_OK) { err = mk_last_error(); rc = -1; }
    mk_dev_free(ctx_[0], d);
    return rc;
}

int DeviceGroup::all_gather_text(const std::string &mine, std::vector<std::string> &all, std::string &err)
{
    const int W = world();
    all.assign(W, std::string());
    if (!comm_) { all[0] = mine; return 0; }
    const uint64_t len = mine.size();
    std::vector<uint8_t> lens;
    if (all_gather_bytes(&len, 8, lens, err)) return -1;
    uint64_t mx = 0;
    std::vector<uint64_t> n(W);
    for (int r = 0; r < W; ++r) { memcpy(&n[r], lens.data() + 8 * r, 8); mx = std::max(mx, n[r]); }
    if (!mx) return 0;
    std::vector<uint8_t> padded(mx, 0), got;
    memcpy(padded.data(), mine.data(), mine.size());
    if (all_gather_bytes(padded.data(), mx, got, err)) return -1;
    for (int r = 0; r < W; ++r) all[r].assign((const char *)got.data() + mx * r, n[r]);
    return 0;
}

// One pass: every rank scans the batch against its shard and the entrant rows travel to rank 0 while the scan goes on
// (mk_qset_run_compact_gather: ncclGather, or grouped send / recv blocks); rank 0 merges on its GPU and tells everybody
// which rows overflowed; those run once more with wide rows, and what overflows even then is answered from dense rows.
int DeviceGroup::query_ranked(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
                              uint32_t min_score, double min_inter, mk_hit *hits, uint32_t *nhits, uint32_t cap, std::string &err)
{
    mk_ctx *c = ctx_[0];
    const int W = world();
    const uint32_t n = (uint32_t)idx.size();
    std::vector<const char *> s(n);
    std::vector<uint64_t> l(n);
    for (uint32_t i = 0; i < n; ++i) { s[i] = seqs[idx[i]]; l[i] = lens[idx[i]]; }
    const uint64_t words = (uint64_t)n * (cap + 1);
    void *d_rows = nullptr, *d_recv = nullptr, *d_hits = nullptr, *d_nh = nullptr, *d_over = nullptr;
    mk_qset *qs = nullptr;
    std::vector<uint32_t> over_msg(n + 1, 0);                         // [count, positions in idx ...]: rank 0 -> everybody
    auto cleanup = [&] {
        if (qs) mk_qset_free(c, qs);
        mk_dev_free(c, d_rows); mk_dev_free(c, d_recv); mk_dev_free(c, d_hits); mk_dev_free(c, d_nh); mk_dev_free(c, d_over);
    };
    auto fail = [&] { err = mk_last_error(); cleanup(); return -1; };
    // what only this rank does -- buffers, the upload and sketch of the queries -- may fail on this rank alone: every rank
    // says how it fared BEFORE the gather, and all leave together if one could not (nobody waits in a collective the failed
    // rank never enters)
    bool ready = mk_dev_alloc(c, words * 8, &d_rows) == MK_OK && mk_dev_alloc(c, (uint64_t)(n + 1) * 4, &d_over) == MK_OK;
    if (ready && root())
        ready = mk_dev_alloc(c, words * 8 * W, &d_recv) == MK_OK && mk_dev_alloc(c, (uint64_t)n * 4, &d_nh) == MK_OK &&
                mk_dev_alloc(c, (uint64_t)n * std::max(nresults, 1u) * sizeof(mk_hit), &d_hits) == MK_OK;
    ready = ready && mk_qset_upload(c, s.data(), l.data(), n, &qs) == MK_OK;
    {
        const std::string mine = ready ? std::string() : std::string("rank ") + std::to_string(rank()) + ": " + mk_last_error();
        std::vector<std::string> all;
        if (all_gather_text(mine, all, err)) { cleanup(); return -1; }
        for (const std::string &e : all)
            if (!e.empty()) { err = e; cleanup(); return -1; }
    }
    if (mk_qset_run_compact_gather(c, comm_, qs, nresults, min_score, min_inter, cap, (uint64_t *)d_rows, (uint64_t *)d_recv, 0) != MK_OK) return fail();
    constexpr uint32_t kRootFailed = 0xFFFFFFFFu;                     // (rank 0's own failure travels in the message everybody waits for)
    if (root()) {
        gather_bytes_ += (uint64_t)(W - 1) * words * 8;
        std::vector<uint32_t> nh(n);
        std::vector<mk_hit> hh((size_t)n * std::max(nresults, 1u));
        const bool merged = mk_merge_compact(c, (const uint64_t *)d_recv, (uint32_t)W, n, cap, nresults, (mk_hit *)d_hits, (uint32_t *)d_nh) == MK_OK &&
                            mk_dev_download(c, nh.data(), d_nh, (uint64_t)n * 4) == MK_OK &&
                            (!nresults || mk_dev_download(c, hh.data(), d_hits, (uint64_t)n * nresults * sizeof(mk_hit)) == MK_OK);
        if (!merged) { err = mk_last_error(); over_msg[0] = kRootFailed; }
        for (uint32_t i = 0; merged && i < n; ++i) {
            if (nh[i] == MK_MERGE_OVERFLOW) { over_msg[++over_msg[0]] = i; continue; }
            nhits[idx[i]] = nh[i];
            std::copy(hh.begin() + (size_t)i * nresults, hh.begin() + (size_t)i * nresults + nh[i], hits + (size_t)idx[i] * nresults);
        }
        if (mk_dev_upload(c, d_over, over_msg.data(), (uint64_t)(n + 1) * 4) != MK_OK) return fail();
    }
    if (mk_comm_broadcast(comm_, d_over, (uint64_t)(n + 1) * 4, 0) != MK_OK ||
        mk_dev_download(c, over_msg.data(), d_over, (uint64_t)(n + 1) * 4) != MK_OK) return fail();
    if (over_msg[0] == kRootFailed) { if (err.empty()) err = "rank 0 could not merge the rows"; cleanup(); return -1; }
    cleanup();
    qs = nullptr; d_rows = d_recv = d_hits = d_nh = d_over = nullptr;
    if (!over_msg[0]) return 0;
    std::vector<uint32_t> over;
    for (uint32_t i = 1; i <= over_msg[0]; ++i) over.push_back(idx[over_msg[i]]);
    static const bool wide = [] { const char *e = getenv("MIEKKI_SHARD_WIDE_ROWS"); return !e || atoi(e) != 0; }();
    if (wide && cap < kCapWide) {
        rerun_queries_ += over.size();
        for (size_t i0 = 0; i0 < over.size(); i0 += 4096) {
            const std::vector<uint32_t> piece(over.begin() + i0, over.begin() + std::min(over.size(), i0 + 4096));
            if (query_ranked(piece, seqs, lens, nresults, min_score, min_inter, hits, nhits, kCapWide, err)) return -1;
        }
        return 0;
    }
    return replay_ranked(over, seqs, lens, nresults, min_score, min_inter, hits, nhits, err);
}

// filter_results over complete score rows (Miekki.cpp:376-397 as written): every rank's dense rows of the queries,
// padded to the largest shard, gathered on rank 0 (ncclGather), which walks them in rank = genome order
int DeviceGroup::replay_ranked(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
                               uint32_t min_score, double min_inter, mk_hit *hits, uint32_t *nhits, std::string &err)
{
    mk_ctx *c = ctx_[0];
    const int W = world();
    const uint32_t step = 64, Gmine = mk_index_size(c);
    replayed_queries_ += idx.size();
    const uint32_t mine = Gmine;
    std::vector<uint8_t> counts;
    if (all_gather_bytes(&mine, 4, counts, err)) return -1;
    std::vector<uint32_t> Gr(W), first(W + 1, 0);
    for (int r = 0; r < W; ++r) { memcpy(&Gr[r], counts.data() + 4 * r, 4); first[r + 1] = first[r] + Gr[r]; }
    const uint64_t pitch = std::max<uint32_t>(largest_shard_, 1);
    void *d_send = nullptr, *d_recv = nullptr;
    if (mk_dev_alloc(c, (uint64_t)step * pitch * 4, &d_send) != MK_OK ||
        (root() && mk_dev_alloc(c, (uint64_t)step * pitch * 4 * W, &d_recv) != MK_OK)) { err = mk_last_error(); mk_dev_free(c, d_send); return -1; }
    int rc = 0;
    std::vector<uint32_t> sc((size_t)step * pitch), all;
    if (root()) all.resize((size_t)step * pitch * W);
    for (size_t i0 = 0; i0 < idx.size() && !rc; i0 += step) {
        const uint32_t n = (uint32_t)std::min<size_t>(step, idx.size() - i0);
        std::vector<const char *> s(n);
        std::vector<uint64_t> l(n);
        for (uint32_t i = 0; i < n; ++i) { s[i] = seqs[idx[i0 + i]]; l[i] = lens[idx[i0 + i]]; }
        std::fill(sc.begin(), sc.end(), 0u);
        if (Gmine) {
            std::vector<uint32_t> dense((size_t)n * Gmine);
            if (mk_query_scores(c, s.data(), l.data(), n, dense.data()) != MK_OK) { err = mk_last_error(); rc = -1; break; }
            for (uint32_t i = 0; i < n; ++i) memcpy(sc.data() + (size_t)i * pitch, dense.data() + (size_t)i * Gmine, (size_t)Gmine * 4);
        }
        const uint64_t bytes = (uint64_t)step * pitch * 4;
        if (mk_dev_upload(c, d_send, sc.data(), bytes) != MK_OK || mk_comm_gather(comm_, d_send, bytes, d_recv, 0) != MK_OK ||
            (root() ? mk_dev_download(c, all.data(), d_recv, bytes * W) : mk_sync(c)) != MK_OK) { err = mk_last_error(); rc = -1; break; }
        if (!root()) continue;
        std::vector<mk_hit> full;
        for (uint32_t i = 0; i < n; ++i) {
            full.clear();
            for (int r = 0; r < W; ++r) {
                const uint32_t *row = all.data() + ((size_t)r * step + i) * pitch;
                for (uint32_t g = 0; g < Gr[r]; ++g) {
                    if (row[g] < min_score) continue;
                    const uint32_t id = first[r] + g;
                    const double jac = (double)row[g] / ss_all_[id];
                    const double inter = jac * gs_all_[id];
                    if (inter < min_inter) continue;
                    full.push_back(mk_hit{id, row[g], jac, inter});
                }
            }
            const uint32_t q = idx[i0 + i];
            nhits[q] = mk_filter_candidates(full.data(), (uint32_t)full.size(), nresults, hits + (size_t)q * nresults);
        }
    }
    mk_dev_free(c, d_send);
    mk_dev_free(c, d_recv);
    return rc;
}

}  // namespace mkhost
