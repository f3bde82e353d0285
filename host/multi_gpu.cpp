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
    for (const std::vector<uint32_t> *part : {&idx_short, &idx_long}) {
        if (part->empty()) continue;
        if (nresults > 64) {
            if (replay(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, err)) return -1;
        } else if (query_part(*part, seqs, lens, nresults, min_score, min_inter, hits, nhits, kCap, err)) {
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

}  // namespace mkhost
