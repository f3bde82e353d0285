// Several GPUs inside the one `miekki` process -- the reference scales inside one
// executable too (-t, main.cpp:190-196; drivers Miekki.cpp:546-581, 430-480).
//
// One mk_ctx per GPU.  The genomes are sharded in LIST order: shard d holds a contiguous
// run of the list, so genome ids (= list positions of the genomes that were kept) and the
// order in which filter_results meets the genomes (Miekki.cpp:379) are those of a single
// context.  The Bloom filter is the one global structure: the shards' filters are folded
// first-writer-wins in shard order (Miekki.cpp:125-129) and handed back to every shard.
// A query batch is scanned by every GPU against its shard; each emits the heap entrants
// of its shard as 8-byte (genome, matches) records (mk_qset_run_compact); the rows are
// copied GPU-to-GPU over xGMI into the first shard's memory (mk_dev_copy: the single
// exchange step) and merged there by filter_results' heap on the device
// (mk_merge_compact).  Plain C++ above the C ABI: no GPU runtime calls in here.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "miekki_hip.h"

namespace mkhost {

// device ordinals to use: MIEKKI_DEVICES=0,1,2 (an ordinal may repeat: several shards on
// one GPU, which is how a one-GPU box rehearses the multi-GPU path), else every visible GPU
std::vector<int> device_list();

// contiguous, ordered split of n items over `parts` shards (sizes differ by at most one)
inline void shard_range(uint64_t n, uint32_t shard, uint32_t parts, uint64_t &b, uint64_t &e)
{
    const uint64_t base = n / parts, rem = n % parts;
    b = shard * base + (shard < rem ? shard : rem);
    e = b + base + (shard < rem ? 1 : 0);
}

// Entrant slots per query of one shard's exchange row (miekki_amd/shard.py: entrant_cap, the same rule).  The
// entrants of filter_results' heap (Miekki.cpp:387) among m candidates in genome order number about
// N (1 + ln(m / N)), more when scores tie; measured at -h 20, N = 10 (profiles/r4_entrant_rows.txt): 12,500-genome
// shards 64 +- 7 (max 95 of 8,192 queries), 50,000: 77 +- 8 (0.6 % over 96 slots, max 110), 100,000: 83 +- 8 (4.8 % over 96).
inline uint32_t entrant_cap(uint32_t nresults, uint64_t shard_genomes)
{
    const double n = nresults ? (double)nresults : 1.0;
    const double ratio = (double)shard_genomes / n;
    const double want = n * (3.0 + std::log(ratio > 1.0 ? ratio : 1.0));
    const uint32_t cap = 32u * (uint32_t)std::ceil(want / 32.0);
    return cap < 64u ? 64u : cap;
}

class DeviceGroup {
public:
    DeviceGroup() = default;
    ~DeviceGroup();
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;

    // takes ownership of the contexts (shard order)
    void adopt(std::vector<mk_ctx *> ctxs) { ctx_ = std::move(ctxs); }
    // The multi-PROCESS form (one process per GPU, RCCL between them): this process holds ONE shard
    // -- rank `rank` of `world`, contiguous runs of the list in rank order -- and a communicator over its context
    // (mk_comm_create; owned from here on).  finish() and query() become collective: every rank calls them with
    // the same arguments; the merged hits arrive on rank 0 (the other ranks' hit arrays stay empty).
    void set_comm(mk_comm *comm) { comm_ = comm; }
    bool ranked() const { return comm_ != nullptr; }
    mk_comm *comm() const { return comm_; }
    int rank() const { return comm_ ? mk_comm_rank(comm_) : 0; }
    int world() const { return comm_ ? mk_comm_world(comm_) : 1; }
    bool root() const { return rank() == 0; }
    // every rank's text on every rank, in rank order (build logs, genome file names)
    int all_gather_text(const std::string &mine, std::vector<std::string> &all, std::string &err);
    size_t shards() const { return ctx_.size(); }
    mk_ctx *ctx(size_t d) const { return ctx_[d]; }
    const std::vector<mk_ctx *> &contexts() const { return ctx_; }
    uint32_t total() const;                                   // genomes in all shards
    // shard that holds genome id g (after finish())
    size_t owner(uint32_t genome) const;

    // After the shards were built (or loaded): id bases by prefix sum, and for more than one
    // shard the global Bloom filter (unless the shards were loaded with it) and the sizes of
    // all genomes on the merging shard.  0 or -1 (+ err).
    int finish(bool merge_bloom, std::string &err);

    // filter_results(query_sequences(batch), nresults, min_score, min_intersection) over the
    // whole sharded index: hits[nq][nresults], nhits[nq].  One shard: mk_query.
    int query(const char *const *seqs, const uint64_t *lens, uint32_t nq, uint32_t nresults, uint32_t min_score,
              double min_intersection, mk_hit *hits, uint32_t *nhits, std::string &err);

    uint64_t gather_bytes() const { return gather_bytes_; }  // bytes copied between GPUs by query() so far
    // queries whose entrant row overflowed the first pass (entrant_cap slots per shard) and were run again with
    // kCapWide slots, and queries answered from dense score rows of every shard (rows that overflowed
    // even then, NaN corners, top-N sizes beyond the device selection): the slow paths, counted so that
    // a run can say how often it took them (MIEKKI_VERBOSE prints them)
    uint64_t rerun_queries() const { return rerun_queries_; }
    uint64_t replayed_queries() const { return replayed_queries_; }

private:
    int query_part(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
                   uint32_t min_score, double min_intersection, mk_hit *hits, uint32_t *nhits, uint32_t cap,
                   std::string &err);
    int replay(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
               uint32_t min_score, double min_intersection, mk_hit *hits, uint32_t *nhits, std::string &err);
    int ensure_buffers(uint32_t nq, uint32_t nresults, uint32_t cap, std::string &err);

    int query_ranked(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
                     uint32_t min_score, double min_intersection, mk_hit *hits, uint32_t *nhits, uint32_t cap, std::string &err);
    int replay_ranked(const std::vector<uint32_t> &idx, const char *const *seqs, const uint64_t *lens, uint32_t nresults,
                      uint32_t min_score, double min_intersection, mk_hit *hits, uint32_t *nhits, std::string &err);
    int all_gather_bytes(const void *mine, uint64_t bytes, std::vector<uint8_t> &all, std::string &err);

    mk_comm *comm_ = nullptr;
    uint32_t my_base_ = 0, largest_shard_ = 0;                // ranked: this shard's first id, the largest shard of all
    bool any_empty_sketch_ = false;                           // ranked: some genome of some rank has sketch_size 0
    std::vector<mk_ctx *> ctx_;
    std::vector<uint32_t> base_;                              // shards() + 1 id boundaries (ranked: {0, total})
    std::vector<uint64_t> gs_all_;                            // sizes of all genomes (for replays)
    std::vector<uint32_t> ss_all_;
    // per-shard row buffers and, on shard 0, the gather buffer / merge output
    std::vector<void *> d_rows_;
    void *d_gather_ = nullptr, *d_hits_ = nullptr, *d_nhits_ = nullptr;
    uint64_t rows_cap_ = 0, hits_cap_ = 0;                    // row words per shard / hit records the buffers hold
    uint64_t nhits_cap_ = 0;
    uint64_t gather_bytes_ = 0, rerun_queries_ = 0, replayed_queries_ = 0;
    static constexpr uint32_t kCapWide = 4096;                // ... for the second pass over rows that overflowed
};

}  // namespace mkhost
