// `miekki` -- drop-in host driver for the -l / -i / -a / -A / -o / -d / -e command
// line of the reference (main.cpp:125-238) on top of libmiekki_hip.so.
//
// Plain C++ above the C ABI of include/miekki_hip.h: flag parsing, FASTA reading,
// batching, output formatting and the index file format stay on the host exactly
// where the reference has them (file drivers Miekki.cpp:426-645, 723-788); every
// sketch / scan / filter / exact-intersection computation is a call into the HIP
// layer.  Differences from the reference, all deliberate:
//   * genome ids follow list order and output follows query order (what the
//     reference does at -t 1) for every -t; -t sets the number of host threads that
//     read and parse input files ahead of the device,
//   * `-i ... -e` reports that genome file names are not stored in an index
//     instead of crashing,
//   * no zlib re-compression of the in-memory columns (main.cpp:198): they live
//     raw in HBM,
//   * where the reference scales with -t threads inside the one process, this one
//     scales over the visible GPUs inside the one process (MIEKKI_DEVICES=0,1,..
//     restricts them): genome shards in list order, one context per GPU
//     (multi_gpu.hpp); ids, hits and files are those of one GPU,
//   * ... or over one PROCESS per GPU with RCCL between them, when a launcher says so:
//         python -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
//             miekki -l genomes.txt -a queries.fa -o out.txt -h 20        (one command line)
//     (RANK / WORLD_SIZE / LOCAL_RANK from the environment, or MIEKKI_RANK / MIEKKI_WORLD / MIEKKI_LOCAL_RANK): rank r
//     indexes the r-th contiguous run of the list on its own GPU, the Bloom filters are folded by one all-reduce,
//     every rank scans all queries against its shard and ONE ncclGather per batch carries the per-query heap
//     entrants to rank 0, which merges them on its GPU and writes the files and the banners (the other ranks stay
//     silent).  -l with -a / -A (and -e); -i and -d need the single-process form.
#include <getopt.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "fasta_reader.hpp"
#include <zlib.h>

#include "index_io.hpp"
#include "miekki_hip.h"
#include "multi_gpu.hpp"
#include "rendezvous.hpp"

using namespace std;

namespace {

string int_to_string(uint64_t n)                 // utils.cpp:145-157: thousands separators
{
    string s = to_string(n), out;
    for (size_t i = 0; i < s.size(); ++i) {
        if (i && (s.size() - i) % 3 == 0) out += ',';
        out += s[i];
    }
    return out;
}

void help()
{
    cout << "miekki (MI355X build): minhash index for k-mer intersection\n"
            "Input\n"
            "  -i <file>  load a constructed index from disk\n"
            "  -l <file>  construct an index from a list of FASTA files\n"
            "  -a <file>  query a FASTA file (one 2-line record per query)\n"
            "  -A <file>  query every FASTA file of a list as one sequence\n"
            "Output\n"
            "  -o <file>  output file name (out.txt)\n"
            "  -d <file>  dump the index on disk\n"
            "Performances\n"
            "  -h <int>   use 2^h minimizers per sequence (17)\n"
            "  -k <int>   k-mer size (31)\n"
            "  -s <num>   minimal estimated intersection to be reported (200)\n"
            "  -t <int>   host threads that read and parse the input files (8)\n"
            "Advanced usage\n"
            "  -f <int>   fingerprint size: 3 (1-byte) or 11 (2-byte)\n"
            "  -b <int>   2^b bits used for the Bloom filter (33)\n"
            "  -e         exact mode, real intersection is computed on hits\n";
}

void die(const string &what)
{
    cout << what << ": " << mk_last_error() << endl;
    exit(1);
}

vector<string> split_lines(const string &text)   // getline semantics
{
    vector<string> lines;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t e = text.find('\n', pos);
        if (e == string::npos) e = text.size();
        lines.emplace_back(text, pos, e - pos);
        pos = e + 1;
    }
    return lines;
}

bool nucleotide_start(const string &s)           // Miekki.cpp:736, 771
{
    return !s.empty() && (s[0] == 'A' || s[0] == 'C' || s[0] == 'G' || s[0] == 'T' || s[0] == 'N');
}

using mkhost::OrderedFastaReader;

// Page-locked memory for the reader pools, in slabs: page-locking is a slow, serialised driver call (a pool of a few
// hundred 2 MB buffers locked one by one costs a visible part of a second at start-up), so buffers are carved out
// of 256 MB slabs and handed back to a free list; the slabs go when the arena goes.  Requests of half a slab or
// more are locked on their own.  Thread-safe (the readers allocate).
class PinnedArena {
public:
    // (a slab is page-locked AHEAD by a thread of the arena's own: sixteen readers that all stand behind one 50 ms driver call
    // whenever a slab runs out -- eleven times while 4,096 gzip'd genomes were read -- lost half a second of inflating)
    explicit PinnedArena(mk_ctx *ctx) : ctx_(ctx), filler_([this] { fill(); }) {}
    ~PinnedArena()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        filler_.join();
        if (spare_) mk_host_free(ctx_, spare_);
        for (void *s : slabs_) mk_host_free(ctx_, s);
    }
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 4095) / 4096 * 4096;
        if (bytes >= kSlab / 2) {
            void *p = nullptr;
            if (mk_host_alloc(ctx_, bytes, &p) != MK_OK) return nullptr;
            std::lock_guard<std::mutex> g(m_);
            big_.insert(p);
            return p;
        }
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            for (size_t i = 0; i < free_.size(); ++i)                // first fit among the blocks handed back
                if (free_[i].second >= bytes) {
                    void *p = free_[i].first;
                    sizes_[p] = free_[i].second;
                    free_.erase(free_.begin() + (long)i);
                    return p;
                }
            if (cur_ && left_ >= bytes) break;
            if (spare_) {                                            // the slab made ahead becomes the current one; the next is asked for
                slabs_.push_back(spare_);
                cur_ = (char *)spare_; left_ = kSlab;
                spare_ = nullptr; want_ = true;
                cv_.notify_all();
                break;
            }
            if (dry_) return nullptr;                                // (no more page-locked memory to be had: the caller takes ordinary memory)
            want_ = true;
            cv_.notify_all();
            const auto t0 = chrono::steady_clock::now();
            const uint64_t seen = released_;
            cv_.wait(g, [&] { return spare_ || dry_ || released_ != seen; });      // (or a block came back that may fit)
            lock_s_ += chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        }
        void *p = cur_;
        cur_ += bytes; left_ -= bytes;
        sizes_[p] = bytes;
        return p;
    }
    void release(void *p)
    {
        std::unique_lock<std::mutex> g(m_);
        if (big_.erase(p)) { g.unlock(); mk_host_free(ctx_, p); return; }
        auto it = sizes_.find(p);
        if (it != sizes_.end()) { free_.emplace_back(p, it->second); sizes_.erase(it); ++released_; g.unlock(); cv_.notify_all(); }
    }
    double lock_s_ = 0;                          // time callers stood waiting for a slab
    size_t n_slabs_ = 0;
private:
    static constexpr size_t kSlab = 256u << 20;
    void fill()                                   // page-locks the next slab whenever the spare one has been taken
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {                                // (nothing until somebody asks: an arena that is hardly used locks one slab, on demand)
            cv_.wait(g, [&] { return stop_ || (want_ && !spare_ && !dry_); });
            if (stop_) return;
            want_ = false;
            g.unlock();
            void *s = nullptr;
            const bool ok = mk_host_alloc(ctx_, kSlab, &s) == MK_OK;
            g.lock();
            if (ok) { spare_ = s; ++n_slabs_; } else dry_ = true;
            cv_.notify_all();
        }
    }
    mk_ctx *ctx_;
    std::mutex m_;
    std::condition_variable cv_;
    void *spare_ = nullptr;
    bool want_ = false, dry_ = false, stop_ = false;
    uint64_t released_ = 0;
    vector<void *> slabs_;
    char *cur_ = nullptr;
    size_t left_ = 0;
    vector<std::pair<void *, size_t>> free_;
    std::unordered_map<void *, size_t> sizes_;
    std::unordered_set<void *> big_;
    std::thread filler_;                         // (last: it runs as soon as it exists)
};
void *pinned_alloc(void *arena, size_t bytes) { return ((PinnedArena *)arena)->alloc(bytes); }
void pinned_free(void *arena, void *p) { ((PinnedArena *)arena)->release(p); }

struct Driver {
    mkhost::DeviceGroup group;                   // one context per GPU, genome shards in list order
    mk_ctx *ctx0() const { return group.ctx(0); }
    unsigned threads = 8;                        // -t: host reader threads
    bool threads_given = false;                  // -t was on the command line: never start more readers than that in total
    uint32_t k = 31, threshold = 200;
    vector<string> file_names;                   // Miekki.h:59, never persisted
    ofstream out;

    // one shard's share of index_file_of_file: files [f0, f1) of the list into ctx, in order.
    // `log` collects what the reference prints meanwhile (one '-' per genome kept, the
    // "Missed file" lines) so that several shards' output can be shown in list order.
    // what the shards' builds leave behind (see build_shard's end): readers first, then the arenas their buffers came from
    vector<std::unique_ptr<OrderedFastaReader>> kept_readers;
    vector<std::shared_ptr<PinnedArena>> kept_arenas;
    std::mutex keep_m;
    ~Driver() { kept_readers.clear(); kept_arenas.clear(); }
    struct ShardBuild { string log, error; vector<string> names; double t_append = 0, t_wait = 0, t_unpack_wait = 0, t_free = 0, t_recycle = 0, t_start = 0, t_total = 0, t_before = 0, t_after = 0, t_lock = 0; size_t slabs = 0; size_t gz_on_device = 0, gz_on_host = 0, from_readers = 0; };
    void build_shard(mk_ctx *ctx, const vector<string> &files, unsigned nthreads, bool live, ShardBuild &sb)
    {
        const double t_enter = chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count();
        // a doubling matrix would hold old and new copy at once: size it for the whole list up front
        if (!files.empty() && mk_reserve(ctx, (uint32_t)files.size()) != MK_OK) { sb.error = string("index build failed: ") + mk_last_error(); return; }
        // readers parse into pinned buffers, three device batches ahead; the append of one
        // batch returns as soon as its copy is done, so parsing, copying and sketching overlap
        // ... and pack as they parse (2 bits per base, mk_index_append_packed): a quarter of the bytes to buffer and
        // to move over PCIe, and the device skips its own packing pass.
        // gzip'd files -- the reference's normal input (zstr, Miekki.cpp:559) -- arrive as their bytes and are inflated,
        // stripped and appended on the device, a few thousand at a time (a deflate stream is decoded by one lane, so it is
        // the number of streams in flight that makes the rate; MIEKKI_GZ_BATCH sets it, 0 = the readers inflate);
        // whatever the device refuses is inflated here
        // the list goes to the device in units of 512 files, three of them ahead of this thread (fixed: what was measured is in
        // profiles/r6_ingest_gz.txt; a -DMK_TUNE_BUILD binary reads MIEKKI_GZ_BATCH -- 0: the readers inflate everything --,
        // MIEKKI_GZ_IN_FLIGHT, and MIEKKI_GZ_SHARE=1: the readers' own zlib takes a unit whenever three are waiting at the device)
#ifdef MK_TUNE_BUILD
        static const size_t gz_batch = [] { const char *e = getenv("MIEKKI_GZ_BATCH"); return e ? (size_t)std::max(0L, atol(e)) : (size_t)512; }();
        static const size_t gz_in_flight = [] { const char *e = getenv("MIEKKI_GZ_IN_FLIGHT"); return e ? (size_t)std::max(1L, atol(e)) : (size_t)3; }();
        static const bool gz_share = [] { const char *e = getenv("MIEKKI_GZ_SHARE"); return e && atoi(e) != 0; }();
#else
        constexpr size_t gz_batch = 512, gz_in_flight = 3;
        constexpr bool gz_share = false;
#endif
        // A unit's files go to the device as the readers read them (RawSink -> mk_gz_open / mk_gz_stage / mk_gz_put): straight
        // from the page cache into page-locked pieces the library lends, a DMA each, into a batch whose layout the files'
        // sizes fixed -- nothing of a gzip'd file waits in host memory.  A unit the device has no memory for is the readers'.
        // A unit RUNS (mk_gz_run, on a thread of its own) as soon as its last file has been put -- the reader that put it says
        // so -- not when this thread gets round to it: the device inflates the units ahead while the ones before them are
        // being appended.
        struct UnitRun { std::future<int> ran; string error; };
        struct Sink {
            mk_ctx *ctx;
            std::mutex m;
            std::unordered_map<void *, std::shared_ptr<UnitRun>> runs;
            static void *open(void *u, const uint64_t *sizes, uint32_t m, uint64_t *offsets)
            {
                mk_gz_batch *b = nullptr;
                if (mk_gz_open(((Sink *)u)->ctx, sizes, m, &b) != MK_OK) return nullptr;
                if (mk_gz_layout(b, offsets) != MK_OK) { mk_gz_free(b); return nullptr; }
                return b;
            }
            static bool put_span(void *, void *batch, uint32_t first, uint32_t count, const void *data, uint64_t bytes, bool staged)
            {
                return mk_gz_put_span((mk_gz_batch *)batch, first, count, data, bytes, staged ? 1 : 0) == MK_OK;
            }
            static void *stage(void *, void *batch, uint64_t *cap) { return mk_gz_stage((mk_gz_batch *)batch, cap); }
            static bool put(void *, void *batch, uint32_t i, uint64_t at, const void *data, uint64_t bytes, bool staged)
            {
                return mk_gz_put((mk_gz_batch *)batch, i, at, data, bytes, staged ? 1 : 0) == MK_OK;
            }
            static void complete(void *u, void *batch)
            {
                std::shared_ptr<UnitRun> r = std::make_shared<UnitRun>();
                UnitRun *rp = r.get();
                r->ran = std::async(std::launch::async, [rp, batch]() -> int {
                    const int rc = mk_gz_run((mk_gz_batch *)batch);
                    if (rc != MK_OK) rp->error = mk_last_error();
                    return rc;
                });
                std::lock_guard<std::mutex> g(((Sink *)u)->m);
                ((Sink *)u)->runs[batch] = std::move(r);
            }
            std::shared_ptr<UnitRun> take(void *batch)
            {
                std::lock_guard<std::mutex> g(m);
                auto it = runs.find(batch);
                if (it == runs.end()) return nullptr;
                std::shared_ptr<UnitRun> r = std::move(it->second);
                runs.erase(it);
                return r;
            }
            ~Sink() { for (auto &kv : runs) if (kv.second && kv.second->ran.valid()) (void)kv.second->ran.get(); }
        } sink_user;
        sink_user.ctx = ctx;
        mkhost::RawSink sink;
        sink.user = &sink_user; sink.open = Sink::open; sink.stage = Sink::stage; sink.put = Sink::put; sink.put_span = Sink::put_span; sink.complete = Sink::complete;
        // (the arena outlives the reader: declared first.  Both are handed to a thread of their own when the shard is built:
        // giving gigabytes of page-locked memory back takes tenths of a second that nothing has to wait for)
        std::shared_ptr<PinnedArena> arena_p = std::make_shared<PinnedArena>(ctx);
        std::unique_ptr<OrderedFastaReader> reader_p(new OrderedFastaReader(files, nthreads, mkhost::HostAllocator{pinned_alloc, pinned_free, arena_p.get()},
                                                                            std::max<size_t>(3 * 64, gz_in_flight * gz_batch + 64), true, gz_batch != 0,
                                                                            gz_batch, gz_in_flight, gz_batch ? sink : mkhost::RawSink(), gz_share));
        OrderedFastaReader &reader = *reader_p;
        auto now = [] { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); };
        auto show = [&]() { if (live) { cout << sb.log << flush_stream(); sb.log.clear(); } };
        // What has been taken from the readers and waits for its turn, in list order: runs of up to 64 sequences the readers
        // made (packed), and whole UNITS that went to the device -- each run by a thread of its own (mk_gz_run works on a
        // stream of its own; a deflate stream is decoded by few lanes, so it is batches side by side that fill the device).
        // Everything is appended in list order, sixty-four at a time.
        struct Pending {
            vector<OrderedFastaReader::Item> items;
            vector<string> names;
            mk_gz_batch *batch = nullptr;                           // a device unit: its files' bytes are there already
            std::shared_ptr<UnitRun> run;                           // ... and it runs, or has run (the readers started it)
        };
        std::deque<std::unique_ptr<Pending>> pending;
        size_t raw_in_flight = 0, host_waiting = 0;
        std::unique_ptr<Pending> cur;                               // the run being collected
        uint64_t cur_bytes = 0;
        // (whatever way this function is left: no thread of a batch still runs, no batch keeps its blocks)
        struct Cleanup {
            std::deque<std::unique_ptr<Pending>> &pending; std::unique_ptr<Pending> &cur;
            ~Cleanup()
            {
                auto drop = [](Pending *p) { if (!p) return; if (p->run && p->run->ran.valid()) (void)p->run->ran.get(); if (p->batch && p->run) mk_gz_free(p->batch); p->batch = nullptr; };
                for (auto &p : pending) drop(p.get());
                drop(cur.get());
            }
        } cleanup{pending, cur};
        auto close_run = [&]() {
            if (!cur) return;
            if (cur->batch) {
                cur->run = sink_user.take(cur->batch);                // (started when the unit's last file had been put: before its item could be taken)
                ++raw_in_flight;
            } else {
                host_waiting += cur->items.size();
            }
            pending.push_back(std::move(cur));
            cur_bytes = 0;
        };
        auto append_items = [&](const OrderedFastaReader::Item *items, size_t n) {      // sequences the readers made, one call
            const double t0 = now();
            int rc;
            if (items[0].packed) {
                vector<mk_packed_seq> p(n);
                for (size_t i = 0; i < n; ++i) {
                    p[i].codes = items[i].codes; p[i].except = items[i].dirty ? items[i].except : nullptr; p[i].len = items[i].len;
                    memcpy(p[i].head, items[i].head, 32);
                }
                rc = mk_index_append_packed(ctx, p.data(), (uint32_t)n);
            } else {
                vector<const char *> p;
                vector<uint64_t> l;
                for (size_t i = 0; i < n; ++i) { p.push_back(items[i].data); l.push_back(items[i].len); }
                rc = mk_index_append(ctx, p.data(), l.data(), (uint32_t)n);
            }
            if (rc != MK_OK) { sb.error = string("index build failed: ") + mk_last_error(); return false; }
            sb.t_append += now() - t0;
            return true;
        };
        auto append_host_run = [&](Pending &run) {
            if (!append_items(run.items.data(), run.items.size())) return false;
            sb.names.insert(sb.names.end(), run.names.begin(), run.names.end());
            sb.log.append(run.items.size(), '-');
            host_waiting -= run.items.size();
            sb.from_readers += run.items.size();
            for (auto &s : run.items) reader.recycle(s);
            return true;
        };
        // a file the device did not take (or a whole unit it failed on): read and inflated here, appended in its place
        auto append_from_disk = [&](const string &fn) {
            vector<char> text, scratch, seq;
            if (!mkhost::read_file(fn, text, scratch)) { sb.error = "cannot read " + fn; return false; }
            seq.resize(text.size() + 1);
            seq.resize(mkhost::strip_fasta(text.data(), text.size(), seq.data()));
            ++sb.gz_on_host;
            if (seq.size() < k) return true;
            const char *p = seq.data();
            const uint64_t l = seq.size();
            if (mk_index_append(ctx, &p, &l, 1) != MK_OK) { sb.error = string("index build failed: ") + mk_last_error(); return false; }
            sb.names.push_back(fn); sb.log += '-';
            return true;
        };
        mk_gz_batch *retired = nullptr;                              // the unit appended last: its text may still be read
        struct Retire { mk_gz_batch *&b; ~Retire() { if (b) mk_gz_free(b); b = nullptr; } } retire{retired};
        auto append_unit = [&](Pending &rb) {
            const double t0 = now();
            const int ran = rb.run ? rb.run->ran.get() : MK_ERR_STATE;
            sb.t_unpack_wait += now() - t0;
            --raw_in_flight;
            // (a batch the device could not run -- no memory for its text, say -- is read again and inflated here, file by
            // file: slower, never wrong; its blocks go back first)
            if (ran != MK_OK) { mk_gz_free(rb.batch); rb.batch = nullptr; }
            vector<uint32_t> which;                                // files of the batch waiting to be appended together
            vector<string> kept;
            bool ok = true;
            auto append = [&]() {
                if (which.empty()) return true;
                const double ta = now();
                if (mk_index_append_gz(ctx, rb.batch, which.data(), (uint32_t)which.size()) != MK_OK) { sb.error = string("index build failed: ") + mk_last_error(); return false; }
                sb.t_append += now() - ta;
                sb.names.insert(sb.names.end(), kept.begin(), kept.end());
                which.clear(); kept.clear();
                return true;
            };
            for (size_t i = 0; i < rb.items.size() && ok; ++i) {
                OrderedFastaReader::Item &it = rb.items[i];
                if (!it.exists) { ok = append(); sb.log += "Missed file: " + rb.names[i] + "\n"; show(); continue; }
                if (it.failed) { ok = append(); if (ok) { sb.error = "cannot read " + rb.names[i]; ok = false; } break; }
                if (!it.raw) {                                     // a file of the unit that came the ordinary way (not gzip'd after all)
                    if (it.len >= k) {
                        ok = append() && append_items(&it, 1);
                        if (ok) { sb.names.push_back(rb.names[i]); sb.log += '-'; ++sb.from_readers; }
                    }
                    continue;
                }
                uint64_t len = 0;
                int32_t st = MK_GZ_INTERNAL;
                if (rb.batch) mk_gz_sequence(rb.batch, it.unit_index, &len, &st);
                if (st != MK_GZ_OK) { ok = append() && append_from_disk(rb.names[i]); continue; }
                if (len >= k) {
                    which.push_back(it.unit_index); kept.push_back(rb.names[i]);
                    sb.log += '-';
                    if (which.size() >= 64) { ok = append(); show(); }
                }
            }
            ok = ok && append();
            // the batch goes when the NEXT unit has been appended (mk_gz_free waits for the strip kernels that read its text:
            // by then they are long done, and this thread has queued the next appends instead of standing here)
            const double tf = now();
            if (retired) mk_gz_free(retired);
            retired = rb.batch;
            rb.batch = nullptr;
            sb.t_free += now() - tf;
            sb.gz_on_device += rb.items.size();
            const double tr = now();
            for (auto &it : rb.items) reader.recycle(it);
            sb.t_recycle += now() - tr;
            reader.raw_consumed(rb.items.size());
            return ok;
        };
        // the front of the queue into the index: everything (all), or whatever is ready now -- and more, waiting for the
        // device, while too much is held back behind it
        auto drain = [&](bool all) {
            while (!pending.empty()) {
                Pending &f = *pending.front();
                if (f.batch && !all && raw_in_flight <= gz_in_flight && host_waiting < 4096 &&
                    f.run && f.run->ran.wait_for(chrono::seconds(0)) != std::future_status::ready) break;
                const bool ok = f.batch ? append_unit(f) : append_host_run(f);
                pending.pop_front();
                show();
                if (!ok) return false;
            }
            return true;
        };
        const double t_loop = now();
        sb.t_before = t_loop - t_enter;
        for (size_t i = 0; i < files.size(); ++i) {
            const string &fn = files[i];
            const double t0 = now();
            OrderedFastaReader::Item item = reader.take(i);
            sb.t_wait += now() - t0;
            if (item.unit_batch) {
                // a file of a device unit, whatever became of it: the unit stays together, in list order, and runs when its
                // last file has been taken (every file has been put then)
                if (cur && cur->batch != (mk_gz_batch *)item.unit_batch) close_run();
                if (!cur) { cur.reset(new Pending()); cur->batch = (mk_gz_batch *)item.unit_batch; }
                cur->items.push_back(item); cur->names.push_back(fn);
                if (item.unit_last) close_run();
                if (!drain(false)) return;
                continue;
            }
            if (!item.exists) { close_run(); if (!drain(true)) return; sb.log += "Missed file: " + fn + "\n"; show(); reader.recycle(item); continue; }
            if (item.failed) { close_run(); (void)drain(true); if (sb.error.empty()) sb.error = "cannot read " + fn; return; }
            if (item.len < k) { reader.recycle(item); continue; }
            if (cur && cur->batch) close_run();
            if (!cur) cur.reset(new Pending());
            cur->items.push_back(item); cur->names.push_back(fn);
            cur_bytes += item.len;
            if (cur->items.size() >= 64 || cur_bytes > (1ull << 30)) close_run();
            if (!drain(false)) return;
        }
        close_run();
        if (!drain(true)) return;
        sb.t_total = now() - t_loop;
        sb.t_lock = arena_p->lock_s_; sb.slabs = arena_p->n_slabs_;
        // the reader's pool and its page-locked arena are kept until the process ends (~Driver), the inflater's blocks until
        // their context goes: giving gigabytes back takes tenths of a second -- inside the index phase when done here, out of the
        // queries' first batches when done beside them (freeing page-locked or device memory stalls the device) -- and a GPU
        // with 288 GB does not miss them
        {
            std::lock_guard<std::mutex> g(keep_m);
            kept_readers.emplace_back(reader_p.release());
            kept_arenas.push_back(std::move(arena_p));
        }
        if (mk_index_size(ctx) != sb.names.size()) sb.error = string("index build failed: ") + mk_last_error();   // settles the last batch
        sb.t_after = now() - t_loop - sb.t_total;
    }

    // ---- Miekki.cpp:540-588.  `make_ctx(device ordinal)` creates one shard's context.
    template <typename MakeCtx>
    void index_file_of_file(const string &list, const vector<int> &devices, MakeCtx make_ctx)
    {
        vector<string> files;
        const bool have_list = mkhost::file_exists(list);
        if (have_list) {
            string text;
            mkhost::read_text(list, text);
            for (const string &fn : split_lines(text))
                if (fn.size() > 3) files.push_back(fn);
        }
        if (rank_world > 1 || forced_rank_mode) { index_file_of_file_ranked(list, have_list, files, devices, make_ctx); return; }
        // genome shards = contiguous runs of the list, one per GPU (never more shards than files)
        const size_t D = std::max<size_t>(1, std::min<size_t>(devices.size(), files.size()));
        vector<mk_ctx *> ctxs;
        for (size_t d = 0; d < D; ++d) ctxs.push_back(make_ctx(devices[d]));
        group.adopt(ctxs);
        if (!have_list) { cout << "Missed file of file: " << list << endl; finish_index(true); return; }
        vector<ShardBuild> sb(D);
        vector<vector<string>> part(D);
        for (size_t d = 0; d < D; ++d) {
            uint64_t b, e;
            mkhost::shard_range(files.size(), (uint32_t)d, (uint32_t)D, b, e);
            part[d].assign(files.begin() + b, files.begin() + e);
        }
        // reader threads per shard: -t as given for one GPU; with several, -t divided by the shards would leave each
        // with a reader or two (the default -t 8 on 8 GPUs: one), far below what a GPU sketches -- so at least eight
        // per shard, as far as the host's cores go
        // ... the CPUs this process may actually use (affinity mask, cgroup quota -- not hardware_concurrency(), which on a
        // 256-thread host with a 16-CPU quota would start 64 readers for 8 shards); an explicit -t is never exceeded in total
        const unsigned hw = mkhost::usable_cpus();
        const unsigned floor_per = threads_given ? 1u : std::min(8u, std::max(1u, hw / (unsigned)D));
        const unsigned per = D == 1 ? std::max(1u, threads) : std::max(1u, std::max(threads / (unsigned)D, floor_per));
        if (D == 1) {
            build_shard(ctxs[0], part[0], per, true, sb[0]);
        } else {                                                   // the shards build side by side
            vector<std::thread> th;
            for (size_t d = 0; d < D; ++d) th.emplace_back([&, d] { build_shard(ctxs[d], part[d], per, false, sb[d]); });
            for (auto &t : th) t.join();
        }
        for (size_t d = 0; d < D; ++d) {                           // what the reference prints, in list order
            cout << sb[d].log;
            if (!sb[d].error.empty()) { cout << endl << sb[d].error << endl; exit(1); }
            file_names.insert(file_names.end(), sb[d].names.begin(), sb[d].names.end());
        }
        cout << endl;
        if (getenv("MIEKKI_VERBOSE"))
            for (size_t d = 0; d < D; ++d)
                cout << "[ingest] shard " << d << ": " << sb[d].names.size() << " genomes, waited for the readers " << sb[d].t_wait
                     << "s, in mk_index_append " << sb[d].t_append << "s, waited for the device's inflater " << sb[d].t_unpack_wait << "s and for its batches' last kernels " << sb[d].t_free << "s (giving the raw files' buffers back " << sb[d].t_recycle << "s, starting batches " << sb[d].t_start << "s, the whole loop " << sb[d].t_total << "s, before it " << sb[d].t_before << "s, after it " << sb[d].t_after << "s; " << sb[d].slabs << " slabs of 256 MB page-locked, readers waited for them " << sb[d].t_lock << "s); gzip'd files inflated on the device " << sb[d].gz_on_device - sb[d].gz_on_host
                     << ", refused by it and inflated here " << sb[d].gz_on_host << "; sequences that came from the readers " << sb[d].from_readers << endl;
        finish_index(true);
        compress_cold();
        cout << "Reference indexed: " << group.total() << endl;
        mk_params p;
        mk_get_params(ctx0(), &p);
        if (p.bloom_log2) cout << "BF size:" << int_to_string(1ull << p.bloom_log2) << endl;
    }

    // ---- one process per GPU (RCCL): this process builds the rank_id-th contiguous run of the list
    int rank_id = 0, rank_world = 1, rank_local = 0;
    bool forced_rank_mode = false;                // a world of one was asked for explicitly (tests: RCCL on one GPU)

    // rank 0 draws the communicator's id and leaves it where the launcher's other children find it (rendezvous.hpp: a file
    // that carries the run's nonce, made exclusively, checked by its readers, gone when everybody has joined)
    mkhost::Rendezvous meet;
    mk_comm *make_comm(mk_ctx *ctx)
    {
        uint8_t id[MK_COMM_ID_BYTES];
        string err;
        if (rank_id == 0) {
            if (mk_comm_unique_id(id) != MK_OK) die("cannot start RCCL");
            static string to_remove;                                   // (whatever ends this process: the file goes)
            to_remove = meet.path;
            atexit([] { if (!to_remove.empty()) unlink(to_remove.c_str()); });
            if (!mkhost::rendezvous_publish(meet, id, sizeof id, err)) { cout << err << endl; exit(1); }
        } else if (!mkhost::rendezvous_fetch(meet, id, sizeof id, 600000, err)) {   // up to ten minutes: rank 0 may still be starting
            cerr << "rank " << rank_id << ": " << err << endl;
            exit(1);
        }
        mk_comm *comm = nullptr;
        if (mk_comm_create(ctx, rank_id, rank_world, id, &comm) != MK_OK) die("cannot create the communicator");
        if (mk_comm_barrier(comm) != MK_OK) die("communicator barrier failed");
        if (rank_id == 0) mkhost::rendezvous_remove(meet);             // everybody has read it
        return comm;
    }

    // Every rank says how it fared (empty = well) before the next collective: if any rank failed, ALL leave here, with the
    // first failure's words on rank 0's stdout -- nobody is left waiting in a collective the failed rank will never enter.
    void agree(const string &mine, const char *what)
    {
        if (!group.ranked()) { if (!mine.empty()) { cout << what << mine << endl; exit(1); } return; }
        vector<string> all;
        string err;
        if (group.all_gather_text(mine, all, err)) { cout << "multi-GPU setup failed: " << err << endl; exit(1); }
        for (const string &e : all)
            if (!e.empty()) { cout << what << e << endl; exit(1); }
    }

    template <typename MakeCtx>
    void index_file_of_file_ranked(const string &list, bool have_list, const vector<string> &files, const vector<int> &devices,
                                   MakeCtx make_ctx)
    {
        // the GPU of this rank: the LOCAL_RANK-th of MIEKKI_DEVICES when given, else that ordinal
        const int device = rank_local < (int)devices.size() ? devices[rank_local] : rank_local;
        mk_ctx *ctx = make_ctx(device);
        group.adopt({ctx});
        group.set_comm(make_comm(ctx));
        if (!have_list) { cout << "Missed file of file: " << list << endl; finish_index(true); return; }
        uint64_t b, e;
        mkhost::shard_range(files.size(), (uint32_t)rank_id, (uint32_t)rank_world, b, e);
        const vector<string> part(files.begin() + b, files.begin() + e);
        ShardBuild sb;
        build_shard(ctx, part, std::max(1u, threads), false, sb);
        // what the reference prints while it indexes, and the names of the genomes that were kept: every rank's, in
        // rank order = list order (an error on any rank ends the run on all of them)
        string err, names;
        for (const string &n : sb.names) { names += n; names += '\n'; }
        vector<string> logs, all_names, errors;
        if (group.all_gather_text(sb.error, errors, err) || group.all_gather_text(sb.log, logs, err) ||
            group.all_gather_text(names, all_names, err)) { cout << "multi-GPU setup failed: " << err << endl; exit(1); }
        for (int r = 0; r < rank_world; ++r) {
            cout << logs[r];
            if (!errors[r].empty()) { cout << endl << errors[r] << endl; exit(1); }
            for (const string &n : split_lines(all_names[r]))
                if (!n.empty()) file_names.push_back(n);
        }
        cout << endl;
        finish_index(true);
        compress_cold();
        cout << "Reference indexed: " << group.total() << endl;
        mk_params p;
        mk_get_params(ctx0(), &p);
        if (p.bloom_log2) cout << "BF size:" << int_to_string(1ull << p.bloom_log2) << endl;
    }

    // INDEX->compress_index(1) of main.cpp:198, for the rows that live in host memory (a collection beyond the GPUs' memory):
    // packed where related genomes sit next to each other in the list, left alone where nothing is to be gained
    void compress_cold()
    {
        for (size_t d = 0; d < group.shards(); ++d) {
            uint64_t raw = 0, packed = 0;
            if (mk_index_compress(group.ctx(d), &raw, &packed) != MK_OK) { cout << "index compression failed: " << mk_last_error() << endl; exit(1); }
            if (raw && getenv("MIEKKI_VERBOSE"))
                cout << "[cold rows] shard " << d << ": " << raw << " bytes in host memory, " << packed << " after packing" << endl;
        }
    }

    // id bases, the global Bloom filter and the sizes of all genomes on the merging GPU
    void finish_index(bool merge_bloom)
    {
        string err;
        if (group.finish(merge_bloom, err) != 0) { cout << "multi-GPU setup failed: " << err << endl; exit(1); }
    }

    static const char *flush_stream() { cout.flush(); return ""; }

    // one output line of query_file / query_whole_file (Miekki.cpp:440-444, 503-505)
    static string hit_text(const mk_hit *h, uint32_t n)
    {
        // to_string(uint) \t to_string(uint) \t to_string(uint(intersection)) \t to_string(double) ";"
        // -- std::to_string(double) is printf's %f -- in one formatting call per hit
        string s;
        char buf[96];
        for (uint32_t i = 0; i < n; ++i) {
            const int len = snprintf(buf, sizeof buf, "%u\t%u\t%u\t%f;", h[i].genome, h[i].matches,
                                     (unsigned)h[i].intersection, h[i].jaccard);
            s.append(buf, (size_t)len);
        }
        return s;
    }

    void run_query(const vector<const char *> &p, const vector<uint64_t> &l, uint32_t nres, uint32_t min_score,
                   double min_inter, vector<mk_hit> &hits, vector<uint32_t> &nhits)
    {
        hits.assign((size_t)p.size() * nres + 1, mk_hit{});
        nhits.assign(p.size() + 1, 0);
        if (p.empty()) return;
        string err;
        if (group.query(p.data(), l.data(), (uint32_t)p.size(), nres, min_score, min_inter, hits.data(), nhits.data(), err) != 0) {
            cout << "query failed: " << err << endl;
            exit(1);
        }
    }

    void run_query(const vector<const string *> &seqs, uint32_t nres, uint32_t min_score, double min_inter,
                   vector<mk_hit> &hits, vector<uint32_t> &nhits)
    {
        vector<const char *> p;
        vector<uint64_t> l;
        for (auto s : seqs) { p.push_back(s->data()); l.push_back(s->size()); }
        run_query(p, l, nres, min_score, min_inter, hits, nhits);
    }

    // strict 2-line records (Miekki.cpp:458-464)
    static void read_records(const string &path, vector<string> &heads, vector<string> &seqs)
    {
        string text;
        mkhost::read_text(path, text);
        vector<string> lines = split_lines(text);
        for (size_t i = 0; i < lines.size(); i += 2) {
            heads.push_back(std::move(lines[i]));
            seqs.push_back(i + 1 < lines.size() ? std::move(lines[i + 1]) : string());
        }
    }

    // 2-line records streamed from a (possibly gzipped) file: the reference reads a query
    // file record by record too (Miekki.cpp:458-464), so its size never has to fit in memory
    class RecordStream {
    public:
        explicit RecordStream(const string &path) : f_(gzopen(path.c_str(), "rb")), buf_(4u << 20)
        {
            if (f_) gzbuffer(f_, 1u << 20);
        }
        ~RecordStream() { if (f_) gzclose(f_); }
        // up to `want` records with at least k bases (shorter ones are dropped, 465-468); false at the end
        bool next(size_t want, uint32_t k, vector<string> &heads, vector<string> &seqs)
        {
            heads.clear(); seqs.clear();
            string head, ref;
            while (seqs.size() < want && !done_) {
                const bool got_head = getline(head);
                const bool got_ref = getline(ref);
                if (!got_head && !got_ref) break;
                if (ref.size() >= k) { heads.push_back(std::move(head)); seqs.push_back(std::move(ref)); }
                head.clear(); ref.clear();
            }
            return !seqs.empty();
        }
    private:
        bool getline(string &out)                  // std::getline: false only when nothing was left
        {
            out.clear();
            bool any = false;
            for (;;) {
                if (pos_ == len_) {
                    if (done_ || !f_) { done_ = true; return any; }
                    const int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
                    if (n <= 0) { done_ = true; return any; }
                    pos_ = 0; len_ = (size_t)n;
                }
                any = true;
                const char *p = buf_.data() + pos_;
                const char *e = (const char *)memchr(p, '\n', len_ - pos_);
                if (e) { out.append(p, (size_t)(e - p)); pos_ += (size_t)(e - p) + 1; return true; }
                out.append(p, len_ - pos_);
                pos_ = len_;
            }
        }
        gzFile f_;
        vector<char> buf_;
        size_t pos_ = 0, len_ = 0;
        bool done_ = false;
    };

    // ---- Miekki.cpp:426-483
    void query_file(const string &path)
    {
        if (!mkhost::file_exists(path)) { cout << "File problem" << endl; return; }
        RecordStream in(path);
        const size_t super = 16384;
        vector<string> heads, seqs, next_heads, next_seqs;
        vector<mk_hit> hits;
        vector<uint32_t> nhits;
        struct WriteJob { vector<string> heads; vector<mk_hit> hits; vector<uint32_t> nhits; size_t n = 0; };
        std::future<void> writer;
        size_t done = 0;
        bool more = in.next(super, k, heads, seqs);
        while (more) {
            // the next super-batch is read and split while the device works on this one
            auto ahead = std::async(std::launch::async, [&] { return in.next(super, k, next_heads, next_seqs); });
            vector<const string *> q;
            for (auto &r : seqs) q.push_back(&r);
            run_query(q, 10, 10, 0.5 * threshold, hits, nhits);
            for (size_t i = 0; i < seqs.size(); ++i)
                if (((done + i) % 201) == 0) cout << "-" << flush_stream();   // one mark per reference batch (345)
            // formatting and writing this batch's lines runs beside the next batch's device work
            // (one writer at a time, in order)
            if (writer.valid()) writer.get();
            auto job = std::make_shared<WriteJob>();
            job->heads.swap(heads); job->hits.swap(hits); job->nhits.swap(nhits);
            job->n = seqs.size();
            writer = std::async(std::launch::async, [this, job] {
                string text;
                for (size_t i = 0; i < job->n; ++i) {
                    text += job->heads[i];
                    text += ':';
                    text += hit_text(job->hits.data() + i * 10, job->nhits[i]);
                    text += '\n';
                }
                out << text;
            });
            done += seqs.size();
            more = ahead.get();
            heads.swap(next_heads); seqs.swap(next_seqs);
        }
        if (writer.valid()) writer.get();
        out << flush;
    }

    // ---- Miekki.cpp:592-612, 487-514
    void query_file_of_file(const string &list)
    {
        if (!mkhost::file_exists(list)) { cout << "Missed file of file: " << list << endl; return; }
        string text;
        mkhost::read_text(list, text);
        vector<string> files;
        for (const string &fn : split_lines(text))
            if (fn.size() > 3) files.push_back(fn);
        // readers parse into pinned buffers, two batches ahead: the upload of a batch is a DMA
        // straight out of them
        PinnedArena arena(ctx0());
        OrderedFastaReader reader(files, threads, mkhost::HostAllocator{pinned_alloc, pinned_free, &arena}, 2 * 64);
        // whole files are queried 64 at a time: the dense kernel scores sixteen per wave and lets the four waves of a workgroup
        // walk the same rows together (the later sets find them in the Infinity Cache): 80 ms for 64 queries against 100,000 genomes, 44 ms for 32; output stays
        // in list order
        vector<string> names;
        vector<OrderedFastaReader::Item> refs;
        uint64_t bytes = 0;
        auto flush = [&]() {
            if (refs.empty()) return;
            vector<const char *> p;
            vector<uint64_t> l;
            for (auto &r : refs) { p.push_back(r.data); l.push_back(r.len); }
            vector<mk_hit> hits;
            vector<uint32_t> nhits;
            run_query(p, l, 10, 10, 0.5 * threshold, hits, nhits);
            for (size_t i = 0; i < refs.size(); ++i)
                if (nhits[i]) out << names[i] << ":" << hit_text(hits.data() + i * 10, nhits[i]) << "\n";   // 506-511
            out << std::flush;
            for (auto &r : refs) reader.recycle(r);
            names.clear(); refs.clear(); bytes = 0;
        };
        for (size_t i = 0; i < files.size(); ++i) {
            OrderedFastaReader::Item item = reader.take(i);
            if (!item.exists) {
                cout << "File problem" << endl;
                reader.recycle(item);
            } else if (item.failed) {
                cout << "cannot read " << files[i] << endl;
                exit(1);
            } else if (item.len >= k) {
                bytes += item.len;
                names.push_back(files[i]); refs.push_back(item);
                if (refs.size() >= 64 || bytes > (1ull << 30)) flush();
            } else {
                reader.recycle(item);
            }
            cout << "-" << flush_stream();
        }
        flush();
    }

    // ---- exact mode -------------------------------------------------------------
    struct Pending { string seq, head; double jaccard, intersection; uint32_t genome; };

    // a genome file as ground_truth_batch sees it: contigs split at '>' lines (Miekki.cpp:805-812),
    // walked line by line over the raw text.  Callable from a helper thread.
    struct GenomeContigs { bool exists = false; vector<string> contigs; };
    GenomeContigs parse_contigs(const string &file) const
    {
        GenomeContigs gc;
        gc.exists = mkhost::file_exists(file);
        if (!gc.exists) return gc;
        vector<char> text, scratch;
        mkhost::read_file(file, text, scratch);
        string ref;
        ref.reserve(text.size());
        const char *pos = text.data(), *const end = text.data() + text.size();
        while (pos <= end) {                                                     // getline semantics
            const char *e = pos < end ? (const char *)memchr(pos, '\n', (size_t)(end - pos)) : nullptr;
            if (!e) e = end;
            if (e != pos && *pos == '>') {
                if (ref.size() >= k) { gc.contigs.push_back(std::move(ref)); ref = string(); }   // short contigs leak (806-812)
            } else {
                ref.append(pos, (size_t)(e - pos));
            }
            pos = e + 1;
        }
        if (ref.size() >= k) gc.contigs.push_back(std::move(ref));
        return gc;
    }

    // ground_truth_batch (Miekki.cpp:792-859) for all pending queries of one genome file.  The
    // reference rebuilds the file's k-mer set at every call (~7 s); here the set stays resident on
    // the GPU that verified the file last, so a file whose queries are flushed in several batches
    // (the flush-at-100 rule) is read, split and hashed once.
    vector<string> resident;                                     // per shard: file whose set B is loaded
    mk_ctx *exact_ctx(const vector<Pending> &v, size_t &shard)
    {
        // K7 runs on the GPU that owns the genome (any would do: the sets come from the file itself)
        shard = v.empty() || group.ranked() ? 0 : group.owner(v[0].genome);
        if (resident.size() != group.shards()) resident.assign(group.shards(), string());
        return group.ctx(shard);
    }
    bool is_resident(const string &file, const vector<Pending> &v)
    {
        size_t shard;
        exact_ctx(v, shard);
        return !file.empty() && resident[shard] == file;
    }

    void ground_truth(const string &file, const vector<Pending> &v)
    {
        if (is_resident(file, v)) ground_truth(file, v, GenomeContigs{});
        else ground_truth(file, v, parse_contigs(file));
    }

    void ground_truth(const string &file, const vector<Pending> &v, const GenomeContigs &gc)
    {
        size_t shard;
        mk_ctx *ctx = exact_ctx(v, shard);
        if (resident[shard] != file) {
            if (!gc.exists) { cout << "File problem: " << file << endl; return; }
            vector<const char *> cp;
            vector<uint64_t> cl;
            for (auto &c : gc.contigs) { cp.push_back(c.data()); cl.push_back(c.size()); }
            resident[shard].clear();
            if (mk_exact_load_genome(ctx, cp.data(), cl.data(), (uint32_t)gc.contigs.size()) != MK_OK) die("exact mode failed");
            resident[shard] = file;
        }
        vector<const char *> qp;
        vector<uint64_t> ql;
        for (auto &q : v) { qp.push_back(q.seq.data()); ql.push_back(q.seq.size()); }
        vector<uint64_t> inter(v.size()), uni(v.size());
        if (mk_exact_query(ctx, qp.data(), ql.data(), (uint32_t)v.size(), inter.data(), uni.data()) != MK_OK)
            die("exact mode failed");
        for (size_t i = 0; i < v.size(); ++i) {
            if (!inter[i]) continue;                                             // 843
            const double real_jax = (double)inter[i] / (double)uni[i];
            out << real_jax << "\t" << v[i].jaccard << "\t" << (double)inter[i] << "\t" << v[i].intersection
                << "\t" << v[i].head << "\t" << file << "\n";                    // 853
        }
    }

    bool need_names()
    {
        if (file_names.size() == group.total()) return true;
        cout << "exact mode needs the genome files of the index: build it with -l in the same run "
                "(file names are not stored in an index file)" << endl;
        return false;
    }

    void exact_collect(const vector<string> &heads, const vector<const string *> &seqs, uint32_t nres,
                       uint32_t min_score)
    {
        vector<mk_hit> hits;
        vector<uint32_t> nhits;
        run_query(seqs, nres, min_score, (double)threshold, hits, nhits);
        if (!group.root()) return;                                   // (one process per GPU: rank 0 verifies and writes)
        // Same container, same insertion sequence and the same flush-at-100 rule as the
        // reference (Miekki.cpp:728-754, 779-785): with the same libstdc++ the lines of the
        // output file then come out in the reference's order, not just as the same set.
        unordered_map<string, vector<Pending>> batch;
        for (size_t q = 0; q < seqs.size(); ++q)
            for (uint32_t i = 0; i < nhits[q]; ++i) {
                const mk_hit &h = hits[q * nres + i];
                const string &file_name = file_names[h.genome];
                vector<Pending> &v = batch[file_name];
                v.push_back(Pending{*seqs[q], heads[q], h.jaccard, h.intersection, h.genome});
                if (v.size() >= 100) { ground_truth(file_name, v); v.clear(); }
            }
        // the genome file of the NEXT entry is read and split while the device works on this one
        vector<const std::pair<const string, vector<Pending>> *> todo;
        for (auto itr = batch.begin(); itr != batch.end(); ++itr)
            if (!itr->second.empty()) todo.push_back(&*itr);
        std::future<GenomeContigs> ahead;
        // (a file whose set is still resident from a flush above is not read again)
        auto parse_unless_resident = [this](const string &f, bool res) { return res ? GenomeContigs{} : parse_contigs(f); };
        if (!todo.empty())
            ahead = std::async(std::launch::async, parse_unless_resident, todo[0]->first, is_resident(todo[0]->first, todo[0]->second));
        for (size_t i = 0; i < todo.size(); ++i) {
            GenomeContigs gc = ahead.get();
            if (i + 1 < todo.size())
                ahead = std::async(std::launch::async, parse_unless_resident, todo[i + 1]->first, false);
            ground_truth(todo[i]->first, todo[i]->second, gc);
        }
        out << flush;
    }

    // ---- Miekki.cpp:723-759
    void query_file_exact(const string &path)
    {
        if (!mkhost::file_exists(path)) { cout << "File problem" << endl; return; }
        if (!need_names()) return;
        vector<string> heads, seqs, kh;
        read_records(path, heads, seqs);
        vector<const string *> kept;
        for (size_t i = 0; i < seqs.size(); ++i)
            if (seqs[i].size() >= k && nucleotide_start(seqs[i])) { kept.push_back(&seqs[i]); kh.push_back(heads[i]); }
        exact_collect(kh, kept, 5, 10);
    }

    // ---- Miekki.cpp:616-645, 763-788
    void query_file_of_file_exact(const string &list)
    {
        if (!mkhost::file_exists(list)) { cout << "Missed file of file: " << list << endl; return; }
        if (!need_names()) return;
        string text;
        mkhost::read_text(list, text);
        vector<string> heads, refs;
        for (const string &fn : split_lines(text)) {
            if (fn.size() <= 3) continue;
            if (!mkhost::file_exists(fn)) {
                cout << "File problem" << endl;
            } else {
                string ftext;
                mkhost::read_text(fn, ftext);
                vector<string> lines = split_lines(ftext);
                string ref;
                for (size_t i = 1; i < lines.size(); ++i)
                    if (lines[i].size() >= k && nucleotide_start(lines[i])) ref += lines[i];   // 771-775
                heads.push_back(lines.empty() ? string() : lines[0]);
                refs.push_back(std::move(ref));
            }
            cout << "-" << flush_stream();
        }
        vector<const string *> ptrs;
        for (auto &r : refs) ptrs.push_back(&r);
        exact_collect(heads, ptrs, 5, 5);
    }
};

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) { help(); return 0; }
    // (the HIP runtime folds its streams onto four hardware queues unless told otherwise; the ingest has a dozen streams with
    // work at a time -- a batch's own, the upload streams, the build's -- and a copy that shares a queue with a long kernel of
    // another stream waits behind it: eight queues, unless the user has said something)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    string index_file, list_file, query_lines, query_list, output_file("out.txt"), index_dump;
    uint64_t H = 17, core_number = 8, kmer_size = 31, bloom_size = 33, fingerprint_size = 3;   // main.cpp:131
    double threshold = 200;
    bool exact_mode = false, threads_given = false;
    int c;
    while ((c = getopt(argc, argv, "i:l:a:h:t:f:k:s:b:o:ed:A:")) != -1) {
        switch (c) {
        case 'i': index_file = optarg; break;
        case 'l': list_file = optarg; break;
        case 'a': query_lines = optarg; break;
        case 'A': query_list = optarg; break;
        case 'o': output_file = optarg; break;
        case 'h': H = stoi(optarg); break;
        case 't': core_number = stoi(optarg); threads_given = true; break;
        case 'k': kmer_size = stoi(optarg); break;
        case 's': threshold = stof(optarg); break;
        case 'f': fingerprint_size = stoi(optarg); break;
        case 'b': bloom_size = stoi(optarg); break;
        case 'e': exact_mode = true; break;
        case 'd': index_dump = optarg; break;
        }
    }
    const vector<int> devices = mkhost::device_list();          // every visible GPU, or MIEKKI_DEVICES
    // one process per GPU?  (a launcher's environment: torch.distributed.run --no-python sets RANK / WORLD_SIZE / LOCAL_RANK)
    auto env_int = [](const char *a, const char *b, int dflt) { const char *e = getenv(a); if (!e) e = getenv(b); return e ? atoi(e) : dflt; };
    const int rank_world = env_int("MIEKKI_WORLD", "WORLD_SIZE", 1), rank_id = env_int("MIEKKI_RANK", "RANK", 0);
    const int rank_local = env_int("MIEKKI_LOCAL_RANK", "LOCAL_RANK", rank_id);
    const bool rank_mode = rank_world > 1 || getenv("MIEKKI_WORLD") != nullptr;
    if (rank_mode && (rank_id < 0 || rank_id >= rank_world)) { cout << "rank " << rank_id << " of " << rank_world << "?" << endl; return 1; }
    if (rank_mode && rank_id != 0) cout.setstate(std::ios_base::badbit);      // rank 0 speaks for all
    const unsigned reader_threads = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(core_number, 64));
    const uint32_t bit_per_min = (uint32_t)(5 + fingerprint_size);                              // main.cpp:184
    cout << "Using " << bit_per_min << " bits per minimizer, " << int_to_string(1ull << H) << " minimizers so "
         << int_to_string((uint32_t)(bit_per_min * (1u << H))) << " bits per sequences" << endl;
    auto start = chrono::system_clock::now();
    Driver drv;
    drv.threads = reader_threads;
    drv.threads_given = threads_given;
    if (rank_mode) {
        drv.rank_id = rank_id; drv.rank_world = rank_world; drv.rank_local = rank_local; drv.forced_rank_mode = true;
        drv.meet = mkhost::rendezvous_from_env();
        setenv("MIEKKI_COMM_BANNER_TO_STDERR", "1", 1);              // (stdout is compared with the reference's: mk_comm_create)
    }
    if (!index_file.empty()) {
        if (!mkhost::file_exists(index_file)) {
            cout << "File problem" << endl;
            return 1;
        }
        string err;
        vector<mk_ctx *> ctxs;
        if (rank_mode) {
            // one process per GPU: every rank reads the file and keeps the columns of its run of genomes (main.cpp:189-194,
            // Miekki.cpp:687-719 per rank).  A rank that cannot load still joins the communicator -- on a context of its
            // own making -- so that all ranks hear of it and leave together.
            const int device = rank_local < (int)devices.size() ? devices[rank_local] : rank_local;
            const bool loaded = mkhost::load_index(index_file, {device}, ctxs, err, reader_threads, rank_id, rank_world) == 0;
            if (!loaded) {
                mk_params p{31, 10, 8, 0, 0, device, 0, 0};
                mk_ctx *ctx = nullptr;
                if (mk_create(&p, &ctx) != MK_OK) die("cannot create a context");
                ctxs.assign(1, ctx);
            }
            drv.group.adopt(ctxs);
            drv.group.set_comm(drv.make_comm(ctxs[0]));
            drv.agree(loaded ? string() : (err.empty() ? string("unknown error") : err), "Index load failed: ");
        } else {
            if (mkhost::load_index(index_file, devices, ctxs, err, reader_threads) != 0) { cout << "Index load failed: " << err << endl; return 1; }
            drv.group.adopt(ctxs);
        }
        drv.finish_index(false);                            // the file holds the global Bloom filter already
        mk_params p;
        mk_get_params(drv.ctx0(), &p);
        drv.k = p.k; drv.threshold = p.threshold;          // -k -h -f -b -s come from the file (main.cpp:189-194)
        if (!rank_mode || rank_id == 0) drv.out.open(output_file.c_str());
        cout << "I output results in " << output_file << endl;
        cout << "Load sucessful" << endl;
    } else if (!list_file.empty()) {
        if (!rank_mode || rank_id == 0) drv.out.open(output_file.c_str());
        cout << "I output results in " << output_file << endl;
        drv.k = (uint32_t)kmer_size; drv.threshold = (uint32_t)threshold;
        drv.index_file_of_file(list_file, devices, [&](int device) {
            mk_params p{(uint32_t)kmer_size, (uint32_t)H, bit_per_min, (uint32_t)bloom_size, (uint32_t)threshold, device, 0, 0};
            mk_ctx *ctx = nullptr;
            const int st = mk_create(&p, &ctx);
            if (st == MK_ERR_UNSUPPORTED) { cout << "not implemented" << endl; exit(0); }          // Miekki.cpp:235-237
            if (st != MK_OK) die("cannot create the index");
            return ctx;
        });
    } else {
        cout << "What am I supposed to index ? use either -i or -l options please" << endl;
        help();
        return 0;
    }
    if (!index_dump.empty()) {
        cout << "I write this index on the disk for later" << endl;
        string err;
        const int rc = rank_mode ? mkhost::dump_index_ranked(drv.ctx0(), drv.group.comm(), index_dump, err, reader_threads)
                                 : mkhost::dump_index(drv.group.contexts(), index_dump, err, reader_threads);
        if (rc != 0) { cout << "Index dump failed: " << err << endl; return 1; }
    }
    auto end_index = chrono::system_clock::now();
    cout << "elapsed time: " << chrono::duration<double>(end_index - start).count() << "s\n";
    if (!query_lines.empty()) {
        if (exact_mode) {
            cout << "running in exact mode, actual intersection will be computed on hits found by the index" << endl;
            drv.query_file_exact(query_lines);
        } else {
            cout << "running in approx mode, intersection is estimated by the index" << endl;
            drv.query_file(query_lines);
        }
    } else if (!query_list.empty()) {
        if (exact_mode) {
            cout << "running in exact mode, actual intersection will be computed on hits found by the index" << endl;
            drv.query_file_of_file_exact(query_list);
        } else {
            cout << "running in approx mode, intersection is estimated by the index" << endl;
            drv.query_file_of_file(query_list);
        }
    } else {
        cout << "No query file, No queries" << endl;
    }
    auto end_query = chrono::system_clock::now();
    cout << "elapsed time: " << chrono::duration<double>(end_query - end_index).count() << "s\n";
    if (getenv("MIEKKI_VERBOSE") && (drv.group.shards() > 1 || drv.group.ranked()))
        cout << "[exchange] " << (drv.group.ranked() ? (size_t)drv.group.world() : drv.group.shards()) << " shards: " << drv.group.gather_bytes() << " bytes gathered, "
             << drv.group.rerun_queries() << " queries rerun with wide rows, " << drv.group.replayed_queries()
             << " answered from dense score rows" << endl;
    cout << "The end" << endl;
    drv.out.close();
    // One process, its work done and written: leave.  Taking the contexts apart first -- dozens of page-locked pieces, the device's
    // blocks, the streams, one runtime call each -- is 60-80 ms of a run that may be 300 (profiles/r6_cli_startup.txt), for memory
    // the driver takes back anyway when the process ends.  Ranks leave in order: their communicator is shared.
    if (!rank_mode) {
        for (mk_ctx *ctx : drv.group.contexts()) (void)mk_sync(ctx);   // (nothing of the library's is under way, on the device or on a thread)
        cout.flush();
        exit(0);                                             // (not _exit: a profiler's exit handlers write its files)
    }
    return 0;                                                // ~DeviceGroup releases the contexts
}
