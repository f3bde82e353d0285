#include "index_io.hpp"

#include "fastz.hpp"
#include "gzpar.hpp"

#include <sys/stat.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace mkhost {

bool file_exists(const std::string &path)
{
    struct stat st;
    return stat(path.c_str(), &st) == 0;
}

bool read_text(const std::string &path, std::string &out)
{
    gzFile f = gzopen(path.c_str(), "rb");          // transparent for non-gzip input
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) out.append(buf.data(), (size_t)n);
    gzclose(f);
    return n == 0;
}

namespace {

#pragma pack(push, 1)
struct Header {                  // 39 bytes, little-endian, unpadded (Miekki.cpp:651-661)
    uint32_t kmer_size, h, fp_bits, mantis_bits, index_size, bloom_log2;
    uint64_t bloom_bits;
    uint8_t jaccard_estimation, containment_estimation;
    uint32_t threshold;
    uint8_t compressed;
};
#pragma pack(pop)
static_assert(sizeof(Header) == 39, "index header layout");

constexpr uint64_t kChunk = 64ull << 20;

}  // namespace

// The shards' columns side by side give the reference's column: row p of the file is the
// concatenation, in shard order, of every shard's row p.
int dump_index(const std::vector<mk_ctx *> &ctxs, const std::string &path, std::string &err, unsigned threads)
{
    mk_params p;
    if (ctxs.empty() || mk_get_params(ctxs[0], &p) != MK_OK) { err = ctxs.empty() ? "no context" : mk_last_error(); return -1; }
    const size_t D = ctxs.size();
    std::vector<uint32_t> Gd(D), at(D + 1, 0);
    for (size_t d = 0; d < D; ++d) { Gd[d] = mk_index_size(ctxs[d]); at[d + 1] = at[d] + Gd[d]; }
    const uint32_t G = at[D], W = p.fp_bits / 8, P = 1u << p.h;
    ParallelGzipWriter w(path, threads);
    if (!w.ok()) { err = "cannot open " + path; return -1; }
    Header hd{p.k, p.h, p.fp_bits, 5, G, p.bloom_log2, p.bloom_log2 ? 1ull << p.bloom_log2 : 0, 0, 0, p.threshold, 1};
    w.write(&hd, sizeof hd);
    w.flush_block();                                               // (the header as a member of its own: the columns start at a block boundary)
    bool ok = true;
    const uint64_t row = (uint64_t)G * W;
    const uint32_t rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, row ? kChunk / row : P));
    std::vector<uint8_t> buf((size_t)std::max<uint64_t>(rows * row, 1)), part;
    w.set_strategy(Z_HUFFMAN_ONLY);                                // fingerprints: nothing for LZ77 to find (gzpar.hpp)
    // MIEKKI_DUMP_LEVEL=0: the columns as stored blocks -- no deflate work (a 105 GB index then dumps at the rate the
    // rows leave the GPU), a file a third larger; any gzip reader, the reference's included, reads it all the same
    if (const char *e = getenv("MIEKKI_DUMP_LEVEL")) w.set_level(atoi(e) == 0 ? 0 : 1);
    // One GPU: the rows of a block are exported straight into PAGE-LOCKED blocks (a DMA at the PCIe rate, no copy on this
    // thread) that the writer deflates and writes from where they lie; a block comes back to the pool once its member is
    // on file.  (Through pageable buffers a 105 GB index left the GPU at 2.7 GB/s: that, not deflate, bounded the dump.)
    const uint32_t rows_blk = row && row <= ParallelGzipWriter::kBlock ? (uint32_t)(ParallelGzipWriter::kBlock / row) : 0;
    struct Pool { std::mutex m; std::condition_variable cv; std::vector<uint8_t *> free; } pool;
    std::vector<void *> pinned;
    if (D == 1 && G && rows_blk) {
        for (unsigned i = 0; i < threads + 4; ++i) {
            void *b = nullptr;
            if (mk_host_alloc(ctxs[0], ParallelGzipWriter::kBlock, &b) != MK_OK) break;
            pinned.push_back(b); pool.free.push_back((uint8_t *)b);
        }
    }
    for (uint32_t pb = 0; ok && pb < P; pb += rows) {
        if (pinned.size() >= 3) {
            uint8_t *blk;
            {
                std::unique_lock<std::mutex> g(pool.m);
                pool.cv.wait(g, [&] { return !pool.free.empty(); });
                blk = pool.free.back(); pool.free.pop_back();
            }
            const uint32_t pe1 = std::min(P, pb + rows_blk);
            auto give_back = [&pool, blk] { { std::lock_guard<std::mutex> g(pool.m); pool.free.push_back(blk); } pool.cv.notify_one(); };
            if (mk_index_export_columns(ctxs[0], pb, pe1, blk) != MK_OK) { err = mk_last_error(); ok = false; give_back(); break; }
            if (!w.write_block(blk, (size_t)(pe1 - pb) * row, give_back)) { w.write(blk, (size_t)(pe1 - pb) * row); give_back(); }
            pb = pe1 - rows;                                       // (the loop adds `rows`)
            continue;
        }
        const uint32_t pe = std::min(P, pb + rows);
        if (D == 1) {
            if (G && mk_index_export_columns(ctxs[0], pb, pe, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; break; }
        } else {
            for (size_t d = 0; ok && d < D; ++d) {
                if (!Gd[d]) continue;
                const uint64_t prow = (uint64_t)Gd[d] * W;
                part.resize((size_t)(pe - pb) * prow);
                if (mk_index_export_columns(ctxs[d], pb, pe, part.data()) != MK_OK) { err = mk_last_error(); ok = false; break; }
                for (uint32_t r = 0; r < pe - pb; ++r)
                    memcpy(buf.data() + (uint64_t)r * row + (uint64_t)at[d] * W, part.data() + (uint64_t)r * prow, prow);
            }
            if (!ok) break;
        }
        w.write(buf.data(), (size_t)(pe - pb) * row);
    }
    std::vector<uint64_t> gs(G);
    std::vector<uint32_t> ss(G);
    for (size_t d = 0; ok && d < D; ++d)
        if (mk_index_export_sizes(ctxs[d], gs.data() + at[d], ss.data() + at[d]) != MK_OK) { err = mk_last_error(); ok = false; }
    w.set_strategy(Z_DEFAULT_STRATEGY);
    w.set_level(1);
    if (ok) w.write(gs.data(), (size_t)G * 8);
    // the filter: the cells a 2k-bit k-mer can reach come from the device (every shard holds the global filter), the rest
    // of the reference's 2^(b-3) bytes -- 960 MiB of the 1 GiB at k = 31, b = 33 -- are zeros by construction
    const uint64_t nb = hd.bloom_bits / 8, reach = std::min<uint64_t>(nb, mk_bloom_reachable_bytes(ctxs[0]));
    buf.resize((size_t)std::min<uint64_t>(kChunk, std::max<uint64_t>(reach, 1)));
    for (uint64_t o = 0; ok && o < reach; o += kChunk) {
        const uint64_t e = std::min(reach, o + kChunk);
        if (mk_index_export_bloom(ctxs[0], o, e, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; break; }
        w.write(buf.data(), (size_t)(e - o));
    }
    if (ok) w.write_zeros((size_t)(nb - reach));
    if (ok) w.write(ss.data(), (size_t)G * 4);
    if (!w.finish()) ok = false;                                   // (every block is back in the pool after this)
    for (void *b : pinned) mk_host_free(ctxs[0], b);
    if (!ok && err.empty()) err = "write error on " + path;
    return ok ? 0 : -1;
}

// ---- -d with one process per GPU
namespace {

// every rank's `bytes` bytes -> all ranks (host to host through the communicator's device buffers).  A rank whose own part
// fails (its upload, say) still ENTERS the collective -- with whatever its buffer holds -- and reports afterwards: the other
// ranks must not be left waiting in it.  Only the buffer itself (a few kilobytes) has to exist before.
bool allgather_host(mk_ctx *ctx, mk_comm *comm, const void *mine, uint64_t bytes, std::vector<uint8_t> &all, std::string &err)
{
    const int W = mk_comm_world(comm);
    all.assign((size_t)bytes * W, 0);
    void *d = nullptr;
    if (mk_dev_alloc(ctx, bytes * (W + 1), &d) != MK_OK) { err = mk_last_error(); return false; }
    std::string local;
    if (mk_dev_upload(ctx, (uint8_t *)d + bytes * W, mine, bytes) != MK_OK) local = mk_last_error();
    bool ok = mk_comm_allgather(comm, (uint8_t *)d + bytes * W, bytes, d) == MK_OK;
    if (!ok) err = mk_last_error();
    if (ok && mk_dev_download(ctx, all.data(), d, bytes * W) != MK_OK) { ok = false; err = mk_last_error(); }
    if (ok && !local.empty()) { ok = false; err = local; }
    mk_dev_free(ctx, d);
    return ok;
}

}  // namespace

int dump_index_ranked(mk_ctx *ctx, mk_comm *comm, const std::string &path, std::string &err, unsigned threads)
{
    mk_params p;
    if (!ctx || !comm || mk_get_params(ctx, &p) != MK_OK) { err = "no context"; return -1; }
    const int world = mk_comm_world(comm), rank = mk_comm_rank(comm);
    const uint32_t W = p.fp_bits / 8, P = 1u << p.h, Gm = mk_index_size(ctx);
    std::vector<uint8_t> all;
    if (!allgather_host(ctx, comm, &Gm, 4, all, err)) return -1;
    std::vector<uint32_t> Gd(world), at(world + 1, 0);
    uint32_t maxG = 0;
    for (int r = 0; r < world; ++r) { memcpy(&Gd[r], all.data() + 4 * r, 4); at[r + 1] = at[r] + Gd[r]; maxG = std::max(maxG, Gd[r]); }
    const uint32_t G = at[world];
    const uint64_t row = (uint64_t)G * W, mrow = (uint64_t)maxG * W;
    const uint32_t rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, mrow ? kChunk / mrow : P));
    const uint64_t blk = std::max<uint64_t>(16, (rows * mrow + 15) / 16 * 16);          // every rank's share of a block of rows, padded alike
    std::string my_err;                                            // (a rank that fails keeps standing in the collectives: all leave together)
    void *d_send = nullptr, *d_recv = nullptr;
    if (mk_dev_alloc(ctx, blk, &d_send) != MK_OK || (rank == 0 && mk_dev_alloc(ctx, blk * world, &d_recv) != MK_OK)) my_err = mk_last_error();
    std::unique_ptr<ParallelGzipWriter> w;
    if (rank == 0) {
        w.reset(new ParallelGzipWriter(path, threads));
        if (!w->ok()) { my_err = "cannot open " + path; w.reset(); }
    }
    if (w) {
        Header hd{p.k, p.h, p.fp_bits, 5, G, p.bloom_log2, p.bloom_log2 ? 1ull << p.bloom_log2 : 0, 0, 0, p.threshold, 1};
        w->write(&hd, sizeof hd);
        w->flush_block();
        w->set_strategy(Z_HUFFMAN_ONLY);
        if (const char *e = getenv("MIEKKI_DUMP_LEVEL")) w->set_level(atoi(e) == 0 ? 0 : 1);
    }
    std::vector<uint8_t> mine((size_t)blk), got(rank == 0 ? (size_t)(blk * world) : 0), buf((size_t)std::max<uint64_t>(rows * row, 1));
    bool coll_ok = true;
    auto exchange = [&]() {                                        // mine -> rank 0's got[world][blk]
        if (mk_dev_upload(ctx, d_send, mine.data(), blk) != MK_OK && my_err.empty()) my_err = mk_last_error();
        if (mk_comm_gather(comm, d_send, blk, d_recv, 0) != MK_OK) { coll_ok = false; return; }
        if (rank == 0 && mk_dev_download(ctx, got.data(), d_recv, blk * world) != MK_OK && my_err.empty()) my_err = mk_last_error();
    };
    // (a rank without buffers cannot take part in a gather of blk bytes: that failure is settled first)
    std::string first;
    {
        char msg[256] = {0};
        snprintf(msg, sizeof msg, "%s", my_err.c_str());
        if (!allgather_host(ctx, comm, msg, sizeof msg, all, err)) { mk_dev_free(ctx, d_send); mk_dev_free(ctx, d_recv); return -1; }
        for (int r = 0; r < world && first.empty(); ++r) first = std::string((const char *)all.data() + 256 * r, strnlen((const char *)all.data() + 256 * r, 255));
    }
    for (uint32_t pb = 0; first.empty() && coll_ok && pb < P; pb += rows) {
        const uint32_t pe = std::min(P, pb + rows);
        if (Gm && my_err.empty() && mk_index_export_columns(ctx, pb, pe, mine.data()) != MK_OK) my_err = mk_last_error();
        exchange();
        if (!coll_ok) break;
        if (w && my_err.empty()) {
            for (int r = 0; r < world; ++r) {
                const uint64_t prow = (uint64_t)Gd[r] * W;
                if (!prow) continue;
                const uint8_t *src = got.data() + (uint64_t)r * blk;
                for (uint32_t i = 0; i < pe - pb; ++i) memcpy(buf.data() + (uint64_t)i * row + (uint64_t)at[r] * W, src + (uint64_t)i * prow, prow);
            }
            w->write(buf.data(), (size_t)(pe - pb) * row);
        }
    }
    // the sizes: every rank's genome sizes and sketch sizes, in blocks of what fits
    std::vector<uint64_t> gs(G), gs_m(Gm);
    std::vector<uint32_t> ss(G), ss_m(Gm);
    if (first.empty() && coll_ok) {
        if (Gm && my_err.empty() && mk_index_export_sizes(ctx, gs_m.data(), ss_m.data()) != MK_OK) my_err = mk_last_error();
        const uint32_t per = (uint32_t)std::max<uint64_t>(1, (blk - 0) / 12);
        for (uint32_t g0 = 0; coll_ok && g0 < maxG; g0 += per) {
            const uint32_t n_m = Gm > g0 ? std::min(per, Gm - g0) : 0;
            if (n_m) { memcpy(mine.data(), gs_m.data() + g0, (size_t)n_m * 8); memcpy(mine.data() + (size_t)per * 8, ss_m.data() + g0, (size_t)n_m * 4); }
            exchange();
            if (!coll_ok || rank != 0) continue;
            for (int r = 0; r < world; ++r) {
                const uint32_t n_r = Gd[r] > g0 ? std::min(per, Gd[r] - g0) : 0;
                if (!n_r) continue;
                memcpy(gs.data() + at[r] + g0, got.data() + (uint64_t)r * blk, (size_t)n_r * 8);
                memcpy(ss.data() + at[r] + g0, got.data() + (uint64_t)r * blk + (size_t)per * 8, (size_t)n_r * 4);
            }
        }
    }
    if (w && first.empty() && coll_ok && my_err.empty()) {
        w->set_strategy(Z_DEFAULT_STRATEGY);
        w->set_level(1);
        w->write(gs.data(), (size_t)G * 8);
        const uint64_t nb = (p.bloom_log2 ? 1ull << p.bloom_log2 : 0) / 8, reach = std::min<uint64_t>(nb, mk_bloom_reachable_bytes(ctx));
        std::vector<uint8_t> bb((size_t)std::min<uint64_t>(kChunk, std::max<uint64_t>(reach, 1)));
        for (uint64_t o = 0; my_err.empty() && o < reach; o += kChunk) {        // (every rank holds the global filter: rank 0's own)
            const uint64_t e = std::min(reach, o + kChunk);
            if (mk_index_export_bloom(ctx, o, e, bb.data()) != MK_OK) { my_err = mk_last_error(); break; }
            w->write(bb.data(), (size_t)(e - o));
        }
        w->write_zeros((size_t)(nb - reach));
        w->write(ss.data(), (size_t)G * 4);
    }
    if (w && !w->finish() && my_err.empty()) my_err = "write error on " + path;
    if (d_send) mk_dev_free(ctx, d_send);
    if (d_recv) mk_dev_free(ctx, d_recv);
    if (!coll_ok) { err = std::string("exchange failed: ") + mk_last_error(); if (rank == 0) remove(path.c_str()); return -1; }
    if (first.empty()) {                                           // how everybody fared: the same answer on every rank
        char msg[256] = {0};
        snprintf(msg, sizeof msg, "%s", my_err.c_str());
        if (!allgather_host(ctx, comm, msg, sizeof msg, all, err)) return -1;
        for (int r = 0; r < world && first.empty(); ++r) first = std::string((const char *)all.data() + 256 * r, strnlen((const char *)all.data() + 256 * r, 255));
    }
    if (!first.empty()) { err = first; if (rank == 0) remove(path.c_str()); return -1; }
    return 0;
}

int load_index(const std::string &path, const std::vector<int> &devices, std::vector<mk_ctx *> &out, std::string &err,
               unsigned threads, int slice_rank, int slice_world)
{
    out.clear();
    // MIEKKI_IO_TRACE=1: where a load spends its time (stderr)
    const bool trace = getenv("MIEKKI_IO_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    const auto t_begin = now();
    ParallelGzipReader f(path, threads);
    if (!f.ok()) { err = "cannot open " + path; return -1; }
    Header hd;
    bool ok = f.read(&hd, sizeof hd);
    if (!ok) { err = "truncated index header"; return -1; }
    if (devices.empty()) { err = "no device to load the index onto"; return -1; }
    // the header decides shapes and the shard table below: check it before it is used (mk_create would
    // reject the same values, but only after P, W and the split were computed from them)
    if (hd.h < 1 || hd.h > 28 || (hd.fp_bits != 8 && hd.fp_bits != 16)) {
        err = "index header is not one this build reads (h " + std::to_string(hd.h) + ", " + std::to_string(hd.fp_bits) +
              " bits per fingerprint)";
        return -1;
    }
    const uint32_t G = hd.index_size, W = hd.fp_bits / 8;
    // genomes split over the devices in id order; never more shards than genomes -- or, one process per GPU (slice_world
    // ranks, this one slice_rank): over the RANKS, every one of which reads the file and keeps the columns of its own run
    // of genomes (a rank beyond the genomes keeps none: it still needs a context to stand in the communicator)
    const bool sliced = slice_world > 0;
    if (sliced && (slice_rank < 0 || slice_rank >= slice_world)) { err = "rank " + std::to_string(slice_rank) + " of " + std::to_string(slice_world); return -1; }
    const size_t D = sliced ? (size_t)slice_world : std::max<size_t>(1, std::min<size_t>(devices.size(), std::max<uint32_t>(G, 1)));
    std::vector<uint32_t> at(D + 1, 0);
    for (size_t d = 0; d < D; ++d) {
        const uint64_t base = G / D, rem = G % D;
        at[d + 1] = at[d] + (uint32_t)(base + (d < rem ? 1 : 0));
    }
    std::vector<size_t> mine;                                      // out[i] holds shard mine[i]
    if (sliced) mine.push_back((size_t)slice_rank);
    else for (size_t d = 0; d < D; ++d) mine.push_back(d);
    for (size_t i = 0; ok && i < mine.size(); ++i) {
        const size_t d = mine[i];
        mk_params p{hd.kmer_size, hd.h, hd.fp_bits, hd.bloom_log2, hd.threshold, devices[sliced ? 0 : d], at[d], 0};
        mk_ctx *ctx = nullptr;
        if (mk_create(&p, &ctx) != MK_OK) { err = mk_last_error(); ok = false; break; }
        out.push_back(ctx);
        if (trace) fprintf(stderr, "[load] context %zu created after %.2f s\n", d, since(t_begin));
        if (mk_index_import_begin(ctx, at[d + 1] - at[d]) != MK_OK) { err = mk_last_error(); ok = false; }
        if (trace) fprintf(stderr, "[load] its matrix allocated after %.2f s\n", since(t_begin));
    }
    const uint32_t P = ok ? 1u << hd.h : 0;
    const uint64_t row = (uint64_t)G * W;
    // The columns pass through TWO page-locked chunks of as many whole writer blocks as there are threads to inflate them
    // (the reader inflates a member straight into the chunk when the request covers it -- and a chunk cut at the writer's
    // block size covers whole members only): while the members of one chunk are read and inflated, the chunk before it
    // goes to the device(s) on a thread of its own.
    const uint64_t blk_rows = row && row <= ParallelGzipWriter::kBlock ? ParallelGzipWriter::kBlock / row : 0;
    const uint64_t want_rows = blk_rows ? blk_rows * std::max(2u, f.parallel() ? f.threads() : 2u) : (row ? std::max<uint64_t>(1, kChunk / row) : P);
    const uint32_t rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, want_rows));
    const size_t chunk_bytes = (size_t)std::max<uint64_t>(rows * row, 1);
    void *pinned[2] = {nullptr, nullptr};
    std::vector<uint8_t> pageable[2];
    uint8_t *cbuf[2];
    cbuf[0] = cbuf[1] = nullptr;
    double t_read = 0, t_wait = 0;
    // ---- the columns, inflated on the GPU.  This program's own dumps keep them as Huffman-only deflate blocks and list
    // the blocks in the members' extra fields (gzpar.hpp, "MH"): when every member of the column stretch is such a one and
    // holds whole rows (the dump's blocks do), the members' bytes go to the device as they are -- read by `threads` threads
    // into one of two page-locked buffers while the chunk before is on the device -- and mk_index_import_columns_huffman
    // inflates them there, one lane per 16 KiB block; the host folds the blocks' CRC remainders and holds them against
    // every member's trailer.  Anything else (the reference's dumps, stored columns, an index spread over several
    // devices, MIEKKI_LOAD_INFLATE=host) is inflated by the reader's threads below.
    uint32_t rows_done = 0;
    {
        std::vector<ParallelGzipReader::Member> mem;
        const char *how = getenv("MIEKKI_LOAD_INFLATE");
        size_t K = 0;
        uint64_t covered = 0;
        if (ok && D == 1 && G && row && f.parallel() && !(how && !strcmp(how, "host")) && f.list_members(mem)) {
            while (K < mem.size() && covered < (uint64_t)P * row && mem[K].indexed && mem[K].isize % row == 0 && mem[K].payload < (1u << 29)) covered += mem[K++].isize;
            if (covered != (uint64_t)P * row) K = 0;
        }
        if (K) {
            const uint64_t cap_out = 768ull << 20;
            void *stage[2] = {nullptr, nullptr};
            for (int q = 0; q < 2; ++q) if (mk_host_alloc(out[0], cap_out, &stage[q]) != MK_OK) stage[q] = nullptr;
            if (stage[0] && stage[1]) {
                struct Chunk {
                    uint32_t pb = 0, pe = 0; size_t m0 = 0, m1 = 0; uint64_t bytes = 0;
                    std::vector<uint64_t> pay_off; std::vector<mk_huff_block> blocks; std::vector<uint8_t> lens; std::vector<uint32_t> crc;
                };
                Chunk ch[2];
                std::thread up;
                bool up_ok = true;
                std::string up_err;
                const uint32_t k16 = crc32_shift_factor(HuffIndex::kSub);
                size_t m = 0;
                uint32_t pb = 0;
                int cur = 0;
                const auto t_gpu = now();
                double t_rd = 0, t_up = 0;
                while (ok && m < K) {
                    Chunk &c = ch[cur];
                    c = Chunk();
                    c.pb = pb; c.m0 = m;
                    uint64_t out_bytes = 0;
                    while (m < K && out_bytes + mem[m].isize <= cap_out && c.bytes + mem[m].payload + 16 <= cap_out) {
                        c.pay_off.push_back(c.bytes);
                        c.bytes += (mem[m].payload + 15) / 16 * 16;
                        out_bytes += mem[m].isize;
                        ++m;
                    }
                    if (m == c.m0) { err = "an index member larger than the loader's chunks"; ok = false; break; }
                    c.m1 = m;
                    c.pe = pb + (uint32_t)(out_bytes / row);
                    pb = c.pe;
                    // the members' streams, by `threads` threads; the block list beside it
                    auto t0 = now();
                    {
                        std::vector<std::thread> rd;
                        std::atomic<size_t> next{c.m0};
                        std::atomic<bool> rd_ok{true};
                        uint8_t *dst = (uint8_t *)stage[cur];
                        for (unsigned t = 0; t < std::max(1u, threads); ++t)
                            rd.emplace_back([&] {
                                for (size_t i; (i = next.fetch_add(1)) < c.m1;)
                                    if (!f.read_raw(mem[i].at, dst + c.pay_off[i - c.m0], (size_t)mem[i].payload)) rd_ok = false;
                            });
                        uint64_t out_at = 0;
                        for (size_t i = c.m0; i < c.m1; ++i) {
                            const ParallelGzipReader::Member &mm = mem[i];
                            size_t sub = 0;
                            for (uint64_t sp = 0; sp * HuffIndex::kSuper < mm.isize; ++sp) {
                                const uint64_t sbytes = std::min<uint64_t>(HuffIndex::kSuper, mm.isize - sp * HuffIndex::kSuper);
                                const uint32_t code = (uint32_t)(c.lens.size() / 257);
                                c.lens.insert(c.lens.end(), mm.lens.begin() + sp * 257, mm.lens.begin() + (sp + 1) * 257);
                                for (uint32_t q = 0; q < 64; ++q) {
                                    mk_huff_block b{0, 0, 0, code};
                                    if ((uint64_t)q * HuffIndex::kSub < sbytes) {
                                        b.bit = c.pay_off[i - c.m0] * 8 + mm.sym_bit[sub++];
                                        b.out = out_at + sp * HuffIndex::kSuper + (uint64_t)q * HuffIndex::kSub;
                                        b.out_len = (uint32_t)std::min<uint64_t>(HuffIndex::kSub, sbytes - (uint64_t)q * HuffIndex::kSub);
                                    }
                                    c.blocks.push_back(b);
                                }
                            }
                            out_at += mm.isize;
                        }
                        c.crc.assign(c.blocks.size(), 0);
                        for (auto &t : rd) t.join();
                        if (!rd_ok) { err = "cannot read the index members"; ok = false; }
                    }
                    t_rd += since(t0);
                    t0 = now();
                    if (up.joinable()) up.join();
                    t_up += since(t0);
                    if (!up_ok) break;
                    if (!ok) break;
                    up = std::thread([&, cur] {
                        Chunk &u = ch[cur];
                        uint32_t bad = 0;
                        if (mk_index_import_columns_huffman(out[0], u.pb, u.pe, (const uint8_t *)stage[cur], u.bytes, u.blocks.data(), (uint32_t)u.blocks.size(),
                                                            u.lens.data(), (uint32_t)(u.lens.size() / 257), u.crc.data(), &bad) != MK_OK) {
                            up_err = mk_last_error(); up_ok = false; return;
                        }
                        if (bad) { up_err = "corrupt index columns (a block does not decode)"; up_ok = false; return; }
                        size_t slot = 0;
                        for (size_t i = u.m0; i < u.m1 && up_ok; ++i) {      // every member's CRC-32 from its blocks' remainders
                            uint32_t r = 0;
                            for (uint64_t left = mem[i].isize; left; slot += 64) {
                                const uint64_t sbytes = std::min<uint64_t>(HuffIndex::kSuper, left);
                                for (uint32_t q = 0; (uint64_t)q * HuffIndex::kSub < sbytes; ++q) {
                                    const uint64_t len = std::min<uint64_t>(HuffIndex::kSub, sbytes - (uint64_t)q * HuffIndex::kSub);
                                    r = (len == HuffIndex::kSub ? crc32_shift_by(k16, r) : crc32_shift(r, len)) ^ u.crc[slot + q];
                                }
                                left -= sbytes;
                            }
                            if (crc32_from_raw(r, mem[i].isize) != mem[i].crc) { up_err = "corrupt index columns (CRC)"; up_ok = false; }
                        }
                    });
                    cur ^= 1;
                }
                if (up.joinable()) up.join();
                if (ok && !up_ok) { err = up_err; ok = false; }
                if (ok) { rows_done = P; ok = f.seek_member(K < mem.size() ? mem[K].begin : mem[K - 1].at + mem[K - 1].payload + 8); }
                if (trace)
                    fprintf(stderr, "[load] columns inflated on the GPU: %.2f s (%zu members; reading them %.2f s, waiting for the device %.2f s)\n", since(t_gpu), K,
                            t_rd, t_up);
            }
            for (int q = 0; q < 2; ++q) if (stage[q]) mk_host_free(out[0], stage[q]);
        }
    }
    if (trace) fprintf(stderr, "[load] columns start after %.2f s\n", since(t_begin));
    const auto t_cols = now();
    for (int k = 0; k < 2 && rows_done < P; ++k) {               // (the chunks of the host's inflate path: only when it runs)
        if (ok && !out.empty() && mk_host_alloc(out[0], chunk_bytes, &pinned[k]) != MK_OK) pinned[k] = nullptr;
        if (!pinned[k]) pageable[k].resize(chunk_bytes);
        cbuf[k] = pinned[k] ? (uint8_t *)pinned[k] : pageable[k].data();
    }
    std::thread importer;
    bool import_ok = true;
    std::string import_err;
    int k = 0;
    for (uint32_t pb = rows_done; ok && pb < P; pb += rows, k ^= 1) {
        const uint32_t pe = std::min(P, pb + rows);
        uint8_t *const b = cbuf[k];
        auto t0 = now();
        ok = f.read(b, (size_t)(pe - pb) * row);
        t_read += since(t0);
        t0 = now();
        if (importer.joinable()) importer.join();                 // the chunk before this one is on the device(s)
        t_wait += since(t0);
        if (!ok) { err = "truncated index columns"; break; }
        if (!import_ok) break;
        importer = std::thread([&, b, pb, pe] {
            if (D == 1) {
                if (G && mk_index_import_columns(out[0], pb, pe, b) != MK_OK) { import_err = mk_last_error(); import_ok = false; }
                return;
            }
            std::vector<uint8_t> part;
            for (size_t i = 0; import_ok && i < mine.size(); ++i) {
                const size_t d = mine[i];
                const uint64_t prow = (uint64_t)(at[d + 1] - at[d]) * W;
                if (!prow) continue;
                part.resize((size_t)(pe - pb) * prow);
                for (uint32_t r = 0; r < pe - pb; ++r)
                    memcpy(part.data() + (uint64_t)r * prow, b + (uint64_t)r * row + (uint64_t)at[d] * W, prow);
                if (mk_index_import_columns(out[i], pb, pe, part.data()) != MK_OK) { import_err = mk_last_error(); import_ok = false; }
            }
        });
    }
    if (importer.joinable()) importer.join();
    if (trace)
        fprintf(stderr, "[load] columns %.2f s (chunks of %.0f MiB: read + inflate %.2f s, waiting for the upload of the chunk before %.2f s)\n",
                since(t_cols), chunk_bytes / 1048576.0, t_read, t_wait);
    if (ok && !import_ok) { err = import_err; ok = false; }
    for (int q = 0; q < 2; ++q) if (pinned[q]) mk_host_free(out[0], pinned[q]);
    std::vector<uint8_t> buf;
    std::vector<uint64_t> gs(G);
    std::vector<uint32_t> ss(G);
    if (ok && !(ok = f.read(gs.data(), (size_t)G * 8))) err = "truncated genome sizes";
    const uint64_t nb = ok ? hd.bloom_bits / 8 : 0;
    const uint64_t bchunk = std::max<uint64_t>(kChunk, (uint64_t)ParallelGzipWriter::kBlock * std::min(8u, f.threads()));
    buf.resize((size_t)std::min<uint64_t>(bchunk, std::max<uint64_t>(nb, 1)));
    for (uint64_t o = 0; ok && o < nb; o += bchunk) {
        const uint64_t e = std::min(nb, o + bchunk);
        ok = f.read(buf.data(), (size_t)(e - o));
        if (!ok) { err = "truncated Bloom filter"; break; }
        for (size_t i = 0; ok && i < out.size(); ++i)               // the one global filter, on every shard
            if (mk_index_import_bloom(out[i], o, e, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    }
    if (ok && !(ok = f.read(ss.data(), (size_t)G * 4))) err = "truncated sketch sizes";
    if (trace) fprintf(stderr, "[load] sizes and Bloom filter read after %.2f s\n", since(t_begin));
    for (size_t i = 0; ok && i < mine.size(); ++i) {
        const size_t d = mine[i];
        if (at[d + 1] > at[d] && mk_index_import_sizes(out[i], gs.data() + at[d], ss.data() + at[d]) != MK_OK) {
            err = mk_last_error();
            ok = false;
        }
    }
    if (!ok) {
        for (mk_ctx *c : out) mk_destroy(c);
        out.clear();
        return -1;
    }
    if (trace) fprintf(stderr, "[load] done after %.2f s\n", since(t_begin));
    return 0;
}

}  // namespace mkhost
