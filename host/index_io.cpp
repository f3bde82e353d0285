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

int load_index(const std::string &path, const std::vector<int> &devices, std::vector<mk_ctx *> &out, std::string &err,
               unsigned threads)
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
    // genomes split over the devices in id order; never more shards than genomes
    const size_t D = std::max<size_t>(1, std::min<size_t>(devices.size(), std::max<uint32_t>(G, 1)));
    std::vector<uint32_t> at(D + 1, 0);
    for (size_t d = 0; d < D; ++d) {
        const uint64_t base = G / D, rem = G % D;
        at[d + 1] = at[d] + (uint32_t)(base + (d < rem ? 1 : 0));
    }
    for (size_t d = 0; ok && d < D; ++d) {
        mk_params p{hd.kmer_size, hd.h, hd.fp_bits, hd.bloom_log2, hd.threshold, devices[d], at[d], 0};
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
            for (size_t d = 0; import_ok && d < D; ++d) {
                const uint64_t prow = (uint64_t)(at[d + 1] - at[d]) * W;
                if (!prow) continue;
                part.resize((size_t)(pe - pb) * prow);
                for (uint32_t r = 0; r < pe - pb; ++r)
                    memcpy(part.data() + (uint64_t)r * prow, b + (uint64_t)r * row + (uint64_t)at[d] * W, prow);
                if (mk_index_import_columns(out[d], pb, pe, part.data()) != MK_OK) { import_err = mk_last_error(); import_ok = false; }
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
        for (size_t d = 0; ok && d < D; ++d)                        // the one global filter, on every shard
            if (mk_index_import_bloom(out[d], o, e, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    }
    if (ok && !(ok = f.read(ss.data(), (size_t)G * 4))) err = "truncated sketch sizes";
    if (trace) fprintf(stderr, "[load] sizes and Bloom filter read after %.2f s\n", since(t_begin));
    for (size_t d = 0; ok && d < D; ++d)
        if (at[d + 1] > at[d] && mk_index_import_sizes(out[d], gs.data() + at[d], ss.data() + at[d]) != MK_OK) {
            err = mk_last_error();
            ok = false;
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
