#include "index_io.hpp"

#include "gzpar.hpp"

#include <sys/stat.h>
#include <zlib.h>

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

int dump_index(mk_ctx *ctx, const std::string &path, std::string &err, unsigned threads)
{
    mk_params p;
    if (mk_get_params(ctx, &p) != MK_OK) { err = mk_last_error(); return -1; }
    const uint32_t G = mk_index_size(ctx), W = p.fp_bits / 8, P = 1u << p.h;
    ParallelGzipWriter w(path, threads);
    if (!w.ok()) { err = "cannot open " + path; return -1; }
    Header hd{p.k, p.h, p.fp_bits, 5, G, p.bloom_log2, p.bloom_log2 ? 1ull << p.bloom_log2 : 0, 0, 0, p.threshold, 1};
    w.write(&hd, sizeof hd);
    bool ok = true;
    const uint64_t row = (uint64_t)G * W;
    const uint32_t rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, row ? kChunk / row : P));
    std::vector<uint8_t> buf((size_t)std::max<uint64_t>(rows * row, 1));
    for (uint32_t pb = 0; ok && pb < P; pb += rows) {
        const uint32_t pe = std::min(P, pb + rows);
        if (G && mk_index_export_columns(ctx, pb, pe, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; break; }
        w.write(buf.data(), (size_t)(pe - pb) * row);
    }
    std::vector<uint64_t> gs(G);
    std::vector<uint32_t> ss(G);
    if (ok && mk_index_export_sizes(ctx, gs.data(), ss.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    if (ok) w.write(gs.data(), (size_t)G * 8);
    const uint64_t nb = hd.bloom_bits / 8;
    buf.resize((size_t)std::min<uint64_t>(kChunk, std::max<uint64_t>(nb, 1)));
    for (uint64_t o = 0; ok && o < nb; o += kChunk) {
        const uint64_t e = std::min(nb, o + kChunk);
        if (mk_index_export_bloom(ctx, o, e, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; break; }
        w.write(buf.data(), (size_t)(e - o));
    }
    if (ok) w.write(ss.data(), (size_t)G * 4);
    if (!w.finish()) ok = false;
    if (!ok && err.empty()) err = "write error on " + path;
    return ok ? 0 : -1;
}

int load_index(const std::string &path, int device, mk_ctx **out, std::string &err, unsigned threads)
{
    *out = nullptr;
    ParallelGzipReader f(path, threads);
    if (!f.ok()) { err = "cannot open " + path; return -1; }
    Header hd;
    mk_ctx *ctx = nullptr;
    bool ok = f.read(&hd, sizeof hd);
    if (!ok) err = "truncated index header";
    if (ok) {
        mk_params p{hd.kmer_size, hd.h, hd.fp_bits, hd.bloom_log2, hd.threshold, device, 0, 0};
        if (mk_create(&p, &ctx) != MK_OK) { err = mk_last_error(); ok = false; }
    }
    const uint32_t G = hd.index_size, W = hd.fp_bits / 8, P = ok ? 1u << hd.h : 0;
    if (ok && mk_index_import_begin(ctx, G) != MK_OK) { err = mk_last_error(); ok = false; }
    const uint64_t row = (uint64_t)G * W;
    const uint32_t rows = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, row ? kChunk / row : P));
    std::vector<uint8_t> buf((size_t)std::max<uint64_t>(rows * row, 1));
    for (uint32_t pb = 0; ok && pb < P; pb += rows) {
        const uint32_t pe = std::min(P, pb + rows);
        ok = f.read(buf.data(), (size_t)(pe - pb) * row);
        if (!ok) { err = "truncated index columns"; break; }
        if (G && mk_index_import_columns(ctx, pb, pe, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    }
    std::vector<uint64_t> gs(G);
    std::vector<uint32_t> ss(G);
    if (ok && !(ok = f.read(gs.data(), (size_t)G * 8))) err = "truncated genome sizes";
    const uint64_t nb = ok ? hd.bloom_bits / 8 : 0;
    buf.resize((size_t)std::min<uint64_t>(kChunk, std::max<uint64_t>(nb, 1)));
    for (uint64_t o = 0; ok && o < nb; o += kChunk) {
        const uint64_t e = std::min(nb, o + kChunk);
        ok = f.read(buf.data(), (size_t)(e - o));
        if (!ok) { err = "truncated Bloom filter"; break; }
        if (mk_index_import_bloom(ctx, o, e, buf.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    }
    if (ok && !(ok = f.read(ss.data(), (size_t)G * 4))) err = "truncated sketch sizes";
    if (ok && G && mk_index_import_sizes(ctx, gs.data(), ss.data()) != MK_OK) { err = mk_last_error(); ok = false; }
    if (!ok) { mk_destroy(ctx); return -1; }
    *out = ctx;
    return 0;
}

}  // namespace mkhost
