// Test helper: run the host driver's OrderedFastaReader over a list of files and print,
// per file in list order, "<exists> <length> <fnv1a64 of the sequence> <failed>".
//   reader_dump <list> <threads> [window] [allocator budget in bytes, 0 = no allocator] [take only the first N] [packed]
// With "packed" the reader packs while it parses (2-bit codes + exception bits, unpacked again here) and the line is
// "<exists> <length> <fnv1a64 of the sequence with every non-ACGT character as '?'> <failed> <dirty> <head as hex>".
// With "share:<unit>:<ahead>" instead of "packed" the reader packs AND shares gzip'd files with the device in units of <unit>
// files (fasta_reader.hpp): a raw item's line is "raw <length> <fnv1a64 of the file's bytes>"; raw items are given back
// (raw_consumed) only when <unit> of them have been seen, as a device batch would.
// With "sink:<unit>:<ahead>" the same through a RawSink (what the driver binds to mk_gz_open / mk_gz_stage / mk_gz_put): the
// files of a device unit are put piece by piece (lent pieces of 3,000 bytes, every fifth request refused) into a batch this
// helper keeps in memory; a raw item has no bytes of its own, its line is made from the batch's when the unit's last file
// has been taken (the lines still come in list order).
// With a budget the reader gets an allocator that hands out at most that many bytes and then
// fails (the page-lock limit of pinned memory); with "take only N" the reader is destroyed
// while workers are still parked on the read-ahead bound (must not hang).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <vector>

#include "fasta_reader.hpp"

static std::atomic<long long> g_budget{0};
static std::atomic<long> g_live{0};
static void *budget_alloc(void *, size_t bytes)
{
    if (g_budget.fetch_sub((long long)bytes) < (long long)bytes) return nullptr;
    ++g_live;
    return malloc(bytes);
}
static void budget_free(void *, void *p) { --g_live; free(p); }

// a stand-in for the device: a unit's files assembled in memory
struct FakeBatch { std::vector<std::vector<unsigned char>> files; std::vector<uint64_t> offsets; };
static std::mutex g_sink_m;
static std::vector<void *> g_lent;
static std::atomic<long> g_stage_calls{0};
static size_t g_piece = 3000;
static std::atomic<long> g_spans{0}, g_span_files{0};
static void *fake_open(void *, const uint64_t *sizes, uint32_t m, uint64_t *offsets)
{
    FakeBatch *b = new FakeBatch();
    uint64_t at = 0;
    for (uint32_t i = 0; i < m; ++i) {
        b->files.emplace_back((size_t)sizes[i], (unsigned char)0);
        offsets[i] = at;
        at += (sizes[i] + 16 + 15) / 16 * 16;                            // (the device's layout: sixteen zeros at least behind every file)
    }
    offsets[m] = at;
    b->offsets.assign(offsets, offsets + m + 1);
    return b;
}
static bool fake_put(void *, void *batch, uint32_t i, uint64_t at, const void *data, uint64_t bytes, bool staged);
// files [first, first + count) laid out as the batch's input is: taken apart again (what lies between two files must be zero)
static bool fake_put_span(void *u, void *batch, uint32_t first, uint32_t count, const void *data, uint64_t bytes, bool staged)
{
    FakeBatch *b = (FakeBatch *)batch;
    ++g_spans; g_span_files += count;
    bool ok = first + count <= b->files.size() && bytes <= b->offsets[first + count] - b->offsets[first];
    const unsigned char *p = (const unsigned char *)data;
    for (uint32_t k = 0; ok && k < count; ++k) {
        const uint64_t o = b->offsets[first + k] - b->offsets[first], n = b->files[first + k].size(), room = b->offsets[first + k + 1] - b->offsets[first + k];
        if (o + room > bytes) { ok = false; break; }
        // (a file that was given up is all zeros at its place; one that went is its bytes, then zeros)
        bool all_zero = true;
        for (uint64_t x = 0; x < n; ++x) all_zero = all_zero && p[o + x] == 0;
        if (!all_zero) memcpy(b->files[first + k].data(), p + o, (size_t)n);
        for (uint64_t x = n; x < room; ++x) ok = ok && p[o + x] == 0;
    }
    return fake_put(u, batch, first, 0, data, 0, staged) && ok;          // (gives the piece back)
}
static void *fake_stage(void *, void *, uint64_t *cap)
{
    if (g_stage_calls.fetch_add(1) % 5 == 4) return nullptr;           // (none to be had: the reader uses its own memory)
    void *p = malloc(g_piece);
    std::lock_guard<std::mutex> g(g_sink_m);
    g_lent.push_back(p);
    *cap = g_piece;
    return p;
}
static bool fake_put(void *, void *batch, uint32_t i, uint64_t at, const void *data, uint64_t bytes, bool staged)
{
    FakeBatch *b = (FakeBatch *)batch;
    bool ok = i < b->files.size() && at + bytes <= b->files[i].size();
    if (ok && bytes) memcpy(b->files[i].data() + at, data, (size_t)bytes);
    if (staged) {
        std::lock_guard<std::mutex> g(g_sink_m);
        bool found = false;
        for (size_t k = 0; k < g_lent.size(); ++k) if (g_lent[k] == data) { g_lent.erase(g_lent.begin() + (long)k); found = true; break; }
        if (!found) ok = false;
        else free(const_cast<void *>(data));
    }
    return ok;
}

static uint64_t fnv(const unsigned char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t j = 0; j < n; ++j) { h ^= p[j]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::vector<std::string> files;
    std::ifstream in(argv[1]);
    for (std::string l; std::getline(in, l);) if (!l.empty()) files.push_back(l);
    const long long budget = argc > 4 ? atoll(argv[4]) : 0;
    const size_t only = argc > 5 ? (size_t)atoll(argv[5]) : files.size();
    g_budget = budget;
    {
        mkhost::HostAllocator a{nullptr, nullptr, nullptr};
        if (budget > 0) a = mkhost::HostAllocator{budget_alloc, budget_free, nullptr};
        const std::string mode = argc > 6 ? argv[6] : "";
        size_t unit = 0, ahead = 0;
        const bool with_sink = sscanf(mode.c_str(), "sink:%zu:%zu", &unit, &ahead) == 2;
        const bool share = with_sink || sscanf(mode.c_str(), "share:%zu:%zu", &unit, &ahead) == 2;
        const bool packed = share || mode == "packed";
        mkhost::RawSink sink;
        if (mode.find(":span") != std::string::npos) g_piece = 20000;        // (room for a few of the test's files: spans form)
        if (with_sink) { sink.open = fake_open; sink.stage = fake_stage; sink.put = fake_put; if (mode.find(":span") != std::string::npos) sink.put_span = fake_put_span; }
        mkhost::OrderedFastaReader reader(files, (unsigned)atoi(argv[2]), a, argc > 3 ? (size_t)atoi(argv[3]) : 4,
                                          packed, share, unit, ahead, sink);
        size_t raw_held = 0;
        std::vector<mkhost::OrderedFastaReader::Item> held;             // a device unit's items until its last file has come
        auto line = [&](mkhost::OrderedFastaReader::Item &it) {
            if (it.raw && it.unit_batch) {                              // (its bytes are in the sink's batch)
                const auto &bytes = ((FakeBatch *)it.unit_batch)->files[it.unit_index];
                printf("raw %zu %016llx\n", bytes.size(), (unsigned long long)fnv(bytes.data(), bytes.size()));
                return;
            }
            if (it.raw) {
                printf("raw %zu %016llx\n", it.len, (unsigned long long)fnv((const unsigned char *)it.data, it.len));
                reader.recycle(it);
                if (++raw_held >= unit) { reader.raw_consumed(raw_held); raw_held = 0; }
                return;
            }
            uint64_t h = 1469598103934665603ull;
            if (packed && it.exists && !it.failed) {
                for (size_t j = 0; j < it.len; ++j) {
                    const bool bad = (it.except[j / 64] >> (j % 64)) & 1u;
                    const unsigned char c = bad ? '?' : "ACGT"[(it.codes[j / 32] >> (2 * (j % 32))) & 3u];
                    h ^= c; h *= 1099511628211ull;
                }
                printf("%d %zu %016llx %d %d ", 1, it.len, (unsigned long long)h, 0, it.dirty ? 1 : 0);
                for (size_t j = 0; j < 32 && j < it.len; ++j) printf("%02x", (unsigned char)it.head[j]);
                printf("\n");
                reader.recycle(it);
                return;
            }
            printf("%d %zu %016llx %d\n", it.exists ? 1 : 0, it.len, (unsigned long long)fnv((const unsigned char *)it.data, it.len), it.failed ? 1 : 0);
            reader.recycle(it);
        };
        for (size_t i = 0; i < files.size() && i < only; ++i) {
            mkhost::OrderedFastaReader::Item it = reader.take(i);
            if (it.unit_batch) {
                held.push_back(it);
                if (!it.unit_last) continue;
                for (auto &u : held) line(u);
                delete (FakeBatch *)it.unit_batch;
                reader.raw_consumed(held.size());
                held.clear();
                continue;
            }
            line(it);
        }
    }                                                  // destructor: joins the workers, releases the pool
    printf("done live=%ld spans=%ld of %ld files lent=%zu\n", (long)g_live.load(), (long)g_spans.load(), (long)g_span_files.load(), g_lent.size());
    return 0;
}
