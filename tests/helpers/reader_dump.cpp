// Test helper: run the host driver's OrderedFastaReader over a list of files and print,
// per file in list order, "<exists> <length> <fnv1a64 of the sequence> <failed>".
//   reader_dump <list> <threads> [window] [allocator budget in bytes, 0 = no allocator] [take only the first N] [packed]
// With "packed" the reader packs while it parses (2-bit codes + exception bits, unpacked again here) and the line is
// "<exists> <length> <fnv1a64 of the sequence with every non-ACGT character as '?'> <failed> <dirty> <head as hex>".
// With "share:<unit>:<ahead>" instead of "packed" the reader packs AND shares gzip'd files with the device in units of <unit>
// files (fasta_reader.hpp): a raw item's line is "raw <length> <fnv1a64 of the file's bytes>"; raw items are given back
// (raw_consumed) only when <unit> of them have been seen, as a device batch would.
// With a budget the reader gets an allocator that hands out at most that many bytes and then
// fails (the page-lock limit of pinned memory); with "take only N" the reader is destroyed
// while workers are still parked on the read-ahead bound (must not hang).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
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
        const bool share = sscanf(mode.c_str(), "share:%zu:%zu", &unit, &ahead) == 2;
        const bool packed = share || mode == "packed";
        mkhost::OrderedFastaReader reader(files, (unsigned)atoi(argv[2]), a, argc > 3 ? (size_t)atoi(argv[3]) : 4,
                                          packed, share, unit, ahead);
        size_t raw_held = 0;
        for (size_t i = 0; i < files.size() && i < only; ++i) {
            mkhost::OrderedFastaReader::Item it = reader.take(i);
            uint64_t h = 1469598103934665603ull;
            if (it.raw) {
                for (size_t j = 0; j < it.len; ++j) { h ^= (unsigned char)it.data[j]; h *= 1099511628211ull; }
                printf("raw %zu %016llx\n", it.len, (unsigned long long)h);
                reader.recycle(it);
                if (++raw_held >= unit) { reader.raw_consumed(raw_held); raw_held = 0; }
                continue;
            }
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
                continue;
            }
            for (size_t j = 0; j < it.len; ++j) { h ^= (unsigned char)it.data[j]; h *= 1099511628211ull; }
            printf("%d %zu %016llx %d\n", it.exists ? 1 : 0, it.len, (unsigned long long)h, it.failed ? 1 : 0);
            reader.recycle(it);
        }
    }                                                  // destructor: joins the workers, releases the pool
    printf("done live=%ld\n", (long)g_live.load());
    return 0;
}
