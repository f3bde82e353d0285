// Test helper: run the host driver's OrderedFastaReader over a list of files and print,
// per file in list order, "<exists> <length> <fnv1a64 of the sequence> <failed>".
//   reader_dump <list> <threads> [window] [allocator budget in bytes, 0 = no allocator] [take only the first N] [packed]
// With "packed" the reader packs while it parses (2-bit codes + exception bits through a plain packer of
// this file's own, same contract as the library's mk_pack_append) and the line is
// "<exists> <length> <fnv1a64 of the sequence with every non-ACGT character as '?'> <failed> <dirty> <head as hex>".
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

// contract of mk_pack_append (include/miekki_hip.h), stated plainly
static int plain_pack(uint64_t *codes, uint64_t *except, uint64_t at, const char *chars, uint64_t n)
{
    int any = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t p = at + i;
        const char c = chars[i];
        const uint64_t code = c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
        const bool bad = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        if (p % 32 == 0) codes[p / 32] = 0;
        if (p % 64 == 0) except[p / 64] = 0;
        codes[p / 32] = (codes[p / 32] & ((1ull << (2 * (p % 32))) - 1)) | (code << (2 * (p % 32)));
        except[p / 64] = (except[p / 64] & ((1ull << (p % 64)) - 1)) | ((uint64_t)bad << (p % 64));
        any |= bad;
    }
    return any;
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
        const bool packed = argc > 6 && std::string(argv[6]) == "packed";
        mkhost::OrderedFastaReader reader(files, (unsigned)atoi(argv[2]), a, argc > 3 ? (size_t)atoi(argv[3]) : 4,
                                          packed ? &plain_pack : nullptr);
        for (size_t i = 0; i < files.size() && i < only; ++i) {
            mkhost::OrderedFastaReader::Item it = reader.take(i);
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
