// -d / -i of an index at BASELINE config 3's size, through the `miekki` binary's own writer and reader
// (host/index_io.cpp: dump_index / load_index) -- without 100,000 FASTA files: the index is built from the synthetic
// genomes of SURVEY.md 8d on the device (mk_index_append_synthetic), dumped, destroyed, loaded back and compared.
//     index_io_bench <genomes> <h> <fp_bits> <threads> <path>
// Prints the build, dump and load times, the file size, and whether the loaded index answers like the built one.
// (not part of the product; make -C miekki_amd/csrc tools)
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "index_io.hpp"
#include "miekki_hip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: index_io_bench <genomes> <h> <fp_bits> <threads> <path> [keep]\n"); return 2; }
    const uint32_t G = (uint32_t)atol(argv[1]);
    mk_params p;
    memset(&p, 0, sizeof p);
    p.k = 31; p.h = (uint32_t)atoi(argv[2]); p.fp_bits = (uint32_t)atoi(argv[3]); p.bloom_log2 = 33; p.threshold = 200;
    const unsigned threads = (unsigned)atoi(argv[4]);
    const std::string path = argv[5];
    mk_ctx *ctx = nullptr;
    if (mk_create(&p, &ctx) != MK_OK || mk_reserve(ctx, G) != MK_OK) { fprintf(stderr, "%s\n", mk_last_error()); return 1; }
    double t = now();
    for (uint32_t g = 0; g < G; g += 4096)
        if (mk_index_append_synthetic(ctx, g, std::min<uint32_t>(4096, G - g), 5000000) != MK_OK) { fprintf(stderr, "%s\n", mk_last_error()); return 1; }
    mk_sync(ctx);
    printf("built %u genomes (-h %u, %u-bit fingerprints) in %.2f s\n", G, p.h, p.fp_bits, now() - t);
    // a probe the loaded index has to answer the same way
    std::vector<char> q(64 * 1000);
    std::vector<const char *> qp(64);
    std::vector<uint64_t> ql(64, 1000);
    std::vector<char> genome(5000000);
    for (uint32_t i = 0; i < 64; ++i) {
        if (i % 16 == 0 && mk_probe_synth_genomes(ctx, (uint64_t)(i / 16) * (G / 4), 1, 5000000, genome.data()) != MK_OK) return 1;
        memcpy(q.data() + i * 1000, genome.data() + 1000 + 7919 * (i % 16), 1000);
        qp[i] = q.data() + i * 1000;
    }
    std::vector<mk_hit> h0(640), h1(640);
    std::vector<uint32_t> n0(64), n1(64);
    if (mk_query(ctx, qp.data(), ql.data(), 64, 10, 10, 100.0, h0.data(), n0.data(), nullptr) != MK_OK) return 1;
    std::string err;
    t = now();
    if (mkhost::dump_index({ctx}, path, err, threads) != 0) { fprintf(stderr, "dump: %s\n", err.c_str()); return 1; }
    const double t_dump = now() - t;
    struct stat st;
    stat(path.c_str(), &st);
    const double raw = 39.0 + (double)(1ull << p.h) * G * (p.fp_bits / 8) + 12.0 * G + (double)(1ull << 30);
    printf("dump: %.2f s, %.2f GB file for a %.2f GB stream (%.2f GB/s of stream, %u threads)\n", t_dump, st.st_size / 1e9, raw / 1e9, raw / 1e9 / t_dump, threads);
    mk_destroy(ctx);
    std::vector<mk_ctx *> loaded;
    t = now();
    if (mkhost::load_index(path, {0}, loaded, err, threads) != 0) { fprintf(stderr, "load: %s\n", err.c_str()); return 1; }
    const double t_load = now() - t;
    printf("load: %.2f s (%.2f GB/s of stream)\n", t_load, raw / 1e9 / t_load);
    if (mk_query(loaded[0], qp.data(), ql.data(), 64, 10, 10, 100.0, h1.data(), n1.data(), nullptr) != MK_OK) return 1;
    bool same = n0 == n1;
    for (uint32_t i = 0; i < 64 && same; ++i) same = memcmp(&h0[i * 10], &h1[i * 10], n0[i] * sizeof(mk_hit)) == 0;
    printf("the loaded index answers 64 probe queries like the built one: %s (first query: %u hits, top genome %u, %u matches)\n", same ? "yes" : "NO",
           n1[0], n1[0] ? h1[0].genome : 0, n1[0] ? h1[0].matches : 0);
    mk_destroy(loaded[0]);
    if (!(argc > 6 && std::string(argv[6]) == "keep")) remove(path.c_str());
    return same ? 0 : 1;
}
