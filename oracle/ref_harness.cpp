// Harness around the UNMODIFIED reference (compiled in place from
// /root/reference by oracle/Makefile, outputs in oracle/_ref/).  Own code: it
// only calls the reference's public members and dumps their results so that
// tests/golden/make_golden.py can turn them into fixtures, and times the
// reference's scan for bench.py's cpu_baseline ("kind": "reference").
//
// TEST INFRASTRUCTURE.  Never linked into the product.
#include <omp.h>          // Miekki.h uses omp_lock_t without including it
#include "Miekki.h"
#include "synth.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

using namespace std;

static void wr(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(1); } }

static void dump_hits(const string& path, Miekki& ix, const vector<vector<uint32_t>>& rows,
                      size_t nres, uint32_t min_score, double min_inter)
{
    FILE* f = fopen(path.c_str(), "wb");
    for (auto& row : rows) {
        matrix::vector<score_t> v(int(row.size()));
        for (size_t i = 0; i < row.size(); ++i) v[i] = row[i];
        auto hits = ix.filter_results(matrix::refvec<score_t>(v), nres, min_score, min_inter);
        uint32_t n = hits.size();
        wr(f, &n, 4);
        for (auto& s : hits) { wr(f, &s.genome, 4); wr(f, &s.matches, 4); wr(f, &s.jaccard, 8); wr(f, &s.intersection, 8); }
    }
    fclose(f);
}

// golden <dir> <k> <h> <f> <b> <threshold> <n_sketch_dump>
//   <dir>/genomes.lst : list of FASTA paths (index_file_of_file input)
//   <dir>/queries.fa  : strict 2-line records
static int cmd_golden(int argc, char** argv)
{
    if (argc < 9) { fprintf(stderr, "usage: golden dir k h f b threshold n_sketch_dump\n"); return 2; }
    string dir = argv[2];
    uint32_t k = atoi(argv[3]), h = atoi(argv[4]), f = atoi(argv[5]), b = atoi(argv[6]);
    uint32_t thr = uint32_t(atof(argv[7]));           // main.cpp:196 truncation of the -s double
    uint32_t nsk = atoi(argv[8]);
    Miekki ix(k, h, 5 + f, 5, 0, dir + "/harness_out.txt", b, thr, 1);
    ix.index_file_of_file(dir + "/genomes.lst");
    ix.dump_disk(dir + "/harness_idx.gz");            // before compress_index: payload identical

    vector<string> names, seqs;
    {
        ifstream in(dir + "/queries.fa");
        string hd, sq;
        while (getline(in, hd)) { if (!getline(in, sq)) sq = ""; if (sq.size() >= k) { names.push_back(hd); seqs.push_back(sq); } }
    }
    uint32_t nq = seqs.size(), G = ix.index_size;
    // raw query sketches (A4) for the first nsk queries: fp (as u16) + hash
    {
        FILE* fo = fopen((dir + "/sketch_q.bin").c_str(), "wb");
        for (uint32_t q = 0; q < nq && q < nsk; ++q) {
            uint32_t act = 0;
            auto sk = ix.minhash_sketch_partition(seqs[q], act);
            wr(fo, &act, 4);
            for (auto v : sk.first) { uint16_t x = v; wr(fo, &x, 2); }
            wr(fo, sk.second.data(), sk.second.size() * 8);
        }
        fclose(fo);
    }
    // A11 in batches of 201 like query_file
    vector<vector<uint32_t>> rows(nq, vector<uint32_t>(G));
    for (uint32_t q0 = 0; q0 < nq; q0 += 201) {
        vector<pair<string, uint32_t>> batch;
        for (uint32_t q = q0; q < nq && q < q0 + 201; ++q) batch.push_back({seqs[q], 0});
        auto m = ix.query_sequences(batch);
        for (uint32_t i = 0; i < batch.size(); ++i)
            for (uint32_t g = 0; g < G; ++g) rows[q0 + i][g] = m(int(i), int(g));
    }
    {
        FILE* fo = fopen((dir + "/scores.u32").c_str(), "wb");
        for (auto& r : rows) wr(fo, r.data(), r.size() * 4);
        fclose(fo);
    }
    // A12
    {
        FILE* fo = fopen((dir + "/qseq_scores.u32").c_str(), "wb");
        FILE* fa = fopen((dir + "/qseq_active.u32").c_str(), "wb");
        for (uint32_t q = 0; q < nq; ++q) {
            uint32_t act = 0;
            auto v = ix.query_sequence(seqs[q], act);
            for (uint32_t g = 0; g < G; ++g) { uint32_t s = v[g]; wr(fo, &s, 4); }
            wr(fa, &act, 4);
        }
        fclose(fo); fclose(fa);
    }
    // A13 with the three parameter sets the drivers use (437/500, 741, 778)
    dump_hits(dir + "/hits_approx.bin", ix, rows, 10, 10, 0.5 * ix.threshold);
    dump_hits(dir + "/hits_exact_a.bin", ix, rows, 5, 10, ix.threshold);
    dump_hits(dir + "/hits_exact_A.bin", ix, rows, 5, 5, ix.threshold);
    dump_hits(dir + "/hits_loose.bin", ix, rows, 3, 0, 0.0);       // forces heap ties / replacement
    dump_hits(dir + "/hits_loose10.bin", ix, rows, 10, 1, 0.0);
    printf("\ngolden: G=%u nq=%u\n", G, nq);
    return 0;
}

// single <dir> <k> <h> <f> <b> <threshold>: every file of <dir>/genomes.lst through index_file (Miekki.cpp:518-536),
// i.e. the one-genome insert_sequence (243-273) whose size estimate differs from insert_sequences'; the index is
// written to <dir>/single_idx.gz
static int cmd_single(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: single dir k h f b threshold\n"); return 2; }
    string dir = argv[2];
    uint32_t k = atoi(argv[3]), h = atoi(argv[4]), f = atoi(argv[5]), b = atoi(argv[6]);
    uint32_t thr = uint32_t(atof(argv[7]));
    Miekki ix(k, h, 5 + f, 5, 0, dir + "/harness_out.txt", b, thr, 1);
    ifstream in(dir + "/genomes.lst");
    string path;
    while (getline(in, path)) if (path.size() > 3) ix.index_file(dir + "/" + path);
    ix.dump_disk(dir + "/single_idx.gz");
    printf("\nsingle: G=%u\n", (unsigned)ix.index_size);
    return 0;
}

// filter <in.bin> <out.bin>: synthetic filter_results cases (tie behaviour).
// in: u32 ncase; per case: u32 G, u32 nres, u32 min_score, f64 min_inter,
//     u32 sketch_size[G], u64 genome_size[G], u32 scores[G]
static int cmd_filter(int argc, char** argv)
{
    if (argc < 4) return 2;
    FILE* fi = fopen(argv[2], "rb");
    FILE* fo = fopen(argv[3], "wb");
    uint32_t nc; if (fread(&nc, 4, 1, fi) != 1) return 1;
    Miekki ix(31, 4, 8, 5, 0, "/dev/null", 32, 0, 1);
    for (uint32_t c = 0; c < nc; ++c) {
        uint32_t G, nres, ms; double mi;
        if (fread(&G, 4, 1, fi) != 1) return 1;
        if (fread(&nres, 4, 1, fi) != 1 || fread(&ms, 4, 1, fi) != 1 || fread(&mi, 8, 1, fi) != 1) return 1;
        ix.sketch_size.assign(G, 0); ix.genome_size.assign(G, 0);
        vector<uint32_t> sc(G);
        if (fread(ix.sketch_size.data(), 4, G, fi) != G) return 1;
        if (fread(ix.genome_size.data(), 8, G, fi) != G) return 1;
        if (fread(sc.data(), 4, G, fi) != G) return 1;
        ix.index_size = G;
        matrix::vector<score_t> v{int(G)};
        for (uint32_t i = 0; i < G; ++i) v[i] = sc[i];
        auto hits = ix.filter_results(matrix::refvec<score_t>(v), nres, ms, mi);
        uint32_t n = hits.size();
        wr(fo, &n, 4);
        for (auto& s : hits) { wr(fo, &s.genome, 4); wr(fo, &s.matches, 4); wr(fo, &s.jaccard, 8); wr(fo, &s.intersection, 8); }
    }
    fclose(fi); fclose(fo);
    return 0;
}

// scanbench <h> <G> <batches per thread> <threads> [queries per batch]: time the reference's query_sequences
// (Miekki.cpp:344-372) on an index of G genomes whose columns hold the real
// fingerprints of a few synthetic genomes, cyclically shifted, with a saturated
// Bloom filter (the >=10^4-genome regime).  Two samples on
// the same index: ONE thread over one batch of 201 queries (the per-core figure),
// then `threads` threads over `batches per thread` batches each, the way query_file
// hands batches to its OpenMP threads (Miekki.cpp:430-480).  Prints one JSON line.
static int cmd_scanbench(int argc, char** argv)
{
    if (argc < 6) return 2;
    uint32_t h = atoi(argv[2]), G = atoi(argv[3]), per = atoi(argv[4]), th = atoi(argv[5]);
    const uint32_t B = argc > 6 ? atoi(argv[6]) : 201;      // queries per batch of the all-thread sample (query_file: 201)
    const uint64_t L = 5000000, QL = 1000;
    const uint32_t NSRC = 4, k = 31;
    Miekki ix(k, h, 8, 5, 0, "/dev/null", 33, 200, th);
    vector<pair<string, string>> gs;
    for (uint32_t g = 0; g < NSRC; ++g) {
        string s(L, 'A');
        mk_genome_fill(g, 0, L, &s[0]);
        gs.push_back({s, "g"});
    }
    ix.insert_sequences(gs);
    uint32_t P = 1u << h;
    {                                                  // pad every column to G genomes (setup, not timed)
        vector<string> cols(P);
        #pragma omp parallel for num_threads(th) schedule(static)
        for (uint32_t p = 0; p < P; ++p) {
            string col(G, 0);
            for (uint32_t g = 0; g < G; ++g) col[g] = ix.index[(p + g / NSRC) & (P - 1)][g % NSRC];
            cols[p].swap(col);
        }
        ix.index.swap(cols);
    }
    ix.sketch_size.resize(G, ix.sketch_size[0]); ix.genome_size.resize(G, L); ix.index_size = G;
    std::fill(ix.Bloom_Filter.begin(), ix.Bloom_Filter.end(), 1);   // saturated
    // batch 0 = the one-thread sample, always 201 queries; batches 1 .. per * th = the all-thread sample, B each
    const uint32_t nb = 1 + per * th, nq = 201 + per * th * B;
    vector<vector<pair<string, uint32_t>>> batches(nb);
    for (uint32_t q = 0; q < nq; ++q) {
        uint64_t g, off; mk_query_origin(q, NSRC, L, QL, &g, &off);
        batches[q < 201 ? 0 : 1 + (q - 201) / B].push_back({gs[g].first.substr(off, QL), 0});
    }
    // comparisons = G * sum over queries of active partitions
    vector<uint64_t> act(nb, 0);
    #pragma omp parallel for num_threads(th) schedule(dynamic)
    for (uint32_t b = 0; b < nb; ++b)
        for (auto& q : batches[b]) { uint32_t a = 0; auto sk = ix.minhash_sketch_partition(q.first, a); act[b] += a; }
    uint64_t act_sum = 0;
    for (uint32_t b = 1; b < nb; ++b) act_sum += act[b];
    uint64_t chk = 0;
    auto t1 = chrono::steady_clock::now();
    {
        auto m = ix.query_sequences(batches[0]);
        for (int i = 0; i < int(batches[0].size()); ++i) chk += m(i, 0);
    }
    double s1 = chrono::duration<double>(chrono::steady_clock::now() - t1).count();
    auto t0 = chrono::steady_clock::now();
    #pragma omp parallel for num_threads(th) schedule(dynamic) reduction(+:chk)
    for (size_t b = 1; b < batches.size(); ++b) {
        auto m = ix.query_sequences(batches[b]);
        for (int i = 0; i < int(batches[b].size()); ++i) chk += m(i, 0);
    }
    double s = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
    printf("\n{\"comparisons\": %llu, \"seconds\": %.6f, \"threads\": %u, \"h\": %u, \"G\": %u, \"queries\": %u, \"batch\": %u, "
           "\"one_thread_comparisons\": %llu, \"one_thread_seconds\": %.6f, \"check\": %llu}\n",
           (unsigned long long)(act_sum * G), s, th, h, G, nq - 201, B, (unsigned long long)(act[0] * G), s1, (unsigned long long)chk);
    return 0;
}

// sketchbench <h> <genomes> <threads>: time the reference's index build on synthetic 5 Mb genomes
// held in memory: `threads` OpenMP threads each take the next genomes of the list and call
// insert_sequences with batches of eleven -- index_file_of_file's structure (Miekki.cpp:546-581)
// without its file reading: sketching runs in parallel, the append to the index is the
// reference's own global critical section (Miekki.cpp:285).  First one thread over a few genomes
// (the per-core figure), then all threads.  Prints one JSON line.
static int cmd_sketchbench(int argc, char** argv)
{
    if (argc < 5) return 2;
    uint32_t h = atoi(argv[2]), n = atoi(argv[3]), th = atoi(argv[4]);
    const uint64_t L = 5000000;
    const uint32_t k = 31, n1 = 8;
    vector<string> seqs(n);
    #pragma omp parallel for num_threads(th) schedule(dynamic)
    for (uint32_t g = 0; g < n; ++g) { seqs[g].assign(L, 'A'); mk_genome_fill(g, 0, L, &seqs[g][0]); }
    auto run = [&](uint32_t count, uint32_t threads) {
        Miekki ix(k, h, 8, 5, 0, "/dev/null", 33, 200, threads);
        uint32_t next = 0;
        auto t0 = chrono::steady_clock::now();
        #pragma omp parallel num_threads(threads)
        {
            vector<pair<string, string>> batch;
            for (;;) {
                uint32_t g;
                #pragma omp critical(fof)
                { g = next++; }
                if (g >= count) break;
                batch.push_back({seqs[g], "g"});
                if (batch.size() > 10) { ix.insert_sequences(batch); batch.clear(); }
            }
            ix.insert_sequences(batch);
        }
        double s = chrono::duration<double>(chrono::steady_clock::now() - t0).count();
        if (ix.index_size != count) { fprintf(stderr, "sketchbench: %u genomes indexed, %u expected\n", (unsigned)ix.index_size, count); exit(1); }
        return s;
    };
    const double s1 = run(std::min(n1, n), 1);
    const double s = run(n, th);
    printf("\n{\"genomes\": %u, \"seconds\": %.6f, \"threads\": %u, \"h\": %u, \"one_thread_genomes\": %u, "
           "\"one_thread_seconds\": %.6f}\n", n, s, th, h, std::min(n1, n), s1);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ref_harness golden|filter|scanbench|sketchbench ...\n"); return 2; }
    if (!strcmp(argv[1], "golden")) return cmd_golden(argc, argv);
    if (!strcmp(argv[1], "filter")) return cmd_filter(argc, argv);
    if (!strcmp(argv[1], "single")) return cmd_single(argc, argv);
    if (!strcmp(argv[1], "scanbench")) return cmd_scanbench(argc, argv);
    if (!strcmp(argv[1], "sketchbench")) return cmd_sketchbench(argc, argv);
    return 2;
}
