// Test helper (GPU): the multi-GPU driver of the `miekki` binary (host/multi_gpu.cpp, DeviceGroup) over SYNTHETIC
// genomes, so that the sharded path can be run at sizes no FASTA directory reaches (BASELINE config 3: 100,000 x 5 Mb).
//
//   group_synth <G> <genome_len> <nq> <query_len> <k> <h> <fp_bits> <nresults> <min_score> <min_inter> <out.bin>
//
// Shards = MIEKKI_DEVICES (an ordinal may repeat: "0,0" = two shards in the one GPU), shard d builds the genomes
// shard_range(G, d, D) with mk_index_append_synthetic, the group folds the Bloom filters and answers the nq
// synthetic queries of SURVEY.md 8d.  Writes nhits u32[nq] then hits mk_hit[nq][nresults] to <out.bin> and prints
//   shards D total G rerun R replayed P gather_bytes B
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "multi_gpu.hpp"

static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// n bases of synthetic genome g from position off (tests/synth.py: genome_bases)
static void genome_bases(uint64_t g, uint64_t off, uint64_t n, char *out)
{
    const uint64_t seed = 0x4D49454B4B490001ull;
    for (uint64_t i = off; i < off + n; ++i) {
        const uint64_t w = splitmix64(seed ^ (g << 32) ^ (i >> 5));
        out[i - off] = "ACGT"[(w >> (62 - 2 * (i & 31))) & 3];
    }
}

int main(int argc, char **argv)
{
    if (argc < 12) { fprintf(stderr, "usage: see the head of group_synth.cpp\n"); return 2; }
    const uint64_t G = strtoull(argv[1], nullptr, 10), L = strtoull(argv[2], nullptr, 10);
    const uint32_t nq = (uint32_t)atol(argv[3]);
    const uint64_t qlen = strtoull(argv[4], nullptr, 10);
    mk_params p;
    memset(&p, 0, sizeof p);
    p.k = (uint32_t)atoi(argv[5]); p.h = (uint32_t)atoi(argv[6]); p.fp_bits = (uint32_t)atoi(argv[7]);
    p.bloom_log2 = 33; p.threshold = 200;
    const uint32_t nresults = (uint32_t)atoi(argv[8]), min_score = (uint32_t)atoi(argv[9]);
    const double min_inter = atof(argv[10]);
    const char *out_path = argv[11];

    const std::vector<int> devs = mkhost::device_list();
    const uint32_t D = (uint32_t)devs.size();
    std::vector<mk_ctx *> ctxs(D, nullptr);
    std::vector<int> rc(D, 0);
    std::vector<std::string> msg(D);
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < D; ++d)                                  // the shards build side by side, as in the binary
        th.emplace_back([&, d] {
            uint64_t b, e;
            mkhost::shard_range(G, d, D, b, e);
            mk_params pd = p;
            pd.device = devs[d];
            int r = mk_create(&pd, &ctxs[d]);
            if (r == MK_OK) r = mk_reserve(ctxs[d], (uint32_t)(e - b));
            for (uint64_t g = b; g < e && r == MK_OK; g += 2048)
                r = mk_index_append_synthetic(ctxs[d], g, (uint32_t)std::min<uint64_t>(2048, e - g), L);
            if (r == MK_OK) r = mk_sync(ctxs[d]);
            if (r != MK_OK) msg[d] = mk_last_error();
            rc[d] = r;
        });
    for (auto &t : th) t.join();
    for (uint32_t d = 0; d < D; ++d)
        if (rc[d] != MK_OK) { fprintf(stderr, "shard %u: %s\n", d, msg[d].c_str()); return 1; }
    mkhost::DeviceGroup group;
    group.adopt(ctxs);
    std::string err;
    if (group.finish(true, err)) { fprintf(stderr, "finish: %s\n", err.c_str()); return 1; }

    std::vector<char> text((size_t)nq * qlen);
    std::vector<const char *> seqs(nq);
    std::vector<uint64_t> lens(nq, qlen);
    for (uint32_t q = 0; q < nq; ++q) {                               // tests/synth.py: query_origin
        const uint64_t g = q % G, o = splitmix64(0x4D49454B4B490002ull ^ q) % (L - qlen);
        genome_bases(g, o, qlen, text.data() + (size_t)q * qlen);
        seqs[q] = text.data() + (size_t)q * qlen;
    }
    std::vector<mk_hit> hits((size_t)nq * std::max(nresults, 1u));
    std::vector<uint32_t> nhits(nq, 0);
    memset(hits.data(), 0, hits.size() * sizeof(mk_hit));
    if (group.query(seqs.data(), lens.data(), nq, nresults, min_score, min_inter, hits.data(), nhits.data(), err)) {
        fprintf(stderr, "query: %s\n", err.c_str());
        return 1;
    }
    FILE *f = fopen(out_path, "wb");
    if (!f) { perror(out_path); return 1; }
    fwrite(nhits.data(), 4, nq, f);
    fwrite(hits.data(), sizeof(mk_hit), (size_t)nq * nresults, f);
    fclose(f);
    printf("shards %u total %u rerun %llu replayed %llu gather_bytes %llu\n", D, group.total(),
           (unsigned long long)group.rerun_queries(), (unsigned long long)group.replayed_queries(),
           (unsigned long long)group.gather_bytes());
    return 0;
}
