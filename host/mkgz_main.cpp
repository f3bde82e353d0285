// mkgz -- the index file's parallel gzip layer as a stand-alone filter (host only):
//   mkgz c <out.gz> [threads] < stream      mkgz d <in.gz> [threads] > stream
//   mkgz i <out.gz> [threads] < stream      the index writer's special forms: the stream as the dump writes a matrix --
//       a 39-byte head as a member of its own, then whole blocks handed over where they lie (write_block) with the
//       Huffman-only strategy, runs of zero bytes of a block or more through write_zeros -- or, with MKGZ_STORED=1, stored
// Lets the CPU tests exercise exactly the writer / reader the `miekki` binary uses.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>

#include "gzpar.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: mkgz c|d|i <file> [threads]\n"); return 2; }
    const unsigned threads = argc > 3 ? (unsigned)atoi(argv[3]) : 4;
    std::vector<char> buf(1 << 22);
    if (!strcmp(argv[1], "c")) {
        mkhost::ParallelGzipWriter w(argv[2], threads);
        if (!w.ok()) return 1;
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0) w.write(buf.data(), n);
        return w.finish() ? 0 : 1;
    }
    if (!strcmp(argv[1], "i")) {
        std::vector<char> all;
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0) all.insert(all.end(), buf.data(), buf.data() + n);
        mkhost::ParallelGzipWriter w(argv[2], threads);
        if (!w.ok()) return 1;
        size_t at = std::min<size_t>(39, all.size());
        w.write(all.data(), at);
        w.flush_block();
        w.set_strategy(Z_HUFFMAN_ONLY);
        if (getenv("MKGZ_STORED")) w.set_level(0);
        const size_t B = mkhost::ParallelGzipWriter::kBlock;
        while (at < all.size()) {
            // a run of zeros of at least a block: the ready-made member
            size_t z = at;
            while (z < all.size() && all[z] == 0) ++z;
            if (z - at >= B) { w.write_zeros(z - at); at = z; continue; }
            const size_t take = std::min(B, all.size() - at);
            if (!w.write_block((const uint8_t *)all.data() + at, take, nullptr)) w.write(all.data() + at, take);
            at += take;
        }
        return w.finish() ? 0 : 1;
    }
    mkhost::ParallelGzipReader r(argv[2], threads);
    if (!r.ok()) return 1;
    size_t n;
    while ((n = r.read_some(buf.data(), buf.size())) > 0) fwrite(buf.data(), 1, n, stdout);
    return r.ok() ? 0 : 1;
}
