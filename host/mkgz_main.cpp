// mkgz -- the index file's parallel gzip layer as a stand-alone filter (host only):
//   mkgz c <out.gz> [threads] < stream      mkgz d <in.gz> [threads] > stream
// Lets the CPU tests exercise exactly the writer / reader the `miekki` binary uses.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gzpar.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: mkgz c|d <file> [threads]\n"); return 2; }
    const unsigned threads = argc > 3 ? (unsigned)atoi(argv[3]) : 4;
    std::vector<char> buf(1 << 22);
    if (!strcmp(argv[1], "c")) {
        mkhost::ParallelGzipWriter w(argv[2], threads);
        if (!w.ok()) return 1;
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0) w.write(buf.data(), n);
        return w.finish() ? 0 : 1;
    }
    mkhost::ParallelGzipReader r(argv[2], threads);
    if (!r.ok()) return 1;
    size_t n;
    while ((n = r.read_some(buf.data(), buf.size())) > 0) fwrite(buf.data(), 1, n, stdout);
    return r.ok() ? 0 : 1;
}
