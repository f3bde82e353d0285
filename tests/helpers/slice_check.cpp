// Test helper: the per-rank index load of the one-process-per-GPU form, without a communicator -- an index file is loaded
// as `world` slices (load_index with slice_rank / slice_world, one context each, all on GPU 0) and the slices, side by side,
// are dumped again: the stream must be the file's.
//   slice_check <index.gz> <world> <out.gz>      prints the slices' genome counts
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "index_io.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int world = atoi(argv[2]);
    std::vector<mk_ctx *> all;
    std::string err;
    for (int r = 0; r < world; ++r) {
        std::vector<mk_ctx *> one;
        if (mkhost::load_index(argv[1], {0}, one, err, 2, r, world) != 0 || one.size() != 1) { printf("load failed: %s\n", err.c_str()); return 1; }
        printf("%u ", mk_index_size(one[0]));
        all.push_back(one[0]);
    }
    printf("\n");
    if (mkhost::dump_index(all, argv[3], err, 2) != 0) { printf("dump failed: %s\n", err.c_str()); return 1; }
    for (mk_ctx *c : all) mk_destroy(c);
    return 0;
}
