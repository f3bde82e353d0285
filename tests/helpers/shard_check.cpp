// Test helper (no GPU needed): prints the device list host/multi_gpu.cpp derives from the environment
// the shard table of `n` items over `parts` shards and the entrant-row width for a shard of n genomes.   shard_check <n> <parts>
#include <cstdio>
#include <cstdlib>

#include "multi_gpu.hpp"

int main(int argc, char **argv)
{
    const unsigned long n = argc > 1 ? strtoul(argv[1], nullptr, 10) : 0;
    const unsigned parts = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    printf("devices");
    for (int d : mkhost::device_list()) printf(" %d", d);
    printf("\n");
    for (unsigned s = 0; s < parts; ++s) {
        uint64_t b, e;
        mkhost::shard_range(n, s, parts, b, e);
        printf("shard %u %llu %llu\n", s, (unsigned long long)b, (unsigned long long)e);
    }
    // entrant slots per exchange row for a top-10 / top-5 heap over a shard of n genomes (multi_gpu.hpp: entrant_cap)
    printf("cap %u %u\n", mkhost::entrant_cap(10, n), mkhost::entrant_cap(5, n));
    return 0;
}
