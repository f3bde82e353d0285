// Test helper: run the host driver's OrderedFastaReader over a list of files and print,
// per file in list order, "<exists> <length> <fnv1a64 of the sequence>".
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "fasta_reader.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::vector<std::string> files;
    std::ifstream in(argv[1]);
    for (std::string l; std::getline(in, l);) if (!l.empty()) files.push_back(l);
    mkhost::OrderedFastaReader reader(files, (unsigned)atoi(argv[2]), mkhost::HostAllocator{nullptr, nullptr, nullptr},
                                      argc > 3 ? (size_t)atoi(argv[3]) : 4);
    for (size_t i = 0; i < files.size(); ++i) {
        mkhost::OrderedFastaReader::Item it = reader.take(i);
        uint64_t h = 1469598103934665603ull;
        for (size_t j = 0; j < it.len; ++j) { h ^= (unsigned char)it.data[j]; h *= 1099511628211ull; }
        printf("%d %zu %016llx\n", it.exists ? 1 : 0, it.len, (unsigned long long)h);
        reader.recycle(it);
    }
    return 0;
}
