// Test helper for host/rendezvous.cpp (the ranks' bootstrap file):
//   rendezvous_check publish <path> <nonce> <payload text> <hold ms>   writes, prints "published", stays alive for <hold ms>
//   rendezvous_check fetch   <path> <nonce> <bytes> <timeout ms>        prints "got <payload>" or "timeout: <why>"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rendezvous.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    mkhost::Rendezvous r{argv[2], argv[3]};
    std::string err;
    if (!strcmp(argv[1], "publish")) {
        if (!mkhost::rendezvous_publish(r, argv[4], strlen(argv[4]), err)) { printf("failed: %s\n", err.c_str()); return 1; }
        printf("published\n");
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(atol(argv[5])));
        return 0;
    }
    std::vector<char> buf((size_t)atol(argv[4]) + 1, 0);
    if (!mkhost::rendezvous_fetch(r, buf.data(), buf.size() - 1, atol(argv[5]), err)) { printf("timeout: %s\n", err.c_str()); return 1; }
    printf("got %s\n", buf.data());
    return 0;
}
