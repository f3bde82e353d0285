// Host check of the build kernels' 32-bit fingerprint arithmetic (mk::mantis_halves, mk::fingerprint_from_top32) against the
// plain 64-bit statement of Miekki::mantis (mk::mantis, Miekki.cpp:91-113) in the same header.
// Compiled by tests/test_host_arith.py with MK_DEVICE_HPP pointing at a copy of mk_device.hpp whose
// HIP include has been removed (the header's host side is plain C++).
#include <cstdint>
#include <cstdio>
#include <random>
#define __host__
#define __device__
#define __forceinline__ inline
#include MK_DEVICE_HPP

int main()
{
    std::mt19937_64 rng(20261004);
    uint64_t bad = 0, n = 0, fast = 0;
    for (uint32_t h = 1; h <= 28; ++h)
        for (uint32_t fpb : {8u, 16u}) {
            const uint32_t f = fpb - 5, empty = fpb == 8 ? 255u : 65535u;
            for (int it = 0; it < 200000; ++it) {
                uint64_t v = rng() >> (rng() % 64);
                v &= (1ULL << (64 - h)) - 1;
                if (it < 300) v = (uint64_t)it;                    // n = 0, and fewer than f bits below the leading one
                else if (it < 1200) {                              // all ones below the leading one at every place (what rounding would push up),
                    const uint32_t top = (uint32_t)(it - 300) % (64 - h);      // and the same with single zeros punched in
                    v = (top == 63 ? ~0ULL : (1ULL << (top + 1)) - 1);
                    if ((it - 300) / 64 % 3 == 1 && top > 10) v &= ~(1ULL << (top - 1 - (uint32_t)(it % 9)));
                    if ((it - 300) / 64 % 3 == 2 && top > 30) v &= ~(1ULL << (top - 24 - (uint32_t)(it % 5)));
                    v &= (1ULL << (64 - h)) - 1;
                }
                const uint32_t a = mk::mantis(v, h, f, empty), b = mk::mantis_halves((uint32_t)(v >> 32), (uint32_t)v, h, f, empty);
                ++n;
                if (a != b && bad++ < 5) printf("h=%u f=%u n=%llx: %u vs %u\n", h, f, (unsigned long long)v, a, b);
                // the float-conversion form, where it applies: v32 = the top 32 of n's 64 - h bits
                const uint32_t v32 = (uint32_t)(v >> (32 - h));
                if (mk::fingerprint_top32_ok(v32, f)) {
                    ++fast;
                    const uint32_t c = mk::fingerprint_from_top32(v32, f);
                    if (a != c && bad++ < 5) printf("h=%u f=%u n=%llx: %u vs %u (float form)\n", h, f, (unsigned long long)v, a, c);
                }
            }
        }
    printf("checked %llu (float form %llu) mismatches %llu\n", (unsigned long long)n, (unsigned long long)fast, (unsigned long long)bad);
    return bad != 0;
}
