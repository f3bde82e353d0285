// Packed sequence codes on the device (the layout of mk_packed_seq, include/miekki_hip.h): 2 bits
// per position -- position i at bits 2 * (i % 32) of 64-bit word i / 32, A C G T = 0 1 2 3 and 0 for
// anything else -- plus one exception bit per position for sequences that hold anything else.
// What the reference's rolling state sees at a position:
//   forward digit  = the code                       (nuc2int, utils.cpp:31-49)
//   reverse digit  = 3 - code, or 0 at an exception (nuc2intrc, utils.cpp:107-125)
// and inside the k-1 seed the digits str2numstrand / rcb produce (utils.cpp:252-272,
// Miekki.cpp:66-76), which seed_fix_kernel (build.hip) writes over the packed ones once per batch.
#pragma once
#include "mk_device.hpp"

namespace mk {

// every bit of a 16-bit mask doubled: bit j -> bits 2j and 2j + 1
__device__ __forceinline__ uint32_t spread_pairs16(uint32_t m)
{
    m = (m | (m << 8)) & 0x00FF00FFu;
    m = (m | (m << 4)) & 0x0F0F0F0Fu;
    m = (m | (m << 2)) & 0x33333333u;
    m = (m | (m << 1)) & 0x55555555u;
    return m * 3u;
}
__device__ __forceinline__ uint64_t spread_pairs32(uint32_t m)
{
    return (uint64_t)spread_pairs16(m & 0xffffu) | ((uint64_t)spread_pairs16(m >> 16) << 32);
}

// order of the 2-bit digits of a 64-bit word reversed (digit j <-> digit 31 - j)
__device__ __forceinline__ uint64_t reverse_digits(uint64_t x)
{
    uint64_t r = __builtin_bitreverse64(x);
    return ((r & 0x5555555555555555ULL) << 1) | ((r >> 1) & 0x5555555555555555ULL);
}

// Canonical k-mer (min of the forward k-mer and the reverse strand's, as minhash_sketch_partition
// compares them, Miekki.cpp:167) that starts at position pos: ONE 16-byte load of codes from an
// 8-byte boundary -- every lane-load is a request of its own at the L2, and the request rate is what
// bounds the callers -- plus, for sequences with exceptions only, the same of the exception bits.
typedef uint64_t __attribute__((ext_vector_type(2), aligned(8))) packed_pair;

__device__ __forceinline__ packed_pair load_packed_pair(const uint64_t *__restrict__ codes, uint64_t pos)
{
    return *reinterpret_cast<const packed_pair *>(codes + (pos >> 5));
}

// (the second half of canon_from_packed, for callers that request the codes of several positions before using the first)
__device__ __forceinline__ uint64_t canon_from_pair(packed_pair x, const uint64_t *__restrict__ except, bool has_except,
                                                    uint64_t pos, uint32_t k)
{
    const uint32_t sh = (uint32_t)(pos & 31u) * 2u;
    const uint64_t kmask = (1ULL << (2 * k)) - 1;                                  // k <= 31
    const uint64_t F = (sh ? (x.x >> sh) | (x.y << (64 - sh)) : x.x) & kmask;      // digit j of the k-mer at bits 2j
    uint64_t RC = ~F & kmask;                                                      // 3 - digit: update_kmer_RC's state
    if (has_except) {
        const packed_pair e = *reinterpret_cast<const packed_pair *>(except + (pos >> 6));
        const uint32_t es = (uint32_t)(pos & 63u);
        const uint32_t bits = (uint32_t)(es ? (e.x >> es) | (e.y << (64 - es)) : e.x);   // 32 >= k exception bits
        RC &= ~spread_pairs32(bits);
    }
    const uint64_t S = reverse_digits(F) >> (64 - 2 * k);                          // digit 0 on top
    return S < RC ? S : RC;
}

__device__ __forceinline__ uint64_t canon_from_packed(const uint64_t *__restrict__ codes, const uint64_t *__restrict__ except,
                                                      bool has_except, uint64_t pos, uint32_t k)
{
    return canon_from_pair(load_packed_pair(codes, pos), except, has_except, pos, k);
}

}  // namespace mk
