// Shared device/host arithmetic of the sketch path (gfx950 only).
// Reference citations: /root/reference/{utils.cpp,Miekki.cpp}; the algorithm is
// restated from its definition in SURVEY.md section 8a, not translated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_HD __host__ __device__ __forceinline__

namespace mk {

constexpr uint32_t kNumHash = 5;        // Miekki.h:79
constexpr uint32_t kMantisBits = 5;     // main.cpp:196
constexpr uint64_t kRevMul = 0xD6E8FEB86659FD93ULL;   // utils.cpp:180

// utils.cpp:179-184
MK_HD uint64_t revhash64(uint64_t x)
{
    x = ((x >> 32) ^ x) * kRevMul;
    x = ((x >> 32) ^ x) * kRevMul;
    return (x >> 32) ^ x;
}

// Forward code of a base outside the k-1 seed (nuc2int, utils.cpp:31-49):
// C,G,T -> 1,2,3; anything else 0.
MK_HD uint32_t fwd_code(uint8_t c) { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; }
// Reverse-strand code outside the seed (nuc2intrc, utils.cpp:107-125):
// A,C,G -> 3,2,1; anything else (T included) 0.
MK_HD uint32_t rc_code(uint8_t c) { return c == 'A' ? 3u : c == 'C' ? 2u : c == 'G' ? 1u : 0u; }
// Seed characters (first k-1 of the sequence) go through str2numstrand
// (utils.cpp:252-272): case-insensitive ACGT, 4 = invalid (whole seed becomes 0).
MK_HD uint32_t seed_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0; case 'C': case 'c': return 1;
    case 'G': case 'g': return 2; case 'T': case 't': return 3;
    default: return 4;
    }
}

// Both 2-bit codes of sequence position j as the rolling state of
// minhash_sketch_partition sees them (Miekki.cpp:158-164): positions inside the
// seed contribute seed digits (all zero when the seed is invalid) and their
// complement through rcb (Miekki.cpp:66-76); later positions go through
// update_kmer / update_kmer_RC.  Returns fwd | rc << 2.
MK_HD uint32_t pos_codes(uint8_t c, uint64_t j, uint32_t k, bool seed_valid)
{
    if (j < k - 1) {
        uint32_t s = seed_valid ? seed_code(c) : 0u;
        return s | ((3u - s) << 2);
    }
    return fwd_code(c) | (rc_code(c) << 2);
}

// HyperMinHash fingerprint (Miekki::mantis, Miekki.cpp:91-113) of the
// (64-h)-bit remainder n, truncated to fp_bits; f = fp_bits - 5.
MK_HD uint32_t mantis(uint64_t n, uint32_t h, uint32_t f, uint32_t empty)
{
    if (n == 0) return empty;
#if defined(__HIP_DEVICE_COMPILE__)
    int prefix = 63 - __clzll((long long)n);
#else
    int prefix = 63 - __builtin_clzll(n);
#endif
    int e = prefix - 32 + (int)h;
    if (e < 0) e = 0;
    int off = prefix - (int)f;
    if (off < 0) off = 0;
    uint64_t suffix = (n - (1ULL << prefix)) >> off;
    return (uint32_t)(suffix + ((uint64_t)e << f)) & empty;
}

// The same fingerprint from the two 32-bit halves of n with 32-bit operations only (64-bit shifts
// and adds cost two to four issue slots each on CDNA; the build's hash loop is issue-bound).
// funnel(hi, lo, s) = low word of ({hi, lo} >> s), 0 < s < 32.
MK_HD uint32_t funnel_shift(uint32_t hi, uint32_t lo, uint32_t s)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
MK_HD uint32_t mantis_halves(uint32_t nhi, uint32_t nlo, uint32_t h, uint32_t f, uint32_t empty)
{
    if ((nhi | nlo) == 0) return empty;
    uint32_t prefix, top;                      // top: n shifted up so that its leading one sits at bit 31
#if defined(__HIP_DEVICE_COMPILE__)
    if (nhi) { const uint32_t lz = (uint32_t)__clz((int)nhi); prefix = 63u - lz; top = lz ? funnel_shift(nhi, nlo, 32u - lz) : nhi; }
    else     { const uint32_t lz = (uint32_t)__clz((int)nlo); prefix = 31u - lz; top = nlo << lz; }
#else
    if (nhi) { const uint32_t lz = (uint32_t)__builtin_clz(nhi); prefix = 63u - lz; top = lz ? funnel_shift(nhi, nlo, 32u - lz) : nhi; }
    else     { const uint32_t lz = (uint32_t)__builtin_clz(nlo); prefix = 31u - lz; top = nlo << lz; }
#endif
    const int e = (int)prefix - 32 + (int)h;
    uint32_t suffix;
    if (prefix >= f) suffix = (top << 1) >> (32u - f);         // the f bits below the leading one
    else suffix = nlo - (1u << prefix);                        // fewer than f bits exist (n < 2^f): all of them
    return (suffix + ((uint32_t)(e < 0 ? 0 : e) << f)) & empty;
}

// The common case of the same fingerprint in ONE conversion (the build's hash loop, build.hip: fingerprint_of): v = the top
// 32 of n's 64 - h bits.  When v >= 2^f the leading one and the f bits below it all lie in v, the exponent
// max(prefix - 32 + h, 0) is v's own bit index e and the suffix the f bits below it -- which is what a float's exponent
// field and the top of its mantissa hold: float(v) = 1.mantissa x 2^e, bits = (127 + e) << 23 | mantissa, so the
// fingerprint e << f | suffix is (bits >> (23 - f)) - (127 << f).  The conversion must not round up into the next power of
// two: with the low eight bits cleared v has at most 24 significant bits and converts exactly, and the f suffix bits are
// untouched by that as long as e >= 8 + f.  fingerprint_top32_ok says whether v qualifies (all but one k-mer in 2^(24 - f)).
MK_HD bool fingerprint_top32_ok(uint32_t v, uint32_t f) { return v >= (256u << f); }
MK_HD uint32_t fingerprint_from_top32(uint32_t v, uint32_t f)
{
    const float x = (float)(v & 0xFFFFFF00u);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t bits = __float_as_uint(x);
#else
    uint32_t bits;
    __builtin_memcpy(&bits, &x, 4);
#endif
    return (bits >> (23u - f)) - (127u << f);                      // <= 31 << f | 2^f - 1 = empty
}

// anc -> (bucket, fingerprint)  (Miekki.cpp:169-171)
MK_HD void bucket_fp(uint64_t anc, uint32_t h, uint32_t f, uint32_t empty, uint32_t &bucket,
                     uint32_t &fp)
{
    bucket = (uint32_t)(anc >> (64 - h));
    fp = mantis(anc & ((1ULL << (64 - h)) - 1), h, f, empty);
}

// i-th Bloom position of a selected k-mer: universal_hash(anc, i) >> b (utils.cpp:197-199, Miekki.cpp:124/138),
// universal_hash(x, i) = unrevhash64(x) + (i * 69 * revhash64(x)) % 1024 called on x = anc = revhash64(canon).
// The first term is the canonical k-mer itself; the multiplier of the second is revhash64(anc) -- the hash of the
// HASH.  That term is below 1024 and b >= 32, so it changes the shifted value only when the low word of canon can
// carry (one k-mer in four million): everybody else is spared the second hash.
MK_HD uint64_t bloom_pos(uint64_t canon, uint64_t anc, uint32_t i, uint32_t bloom_log2)
{
    if ((uint32_t)canon <= 0xFFFFFC00u) return canon >> bloom_log2;
    return (canon + (((uint64_t)(i * 69u) * revhash64(anc)) % 1024u)) >> bloom_log2;
}

// splitmix64 and the synthetic genome generator of SURVEY.md 8d
MK_HD uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
constexpr uint64_t kSeedG = 0x4D49454B4B490001ULL;
constexpr uint64_t kSeedQ = 0x4D49454B4B490002ULL;

MK_HD uint64_t genome_word(uint64_t g, uint64_t word) { return splitmix64(kSeedG ^ (g << 32) ^ word); }

// A collection of RELATED genomes (measurement aid: the column codec of SURVEY.md 8f row N4 pays on strains of one
// species, README.md:136-138): genome g is strain g % strains of species g / strains.  Strain 0 is the species' genome
// -- the synthetic genome `species` of SURVEY.md 8d itself -- the others are independent descendants of it: each
// 32-base word carries, with probability 32 x rate, ONE substitution (position and new base from a hash of (g, word)).
constexpr uint64_t kSeedS = 0x4D49454B4B490004ULL;
MK_HD uint64_t strain_word(uint64_t g, uint64_t word, uint32_t strains, uint32_t rate_ppm)
{
    const uint64_t species = g / strains, t = g % strains;
    uint64_t w = genome_word(species, word);
    if (t == 0) return w;
    const uint64_t r = splitmix64(kSeedS ^ (g << 32) ^ word);
    // (a word mutates when the low 32 bits of r fall below 32 x rate x 2^32)
    if ((r & 0xffffffffu) < (uint64_t)rate_ppm * 32u * 4295u) {                 // 4295 = 2^32 / 10^6
        const uint32_t pos = (uint32_t)(r >> 32) & 31u, sh = 62u - 2u * pos;
        const uint64_t old = (w >> sh) & 3u, nw = (old + 1u + ((r >> 40) % 3u)) & 3u;
        w = (w & ~(3ULL << sh)) | (nw << sh);
    }
    return w;
}

#if defined(__HIPCC__)
// inclusive prefix sum over the wave's 64 lanes by data-parallel-primitive moves (no LDS round trips: __shfl_up is one each)
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x)
{
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1 (rows of 16 lanes)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
#endif

}  // namespace mk
