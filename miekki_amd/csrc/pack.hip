// Host side of the packed ingest format (include/miekki_hip.h, mk_packed_seq): characters ->
// 2-bit codes + exception bitmap, appended at any base position so that a reader can pack a
// FASTA file line by line straight into the (pinned) buffer the copy to the GPU starts from.
// Pure host code: no device, no context.
//
// What has to survive: outside the k-1 seed characters a base enters the
// rolling state through nuc2int (utils.cpp:31-49: C, G, T -> 1, 2, 3, ANYTHING else -> 0) and
// nuc2intrc (utils.cpp:107-125: A, C, G -> 3, 2, 1, ANYTHING else -> 0).  For A, C, G, T the second
// is 3 minus the first; for every other character (N, lower case, junk) both are 0.  So a position
// is its forward code plus one bit "not one of ACGT".  The seed characters go through
// str2numstrand (utils.cpp:252-272) instead, which is case-insensitive and zeroes the whole seed
// on any other character: they travel as characters (mk_packed_seq::head).
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/miekki_hip.h"

namespace {

// 32 characters -> 64 bits of codes (character j at bits 2j) and 32 exception bits
struct Block { uint64_t codes; uint32_t except; };

inline Block pack32_scalar(const unsigned char *c)
{
    Block b{0, 0};
    for (unsigned j = 0; j < 32; ++j) {
        const unsigned ch = c[j];
        const unsigned t = (ch >> 1) & 3u;                         // A C G T -> 0 1 3 2
        const bool ok = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        if (ok) b.codes |= (uint64_t)(t ^ (t >> 1)) << (2 * j);    // -> 0 1 2 3
        else b.except |= 1u << j;
    }
    return b;
}

__attribute__((target("avx2"))) inline Block pack32_avx2(const unsigned char *c)
{
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(c));
    const __m256i ok = _mm256_or_si256(
        _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('C'))),
        _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('T'))));
    // (c >> 1) & 3: A C G T -> 0 1 3 2; t ^ (t >> 1) -> 0 1 2 3; anything else -> 0
    const __m256i t = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));
    __m256i v = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), _mm256_set1_epi8(1)));
    v = _mm256_and_si256(v, ok);
    // squeeze the eight 2-bit codes of every 64-bit lane into its low 16 bits
    v = _mm256_and_si256(_mm256_or_si256(v, _mm256_srli_epi64(v, 6)), _mm256_set1_epi64x(0x000F000F000F000FLL));
    v = _mm256_and_si256(_mm256_or_si256(v, _mm256_srli_epi64(v, 12)), _mm256_set1_epi64x(0x000000FF000000FFLL));
    v = _mm256_or_si256(v, _mm256_srli_epi64(v, 24));
    Block b;
    b.codes = ((uint64_t)_mm256_extract_epi64(v, 0) & 0xffffu) | (((uint64_t)_mm256_extract_epi64(v, 1) & 0xffffu) << 16) |
              (((uint64_t)_mm256_extract_epi64(v, 2) & 0xffffu) << 32) | (((uint64_t)_mm256_extract_epi64(v, 3) & 0xffffu) << 48);
    b.except = ~(uint32_t)_mm256_movemask_epi8(ok);
    return b;
}

bool have_avx2()
{
    // MIEKKI_PACK_SCALAR=1 forces the portable form (tests cover both)
    static const bool yes = __builtin_cpu_supports("avx2") && !(getenv("MIEKKI_PACK_SCALAR") && atoi(getenv("MIEKKI_PACK_SCALAR")));
    return yes;
}

// bits [at, at + n) of a little-endian bit string held in 64-bit words := the low n bits of v (n <= 64); the word
// that holds `at` keeps its bits below `at`, every bit above at + n in the words touched becomes zero
inline void put_bits(uint64_t *w, uint64_t at, uint64_t v, unsigned n)
{
    if (n < 64) v &= (1ull << n) - 1;
    const uint64_t i = at >> 6;
    const unsigned s = (unsigned)(at & 63u);
    if (s == 0) { w[i] = v; return; }
    w[i] = (w[i] & ((1ull << s) - 1)) | (v << s);
    if (s + n > 64) w[i + 1] = v >> (64 - s);
}

}  // namespace

extern "C" {

uint64_t mk_pack_code_words(uint64_t len) { return (len + 31) / 32 + 1; }      // one word of slack: appends write whole words
uint64_t mk_pack_except_words(uint64_t len) { return (len + 63) / 64 + 1; }

int mk_pack_append(uint64_t *codes, uint64_t *except, uint64_t at, const char *chars, uint64_t n)
{
    if (!codes || !except || (n && !chars)) return -1;
    const unsigned char *c = reinterpret_cast<const unsigned char *>(chars);
    const bool avx2 = have_avx2();
    uint32_t any = 0;
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const Block b = avx2 ? pack32_avx2(c + i) : pack32_scalar(c + i);
        put_bits(codes, 2 * (at + i), b.codes, 64);
        put_bits(except, at + i, b.except, 32);
        any |= b.except;
    }
    if (i < n) {
        unsigned char tail[32];
        const unsigned m = (unsigned)(n - i);
        memset(tail, 'A', sizeof tail);                            // padding packs to code 0, no exception
        memcpy(tail, c + i, m);
        const Block b = avx2 ? pack32_avx2(tail) : pack32_scalar(tail);
        put_bits(codes, 2 * (at + i), b.codes, 2 * m);
        put_bits(except, at + i, b.except, m);
        any |= b.except;
    }
    return any ? 1 : 0;
}

}  // extern "C"
