/* Deterministic synthetic genome / query generator.
 *
 * TEST INFRASTRUCTURE.  Shared by the CPU oracle, the reference harness and the
 * bench's cpu_baseline leg so that every side sees byte-identical inputs.  The
 * product has its own device-side statement of the same formula
 * (miekki_amd/csrc/synth.hip); tests compare the two.
 */
#ifndef MIEKKI_ORACLE_SYNTH_H
#define MIEKKI_ORACLE_SYNTH_H
#include <stdint.h>
#include <stddef.h>

#define MK_SEED_G 0x4D49454B4B490001ULL
#define MK_SEED_Q 0x4D49454B4B490002ULL
#define MK_SEED_M 0x4D49454B4B490003ULL

static inline uint64_t mk_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* base i of genome g: 32 bases per 64-bit word, most significant pair first */
static inline char mk_genome_base(uint64_t g, uint64_t i)
{
    uint64_t w = mk_splitmix64(MK_SEED_G ^ (g << 32) ^ (i >> 5));
    return "ACGT"[(w >> (62 - 2 * (i & 31))) & 3];
}

static inline void mk_genome_fill(uint64_t g, uint64_t off, uint64_t n, char *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = mk_genome_base(g, off + i);
}

/* query q of a collection of G genomes of length L: qlen bases cut from genome
 * q mod G at offset splitmix64(SEED_Q ^ q) mod (L - qlen) */
static inline void mk_query_origin(uint64_t q, uint64_t G, uint64_t L, uint64_t qlen,
                                   uint64_t *g, uint64_t *off)
{
    *g = q % G;
    *off = mk_splitmix64(MK_SEED_Q ^ q) % (L - qlen);
}

#endif
