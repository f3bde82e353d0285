// The index build from PACKED sequences (gfx950 only): 2 bits per base + exception bits, the
// form genomes arrive in through mk_index_append_packed, are generated in by the synthetic
// source, and are brought into first when they arrive as characters (mk_index_append).
//
// Replaces Miekki::minhash_sketch_partition (Miekki.cpp:150-197) and the body of
// insert_sequences (Miekki.cpp:287-311) for whole batches of genomes:
//   pack_kernel        characters -> codes + exception bits + "has exceptions" flag    (A1)
//   seed_fix_kernel    the k-1 seed digits as str2numstrand / rcb make them            (A2)
//   build_scatter_kernel   the k-mers (windows of the packed digits), revhash64, partition +
//                      fingerprint; the items (fingerprint, partition, position) of a
//                      workgroup's 4096 k-mers sorted by BIN (the partition's high bits) in LDS
//                      and written as they lie -- a dense array plus one meta word per
//                      (workgroup, bin); k-mers whose Bloom cell may still be empty are flagged
//                      and lead their run                                                (A3-A6, A9)
//   meta_transpose_kernel  the meta words bin-major, as the reduce workgroups read them
//   build_reduce_kernel    one workgroup per (genome, bin): per-partition minimum of
//                      (fingerprint, position) in LDS, then for the winners: fingerprint
//                      bytes, sketch_size / cardinality sums, Bloom pass A for the
//                      flagged ones                                                     (A4, A7, A9)
// followed by fp_transpose_kernel, bloom_sweep_kernel and bloom_summary_kernel (sketch.hip).
// Runs of long queries take the same kernels (launch_query_tables): no flags, the reduce kernel leaves the keys in a table.
//
// Selection rule: in every partition the k-mer with the smallest
// fingerprint wins, the earliest position among equals; "empty" can never be stored.
#include <cmath>
#include <cstdlib>

#include "codes.hpp"
#include "mk_internal.hpp"

namespace mk {

namespace {

constexpr uint32_t kSeg = 4096;            // k-mers per scatter workgroup
constexpr uint32_t kPer = 16;              // consecutive k-mers per thread
constexpr uint32_t kBin = 12;              // log2 partitions per bin = entries of the reduce table
constexpr uint32_t kBins = 1024;           // most bins (h <= 22)
constexpr uint32_t kAllFlagged = 127;      // meta word: the run's flagged count when it is >= this, i.e. "all of them"
}  // namespace

// ---------------------------------------------------------------- characters -> packed
// One thread packs 32 positions: one 64-bit word of codes, 32 exception bits.  A character is
// classified four at a time: (c >> 1) & 3 maps A C G T to 0 1 3 2, x ^ (x >> 1) to 0 1 2 3; a byte
// permute looks "ACGT"[code] up again, and a character that is not what its code stands for is an
// exception (code 0, reverse digit 0: nuc2int / nuc2intrc, utils.cpp:31-49, 107-125).
__device__ __forceinline__ void classify4(uint32_t x, uint32_t &codes8, uint32_t &bad4)
{
    const uint32_t t = (x >> 1) & 0x03030303u;
    uint32_t code = t ^ ((t >> 1) & 0x01010101u);
    const uint32_t back = __builtin_amdgcn_perm(0u, 0x54474341u, code);          // 'A' 'C' 'G' 'T' by code
    const uint32_t d = x ^ back;
    const uint32_t nz = ((((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) >> 7) & 0x01010101u;   // 1 per differing byte
    code &= ~(nz * 3u);
    codes8 = (code * 0x01041040u) >> 24;                                         // four 2-bit codes side by side
    bad4 = ((nz * 0x01020408u) >> 24) & 0xfu;
}

__global__ __launch_bounds__(256) void pack_kernel(const char *__restrict__ seq, const uint64_t *__restrict__ off, uint32_t n,
                                                   uint8_t *__restrict__ codes, uint8_t *__restrict__ except,
                                                   const uint64_t *__restrict__ code_off, uint32_t *__restrict__ dirty)
{
    const uint32_t g = blockIdx.y;
    const uint64_t len = off[g + 1] - off[g];
    const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (w * 32 >= len) return;
    const char *__restrict__ s = seq + off[g] + w * 32;
    const uint32_t m = (uint32_t)min((uint64_t)32, len - w * 32);
    // 32 characters from wherever the sequence starts: whole 8-byte words around them (the buffer
    // starts 256-byte aligned and carries 64 bytes of slack, so the window stays inside it)
    const uintptr_t a = reinterpret_cast<uintptr_t>(s);
    const uint64_t *__restrict__ q = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
    const uint32_t sh = (uint32_t)(a & 7u) * 8u;
    uint64_t v[5];
#pragma unroll
    for (uint32_t j = 0; j < 5; ++j) v[j] = q[j];
    uint64_t word = 0;
    uint32_t bad = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t c8 = sh ? (v[j] >> sh) | (v[j + 1] << (64 - sh)) : v[j];
        uint32_t c0, b0, c1, b1;
        classify4((uint32_t)c8, c0, b0);
        classify4((uint32_t)(c8 >> 32), c1, b1);
        word |= (uint64_t)(c0 | (c1 << 8)) << (16 * j);
        bad |= (b0 | (b1 << 4)) << (8 * j);
    }
    if (m < 32) {                                                   // past the end: code 0, no exception
        word &= (1ULL << (2 * m)) - 1;
        bad &= (1u << m) - 1u;
    }
    reinterpret_cast<uint64_t *>(codes + code_off[g])[w] = word;
    reinterpret_cast<uint32_t *>(except + code_off[g] / 2)[w] = bad;
    if (bad) atomicOr(&dirty[g], 1u);
}

// ---------------------------------------------------------------- the seed
// The first k-1 characters of a sequence enter the rolling state through str2numstrand
// (utils.cpp:252-272: case-insensitive; ANY other character makes the whole seed zero) and their
// reverse digits through rcb (Miekki.cpp:66-76: always 3 - digit).  One thread per sequence
// rewrites those positions of the packed form accordingly -- they lie in the first word -- and
// records the seed's validity for the kernels that work from characters (the fallback path).
// heads: the first 32 characters of every sequence, 32 bytes apart (packed input), or null:
// then they are read from seq + off[g] (both null: nothing to do).
__global__ void seed_fix_kernel(const char *__restrict__ seq, const uint64_t *__restrict__ off, const char *__restrict__ heads,
                                uint32_t n, uint32_t k, uint8_t *__restrict__ codes, uint8_t *__restrict__ except,
                                const uint64_t *__restrict__ code_off, uint32_t *__restrict__ valid)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    if (!heads && !seq) { valid[g] = 1; return; }                   // generated sequences: plain ACGT, the codes are the digits
    const uint64_t len = off[g + 1] - off[g];
    const char *__restrict__ s = heads ? heads + 32u * g : seq + off[g];
    const uint32_t ns = (uint32_t)min((uint64_t)(k - 1), len);
    uint32_t ok = 1;
    uint64_t digits = 0;
    for (uint32_t j = 0; j < ns; ++j) {
        const uint32_t sc = seed_code((uint8_t)s[j]);
        ok &= sc != 4u;
        digits |= (uint64_t)(sc & 3u) << (2 * j);
    }
    if (!ok) digits = 0;
    valid[g] = ok;
    const uint64_t m2 = ns ? ((1ULL << (2 * ns)) - 1) : 0;
    uint64_t *cw = reinterpret_cast<uint64_t *>(codes + code_off[g]);
    cw[0] = (cw[0] & ~m2) | digits;
    uint32_t *xw = reinterpret_cast<uint32_t *>(except + code_off[g] / 2);
    xw[0] &= ~(uint32_t)((1ULL << ns) - 1);
}

// ---------------------------------------------------------------- synthetic genomes, packed
// base i of genome g is the 2-bit field (62 - 2 * (i % 32)) of genome_word(g, i / 32),
// i.e. the packed word is that word with the order of its digits reversed.
__global__ void synth_packed_kernel(uint64_t first_id, uint32_t n, uint64_t len, uint8_t *__restrict__ codes,
                                    const uint64_t *__restrict__ code_off, uint32_t strains, uint32_t rate_ppm)
{
    const uint64_t words = (len + 31) / 32;
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = blockIdx.y;
    if (w >= words || g >= n) return;
    uint64_t v = reverse_digits(strains ? strain_word(first_id + g, w, strains, rate_ppm) : genome_word(first_id + g, w));
    const uint64_t m = len - w * 32;
    if (m < 32) v &= (1ULL << (2 * m)) - 1;
    reinterpret_cast<uint64_t *>(codes + code_off[g])[w] = v;
}

// packed -> characters with the same meaning to every kernel that works from characters: "ACGT"
// by code, 'N' at an exception, and the head characters as they came (they decide the seed).
__global__ void unpack_kernel(const uint8_t *__restrict__ codes, const uint8_t *__restrict__ except,
                              const uint64_t *__restrict__ code_off, const uint32_t *__restrict__ dirty,
                              const char *__restrict__ heads, const uint64_t *__restrict__ off, char *__restrict__ seq)
{
    const uint32_t g = blockIdx.y;
    const uint64_t len = off[g + 1] - off[g];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const uint64_t word = reinterpret_cast<const uint64_t *>(codes + code_off[g])[i >> 5];
    char ch = "ACGT"[(word >> (2 * (i & 31u))) & 3u];
    if (dirty[g] && ((reinterpret_cast<const uint32_t *>(except + code_off[g] / 2)[i >> 5] >> (i & 31u)) & 1u)) ch = 'N';
    if (heads && i < 32) ch = heads[32u * g + i];
    seq[off[g] + i] = ch;
}

// ---------------------------------------------------------------- scatter
struct BuildShape {
    uint32_t nbins, low_bits;      // low_bits = min(h, 12) partitions per bin (log2); nbins = P >> low_bits
    uint32_t lpr;                  // lanes of a reduce wave that share one run, 16 bytes of items each (a power of two)
    uint32_t nwg;                  // scatter workgroups per genome
    uint32_t tune;                 // 0 in the shipped library; timing experiments of a -DMK_TUNE_BUILD build: 1 no Bloom pass A, 2 no item reads
    uint32_t sum_words;            // 64-bit words of the Bloom summary the scatter kernel consults (one bit per 2048 cells); 0: none
    uint32_t bloom_on;             // the index has a Bloom filter (and pass A is not switched off): items get flagged
    uint32_t tables_only;          // query sketches: the reduce kernel leaves the minimum keys in `tables` and nothing else
};

// Item: W == 1: fingerprint << 24 | partition in bin << 12 | position in segment  (32 bits);
//       W == 2: 40 bits of information, FIVE bytes in two arrays (round 3 stored a 64-bit word: twice the bytes of the
//               one-byte build both ways, twice the LDS stage, half the scatter kernel's occupancy):
//                   main  u32  fingerprint << 16 | partition in bin << 4 | position >> 8
//                   low   u8   position & 255
//               items[(genome, workgroup)][i] and low[(genome, workgroup)][i] belong together.
// The segment (= scatter workgroup) supplies the upper bits of the position.  A scatter workgroup leaves its items
// SORTED BY BIN and dense -- items[(genome, workgroup)][0 .. total) in a region of kSeg -- plus one word per bin,
// meta[(genome, workgroup)][bin] = run start << 20 | run length << 7 | flagged: no capacities, no padding, no overflow.
// FLAGGED items are those whose k-mer may still have work to do in the Bloom filter (the scatter kernel has the canonical
// k-mer at hand and asks a summary of the filter, below); they come first in their run, so that the item itself needs no
// bit for it.  kAllFlagged stands for "every item of the run" (and is what a count that does not fit becomes: a flag
// too many only costs the reduce kernel a look at the filter).
template <int W> struct ItemOf { using type = uint32_t; };          // the main word in memory, either width

// HyperMinHash fingerprint of anc's low 64 - h bits (Miekki::mantis, Miekki.cpp:91-113): the float-conversion form of
// mk_device.hpp for all but one k-mer in 2^(24 - f), the general form for those.
__device__ __forceinline__ uint32_t fingerprint_of(uint32_t ahi, uint32_t alo, uint32_t h, uint32_t f, uint32_t empty)
{
    const uint32_t v = __builtin_amdgcn_alignbit(ahi, alo, 32u - h);
    if (__builtin_expect(fingerprint_top32_ok(v, f), 1)) return fingerprint_from_top32(v, f);
    return mantis_halves(ahi & ((1u << (32u - h)) - 1u), alo, h, f, empty);
}
// The scatter kernel converts with rounding TOWARD ZERO: the conversion then drops the low bits itself and the mask that made
// it exact (fingerprint_from_top32) is not needed.  MODE register (hardware register 1), bits 1:0 = rounding of
// single-precision results, 3 = toward zero; until the wave ends.
__device__ __forceinline__ void round_toward_zero() { __builtin_amdgcn_s_setreg(1 | (0 << 6) | ((2 - 1) << 11), 3); }

// LDS by ADDRESS (a 32-bit integer) instead of by pointer: the scatter kernel's block is dynamic, so its base is a symbol the
// compiler adds to every pointer it derives from it, one v_add_u32 per access; an address carried in a register -- in the
// items' keys, in the counters' starts -- is added to once.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ __forceinline__ uint32_t lds_address(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char *)p;
}
__device__ __forceinline__ lds_u32 *lds_word(uint32_t a) { return (lds_u32 *)(uintptr_t)a; }

// The hash loop of one thread: sixteen consecutive k-mers rolled out of three words of packed digits (w0, w1, w2:
// the thread's 48 positions; x*: their exception bits), each one's item and, from an LDS counter, its rank in its bin.
// KBIG: k >= 17 -- the entering reverse digit then always lands in the state's high word and the forward state's low
// word needs no mask; FULL: all sixteen k-mers exist (every workgroup but a sequence's last).  Both are the same for
// the whole workgroup; they only take instructions out of a loop that is bound by instruction issue.
template <int W, bool KBIG, bool FULL>
__device__ __forceinline__ void hash16(uint32_t w0, uint32_t w1, uint32_t w2, bool has_x, uint32_t x01, uint32_t x2, uint32_t i0,
                                       uint32_t cnt, uint32_t ctr_a, uint32_t sum_a, typename ItemOf<W>::type (&it)[kPer],
                                       uint32_t (&key)[kPer], const SketchParams &sp, const BuildShape &bs)
{
    // digit j of the thread's 48 positions at bits 2j: forward digits F, reverse-strand digits R
    const uint64_t F = ((uint64_t)w1 << 32) | w0;
    uint64_t R = ~F;
    uint32_t R2 = ~w2;
    if (has_x) {
        R &= ~spread_pairs32(x01);
        R2 &= ~spread_pairs16(x2);
    }
    const uint32_t km1 = sp.k - 1;                                // 1..30 digits of seed
    const uint64_t seedmask = (1ULL << (2 * km1)) - 1;
    // state after k-1 digits, as the reference's loop leaves it (Miekki.cpp:158-164)
    const uint64_t S0 = reverse_digits(F & seedmask) >> (64 - 2 * km1), RC0 = (R & seedmask) << 2;
    // the sixteen digits that enter, one per k-mer
    const uint32_t fnew = (uint32_t)((F >> (2 * km1)) | ((uint64_t)w2 << (64 - 2 * km1)));
    const uint32_t rnew = (uint32_t)((R >> (2 * km1)) | ((uint64_t)R2 << (64 - 2 * km1)));
    // rolling state in 32-bit halves (a 64-bit shift / and is several issue slots, a funnel shift one)
    uint32_t Slo = (uint32_t)S0, Shi = (uint32_t)(S0 >> 32), Rlo = (uint32_t)RC0, Rhi = (uint32_t)(RC0 >> 32);
    const uint32_t mlo = (uint32_t)sp.kmask, mhi = (uint32_t)(sp.kmask >> 32);
    const uint32_t topshift = 2 * sp.k - 2;                       // even: the entering reverse digit lies within ONE half
    const bool top_hi = KBIG || topshift >= 32;
    const uint32_t tsh = top_hi ? topshift - 32 : topshift;
    // KBIG: no state is carried from k-mer to k-mer -- k-mer u is a 64-bit window of a 96-bit word that holds the seed and all
    // sixteen entering digits, two funnel shifts by a constant and one mask each way (update_kmer / update_kmer_RC,
    // Miekki.cpp:51-62, sixteen times over): forward {S0 | entering digits, first on top, below it}, read 2 (u + 1) bits further down per
    // k-mer; reverse {entering digits << 2k | RC0}, read 2 (u + 1) bits further up.
    const uint32_t S0lo = Slo, S0hi = Shi;
    const uint32_t FR = (uint32_t)(reverse_digits((uint64_t)fnew) >> 32);
    const uint32_t Vlo = Rlo;
    const uint32_t Vmid = KBIG ? Rhi | (rnew << (2 * sp.k - 32)) : 0u, Vtop = KBIG ? rnew >> (64 - 2 * sp.k) : 0u;   // (k >= 17: 2 <= 64 - 2k <= 30)
    const uint32_t bshift = 32u - sp.h;                           // bucket = anc >> (64 - h)  (Miekki.cpp:169)
    const uint32_t sumshift = bs.sum_words ? sp.bloom_log2 - 32u + 3u + 11u : 31u;   // canon's high word -> group of 2048 cells
    const uint32_t wordshift = min(sumshift + 5u, 31u);           // ... -> the summary word of the group (canon < 2^62: >> 31 leaves 0)
    const uint32_t binshift = bshift + bs.low_bits;                // the bin: the partition's bits above the low 12
    // (four loop constants pinned in vector registers: v_bfe_u32 takes one scalar operand, not two, and the compiler would
    // rather make a constant again with a v_mov_b32 per k-mer than keep it)
    uint32_t binbits = sp.h - bs.low_bits, partbits = bs.low_bits, four = 4u, fp_bias = 0x08000000u;
    asm volatile("" : "+v"(binbits), "+v"(partbits), "+v"(four), "+v"(fp_bias));
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        if (KBIG) {
            Slo = __builtin_amdgcn_alignbit(S0lo, FR, 30 - 2 * u);
            Shi = __builtin_amdgcn_alignbit(S0hi, S0lo, 30 - 2 * u) & mhi;
            Rlo = u == kPer - 1 ? Vmid : __builtin_amdgcn_alignbit(Vmid, Vlo, 2 * u + 2);
            Rhi = (u == kPer - 1 ? Vtop : __builtin_amdgcn_alignbit(Vtop, Vmid, 2 * u + 2)) & mhi;
        } else {
            const uint32_t fd = (fnew >> (2 * u)) & 3u, rd = (rnew >> (2 * u)) & 3u;
            Shi = __builtin_amdgcn_alignbit(Shi, Slo, 30) & mhi;                  // update_kmer, Miekki.cpp:51-55
            Slo = ((Slo << 2) | fd) & mlo;
            Rlo = __builtin_amdgcn_alignbit(Rhi, Rlo, 2);                         // update_kmer_RC, Miekki.cpp:59-62
            Rhi >>= 2;
            if (top_hi) Rhi |= rd << tsh; else Rlo |= rd << tsh;
        }
        const uint64_t S = ((uint64_t)Shi << 32) | Slo, RC = ((uint64_t)Rhi << 32) | Rlo;
        const uint64_t canon = S < RC ? S : RC;
        // Would inserting this k-mer into the Bloom filter still change anything (Miekki.cpp:121-131)?  Its five positions
        // are (canon + t_i) >> b with t_i < 1024 and b >= 32 (universal_hash, utils.cpp:197-199): ONE cell, or, when the low
        // word carries, that cell and the next; the summary bit of a group of 2048 cells says that all of them AND the first
        // cell of the next group are taken (bloom_summary_kernel, sketch.hip) -- no test of the low word here.  Cells are never
        // emptied, so a summary older than the filter only flags more.
        // (asked before the hash so that the LDS read is under way while the hash is computed)
        // (no branch here: without a summary the shifts leave group 0 and the one word there says "nothing is settled", or,
        // for an index without a filter, "everything is")
        const uint32_t chi = (uint32_t)(canon >> 32);
        const uint32_t s3 = chi >> sumshift;                                      // cell >> 11
        const uint32_t sword = *lds_word(sum_a + ((chi >> wordshift) << 2));
        const uint32_t settled = __builtin_amdgcn_ubfe(sword, s3, 1u);            // bit s3 & 31 of the word (v_bfe_u32 takes the offset's low five bits)
        const uint64_t anc = revhash64(canon);                                    // Miekki.cpp:167-168
        const uint32_t ahi = (uint32_t)(anc >> 32);
        // The fingerprint where the item wants it, on top (kFpShift = 24 or 16), straight from the float's bits: fp =
        // (bits >> (23 - f)) - (127 << f) (mk_device.hpp) and f = 3 or 11 with the width, so fp << kFpShift is bits << 4 minus
        // 127 << 27, i.e. plus 1 << 27 mod 2^32, with mantissa bits that do not belong to it below kFpShift (the v_bfi_b32
        // below drops them).  The empty fingerprint is all ones.
        constexpr uint32_t kFpShift = W == 1 ? 24 : 16, kFpMask = ~0u << kFpShift;
        const uint32_t v = __builtin_amdgcn_alignbit(ahi, (uint32_t)anc, 32u - sp.h);
        uint32_t fptop;
        if (__builtin_expect(fingerprint_top32_ok(v, sp.f), 1)) fptop = (__float_as_uint((float)v) << 4) + fp_bias;   // (rounding toward zero: see the kernel)
        else fptop = mantis_halves(ahi & ((1u << (32u - sp.h)) - 1u), (uint32_t)anc, sp.h, sp.f, sp.empty) << kFpShift;
        if (fptop >= kFpMask || (!FULL && i0 + u >= cnt)) continue;  // (past the segment's end only in a sequence's last workgroup)
        const uint32_t part = __builtin_amdgcn_ubfe(ahi, bshift, partbits);
        // the address of the bin's counter: flagged, settled (a field of no bits is 0: -h 12 and below have one bin).  The
        // counters count BYTES of items (4 per item): what comes back is the item's place in its run as the stage wants it
        const uint32_t caddr = (ctr_a + (__builtin_amdgcn_ubfe(ahi, binshift, binbits) << 3)) | (settled << 2);
        const uint32_t rank4 = __hip_atomic_fetch_add(lds_word(caddr), four, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (W == 2: the low eight position bits are (i0 + u) & 255, known to whoever stores the item: they are not kept here)
        const uint32_t below = W == 1 ? (part << kBin) | (i0 + u) : (part << 4) | ((i0 + u) >> 8);
        {
            uint32_t word;                                           // (fptop & kFpMask) | (below & ~kFpMask), which the compiler would make of three
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(word) : "s"(kFpMask), "v"(fptop), "v"(below));
            it[u] = word;
        }
        key[u] = (caddr << 14) | rank4;                              // rank < 4096 (among the bin's flagged / settled items); addresses < 2^18
    }
}

// What bounds this kernel is instruction issue and, behind it, the LDS round trips of its four random LDS accesses per
// k-mer (round 2: ~122 vector instructions per k-mer in its predecessor, round 5: 73.5, most of the difference around the
// hash loop, not in it; 51.6 here, PMC, at 4.4 cycles each -- profiles/r6_pmc_build_sq.txt).
// So: the positions arrive packed -- a thread's 16 + k-1 digits are three LDS words, no
// classification, no squeezing; the items of the whole workgroup are sorted by (bin, flagged |
// settled) through ONE 4096-entry stage (4-byte items: 16 KiB, which also lends its first 6 KiB to the positions' words
// and the Bloom summary while nothing is staged yet: eight workgroups per CU, and the waves are what hides the LDS) and
// leave as they lie, 16 bytes per lane -- no per-item address arithmetic.
template <int W, bool KBIG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void build_scatter_kernel(
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ except, const uint64_t *__restrict__ code_off,
    const uint32_t *__restrict__ dirty, const uint64_t *__restrict__ off, uint32_t *__restrict__ items, uint8_t *__restrict__ low,
    uint32_t *__restrict__ meta, const uint32_t *summary, SketchParams sp, BuildShape bs)
{
    using item_t = typename ItemOf<W>::type;
    round_toward_zero();                                              // (the one conversion of this kernel: the fingerprint in hash16)
    constexpr uint32_t kIPV = 4;                                      // items per 16-byte store of the main array
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (the positions' words, cw and xw below, and the Bloom summary lie IN the stage: the hash loop is the last to read them,
    // two barriers before the first item is stored there -- 6 KiB that make the difference between six and eight workgroups
    // per CU, five and seven at 2-byte fingerprints: the kernel runs 557 us at four, 507 at five, 480 at six, 460 at seven)
    uint32_t *stage = reinterpret_cast<uint32_t *>(smem);             // kSeg + kIPV: the workgroup's items (main words), sorted by bin
    uint8_t *stage_low = smem + (kSeg + kIPV) * 4;                    // W == 2: their low position bytes, kSeg + 16
    constexpr uint32_t kStageBytes = (kSeg + kIPV) * 4 + (W == 2 ? kSeg + 16 : 0);
    // 2 * nbins counters (per bin: flagged items, then settled ones; padded with zeros to a multiple of eight), later the
    // places where their runs start -- as LDS addresses in the stage -- and behind them one word that no k-mer counts in (the "dump
    // slot": where the items that do not exist go, below).
    const uint32_t nctr = 2 * bs.nbins, pad8 = (nctr + 7u) & ~7u;
    uint32_t *ctr = reinterpret_cast<uint32_t *>(smem + kStageBytes);   // (16-byte aligned)
    // the Bloom summary (one bit per 2048 cells: all taken), when there is one: 4 KiB at -b 33 -- as a snapshot of whatever
    // the array holds right now (the summary kernel of the batch before may be writing it: bits only ever get set)
    uint32_t *wave_sum = ctr + pad8 + 4;                              // [4] (behind the dump slot's 16 bytes)
    constexpr uint32_t kWordsBytes = 2 * (kSeg / 16 + 4) * 4;         // cw, xw: 2,080 bytes
    uint32_t *sum32 = reinterpret_cast<uint32_t *>(smem + ((kWordsBytes + 15u) & ~15u));   // (at least one word; at most 8 KiB: build_setup)
    static_assert(((kWordsBytes + 15u) & ~15u) + 8192 <= kSeg * 4, "the positions' words and the summary fit in the stage");
    uint32_t *cw = reinterpret_cast<uint32_t *>(smem);                // [kSeg / 16 + 4] the workgroup's positions, 16 per word (see above)
    uint32_t *xw = cw + kSeg / 16 + 4;                                // [kSeg / 16 + 4] their exception bits (low 16)
    const uint32_t g = blockIdx.y, wg = blockIdx.x, tid = threadIdx.x;
    const uint64_t len = off[g + 1] - off[g];
    const uint64_t nk = len > sp.k ? len - sp.k : 0;                  // Miekki.cpp:162: the last k-mer is skipped
    const uint64_t seg0 = (uint64_t)wg * kSeg;
    for (uint32_t b = tid; b < pad8; b += 256) ctr[b] = 0;
    if (tid == 0) ctr[pad8] = lds_address(stage) + kSeg * 4u;         // (the stage's slack behind its last item)
    // (the summary: at most 8 KiB = two 16-byte pieces per thread, both requested before the first is stored -- the workgroup
    // lives ~12 us, a chain of dependent loads at its start would be a third of that)
    const uint4 *__restrict__ s4 = reinterpret_cast<const uint4 *>(summary);
    const uint32_t pieces = (bs.sum_words + 1) / 2;                  // of 16 bytes (the array is padded)
    uint4 sum_a = make_uint4(0, 0, 0, 0), sum_b = sum_a;
    if (tid < pieces) sum_a = s4[tid];
    if (tid + 256 < pieces) sum_b = s4[tid + 256];
    const uint32_t cnt = seg0 < nk ? (uint32_t)min((uint64_t)kSeg, nk - seg0) : 0u;
    const bool has_x = dirty[g] != 0;                                 // workgroup-uniform
    if (cnt) {
        // positions [seg0, seg0 + cnt + k - 1): words seg0 / 16 .. of the sequence's codes (the arrays carry slack)
        const uint32_t *__restrict__ c32 = reinterpret_cast<const uint32_t *>(codes + code_off[g]) + seg0 / 16;
        const uint16_t *__restrict__ x16 = reinterpret_cast<const uint16_t *>(except + code_off[g] / 2) + seg0 / 16;
        const uint64_t last_word = (len + 15) / 16;                   // words that hold positions of this sequence
        for (uint32_t j = tid; j < kSeg / 16 + 2; j += 256) {
            const bool in = seg0 / 16 + j < last_word;
            cw[j] = in ? c32[j] : 0u;
            xw[j] = in && has_x ? (uint32_t)x16[j] : 0u;
        }
    }
    if (tid < pieces) reinterpret_cast<uint4 *>(sum32)[tid] = sum_a;
    if (tid + 256 < pieces) reinterpret_cast<uint4 *>(sum32)[tid + 256] = sum_b;
    if (!bs.sum_words && tid == 0) sum32[0] = bs.bloom_on ? 0u : 1u;
    __syncthreads();
    const uint32_t i0 = tid * kPer;
    item_t it[kPer];
    // LDS address of the bin's counter << 14 | 4 x rank in bin.  no_item: the dump slot's address, rank 0 -- it goes where that
    // slot says, behind the stage's last item (no test per item below; there is one such k-mer in 2^(24 - f), and a sequence's tail)
    uint32_t key[kPer];
    const uint32_t ctr_a = lds_address(ctr), sum32_a = lds_address(sum32), stage_a = lds_address(stage);
    const uint32_t no_item = (ctr_a + pad8 * 4u) << 14;
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) key[u] = no_item;
    if (i0 < cnt) {
        const uint32_t w0 = cw[tid], w1 = cw[tid + 1], w2 = cw[tid + 2];
        const uint32_t x01 = has_x ? xw[tid] | (xw[tid + 1] << 16) : 0u, x2 = has_x ? xw[tid + 2] : 0u;
        if (cnt == kSeg) hash16<W, KBIG, true>(w0, w1, w2, has_x, x01, x2, i0, cnt, ctr_a, sum32_a, it, key, sp, bs);
        else             hash16<W, KBIG, false>(w0, w1, w2, has_x, x01, x2, i0, cnt, ctr_a, sum32_a, it, key, sp, bs);
    }
    __syncthreads();
    // ---- where each run starts among the workgroup's sorted items: exclusive prefix of the counts.  A thread: eight
    // counters = four bins, two 16-byte reads; only the waves that have counters take part (-h 20: 512 counters, one wave
    // of the four -- the scan's six steps are then paid once per workgroup); the prefix within a wave by DPP moves.
    uint32_t c8[8], start4[4];
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const bool scanning = wave * 512u < nctr;                         // (wave-uniform)
    uint32_t sum = 0, incl = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; ++e) c8[e] = 0;
    if (scanning) {
        if (tid * 8 < pad8) {
            const uint4 a = reinterpret_cast<const uint4 *>(ctr)[2 * tid], b = reinterpret_cast<const uint4 *>(ctr)[2 * tid + 1];
            c8[0] = a.x; c8[1] = a.y; c8[2] = a.z; c8[3] = a.w; c8[4] = b.x; c8[5] = b.y; c8[6] = b.z; c8[7] = b.w;
        }
        sum = ((c8[0] + c8[1]) + (c8[2] + c8[3])) + ((c8[4] + c8[5]) + (c8[6] + c8[7]));
        incl = wave_prefix_sum(sum);
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();                                                  // (every count has been read: the starts may overwrite them)
    if (scanning) {
        uint32_t base = stage_a + incl - sum;
        for (uint32_t w = 0; w < wave; ++w) base += wave_sum[w];
        uint32_t s8[8];
#pragma unroll
        for (uint32_t e = 0; e < 8; ++e) {
            s8[e] = base;
            if (!(e & 1u)) start4[e >> 1] = base;
            base += c8[e];
        }
        if (tid * 8 < pad8) {
            reinterpret_cast<uint4 *>(ctr)[2 * tid] = make_uint4(s8[0], s8[1], s8[2], s8[3]);
            reinterpret_cast<uint4 *>(ctr)[2 * tid + 1] = make_uint4(s8[4], s8[5], s8[6], s8[7]);
        }
    }
    __syncthreads();
    const uint32_t total = (wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3]) >> 2;
    // (all sixteen starts asked for before the first item is stored: a read behind a write would wait for it)
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) key[u] = *lds_word(key[u] >> 14) + (key[u] & 0x3FFCu);
    const uint32_t low_bias = lds_address(stage_low) - (stage_a >> 2);    // (W == 2: item at stage_a + 4 i <-> byte at stage_low + i)
#pragma unroll
    for (uint32_t u = 0; u < kPer; ++u) {
        *lds_word(key[u]) = it[u];
        if (W == 2) *(lds_u8 *)(uintptr_t)((key[u] >> 2) + low_bias) = (uint8_t)(i0 + u);
    }
    __syncthreads();
    // ---- out: the sorted items as they lie, 16 bytes per lane (a pure stream), and one word per bin
    const uint64_t seg = (uint64_t)g * bs.nwg + wg;
    uint4 *__restrict__ dst = reinterpret_cast<uint4 *>(items + seg * kSeg);
    const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(stage);
    for (uint32_t i = tid; i * kIPV < total; i += 256) dst[i] = src[i];
    if (W == 2) {
        uint4 *__restrict__ dl = reinterpret_cast<uint4 *>(low + seg * kSeg);
        const uint4 *__restrict__ sl = reinterpret_cast<const uint4 *>(stage_low);
        for (uint32_t i = tid; i * 16 < total; i += 256) dl[i] = sl[i];
    }
    if (tid * 4 < bs.nbins) {                                         // (a thread's eight counters are four bins)
        uint32_t *__restrict__ m = meta + seg * bs.nbins + tid * 4;
        uint32_t w4[4];
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e)                              // run start << 20 | run length << 7 | flagged, from byte counts; length <= 4096
            w4[e] = (((start4[e] - stage_a) & (4u * kSeg - 4u)) << 18) | ((c8[2 * e] + c8[2 * e + 1]) << 5) | min(c8[2 * e] >> 2, kAllFlagged);
        if (bs.nbins >= 4) *reinterpret_cast<uint4 *>(m) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        else {
            m[0] = w4[0];
            if (bs.nbins > 1) m[1] = w4[1];
        }
    }
}

// ---------------------------------------------------------------- the meta words, bin-major
// A reduce workgroup wants the words of ONE bin from every scatter workgroup of its genome: in the order the scatter kernel
// leaves them -- meta[(genome, workgroup)][bin], a workgroup's 16-byte stores -- that is 1,221 loads of four bytes from 1,221
// lines at the head of every reduce workgroup, 20 M requests per 64 x 5 Mb batch and 0.15 of the reduce kernel's 0.49 ms
// (measured with the words made up instead of loaded).  Written bin-major by the scatter kernel they were as many four-byte
// STORES to as many lines and cost what they saved.  This pass turns them once, 64 x 64 words through LDS, rows in and rows
// out: 160 MB of traffic per batch for metaT[(genome, bin)][workgroup], which the reduce workgroup reads as 39 lines.
// (pitch: words per row of metaT, nwg rounded up to a multiple of four, so that rows start on 16 bytes.  VEC: nbins is a
// multiple of four -- 16 bytes per lane both ways; one to three bins, -h 13 and below, take the word-by-word form.)
template <bool VEC>
__global__ __launch_bounds__(256) void meta_transpose_kernel(const uint32_t *__restrict__ meta, uint32_t *__restrict__ metaT,
                                                             uint32_t nwg, uint32_t nbins, uint32_t pitch)
{
    __shared__ uint32_t tile[64][65];                                 // [workgroup][bin]; 65: a column is read across 64 banks
    const uint32_t g = blockIdx.z, w0 = blockIdx.x * 64u, b0 = blockIdx.y * 64u;
    const uint32_t *__restrict__ src = meta + (uint64_t)g * nwg * nbins;
    uint32_t *__restrict__ dst = metaT + (uint64_t)g * nbins * pitch;
    if (VEC) {
        const uint32_t q = (threadIdx.x & 15u) * 4u, r0 = threadIdx.x >> 4;   // four words of a row; sixteen rows per step
#pragma unroll
        for (uint32_t r = r0; r < 64u; r += 16u)
            if (w0 + r < nwg && b0 + q < nbins) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + (uint64_t)(w0 + r) * nbins + b0 + q);
                tile[r][q] = v.x; tile[r][q + 1] = v.y; tile[r][q + 2] = v.z; tile[r][q + 3] = v.w;
            }
        __syncthreads();
#pragma unroll
        for (uint32_t r = r0; r < 64u; r += 16u)
            if (b0 + r < nbins && w0 + q < nwg)                       // (the row's last words may lie in its padding: pitch >= nwg rounded up)
                *reinterpret_cast<uint4 *>(dst + (uint64_t)(b0 + r) * pitch + w0 + q) = make_uint4(tile[q][r], tile[q + 1][r], tile[q + 2][r], tile[q + 3][r]);
    } else {
        const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
        for (uint32_t r = ty; r < 64u; r += 4u)
            if (w0 + r < nwg && b0 + tx < nbins) tile[r][tx] = src[(uint64_t)(w0 + r) * nbins + b0 + tx];
        __syncthreads();
        for (uint32_t r = ty; r < 64u; r += 4u)
            if (b0 + r < nbins && w0 + tx < nwg) dst[(uint64_t)(b0 + r) * pitch + w0 + tx] = tile[tx][r];
    }
}

// ---------------------------------------------------------------- reduce + fingerprints + sizes + Bloom pass A
// One 512-thread workgroup per (genome, bin) -- 38 KiB of LDS, so four of them share a CU: the kernel is a chain of
// latencies (meta words, then items) and what hides them is other workgroups.  The bin's runs -- one per scatter
// workgroup, found through the workgroup's meta word (run start, run length, flagged items) -- are read with `lpr`
// lanes per run, 16 bytes per lane.  The items of a genome are one dense array that the 2^(h-12) reduce workgroups of
// the genome walk side by side (neighbouring bins on one XCD, so that a line that holds the end of one run and the
// start of the next is fetched into one L2): HBM sees it once.  Minimum per partition with LDS atomics; KEY32 packs
// (fingerprint, position) into 32 bits when the sequence is shorter than 2^24 (a 16 KiB table).
// Bloom pass A works on FLAGGED winners only (the scatter kernel's flag: the k-mer's cell may still be empty): flagged
// items are noted in an LDS list while they pass, and those that turn out to be their partition's minimum get the
// filter looked at.  Once the filter has filled up nothing is flagged and a winner costs no memory request at all
// (before: its codes, 16 bytes from a random place of the genome -- 67 M L2 requests per 64 x 5 Mb batch).
#ifndef MK_REDUCE_STAGES
#define MK_REDUCE_STAGES 4
#define MK_REDUCE_UN 1
#endif
template <int W, bool KEY32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((W == 1 && KEY32) ? 8 : 6, 8))) void build_reduce_kernel(
    const typename ItemOf<W>::type *__restrict__ items, const uint8_t *__restrict__ low, const uint32_t *__restrict__ meta,
    const uint8_t *__restrict__ codes, const uint8_t *__restrict__ except, const uint64_t *__restrict__ code_off,
    const uint32_t *__restrict__ dirty, const uint8_t *bloom, uint64_t bloom_dev_bytes, uint32_t *order,
    const uint32_t *__restrict__ full, uint8_t *__restrict__ fp_out, uint64_t *__restrict__ tables,
    uint8_t *__restrict__ touched, uint32_t *__restrict__ active, unsigned long long *__restrict__ cardsum,
    SketchParams sp, BuildShape bs)
{
    using item_t = typename ItemOf<W>::type;
    using key_t = typename std::conditional<KEY32, uint32_t, unsigned long long>::type;
    using fp_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;
    constexpr uint32_t kKeyPos = KEY32 ? 24 : 40;                    // key = fingerprint << kKeyPos | position
    constexpr key_t kNoKey = (key_t)~(key_t)0;
    constexpr uint32_t kThreads = 512, kWin = (1u << kBin) / kThreads;   // winners per thread
    constexpr uint32_t kMetaChunk = 1280;                             // scatter workgroups whose meta words sit in LDS at a time (5 Mb: 1,221)
    // flagged items seen: partition in bin << 32 | position.  More than fit: every winner of the bin gets the filter
    // looked at (that is the state of a young filter, where nearly everything is flagged anyway)
    constexpr uint32_t kNoted = KEY32 ? 2048 : 1024;                  // (64-bit keys: a 32 KiB table -- 46 KiB in all, three workgroups per CU)
    // (one object, the table first: at LDS address 0 the atomic minimum's address is the read's -- with the table behind the
    // list the compiler added the table's offset once more for every item)
    struct Shared {
        key_t table[1u << kBin];
        unsigned long long noted[kNoted];
        unsigned long long card;
        uint32_t meta[kMetaChunk];
        uint32_t to_check[(1u << kBin) / 32];                         // partitions whose winner is a flagged item
        uint32_t act, n_noted;
    };
    __shared__ Shared sh;
    key_t (&table)[1u << kBin] = sh.table;
    unsigned long long (&noted)[kNoted] = sh.noted;
    uint32_t (&s_meta)[kMetaChunk] = sh.meta;
    uint32_t (&to_check)[(1u << kBin) / 32] = sh.to_check;
    uint32_t &s_act = sh.act, &n_noted = sh.n_noted;
    unsigned long long &s_card = sh.card;
    // workgroups are dealt to the XCDs round robin: XCD x gets the bins x * nbins / 8 ... of a genome, i.e. neighbours
    const uint32_t g = blockIdx.y;
    const uint32_t bin = (bs.nbins & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (bs.nbins >> 3) + (blockIdx.x >> 3);
    const uint32_t R = 1u << bs.low_bits;
    // (the table and the counters are set up while the first meta words are on their way: below)
    auto init_shared = [&] {
        for (uint32_t i = threadIdx.x; i < R; i += kThreads) table[i] = kNoKey;
        if (threadIdx.x < (1u << kBin) / 32) to_check[threadIdx.x] = 0;
        if (threadIdx.x == 0) { s_act = 0; s_card = 0; n_noted = 0; }
    };
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    {
        // lpr lanes per run, 16 bytes (kIPL items) per lane.  A run starts wherever the bin's items start in their
        // workgroup's sorted array; the lanes load from the 16-byte boundary below it and drop what lies before the run
        constexpr uint32_t kIPL = 16 / sizeof(item_t);
        typedef item_t vec_t __attribute__((ext_vector_type(kIPL)));
        const uint32_t lpr = bs.lpr, per_wave = 64u / lpr;            // runs per wave-instruction
        const uint32_t sub = lane / lpr, j0 = (lane % lpr) * kIPL;
        const uint32_t nwg = (bs.tune & 2u) ? 0u : bs.nwg;
        // KR stages of UN wave-loads each are in flight per wave (see the loop at the end of this block)
        constexpr uint32_t NW = kThreads / 64, UN = MK_REDUCE_UN, KR = MK_REDUCE_STAGES;
        constexpr uint32_t kMetaPer = (kMetaChunk + kThreads - 1) / kThreads;
        if (!nwg) init_shared();
        for (uint32_t c0 = 0; c0 < nwg; c0 += kMetaChunk) {
            const uint32_t cn = min(kMetaChunk, nwg - c0);
            if (c0) __syncthreads();                                  // the previous chunk's words have been used
            const item_t *__restrict__ chunk_items = items + ((uint64_t)g * bs.nwg + c0) * kSeg;
            const uint8_t *__restrict__ chunk_low = low + ((uint64_t)g * bs.nwg + c0) * kSeg;
            uint32_t mw[kMetaPer];
#pragma unroll
            for (uint32_t u = 0; u < kMetaPer; ++u) {                 // (all of a thread's words requested before the first is stored)
                const uint32_t i = threadIdx.x + u * kThreads;
                mw[u] = meta[((uint64_t)g * bs.nbins + bin) * ((bs.nwg + 3u) & ~3u) + c0 + min(i, cn - 1u)];   // bin-major (meta_transpose_kernel); (no branch: see the item loads below)
            }
            if (!c0) init_shared();                                   // (under the loads' latency)
#pragma unroll
            for (uint32_t u = 0; u < kMetaPer; ++u) {
                const uint32_t i = threadIdx.x + u * kThreads;
                if (i < cn) s_meta[i] = mw[u];
            }
            __syncthreads();
            // two stages of UN wave-loads each: the runs of the next stage are requested before the items of the
            // current one go to the table, so that the LDS atomics of one stage drain under the loads of the next
            struct Stage { uint32_t m[UN]; vec_t v[UN]; uint32_t lo[W == 2 ? UN : 1]; };   // (lo: W == 2, the items' low position bytes)
            auto place = [&](uint32_t m, uint32_t &a0, uint32_t &first, uint32_t &end) {
                first = m >> 20;
                end = first + ((m >> 7) & 0x1fffu);
                a0 = first & ~(kIPL - 1u);
            };
            // (no branch around the loads: every lane requests its 16 bytes, needed or not -- they lie in lines the run's
            // neighbours need anyway -- so that the loads of the two stages are plain straight-line code whose order the
            // compiler's wait counts can follow; a lane without a run asks for the chunk's last run)
            auto fetch = [&](uint32_t r0, Stage &q) {
#pragma unroll
                for (uint32_t u = 0; u < UN; ++u) {
                    const uint32_t r = r0 + u * NW * per_wave + sub;
                    q.m[u] = r < cn ? s_meta[r] : 0u;
                    uint32_t a0, first, end;
                    place(q.m[u], a0, first, end);
                    // (a load may reach up to lpr * kIPL places past the run's start: inside the array, whose last workgroup
                    // is followed by the meta words)
                    // (a 32-bit offset from the chunk's first workgroup -- at most 1,280 x 4,096 items -- so that the address is the
                    // load's scalar base plus one register instead of four 64-bit additions per lane)
                    const uint32_t at = min(r, cn - 1u) * kSeg + a0 + j0;
                    q.v[u] = *reinterpret_cast<const vec_t *>(reinterpret_cast<const unsigned char *>(chunk_items) + at * (uint32_t)sizeof(item_t));
                    if (W == 2) q.lo[u] = *reinterpret_cast<const uint32_t *>(chunk_low + at);   // (a0 + j0 is a multiple of four)
                }
            };
            auto consume = [&](uint32_t r0, const Stage &q) {
#pragma unroll
                for (uint32_t u = 0; u < UN; ++u) {
                    const uint32_t w = c0 + r0 + u * NW * per_wave + sub;
                    uint32_t a0, first, end;
                    place(q.m[u], a0, first, end);
                    // the flagged items are the first ones of their run (rare once the filter has filled up: none)
                    // (a run without flagged items has 0 here: the wave asks that alone before anything is made of it)
                    const uint32_t nf = q.m[u] & 127u;
                    auto flagged_end = [](uint32_t m) { const uint32_t f = m >> 20, n = m & 127u; return n == kAllFlagged ? f + ((m >> 7) & 0x1fffu) : f + n; };
                    // (lob: the item's low position byte, W == 2 only)
                    // The run's flagged items -- the first (fend - first) of it -- get their places in the list from ONE atomic per
                    // run, by the run's first lane (a young filter flags every item, 18,700 per workgroup: one atomic WITH return
                    // each on one LDS word cost 0.15 ms of LDS time per CU and batch; the meta word already says how many there are)
                    // (Once the filter has filled up no run of a wave-load has a flagged item -- the usual case of a large build: the
                    // whole wave then takes the form of the item loop without slots, bounds against `fend` and the list's store.)
                    const bool noting = __ballot(nf != 0u) != 0ull;                     // wave-uniform (fend != first)
                    uint32_t nbase = 0;
                    const uint32_t fend = noting ? flagged_end(q.m[u]) : first;
                    if (noting) {
                        const uint32_t nflag = fend - first;                        // (0 for a lane without a run)
                        if (nflag && lane % lpr == 0) nbase = atomicAdd(&n_noted, nflag);
                        nbase = (uint32_t)__shfl((int)nbase, (int)(lane - lane % lpr));
                    }
                    // slot: the item's place in the list of flagged items, or beyond it for an item that is not flagged
                    auto fold = [&](auto with_list, uint32_t w, item_t item, uint32_t lob, uint32_t slot) {
                        // the partition's table entry as a BYTE offset: one shift and one mask from the item
                        constexpr uint32_t kKeyLog = KEY32 ? 2 : 3, kPartShift = (W == 1 ? 12 : 4) - kKeyLog;
                        key_t *const entry = reinterpret_cast<key_t *>(reinterpret_cast<unsigned char *>(table) + ((item >> kPartShift) & ((R - 1u) << kKeyLog)));
                        key_t key;
                        if (KEY32) key = (key_t)((item & 0xff000fffu) | (w << 12));                // fingerprint << 24 | position
                        else {
                            const uint32_t pos = W == 1 ? item & (kSeg - 1u) : ((item & 0xfu) << 8) | lob;
                            key = ((key_t)(item >> (W == 1 ? 24 : 16)) << kKeyPos) | (key_t)((uint64_t)w * kSeg + pos);
                        }
                        // (a look before the atomic: repeat-rich sequence sends the same k-mer -- same partition, same fingerprint,
                        // a later position -- again and again, and 64 lanes of an atomic minimum on ONE entry cost 620 cycles where
                        // a read of one entry is a broadcast, profiles/r3_ubench.txt; a key that cannot win is dropped here.
                        // profiles/r4_repeat_rich.txt: 20 % of a genome in tandem repeats tripled this kernel's time without it)
                        if (key < *entry) atomicMin(entry, key);
                        if constexpr (decltype(with_list)::value) {
                            const uint32_t part = W == 1 ? (item >> 12) & (R - 1u) : (item >> 4) & (R - 1u);
                            const uint32_t pos = W == 1 ? item & (kSeg - 1u) : ((item & 0xfu) << 8) | lob;
                            if (slot < kNoted) noted[slot] = ((unsigned long long)part << 32) | (uint32_t)((uint64_t)w * kSeg + pos);
                        }
                    };
                    // (the lane's items e = 0 .. kIPL-1 lie at a0 + j0 + e: inside the run for lo <= e < hi -- compared as they are,
                    // against two numbers per lane, instead of one more addition per item)
                    const int lo = (int)first - (int)(a0 + j0), hi = (int)end - (int)(a0 + j0);
                    if (noting) {
#pragma unroll
                        for (uint32_t e = 0; e < kIPL; ++e) {
                            const uint32_t at = a0 + j0 + e;
                            if ((int)e >= lo && (int)e < hi)
                                fold(std::true_type{}, w, q.v[u][e], W == 2 ? (q.lo[u] >> (8 * e)) & 0xffu : 0u, at < fend ? nbase + (at - first) : ~0u);
                        }
                    } else {
#pragma unroll
                        for (uint32_t e = 0; e < kIPL; ++e)
                            if ((int)e >= lo && (int)e < hi) fold(std::false_type{}, w, q.v[u][e], W == 2 ? (q.lo[u] >> (8 * e)) & 0xffu : 0u, ~0u);
                    }
                    // Runs longer than their lanes reach.  Rare on ordinary sequence (the lanes cover mean + 4 sigma), the rule on
                    // repeat-rich sequence: a tandem repeat sends a whole stretch of k-mers into ONE partition, i.e. hundreds or
                    // thousands of items into one run.  Left to the run's own lpr lanes that is a chain of dependent loads, 32 items
                    // at a time, while the rest of the workgroup waits at the barrier (profiles/r4_repeat_rich.txt: 20 % of a
                    // genome in repeats made this kernel 3.5 x slower); the WHOLE WAVE takes such a tail, 256 items per load.
                    unsigned long long tails = __ballot(end > a0 + lpr * kIPL);
                    while (tails) {                                                             // (wave-uniform)
                        const uint32_t l = (uint32_t)__ffsll((long long)tails) - 1u;            // first lane of a sub-group with a tail
                        tails &= ~(((lpr == 64 ? ~0ull : ((1ull << lpr) - 1ull))) << l);
                        // (the run's meta word says where it starts and ends and how many of it are flagged)
                        const uint32_t t_m = (uint32_t)__shfl((int)q.m[u], (int)l), t_w = (uint32_t)__shfl((int)w, (int)l);
                        const uint32_t t_nbase = (uint32_t)__shfl((int)nbase, (int)l);
                        uint32_t t_a0, t_first, t_end;
                        place(t_m, t_a0, t_first, t_end);
                        const uint32_t t_fend = flagged_end(t_m);
                        for (uint32_t j = t_a0 + lpr * kIPL + lane * kIPL; j < t_end; j += 64u * kIPL) {
                            const uint64_t at = ((uint64_t)g * bs.nwg + t_w) * kSeg + j;
                            const vec_t x = *reinterpret_cast<const vec_t *>(items + at);
                            const uint32_t xl = W == 2 ? *reinterpret_cast<const uint32_t *>(low + at) : 0u;
#pragma unroll
                            for (uint32_t e = 0; e < kIPL; ++e)
                                if (j + e < t_end) fold(std::true_type{}, t_w, x[e], (xl >> (8 * e)) & 0xffu, j + e < t_fend ? t_nbase + (j + e - t_first) : ~0u);
                        }
                    }
                }
            };
            // A ring of KR stages: while one stage's items go to the table the loads of the KR - 1 after it are in flight, and a
            // stage is requested again the moment it has been consumed.  (Round 4 had two stages of two wave-loads: the loads a
            // consume waited for had been requested ONE consume earlier, and half the kernel's wave-cycles were such waits,
            // profiles/r5_pmc_build_sq.txt; with four stages of one wave-load -- the same registers -- they are three
            // consumes old.)  A stage past the chunk's end holds no items.
            constexpr uint32_t kStep = NW * UN;                       // wave-loads a stage of the whole workgroup covers
            const uint32_t step = kStep * per_wave;
            Stage S[KR];
            uint32_t r0 = wave * per_wave;
#pragma unroll
            for (uint32_t i = 0; i + 1 < KR; ++i) fetch(r0 + i * step, S[i]);
            for (; r0 < cn; r0 += KR * step) {
#pragma unroll
                for (uint32_t i = 0; i < KR; ++i) {
                    fetch(r0 + (i + KR - 1) * step, S[(i + KR - 1) % KR]);
                    consume(r0 + i * step, S[i]);
                }
            }
        }
    }
    __syncthreads();
    if (bs.tables_only) {
        // a run of long queries: the minimum keys as the gating kernels of sketch.hip read them, nothing else
#pragma unroll
        for (uint32_t j = 0; j < kWin; ++j) {
            const uint32_t i = threadIdx.x + kThreads * j;
            if (i >= R) continue;
            const key_t key = table[i];
            tables[(uint64_t)g * sp.P + (uint64_t)bin * R + i] =
                key == kNoKey ? kEmptyKey : ((uint64_t)(key >> kKeyPos) << kPosBits) | ((uint64_t)key & ((1ULL << kKeyPos) - 1));
        }
        return;
    }
    // which winners are flagged items: a noted item whose position is its partition's minimum
    const uint32_t n_seen = n_noted;
    const bool check_all = n_seen > kNoted;
    if (n_seen && !check_all) {
        for (uint32_t e = threadIdx.x; e < n_seen; e += kThreads) {
            const unsigned long long x = noted[e];
            const uint32_t part = (uint32_t)(x >> 32);
            if ((uint32_t)((uint64_t)table[part] & ((1ULL << kKeyPos) - 1)) == (uint32_t)x) atomicOr(&to_check[part >> 5], 1u << (part & 31u));
        }
        __syncthreads();
    }
    const uint64_t row0 = (uint64_t)g * sp.P + (uint64_t)bin * R;
    fp_t *__restrict__ fpo = reinterpret_cast<fp_t *>(fp_out) + row0;
    uint32_t act = 0, undecided = 0;
    unsigned long long card = 0;
    const bool pass_a = bloom && !(bs.tune & 1u);
#pragma unroll
    for (uint32_t j = 0; j < kWin; ++j) {
        const uint32_t i = threadIdx.x + kThreads * j;
        if (i >= R) continue;
        const key_t key = table[i];
        const bool has = key != kNoKey;
        const uint32_t fp = has ? (uint32_t)(key >> kKeyPos) : sp.empty;
        fpo[i] = (fp_t)fp;
        act += has ? 1u : 0u;
        card += has ? (unsigned long long)(1u << (31u - (fp >> sp.f))) : 0ull;   // Miekki.cpp:293: sum of 2^-exp, in units of 2^-31
        if (has && pass_a && (check_all || ((to_check[i >> 5] >> (i & 31u)) & 1u))) undecided |= 1u << j;
    }
    // The Bloom insert's first half (Miekki.cpp:121-131) for those, one at a time: the canonical k-mer from the packed codes,
    // first-level summary, the cell itself, and a first-writer key for the cell when it is still empty -- (genome in batch,
    // partition) and, below them, the BIT this k-mer would set: 1 << (hash % 8) -- with an atomic minimum.  The second half
    // is a sweep over the cells that took a key (bloom_sweep_kernel, sketch.hip): the k-mer is not needed again.  The five
    // positions of a k-mer are (canon + t_i) >> b with t_i < 1024 (universal_hash, utils.cpp:197-199) and b >= 32: when the
    // low word cannot carry they are ONE position -- all but one k-mer in four million; otherwise at most two cells, each
    // taking the bit of the lowest hash index that names it, which is what the reference's loop over the indices leaves.
    // (Posting without the two looks while the filter is young was tried: at -h 20 two thirds of the cells are taken after the
    // first batch, and an atomic on a taken cell costs more than the read that avoids it -- 2.22 against 2.11 ms per batch.)
    const uint64_t *__restrict__ gcodes = reinterpret_cast<const uint64_t *>(codes + code_off[g]);
    const uint64_t *__restrict__ gexcept = reinterpret_cast<const uint64_t *>(except + code_off[g] / 2);
    const bool has_x = dirty[g] != 0;                                 // workgroup-uniform
    auto post = [&](uint64_t hsh, uint32_t p) {
        const uint64_t cell = hsh >> 3;
        if (cell >= bloom_dev_bytes) return;
        const uint32_t grp = (uint32_t)(cell >> 3), sidx = grp >> 5;                              // 8, 256 cells
        if ((full[sidx] >> (grp & 31u)) & 1u) return;
        if (bloom[cell] != 0) return;
        atomicMin(&order[cell], (((g << sp.h) | p) << 3) | (uint32_t)(hsh & 7u));
        touched[cell >> kBloomRegionLog2] = 1;                       // the sweep looks at this region of cells
    };
    while (undecided) {
        const uint32_t j = (uint32_t)__builtin_ctz(undecided), i = threadIdx.x + kThreads * j;
        undecided &= undecided - 1u;
        const uint64_t pos = (uint64_t)table[i] & ((1ULL << kKeyPos) - 1);
        const uint64_t cn = canon_from_packed(gcodes, gexcept, has_x, pos, sp.k);
        const uint32_t p = bin * R + i;
        if ((uint32_t)cn <= 0xFFFFFC00u) {
            post(cn >> sp.bloom_log2, p);
        } else {
            const uint64_t anc = revhash64(cn);
            uint64_t seen[2] = {~0ull, ~0ull};
            for (uint32_t hi = 0; hi < kNumHash; ++hi) {
                const uint64_t hsh = bloom_pos(cn, anc, hi, sp.bloom_log2);
                if ((hsh >> 3) == seen[0] || (hsh >> 3) == seen[1]) continue;
                seen[seen[0] == ~0ull ? 0 : 1] = hsh >> 3;
                post(hsh, p);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { act += __shfl_xor(act, o); card += __shfl_xor(card, o); }
    if (lane == 0 && act) { atomicAdd(&s_act, act); atomicAdd(&s_card, card); }
    __syncthreads();
    if (threadIdx.x == 0 && s_act) { atomicAdd(&active[g], s_act); atomicAdd(&cardsum[g], s_card); }
}

// ---------------------------------------------------------------- host side
// shape of the build for a batch, and side b's slot memory; *fits = false when the batch does not suit the
// bins (the caller then takes the atomic kernel, from characters)
static int build_setup(mk_ctx *c, int b, const uint64_t *h_off, uint32_t n, BuildShape &bs, bool *fits, bool *key32, bool for_queries)
{
    *fits = false;
    uint64_t max_nk = 0, max_len = 0;
    for (uint32_t g = 0; g < n; ++g) {
        const uint64_t len = h_off[g + 1] - h_off[g];
        max_len = std::max(max_len, len);
        if (len > c->p.k) max_nk = std::max(max_nk, len - c->p.k);
    }
    // timing experiments (1: no Bloom pass A, 2: no item reads -- both build a WRONG index) exist only in a library compiled
    // with -DMK_TUNE_BUILD (tools/); the shipped one has no such switch
#ifdef MK_TUNE_BUILD
    static const uint32_t tune = [] { const char *e = getenv("MIEKKI_TUNE_BUILD"); return e ? (uint32_t)atoi(e) : 0u; }();
#else
    constexpr uint32_t tune = 0;
#endif
    bs.tune = tune;
    // the scatter kernel flags the k-mers that may still have work to do in the Bloom filter; it asks the filter's coarse
    // summary (one bit per 2048 cells, bloom_summary_bytes: 4 KiB at -b 33), which rides in its LDS when it is at most 8 KiB
    // -- a larger one is not consulted and every item counts as flagged
    bs.tables_only = for_queries ? 1u : 0u;
    bs.bloom_on = c->d_bloom && !(tune & 1u) && !for_queries ? 1u : 0u;   // (a query's k-mers are gated, not inserted)
    bs.sum_words = 0;
    if (bs.bloom_on && bloom_summary_bytes(c) <= 8192) bs.sum_words = (uint32_t)((bloom_summary_bytes(c) + 7) / 8);
    // partitions per bin (log2): 2^12, fewer only when the sketch has fewer.  (Smaller bins at small h -- 256 bins per genome
    // whatever h -- were measured at -h 17: 2^9-partition bins 47.0k sketches/s, 2^10 52.8k, 2^11 55.1k, 2^12 54.2k; and at
    // -h 20 2^10 32.4k, 2^11 40.8k against 53.5k: a reduce workgroup's LDS table wants many entries per lane.)
    bs.low_bits = std::min<uint32_t>(kBin, c->p.h);
    bs.nbins = c->P >> bs.low_bits;
    if (bs.nbins > kBins || max_len >= (1ULL << 35) || max_nk == 0) return MK_OK;
    bs.nwg = (uint32_t)((max_nk + kSeg - 1) / kSeg);
    // a run (one workgroup's items of one bin) holds ~Poisson(kSeg / nbins * 15/16) items: the lanes of a reduce wave that share
    // it, 16 bytes each, cover mean + 4 sigma, so that a second, dependent load per run is the exception (at h = 20: mean 15,
    // eight lanes x four items; with a reach of 16 items nearly every wave-load had a run with a tail and paid its latency)
    const double mean = (double)kSeg / bs.nbins * 15.0 / 16.0, reach = mean + 4.0 * std::sqrt(mean);
    const uint32_t ipl = 4;                                          // items per lane (16 bytes of main words, either width)
    bs.lpr = 1;
    while (bs.lpr < 64 && bs.lpr * ipl < reach) bs.lpr <<= 1;
    *key32 = c->W == 1 && max_len < (1ULL << 24);              // (a stored fingerprint is below the all-ones byte: no key equals "none")
    const uint64_t isz = c->W == 1 ? 4 : 5;                          // (W == 2: the main words, then the low position bytes)
    // the dense item array (kSeg item places per scatter workgroup) and, behind it, one meta word per (workgroup, bin), twice
    const uint64_t item_bytes = ((uint64_t)n * bs.nwg * kSeg * isz + 255) / 256 * 256;
    const uint64_t need = item_bytes + 2 * (uint64_t)n * (bs.nwg + 3) * bs.nbins * 4;   // (the words as the scatter kernel leaves them, and bin-major)
    if (need > (12ull << 30)) return MK_OK;                          // scratch budget
    mk_ctx::BuildSide &sd = c->side[b];
    if (need > sd.slots_bytes) {
        if (sd.d_slots) (void)hipFree(sd.d_slots);
        sd.d_slots = nullptr; sd.slots_bytes = 0;
        // (+ 1 KiB: a reduce lane loads 16 bytes up to lpr * 16 bytes past its run's start without a bounds test; behind the
        // last scatter workgroup's items that reaches into the meta words, and behind a very small meta array -- one short
        // query -- it would leave the allocation)
        MK_HIP(hipMalloc(&sd.d_slots, need + 1024 + 64));
        sd.slots_bytes = need;
    }
    *fits = true;
    return MK_OK;
}


// one side's own arrays (everything a front stage writes)
int ensure_build_side(mk_ctx *c, int b)
{
    mk_ctx::BuildSide &sd = c->side[b];
    if (!sd.d_counters) {
        MK_HIP(hipMalloc((void **)&sd.d_counters, sizeof *sd.d_counters));
        // (done before this returns: queued on c->stream it could run -- behind the Bloom arrays' gigabytes of memset -- AFTER a
        // front stage on the front stream had set the batch's exception flags, and wipe them)
        MK_HIP(hipMemsetAsync(sd.d_counters, 0, sizeof *sd.d_counters, c->front_stream));
        MK_HIP(hipStreamSynchronize(c->front_stream));
    }
    if (!sd.h_back) MK_HIP(hipHostMalloc((void **)&sd.h_back, sizeof *sd.h_back, hipHostMallocDefault));
    if (!sd.d_seq_off) MK_HIP(hipMalloc((void **)&sd.d_seq_off, (kBuildBatch + 1) * 8));
    if (!sd.d_seed_valid) MK_HIP(hipMalloc((void **)&sd.d_seed_valid, kBuildBatch * 4));
    if (!sd.d_ovf) MK_HIP(hipMalloc((void **)&sd.d_ovf, (uint64_t)(1u << 20) * 16));   // sketch.hip's overflow list (long-query path), 2^20 entries
    if (!sd.ev_front) MK_HIP(hipEventCreateWithFlags(&sd.ev_front, hipEventDisableTiming));
    if (!sd.ev_back) MK_HIP(hipEventCreateWithFlags(&sd.ev_back, hipEventDisableTiming));
    return MK_OK;
}

void use_build_side(mk_ctx *c, int b)
{
    mk_ctx::BuildSide &sd = c->side[b];
    c->d_counters = sd.d_counters; c->h_back = sd.h_back; c->d_seq_off = sd.d_seq_off; c->d_seed_valid = sd.d_seed_valid;
    c->d_ovf = sd.d_ovf;
    c->d_ovf_count = &sd.d_counters->ovf;
    c->d_dirty = sd.d_counters->dirty;
    c->d_active = sd.d_counters->act;
    c->d_cardsum = sd.d_counters->card;
}

int launch_pack(mk_ctx *c, int b, const char *d_seq, const uint64_t *h_off, uint32_t n, uint8_t *d_codes, uint8_t *d_except,
                const uint64_t *d_code_off, hipStream_t st)
{
    uint64_t max_len = 0;
    for (uint32_t g = 0; g < n; ++g) max_len = std::max(max_len, h_off[g + 1] - h_off[g]);
    if (!n || !max_len) return MK_OK;
    const uint64_t words = (max_len + 31) / 32;
    hipLaunchKernelGGL(pack_kernel, dim3((uint32_t)((words + 255) / 256), n), dim3(256), 0, st ? st : c->front_stream, d_seq, c->side[b].d_seq_off,
                       n, d_codes, d_except, d_code_off, c->side[b].d_counters->dirty);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_seed_fix(mk_ctx *c, int b, const char *d_seq, const char *d_heads, uint32_t n, uint8_t *d_codes, uint8_t *d_except,
                    const uint64_t *d_code_off, hipStream_t st)
{
    if (!n) return MK_OK;
    hipLaunchKernelGGL(seed_fix_kernel, dim3((n + 63) / 64), dim3(64), 0, st ? st : c->front_stream, d_seq, c->side[b].d_seq_off, d_heads, n,
                       c->p.k, d_codes, d_except, d_code_off, c->side[b].d_seed_valid);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_synth_packed(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t len, uint8_t *d_codes, const uint64_t *d_code_off,
                        uint32_t strains, uint32_t rate_ppm)
{
    if (!n || !len) return MK_OK;
    const uint64_t words = (len + 31) / 32;
    hipLaunchKernelGGL(synth_packed_kernel, dim3((uint32_t)((words + 255) / 256), n), dim3(256), 0, c->front_stream, first_id, n, len,
                       d_codes, d_code_off, strains, rate_ppm);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_unpack(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off, const char *d_heads,
                  const uint64_t *h_off, uint32_t n, char *d_seq)
{
    uint64_t max_len = 0;
    for (uint32_t g = 0; g < n; ++g) max_len = std::max(max_len, h_off[g + 1] - h_off[g]);
    if (!n || !max_len) return MK_OK;
    hipLaunchKernelGGL(unpack_kernel, dim3((uint32_t)((max_len + 255) / 256), n), dim3(256), 0, c->stream, d_codes, d_except,
                       d_code_off, c->side[b].d_counters->dirty, d_heads, c->side[b].d_seq_off, d_seq);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

static void shape_store(mk_ctx::BuildSide &sd, const BuildShape &bs)
{
    sd.shape[0] = bs.nbins; sd.shape[1] = bs.low_bits; sd.shape[2] = bs.lpr; sd.shape[3] = bs.nwg; sd.shape[4] = bs.tune;
    sd.shape[5] = bs.sum_words; sd.shape[6] = bs.bloom_on; sd.shape[7] = bs.tables_only;
}
static BuildShape shape_load(const mk_ctx::BuildSide &sd)
{
    BuildShape bs;
    bs.nbins = sd.shape[0]; bs.low_bits = sd.shape[1]; bs.lpr = sd.shape[2]; bs.nwg = sd.shape[3]; bs.tune = sd.shape[4];
    bs.sum_words = sd.shape[5]; bs.bloom_on = sd.shape[6]; bs.tables_only = sd.shape[7];
    return bs;
}
// the meta words lie behind the side's item array (build_setup)
static uint32_t *meta_of(const mk_ctx *c, const mk_ctx::BuildSide &sd, const BuildShape &bs, uint32_t n)
{
    const uint64_t isz = c->W == 1 ? 4 : 5;
    const uint64_t item_bytes = ((uint64_t)n * bs.nwg * kSeg * isz + 255) / 256 * 256;
    return reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(sd.d_slots) + item_bytes);
}
static uint32_t *metaT_of(const mk_ctx *c, const mk_ctx::BuildSide &sd, const BuildShape &bs, uint32_t n)
{
    return meta_of(c, sd, bs, n) + (uint64_t)n * bs.nwg * bs.nbins;
}
// W == 2: the items' low position bytes lie behind the main words
static uint8_t *low_of(const mk_ctx::BuildSide &sd, const BuildShape &bs, uint32_t n)
{
    return static_cast<uint8_t *>(sd.d_slots) + (uint64_t)n * bs.nwg * kSeg * 4;
}

// Front stage: the scatter kernel of a batch, on the front stream, into side b's slots.
int launch_build_front(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off,
                       const uint64_t *h_off, uint32_t n, bool *used, hipStream_t st, bool for_queries)
{
    *used = false;
    mk_ctx::BuildSide &sd = c->side[b];
    sd.fits = false;
    if (!n) return MK_OK;
    BuildShape bs;
    MK_TRY(build_setup(c, b, h_off, n, bs, &sd.fits, &sd.key32, for_queries));
    if (!sd.fits) return MK_OK;
    shape_store(sd, bs);
    const SketchParams sp = make_sp(c);
    MK_TRY(ensure_bloom_summary_arrays(c));                       // (all zero until the first summary: everything flagged)
    const size_t stage_bytes = (kSeg + 4) * 4 + (c->W == 2 ? kSeg + 16 : 0);
    // stage (the positions' words and the Bloom summary in it), counters, dump slot, wave sums
    size_t lds = stage_bytes + (((size_t)2 * bs.nbins + 7) & ~(size_t)7) * 4 + 32;
#define MK_SCATTER(Wv, KB)                                                                                                      \
    hipLaunchKernelGGL((build_scatter_kernel<Wv, KB>), dim3(bs.nwg, n), dim3(256), lds, st ? st : c->front_stream, d_codes,     \
                       d_except,                                                                                                \
                       d_code_off, sd.d_counters->dirty, sd.d_seq_off, reinterpret_cast<uint32_t *>(sd.d_slots), low_of(sd, bs, n),  \
                       meta_of(c, sd, bs, n), reinterpret_cast<const uint32_t *>(c->d_bloom_full2), sp, bs)
    const bool kbig = c->p.k >= 17;
    if (c->W == 1) { if (kbig) MK_SCATTER(1, true); else MK_SCATTER(1, false); }
    else           { if (kbig) MK_SCATTER(2, true); else MK_SCATTER(2, false); }
#undef MK_SCATTER
    if (bs.nbins % 4 == 0)
        hipLaunchKernelGGL(meta_transpose_kernel<true>, dim3((bs.nwg + 63) / 64, (bs.nbins + 63) / 64, n), dim3(256), 0, st ? st : c->front_stream,
                           meta_of(c, sd, bs, n), metaT_of(c, sd, bs, n), bs.nwg, bs.nbins, (bs.nwg + 3u) & ~3u);
    else
        hipLaunchKernelGGL(meta_transpose_kernel<false>, dim3((bs.nwg + 63) / 64, (bs.nbins + 63) / 64, n), dim3(256), 0, st ? st : c->front_stream,
                           meta_of(c, sd, bs, n), metaT_of(c, sd, bs, n), bs.nwg, bs.nbins, (bs.nwg + 3u) & ~3u);
    MK_HIP(hipGetLastError());
    *used = true;
    return MK_OK;
}

int launch_query_tables(mk_ctx *c, const char *d_seq, const uint64_t *d_off, const uint64_t *h_off, uint32_t n, uint64_t *d_tables,
                        bool *used)
{
    *used = false;
    if (!n || n > c->build_batch) return MK_OK;
    const int b = 0;
    MK_TRY(ensure_build_side(c, b));
    mk_ctx::BuildSide &sd = c->side[b];
    if (!c->d_pk_off[b]) MK_HIP(hipMalloc((void **)&c->d_pk_off[b], (kBuildBatch + 1) * 8));
    uint64_t pk_off[kBuildBatch + 1];
    MK_TRY(ensure_packed(c, b, packed_offsets(h_off, n, pk_off)));
    uint8_t *codes = c->d_pk[b], *except = c->d_pk[b] + c->pk_cap[b];
    hipStream_t st = c->stream;
    MK_HIP(hipMemcpyAsync(c->d_pk_off[b], pk_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, st));   // (pageable: the call returns when it is on its way)
    MK_HIP(hipMemcpyAsync(sd.d_seq_off, d_off, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, st));
    MK_HIP(hipMemsetAsync(sd.d_counters, 0, sizeof *sd.d_counters, st));
    MK_TRY(launch_pack(c, b, d_seq, h_off, n, codes, except, c->d_pk_off[b], st));
    MK_TRY(launch_seed_fix(c, b, d_seq, nullptr, n, codes, except, c->d_pk_off[b], st));
    bool fits = false;
    MK_TRY(launch_build_front(c, b, codes, except, c->d_pk_off[b], h_off, n, &fits, st, true));
    if (!fits) return MK_OK;
    const BuildShape bs = shape_load(sd);
    const SketchParams sp = make_sp(c);
#define MK_QREDUCE(Wv, K32)                                                                                                     \
    hipLaunchKernelGGL((build_reduce_kernel<Wv, K32>), dim3(bs.nbins, n), dim3(512), 0, st,                                     \
                       reinterpret_cast<const uint32_t *>(sd.d_slots), low_of(sd, bs, n), metaT_of(c, sd, bs, n), codes, except,   \
                       c->d_pk_off[b], sd.d_counters->dirty, (const uint8_t *)nullptr, (uint64_t)0, (uint32_t *)nullptr,         \
                       (const uint32_t *)nullptr, (uint8_t *)nullptr, d_tables, (uint8_t *)nullptr, (uint32_t *)nullptr,        \
                       (unsigned long long *)nullptr, sp, bs)
    if (c->W == 1) { if (sd.key32) MK_QREDUCE(1, true); else MK_QREDUCE(1, false); }
    else MK_QREDUCE(2, false);
#undef MK_QREDUCE
    MK_HIP(hipGetLastError());
    *used = true;
    return MK_OK;
}

// Back stage: reduce + fingerprints + sizes + Bloom pass A in one kernel, the matrix rows, Bloom pass B over the
// blocks that need it, the summary -- on c->stream, from side b's slots.
int launch_build_back(mk_ctx *c, int b, const uint8_t *d_codes, const uint8_t *d_except, const uint64_t *d_code_off, uint32_t n,
                      uint32_t g0)
{
    mk_ctx::BuildSide &sd = c->side[b];
    if (!n || !sd.fits) return MK_OK;
    const BuildShape bs = shape_load(sd);
    const uint64_t fp_bytes = (uint64_t)c->build_batch * c->P * c->W;
    if (!c->d_fpT) {
        MK_HIP(hipMalloc((void **)&c->d_fpT, fp_bytes + 64));
    }
    const SketchParams sp = make_sp(c);
    // (more LDS asked for, so that three workgroups share a CU instead of four and the front stage's scatter workgroups find wave
    // slots beside them: measured and left -- 55.1k against 57.7k sketches/s, the two kernels are better off taking turns)
#define MK_REDUCE(Wv, K32)                                                                                                      \
    hipLaunchKernelGGL((build_reduce_kernel<Wv, K32>), dim3(bs.nbins, n), dim3(512), 0, c->stream,                              \
                       reinterpret_cast<const uint32_t *>(sd.d_slots), low_of(sd, bs, n), metaT_of(c, sd, bs, n), d_codes,         \
                       d_except, d_code_off, sd.d_counters->dirty, c->d_bloom, c->bloom_dev_bytes,                             \
                       c->d_bloom_order, c->d_bloom_full, c->d_fpT, c->d_tables, c->d_bloom_touched,                           \
                       sd.d_counters->act, (unsigned long long *)sd.d_counters->card, sp, bs)
    if (c->W == 1) { if (sd.key32) MK_REDUCE(1, true); else MK_REDUCE(1, false); }
    else MK_REDUCE(2, false);
#undef MK_REDUCE
    MK_HIP(hipGetLastError());
    return launch_build_tail(c, n, g0);
}

}  // namespace mk
