// K1-K4: k-mer hashing, per-partition minimizer selection, matrix/Bloom update,
// and the sparse query sketch.  gfx950 only.
//
// Replaces Miekki::minhash_sketch_partition (Miekki.cpp:150-197), the body of
// insert_sequences (Miekki.cpp:287-311), insert_bloom/check_bloom
// (Miekki.cpp:121-146) and minhash_sketch_partition_solid_kmers (Miekki.cpp:214-224).
//
// Selection rule restated: in every partition the winner is
// the k-mer with the smallest fingerprint, earliest position first among equals,
// and a fingerprint equal to the "empty" value can never be stored.  That is the
// minimum of the key (fingerprint, position), which is what every path computes:
// genomes and long queries by binning (fingerprint, position, partition) items and
// taking per-partition minima in LDS (or, for shapes the bins do not fit, by 64-bit
// atomic minimum into a table in HBM), short queries by an in-LDS sort of
// (partition, fingerprint, position) keys.
#include <cmath>
#include <cstdlib>

#include "codes.hpp"
#include <cstring>

#include "mk_internal.hpp"

namespace mk {

constexpr uint32_t kSegKmers = 4096;   // k-mers per workgroup of the genome sketch
constexpr uint32_t kPerThread = 16;    // consecutive k-mers per thread (rolling update)


// canonical k-mer starting at sequence position i, read straight from the characters
__device__ __forceinline__ uint64_t canon_at_bytes(const char *__restrict__ seq, uint64_t i, uint32_t k,
                                                   bool seed_valid)
{
    uint64_t S = 0, RC = 0;
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t cd = pos_codes((uint8_t)seq[i + j], i + j, k, seed_valid);
        S = (S << 2) | (cd & 3u);
        RC |= (uint64_t)(cd >> 2) << (2 * j);
    }
    return S < RC ? S : RC;
}

// Same value for k-mers clear of the k-1 seed, fetched as five aligned 64-bit words
// (k <= 31 characters span at most 38 bytes) instead of k byte loads: the callers
// read winners at random positions, so the load count is what they pay for.  The
// sequence buffers carry 64 bytes of slack, and start 256-byte aligned, so the
// aligned window never leaves the allocation.
__device__ __forceinline__ uint64_t canon_at(const char *__restrict__ seq, uint64_t i, uint32_t k,
                                             bool seed_valid)
{
    if (i + 1 < k) return canon_at_bytes(seq, i, k, seed_valid);      // touches the seed (Miekki.cpp:158)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(seq) + i;
    const uint64_t *__restrict__ w = reinterpret_cast<const uint64_t *>(addr & ~(uintptr_t)7);
    const uint32_t sh = (uint32_t)(addr & 7u) * 8u;
    const uint64_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    uint64_t t[4];
    t[0] = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    t[1] = sh ? (w1 >> sh) | (w2 << (64 - sh)) : w1;
    t[2] = sh ? (w2 >> sh) | (w3 << (64 - sh)) : w2;
    t[3] = sh ? (w3 >> sh) | (w4 << (64 - sh)) : w3;
    uint64_t S = 0, RC = 0;
#pragma unroll
    for (uint32_t j = 0; j < 31; ++j) {
        if (j < k) {
            const uint8_t c = (uint8_t)(t[j >> 3] >> ((j & 7u) * 8u));
            S = (S << 2) | fwd_code(c);
            RC |= (uint64_t)rc_code(c) << (2 * j);
        }
    }
    return S < RC ? S : RC;
}

// ---------------------------------------------------------------- seed validity
__global__ void seed_valid_kernel(const char *__restrict__ seq, const uint64_t *__restrict__ off,
                                  uint32_t n, uint32_t k, uint32_t *__restrict__ valid)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const char *s = seq + off[g];
    const uint64_t len = off[g + 1] - off[g];
    uint32_t ok = 1;
    for (uint32_t j = 0; j + 1 < k && j < len; ++j) ok &= seed_code((uint8_t)s[j]) != 4u;
    valid[g] = ok;
}

// ---------------------------------------------------------------- K1 genome sketch
// grid = (segments of 4096 k-mers, genomes).  Each thread rolls 16 consecutive
// k-mers from 2-bit codes staged in LDS and posts (fingerprint, position) keys to
// the genome's table with a 64-bit atomic minimum, skipping keys that a (possibly
// stale) plain read already shows to be losers.
__global__ __launch_bounds__(256) void genome_sketch_kernel(const char *__restrict__ seq,
                                                            const uint64_t *__restrict__ off,
                                                            const uint32_t *__restrict__ valid,
                                                            uint64_t *__restrict__ tables,
                                                            SketchParams sp)
{
    __shared__ uint8_t codes[kSegKmers + 64];
    const uint32_t g = blockIdx.y;
    const uint64_t len = off[g + 1] - off[g];
    const uint64_t nk = len > sp.k ? len - sp.k : 0;         // Miekki.cpp:162: last k-mer skipped
    const uint64_t seg0 = (uint64_t)blockIdx.x * kSegKmers;
    if (seg0 >= nk) return;
    const uint32_t cnt = (uint32_t)min((uint64_t)kSegKmers, nk - seg0);
    const char *__restrict__ s = seq + off[g];
    const bool sv = valid[g] != 0;
    const uint32_t nchar = cnt + sp.k - 1;
    for (uint32_t j = threadIdx.x; j < nchar; j += 256)
        codes[j] = (uint8_t)pos_codes((uint8_t)s[seg0 + j], seg0 + j, sp.k, sv);
    __syncthreads();
    const uint32_t i0 = threadIdx.x * kPerThread;
    if (i0 >= cnt) return;
    const uint32_t i1 = min(i0 + kPerThread, cnt);
    uint64_t *__restrict__ table = tables + (uint64_t)g * sp.P;
    uint64_t S = 0, RC = 0;
    for (uint32_t j = 0; j + 1 < sp.k; ++j) {                // first k-1 digits of k-mer i0
        const uint32_t cd = codes[i0 + j];
        S = (S << 2) | (cd & 3u);
        RC |= (uint64_t)(cd >> 2) << (2 * (j + 1));
    }
    const uint32_t topshift = 2 * sp.k - 2;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t cd = codes[i + sp.k - 1];
        S = ((S << 2) | (cd & 3u)) & sp.kmask;               // update_kmer, Miekki.cpp:51-55
        RC = (RC >> 2) | ((uint64_t)(cd >> 2) << topshift);  // update_kmer_RC, Miekki.cpp:59-62
        const uint64_t anc = revhash64(S < RC ? S : RC);
        uint32_t bucket, fp;
        bucket_fp(anc, sp.h, sp.f, sp.empty, bucket, fp);
        if (fp != sp.empty) {                                // `fp < 255` can never hold for 255
            const uint64_t key = ((uint64_t)fp << kPosBits) | (seg0 + i);
            if (key < table[bucket]) atomicMin((unsigned long long *)&table[bucket], (unsigned long long)key);
        }
    }
}

int launch_genome_sketch(mk_ctx *c, const char *d_seq, const uint64_t *d_off, const uint64_t *h_off,
                         const uint32_t *d_valid, uint32_t n, uint64_t *d_tables)
{
    if (!n) return MK_OK;
    uint64_t max_nk = 0;
    for (uint32_t g = 0; g < n; ++g) {
        const uint64_t len = h_off[g + 1] - h_off[g];
        if (len > c->p.k) max_nk = std::max(max_nk, len - c->p.k);
        if (len >= (1ULL << kPosBits)) { set_error("sequence too long"); return MK_ERR_ARG; }
    }
    MK_HIP(hipMemsetAsync(d_tables, 0xFF, (size_t)n * c->P * sizeof(uint64_t), c->stream));
    if (!max_nk) return MK_OK;
    const uint64_t segs = (max_nk + kSegKmers - 1) / kSegKmers;
    if (segs > 0x7fffffffULL) { set_error("sequence too long"); return MK_ERR_ARG; }
    hipLaunchKernelGGL(genome_sketch_kernel, dim3((uint32_t)segs, n), dim3(256), 0, c->stream, d_seq, d_off,
                       d_valid, d_tables, make_sp(c));
    MK_HIP(hipGetLastError());
    return MK_OK;
}

constexpr uint32_t kOvfCap = 1u << 20;            // entries of a build side's scratch list (mk_ctx::BuildSide::d_ovf)
constexpr uint32_t kOvfScan = 1u << 15;           // overflow items every reduce workgroup of the long-query sketch folds in

// fp_out[n][P] (genome-major, W bytes) -> M[p][g0 .. g0+n): 16-byte loads of one genome's run of
// partitions, transposed through LDS, 16-byte row pieces out (K2's layout work; its sums are done).
template <int W>
__global__ __launch_bounds__(1024) void fp_transpose_kernel(const uint8_t *__restrict__ fp_in, uint32_t n, uint32_t g0,
                                                           MatRef M, uint64_t ld,
                                                           const uint32_t *__restrict__ ovf_count, SketchParams sp)
{
    if (*ovf_count > kOvfScan) return;
    using fp_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;
    constexpr uint32_t kRows = 1024 / W;                           // partitions per tile = one wave-load of 16 B per lane
    constexpr uint32_t kPitch = kBuildBatch * W + 16;              // bytes per LDS row, 16-byte aligned
    __shared__ __attribute__((aligned(16))) uint8_t tile[kRows * kPitch];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t p0 = blockIdx.x * kRows;
    const uint32_t rows = min(kRows, sp.P - p0);
    for (uint32_t g = wave; g < n; g += 16) {                       // sixteen waves: the tile is little work, what it needs is loads in flight
        const uint32_t e0 = lane * (16 / W);                       // first partition (within the tile) of this lane
        fp_t v[16 / W];
        if (e0 + 16 / W <= rows) {
            *reinterpret_cast<uint4 *>(v) = *reinterpret_cast<const uint4 *>(fp_in + ((uint64_t)g * sp.P + p0 + e0) * W);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 16 / W; ++e)
                v[e] = e0 + e < rows ? reinterpret_cast<const fp_t *>(fp_in)[(uint64_t)g * sp.P + p0 + e0 + e] : (fp_t)0;
        }
#pragma unroll
        for (uint32_t e = 0; e < 16 / W; ++e)
            *reinterpret_cast<fp_t *>(tile + (e0 + e) * kPitch + g * W) = v[e];
    }
    __syncthreads();
    const uint64_t col0 = (uint64_t)g0 * W;
    if (((col0 | ((uint64_t)n * W)) & 15u) == 0) {
        const uint32_t vec_per_row = n * W / 16;
        for (uint32_t idx = threadIdx.x; idx < rows * vec_per_row; idx += 1024) {
            const uint32_t r = idx / vec_per_row, v = idx - r * vec_per_row;
            *reinterpret_cast<uint4 *>(mat_row(M, p0 + r, ld) + col0 + v * 16) =
                *reinterpret_cast<const uint4 *>(tile + r * kPitch + v * 16);
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < rows * n; idx += 1024) {
            const uint32_t r = idx / n, g = idx - r * n;
            *reinterpret_cast<fp_t *>(mat_row(M, p0 + r, ld) + col0 + (uint64_t)g * W) =
                *reinterpret_cast<const fp_t *>(tile + r * kPitch + g * W);
        }
    }
}

// The same without LDS, for the usual batch (first genome and count multiples of four): a thread loads 16 bytes of
// 4 / W genomes' fingerprints (16 / W partitions each), transposes them in registers with byte permutes and stores one
// 32-bit word -- 4 / W genomes side by side -- per partition; the lanes of a wave are laid out so that a row piece
// leaves as 64 (128) contiguous bytes.  (The LDS form wrote single bytes at a pitch that put all 64 lanes on one
// bank: 0.117 ms per batch for 128 MB of traffic; round 3.)
template <int W>
__global__ __launch_bounds__(256) void fp_transpose_reg_kernel(const uint8_t *__restrict__ fp_in, uint32_t n, uint32_t g0, MatRef M,
                                                               uint64_t ld, const uint32_t *__restrict__ ovf_count, SketchParams sp)
{
    if (*ovf_count > kOvfScan) return;
    constexpr uint32_t GPT = 4 / W, PPT = 16 / W;                  // genomes, partitions per thread
    constexpr uint32_t NG = kBuildBatch / GPT, NPB = 64 / NG;      // genome groups, partition blocks per wave
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t gq = lane % NG, pb = lane / NG;
    const uint32_t p0 = (wave * NPB + pb) * PPT;                   // first partition of this thread
    const uint32_t gfirst = gq * GPT;
    if (p0 >= sp.P || gfirst >= n) return;
    uint4 in[GPT];
#pragma unroll
    for (uint32_t e = 0; e < GPT; ++e)                             // (P is a multiple of PPT whenever P >= 16; smaller P: the LDS form)
        in[e] = *reinterpret_cast<const uint4 *>(fp_in + ((uint64_t)(gfirst + e) * sp.P + p0) * W);
    uint32_t out[PPT];
    if constexpr (W == 1) {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {                          // one 4 x 4 byte block per dword position
            const uint32_t a = (&in[0].x)[q], b = (&in[1].x)[q], c = (&in[2].x)[q], d = (&in[3].x)[q];
            const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u), t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(d, c, 0x05010400u), u1 = __builtin_amdgcn_perm(d, c, 0x07030602u);
            out[4 * q + 0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
            out[4 * q + 1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            out[4 * q + 2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            out[4 * q + 3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        }
    } else {
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {                          // two 16-bit values of each of the two genomes per dword
            const uint32_t a = (&in[0].x)[q], b = (&in[1].x)[q];
            out[2 * q + 0] = __builtin_amdgcn_perm(b, a, 0x05040100u);
            out[2 * q + 1] = __builtin_amdgcn_perm(b, a, 0x07060302u);
        }
    }
    const uint64_t col = ((uint64_t)g0 + gfirst) * W;
#pragma unroll
    for (uint32_t j = 0; j < PPT; ++j)
        *reinterpret_cast<uint32_t *>(mat_row(M, p0 + j, ld) + col) = out[j];
}

__global__ void bloom_sweep_kernel(const uint32_t *__restrict__ order, uint8_t *__restrict__ bloom, uint8_t *__restrict__ touched,
                                   uint64_t bloom_dev_bytes);
static int launch_bloom_sweep(mk_ctx *c);

// What follows the fused reduce kernel of a batch (this file's or build.hip's): the batch's fingerprints
// (d_fpT, genome-major) into the matrix rows, Bloom pass B over the blocks that posted a key, the summary.
int launch_build_tail(mk_ctx *c, uint32_t n, uint32_t g0)
{
    const SketchParams sp = make_sp(c);
    const uint32_t rows = 1024 / c->W;
    if (g0 % 4 == 0 && n % 4 == 0 && c->P >= 16) {
        // register form: a wave covers 64 (W = 1) or 16 (W = 2) partitions of all the batch's genomes
        const uint32_t per_wave = c->W == 1 ? 64 : 16;
        const uint32_t waves = (c->P + per_wave - 1) / per_wave;
        if (c->W == 1)
            hipLaunchKernelGGL(fp_transpose_reg_kernel<1>, dim3((waves + 3) / 4), dim3(256), 0, c->stream, c->d_fpT, n, g0, mat_ref(c),
                               c->ld, c->d_ovf_count, sp);
        else
            hipLaunchKernelGGL(fp_transpose_reg_kernel<2>, dim3((waves + 3) / 4), dim3(256), 0, c->stream, c->d_fpT, n, g0, mat_ref(c),
                               c->ld, c->d_ovf_count, sp);
    } else if (c->W == 1)
        hipLaunchKernelGGL(fp_transpose_kernel<1>, dim3((c->P + rows - 1) / rows), dim3(1024), 0, c->stream, c->d_fpT, n, g0,
                           mat_ref(c), c->ld, c->d_ovf_count, sp);
    else
        hipLaunchKernelGGL(fp_transpose_kernel<2>, dim3((c->P + rows - 1) / rows), dim3(1024), 0, c->stream, c->d_fpT, n, g0,
                           mat_ref(c), c->ld, c->d_ovf_count, sp);
    if (c->d_bloom) {
        MK_TRY(launch_bloom_sweep(c));                               // the cells that took a key get their byte
        MK_TRY(launch_bloom_summary(c, true));
    }
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- K2 finalize
// tables[n][P] -> M[p][g0 .. g0+n) (transposed through LDS), plus per-genome
// active count and HyperLogLog-style cardinality sum (Miekki.cpp:290-301).  The
// sum of 2^-exp is kept as an exact integer multiple of 2^-31.
constexpr uint32_t kFinRows = 64;        // partitions per tile
constexpr uint32_t kFinTiles = 16;       // tiles per workgroup

template <int W>
__global__ __launch_bounds__(256) void finalize_kernel(const uint64_t *__restrict__ tables, uint32_t n,
                                                       uint32_t g0, MatRef M, uint64_t ld,
                                                       uint32_t *__restrict__ active,
                                                       unsigned long long *__restrict__ cardsum,
                                                       const uint32_t *__restrict__ ovf_count, SketchParams sp)
{
    if (ovf_count && *ovf_count > kOvfCap) return;      // the binned sketch lost items: the host redoes the batch
    using fp_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;
    __shared__ fp_t tile[kFinRows][kBuildBatch + 2];
    __shared__ uint32_t s_act[kBuildBatch];
    __shared__ unsigned long long s_card[kBuildBatch];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < kBuildBatch) { s_act[threadIdx.x] = 0; s_card[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t t = 0; t < kFinTiles; ++t) {
        const uint32_t p0 = (blockIdx.x * kFinTiles + t) * kFinRows;
        if (p0 >= sp.P) break;
        for (uint32_t g = wave; g < n; g += 4) {
            const uint32_t p = p0 + lane;
            uint32_t fp = sp.empty;
            if (p < sp.P) {
                const uint64_t key = tables[(uint64_t)g * sp.P + p];
                if (key != kEmptyKey) fp = (uint32_t)(key >> kPosBits);
            }
            tile[lane][g] = (fp_t)fp;
            const bool act = fp != sp.empty;
            unsigned long long term = act ? (1ull << (31u - (fp >> sp.f))) : 0ull;
            uint32_t a = act ? 1u : 0u;
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_down(a, o);
                term += __shfl_down(term, o);
            }
            if (lane == 0) { atomicAdd(&s_act[g], a); atomicAdd(&s_card[g], term); }
        }
        __syncthreads();
        const uint32_t rows = min(kFinRows, sp.P - p0);
        const uint64_t col0 = (uint64_t)g0 * W;
        if (((col0 | ((uint64_t)n * W)) & 15u) == 0) {          // 16-byte row pieces
            const uint32_t vec_per_row = n * W / 16;
            for (uint32_t idx = threadIdx.x; idx < rows * vec_per_row; idx += 256) {
                const uint32_t r = idx / vec_per_row, v = idx - r * vec_per_row;
                alignas(16) fp_t tmp[16 / W];
#pragma unroll
                for (uint32_t e = 0; e < 16 / W; ++e) tmp[e] = tile[r][v * (16 / W) + e];
                *reinterpret_cast<uint4 *>(mat_row(M, p0 + r, ld) + col0 + v * 16) =
                    *reinterpret_cast<const uint4 *>(tmp);
            }
        } else {
            for (uint32_t idx = threadIdx.x; idx < rows * n; idx += 256) {
                const uint32_t r = idx / n, g = idx - r * n;
                *reinterpret_cast<fp_t *>(mat_row(M, p0 + r, ld) + col0 + (uint64_t)g * W) = tile[r][g];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < n) {
        atomicAdd(&active[threadIdx.x], s_act[threadIdx.x]);
        atomicAdd(&cardsum[threadIdx.x], s_card[threadIdx.x]);
    }
}

int launch_finalize(mk_ctx *c, const uint64_t *d_tables, uint32_t n, uint32_t g0, const uint32_t *d_abort)
{
    if (!n) return MK_OK;
    MK_HIP(hipMemsetAsync(c->d_active, 0, kBuildBatch * sizeof(uint32_t), c->stream));
    MK_HIP(hipMemsetAsync(c->d_cardsum, 0, kBuildBatch * sizeof(uint64_t), c->stream));
    const uint32_t per_block = kFinRows * kFinTiles;
    const uint32_t blocks = (c->P + per_block - 1) / per_block;
    if (c->W == 1)
        hipLaunchKernelGGL(finalize_kernel<1>, dim3(blocks), dim3(256), 0, c->stream, d_tables, n, g0, mat_ref(c),
                           c->ld, c->d_active, (unsigned long long *)c->d_cardsum, d_abort, make_sp(c));
    else
        hipLaunchKernelGGL(finalize_kernel<2>, dim3(blocks), dim3(256), 0, c->stream, d_tables, n, g0, mat_ref(c),
                           c->ld, c->d_active, (unsigned long long *)c->d_cardsum, d_abort, make_sp(c));
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- K3 Bloom insert
// insert_bloom (Miekki.cpp:121-131) sets a zero cell to 1 << (hash % 8) of its FIRST inserter in (genome, partition,
// hash index) order.  First half (here for the character path; build.hip's reduce kernel has its own): every selected
// k-mer posts  (genome in batch << h | partition) << 3 | bit  for its cell with an atomic minimum -- at most 30 bits: a
// batch holds at most 2^30 / (8 x 2^h) genomes -- when the cell is still empty.  Second half: a SWEEP over the regions of
// cells that took a key (bloom_sweep_kernel): an empty cell with a key gets the key's bit.  No k-mer is looked at twice,
// nothing is sorted, and the result does not depend on the order the k-mers arrive in.  Round 3 had the winner itself
// come back for its cell (a second pass over the k-mers: one random read of an 8-byte key, a random byte written and the
// key reset, per k-mer -- 5.4 ms for the first batch of a collection at -h 20, 0.7 ms per batch at -h 17); the sweep
// streams 5 bytes per cell of the regions touched: 0.1 ms for all 64 MiB of cells, nothing once the filter has filled up.
// Keys are never reset: a cell with a key is set by the sweep of that very batch, and a key on a cell that is set is
// ignored (cells that are replaced wholesale -- an import -- take fresh keys with them, forget_bloom_summary).
__global__ __launch_bounds__(256) void bloom_post_kernel(const uint64_t *__restrict__ tables, const char *__restrict__ seq,
                                                         const uint64_t *__restrict__ off, const uint32_t *__restrict__ valid,
                                                         const uint8_t *__restrict__ bloom, uint64_t bloom_dev_bytes,
                                                         uint32_t *__restrict__ order, uint8_t *__restrict__ touched,
                                                         const uint32_t *__restrict__ ovf_count, const uint32_t *__restrict__ full,
                                                         uint32_t ovf_limit, SketchParams sp)
{
    if (ovf_count && *ovf_count > ovf_limit) return;    // see finalize_kernel
    const uint32_t g = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= sp.P) return;
    const uint64_t key = __builtin_nontemporal_load(tables + (uint64_t)g * sp.P + p);   // fingerprint << kPosBits | position
    if (key == kEmptyKey) return;
    const uint64_t canon = canon_at(seq + off[g], key & ((1ULL << kPosBits) - 1), sp.k, valid[g] != 0);
    auto post = [&](uint64_t hsh) {
        const uint64_t cell = hsh >> 3, grp = cell >> 3;
        if (cell >= bloom_dev_bytes) return;                // unreachable by construction
        // the summary first (one bit per 8 cells, L2-resident): a full group has no zero cell
        if (((full[grp >> 5] >> (grp & 31u)) & 1u) || bloom[cell] != 0) return;
        atomicMin(&order[cell], (((g << sp.h) | p) << 3) | (uint32_t)(hsh & 7u));
        touched[cell >> kBloomRegionLog2] = 1;
    };
    if (__builtin_expect((uint32_t)canon <= 0xFFFFFC00u, 1)) {
        post(canon >> sp.bloom_log2);                       // the five positions are ONE (mk_device.hpp: bloom_pos)
    } else {
        const uint64_t anc = revhash64(canon);
        uint64_t seen[2] = {~0ull, ~0ull};                  // at most two cells; each takes the bit of the lowest index that names it
        for (uint32_t i = 0; i < kNumHash; ++i) {
            const uint64_t hsh = bloom_pos(canon, anc, i, sp.bloom_log2);
            if ((hsh >> 3) == seen[0] || (hsh >> 3) == seen[1]) continue;
            seen[seen[0] == ~0ull ? 0 : 1] = hsh >> 3;
            post(hsh);
        }
    }
}

// One workgroup per region of 2^kBloomRegionLog2 cells; a region nobody posted to costs one byte read, one that took a key
// is ONE pass of its workgroup, sixteen cells per thread in four independent steps.  (Round 5 had regions of 2^16 cells: once
// the filter has filled up the regions that still take keys -- the top twentieth of the range, where canonical k-mers are
// rare -- were 64 dependent steps of one workgroup each, 40 us per batch on fifty CUs with the rest idle.)
// A swept region is noted for the summary (`swept`), which then redoes those regions only.
__global__ __launch_bounds__(256) void bloom_sweep_kernel(const uint32_t *__restrict__ order, uint8_t *__restrict__ bloom,
                                                          uint8_t *__restrict__ touched, uint8_t *__restrict__ swept, uint64_t bloom_dev_bytes)
{
    const uint32_t r = blockIdx.x;
    if (!touched[r]) return;                                // (uniform over the workgroup)
    __syncthreads();
    if (threadIdx.x == 0) { touched[r] = 0; swept[r] = 1; }
    const uint64_t base = (uint64_t)r << kBloomRegionLog2;
    constexpr uint32_t kCells = 1u << kBloomRegionLog2;
#pragma unroll
    for (uint32_t i = threadIdx.x * 4u; i < kCells; i += 256u * 4u) {
        const uint64_t cell = base + i;
        if (cell + 4 <= bloom_dev_bytes) {
            const uint4 k = *reinterpret_cast<const uint4 *>(order + cell);       // (cell is a multiple of four)
            const uint32_t b = *reinterpret_cast<const uint32_t *>(bloom + cell);
            uint32_t nb = b;
            if (!(b & 0x000000ffu) && k.x != 0xffffffffu) nb |= (1u << (k.x & 7u));
            if (!(b & 0x0000ff00u) && k.y != 0xffffffffu) nb |= (1u << (k.y & 7u)) << 8;
            if (!(b & 0x00ff0000u) && k.z != 0xffffffffu) nb |= (1u << (k.z & 7u)) << 16;
            if (!(b & 0xff000000u) && k.w != 0xffffffffu) nb |= (1u << (k.w & 7u)) << 24;
            if (nb != b) *reinterpret_cast<uint32_t *>(bloom + cell) = nb;
        } else {
            for (uint64_t c2 = cell; c2 < bloom_dev_bytes && c2 < cell + 4; ++c2)
                if (bloom[c2] == 0 && order[c2] != 0xffffffffu) bloom[c2] = (uint8_t)(1u << (order[c2] & 7u));
        }
    }
}

static int launch_bloom_sweep(mk_ctx *c)
{
    if (!c->d_bloom || !c->d_bloom_order) return MK_OK;
    const uint64_t regions = bloom_regions(c);
    hipLaunchKernelGGL(bloom_sweep_kernel, dim3((uint32_t)((c->bloom_dev_bytes >> kBloomRegionLog2) + 1)), dim3(256), 0, c->stream, c->d_bloom_order,
                       c->d_bloom, c->d_bloom_touched, c->d_bloom_touched + regions, c->bloom_dev_bytes);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// full[grp] = all eight cells of group grp are non-zero.  Cells never go back to zero, so a
// summary taken before a batch stays true during it; it is refreshed after every pass B.
// `swept` (or null: everything): the regions of 2^kBloomRegionLog2 cells that changed since the summary was last made
// (bloom_sweep_kernel); a thread whose cells lie in another region keeps its two words as they are.
__global__ __launch_bounds__(256) void bloom_summary_kernel(const uint8_t *__restrict__ bloom, uint64_t bloom_dev_bytes,
                                                            uint32_t *__restrict__ full, uint64_t nwords,
                                                            uint16_t *__restrict__ full2, uint64_t n16, uint8_t *__restrict__ swept)
{
    // a thread makes TWO summary words (each = 32 groups = 256 cells): words 2t and 2t + 1
    static_assert(kBloomRegionLog2 >= 11, "the four threads of a coarse bit share a region");
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t region = (t * 512) >> kBloomRegionLog2;
    const bool redo = !swept || (2 * t < nwords && swept[region] != 0);
    bool both = true;
    for (uint32_t half = 0; half < 2; ++half) {
        const uint64_t w = 2 * t + half;
        uint32_t bits = 0;
        const uint64_t base = w * 256;
        if (w >= nwords) {
            bits = 0;
        } else if (!redo) {
            bits = full[w];
        } else if (base + 256 <= bloom_dev_bytes) {
            const uint4 *__restrict__ p = reinterpret_cast<const uint4 *>(bloom + base);
#pragma unroll 4
            for (uint32_t i = 0; i < 16; ++i) {                               // 16 bytes = 2 groups per load
                const uint4 v = p[i];
                const uint64_t a = ((uint64_t)v.y << 32) | v.x, b = ((uint64_t)v.w << 32) | v.z;
                // exact zero-byte test: high bit of each byte set iff the byte is zero
                const uint64_t za = ~(((a & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL) | a | 0x7f7f7f7f7f7f7f7fULL);
                const uint64_t zb = ~(((b & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL) | b | 0x7f7f7f7f7f7f7f7fULL);
                bits |= (za == 0 ? 1u : 0u) << (2 * i);
                bits |= (zb == 0 ? 1u : 0u) << (2 * i + 1);
            }
        } else {                                                              // ragged end of the reachable region
            for (uint32_t gI = 0; gI < 32; ++gI) {
                bool all = true;
                for (uint32_t j = 0; j < 8; ++j) {
                    const uint64_t cell = base + gI * 8 + j;
                    all = all && cell < bloom_dev_bytes && bloom[cell] != 0;
                }
                bits |= (all ? 1u : 0u) << gI;
            }
        }
        if (w < nwords && redo) full[w] = bits;
        both = both && bits == 0xffffffffu;
    }
    // coarse level: one bit per EIGHT summary words = "all 2048 cells taken, and the cell after them" -- 4 KiB for the 64 MiB of reachable cells at
    // -b 33, small enough to ride in the LDS of the build's scatter kernel (build.hip), which asks it for every k-mer.
    // Four neighbouring threads hold those eight words; a wave writes its 16 bits.
    // (... and the first cell of the NEXT 2048: a k-mer whose Bloom positions straddle a carry of the low word lands in its
    // cell or the one after it, and the scatter kernel does not look which)
    if ((t & 3u) == 3u) {
        const uint64_t next = (t + 1) * 512;
        both = both && next < bloom_dev_bytes && bloom[next] != 0;
    }
    unsigned long long m = __ballot(both);
    m &= m >> 1; m &= m >> 2;                                                 // bit 4i: threads 4i .. 4i+3 all full
    uint32_t packed = 0;
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) packed |= (uint32_t)((m >> (4 * i)) & 1ull) << i;
    if ((threadIdx.x & 63u) == 0 && (t >> 6) < n16) full2[t >> 6] = (uint16_t)packed;
    // (a region's threads are neighbours in one wave: all of them have read the flag by now)
    if (swept && redo && (t * 512) % (1u << kBloomRegionLog2) == 0) swept[region] = 0;
}

int launch_bloom_summary(mk_ctx *c, bool after_sweep)
{
    if (!c->d_bloom) return MK_OK;
    const uint64_t nwords = (c->bloom_dev_bytes / 8 + 31) / 32 + 1, threads = (nwords + 1) / 2;
    // after a sweep, with a summary that was current before it: the swept regions only (a young filter: all of them; a filter
    // that has filled up: the few at the top of the range -- 19 us of reading 64 MiB per batch otherwise)
    uint8_t *swept = nullptr;
    if (c->d_bloom_touched) {
        swept = c->d_bloom_touched + bloom_regions(c);
        if (!after_sweep || c->bloom_full_stale) {
            MK_HIP(hipMemsetAsync(swept, 0, bloom_regions(c), c->stream));   // (everything is redone: nothing is pending after it)
            swept = nullptr;
        }
    }
    hipLaunchKernelGGL(bloom_summary_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, c->d_bloom,
                       c->bloom_dev_bytes, c->d_bloom_full, nwords, (uint16_t *)c->d_bloom_full2, bloom_summary_bytes(c) / 2, swept);
    MK_HIP(hipGetLastError());
    c->bloom_full_stale = false;
    return MK_OK;
}

// Genome-sharded builds: the reference has ONE filter whose cells keep the byte of their FIRST
// inserter in genome order (Miekki.cpp:125-129).  Folding the shards' filters in shard order
// with "an occupied cell keeps its byte, an empty one takes the later shard's" gives exactly that.
__global__ void bloom_merge_kernel(uint8_t *__restrict__ cells, const uint8_t *__restrict__ later, uint64_t n)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i + 16 <= n) {
        uint4 a = *reinterpret_cast<const uint4 *>(cells + i);
        const uint4 b = *reinterpret_cast<const uint4 *>(later + i);
        auto fold = [](uint32_t x, uint32_t y) {
            // per byte: x != 0 ? x : y  (mask = 0xff where the byte of x is zero)
            const uint32_t nz = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;          // high bit set iff byte non-zero
            const uint32_t zero_mask = (((~nz) >> 7) & 0x01010101u) * 0xffu;
            return x | (y & zero_mask);
        };
        a.x = fold(a.x, b.x); a.y = fold(a.y, b.y); a.z = fold(a.z, b.z); a.w = fold(a.w, b.w);
        *reinterpret_cast<uint4 *>(cells + i) = a;
    } else {
        for (uint64_t j = i; j < n; ++j)
            if (cells[j] == 0) cells[j] = later[j];
    }
}

int launch_bloom_merge(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_later)
{
    if (!c->d_bloom || end <= begin) return MK_OK;
    const uint64_t n = end - begin;
    if ((begin & 15u) || ((uintptr_t)d_later & 15u)) { set_error("Bloom merge needs 16-byte aligned ranges"); return MK_ERR_ARG; }
    const uint64_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(bloom_merge_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, c->d_bloom + begin,
                       d_later, n);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_bloom_insert(mk_ctx *c, uint64_t *d_tables, const char *d_seq, const uint64_t *d_off,
                        const uint32_t *d_valid, uint32_t n, const uint32_t *d_abort)
{
    if (!n || !c->d_bloom) return MK_OK;
    hipLaunchKernelGGL(bloom_post_kernel, dim3((c->P + 255) / 256, n), dim3(256), 0, c->stream, d_tables, d_seq, d_off, d_valid,
                       c->d_bloom, c->bloom_dev_bytes, c->d_bloom_order, c->d_bloom_touched, d_abort, c->d_bloom_full, kOvfCap, make_sp(c));
    MK_HIP(hipGetLastError());
    MK_TRY(launch_bloom_sweep(c));
    return launch_bloom_summary(c, true);
}

__device__ __forceinline__ bool bloom_check(const uint8_t *__restrict__ bloom, uint64_t bloom_dev_bytes,
                                            uint64_t canon, uint64_t anc, uint32_t bloom_log2,
                                            const uint32_t *__restrict__ full = nullptr)
{
    // check_bloom, Miekki.cpp:135-146: `cell && mask[hit]` is a logical and -> byte != 0.
    // The five positions differ by less than 1024 >> b, i.e. nearly always name ONE cell:
    // it is looked at once (every lane-load is a request of its own at the L2), and in the
    // "all eight cells set" summary first when there is one (1 MiB, L2-resident).
    uint64_t prev = ~0ull;
    for (uint32_t i = 0; i < kNumHash; ++i) {
        const uint64_t cell = bloom_pos(canon, anc, i, bloom_log2) >> 3;
        if (cell == prev) continue;
        prev = cell;
        if (cell >= bloom_dev_bytes) return false;
        if (full && ((full[cell >> 8] >> ((cell >> 3) & 31u)) & 1u)) continue;
        if (bloom[cell] == 0) return false;
    }
    return true;
}

// ---------------------------------------------------------------- K4 query sketch (short)
// One workgroup per query of at most kShortMax k-mers.  Keys
// (partition << 34 | fingerprint << 18 | position << 1 | Bloom gate) are sorted in
// LDS -- grouped by the partition's top bits with one LDS atomic per key, then a few
// keys per group sorted where they lie; a full bitonic sort (55 barriers for 1,024
// keys) only when a group is crowded.  The first key of every partition run is that
// partition's winner.  Winners that pass the Bloom gate are compacted, in ascending
// partition order, into the query's slice of the entry list -- the sparse form of the
// reference's dense 2^h vector.
__global__ __launch_bounds__(256) void query_sketch_kernel(const char *__restrict__ seq,
                                                           const uint64_t *__restrict__ off,
                                                           const uint64_t *__restrict__ ent_off,
                                                           uint64_t *__restrict__ entries,
                                                           uint32_t *__restrict__ nent,
                                                           const uint8_t *__restrict__ bloom,
                                                           uint64_t bloom_dev_bytes,
                                                           const uint32_t *__restrict__ bloom_full, uint32_t npad_max,
                                                           SketchParams sp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + (size_t)npad_max * sizeof(uint64_t));   // npad + 1 entries
    uint8_t *codes = smem + (size_t)npad_max * (sizeof(uint64_t) + sizeof(uint32_t)) + 16;
    __shared__ uint32_t s_seed_bad;
    __shared__ uint32_t s_max_bucket;
    __shared__ uint32_t s_wave_tot[4];
    const uint32_t q = blockIdx.x;
    const uint64_t len = off[q + 1] - off[q];
    const uint64_t nk64 = len > sp.k ? len - sp.k : 0;
    if (nk64 > kShortMax) return;                            // long path handles it
    const uint32_t nk = (uint32_t)nk64;
    if (nk == 0) { if (threadIdx.x == 0) nent[q] = 0; return; }
    const char *__restrict__ s = seq + off[q];
    uint32_t npad = 256;
    while (npad < nk) npad <<= 1;

    if (threadIdx.x == 0) s_seed_bad = 0;
    __syncthreads();
    if (threadIdx.x + 1 < sp.k && seed_code((uint8_t)s[threadIdx.x]) == 4u) atomicOr(&s_seed_bad, 1u);
    __syncthreads();
    const bool sv = s_seed_bad == 0;
    const uint32_t nchar = nk + sp.k - 1;
    for (uint32_t j = threadIdx.x; j < nchar; j += 256) codes[j] = (uint8_t)pos_codes((uint8_t)s[j], j, sp.k, sv);
    __syncthreads();

    // ---- keys, grouped by bucket = the top log2(npad) bits of the partition -----------------
    // Thread t rolls the k-mers [t * per, (t + 1) * per) and keeps their keys in registers;
    // one LDS atomic per key gives both the bucket histogram and the key's rank in its bucket.
    constexpr uint32_t kMaxPer = kShortMax / 256;
    const uint32_t per = npad / 256;
    uint32_t lg = 8;
    while ((1u << lg) < npad) ++lg;
    const uint32_t bshift = sp.h > lg ? sp.h - lg : 0;             // bucket = partition >> bshift  (< npad)
    for (uint32_t i = threadIdx.x; i <= npad; i += 256) hist[i] = 0;
    if (threadIdx.x == 0) s_max_bucket = 0;
    __syncthreads();
    uint64_t kreg[kMaxPer];
    uint32_t rreg[kMaxPer];
    {
        const uint32_t i0 = threadIdx.x * per;
        uint64_t S = 0, RC = 0;
        if (i0 < nk)
            for (uint32_t j = 0; j + 1 < sp.k; ++j) {              // first k-1 digits of k-mer i0
                const uint32_t cd = codes[i0 + j];
                S = (S << 2) | (cd & 3u);
                RC |= (uint64_t)(cd >> 2) << (2 * (j + 1));
            }
        const uint32_t topshift = 2 * sp.k - 2;
#pragma unroll
        for (uint32_t e = 0; e < kMaxPer; ++e) {
            kreg[e] = kEmptyKey; rreg[e] = 0;
            const uint32_t i = i0 + e;
            if (e < per && i < nk) {
                const uint32_t cd = codes[i + sp.k - 1];
                S = ((S << 2) | (cd & 3u)) & sp.kmask;
                RC = (RC >> 2) | ((uint64_t)(cd >> 2) << topshift);
                const uint64_t canon = S < RC ? S : RC;
                const uint64_t anc = revhash64(canon);
                uint32_t bucket, fp;
                bucket_fp(anc, sp.h, sp.f, sp.empty, bucket, fp);
                if (fp != sp.empty) {
                    // the Bloom gate is asked here, where the k-mer is at hand, and rides in the
                    // key's lowest bit (below the position: it cannot change who wins a partition)
                    const uint32_t pass = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, anc, sp.bloom_log2, bloom_full);
                    kreg[e] = ((uint64_t)bucket << 34) | ((uint64_t)fp << 18) | (i << 1) | pass;
                    rreg[e] = atomicAdd(&hist[bucket >> bshift], 1u);
                }
            }
        }
    }
    __syncthreads();
    // exclusive prefix of the histogram (per consecutive entries per thread), total in hist[npad]
    {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        uint32_t c[kMaxPer], tsum = 0, cmax = 0;
#pragma unroll
        for (uint32_t e = 0; e < kMaxPer; ++e) {
            c[e] = e < per ? hist[threadIdx.x * per + e] : 0u;
            tsum += c[e];
            cmax = max(cmax, c[e]);
        }
        if (cmax > 16) atomicMax(&s_max_bucket, cmax);
        uint32_t v = tsum;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        if (lane == 63) s_wave_tot[wave] = v;
        __syncthreads();
        uint32_t excl = v - tsum;
        for (uint32_t w = 0; w < wave; ++w) excl += s_wave_tot[w];
#pragma unroll
        for (uint32_t e = 0; e < kMaxPer; ++e)
            if (e < per) { hist[threadIdx.x * per + e] = excl; excl += c[e]; }
        if (threadIdx.x == 255) hist[npad] = excl;
    }
    __syncthreads();
    const uint32_t nvalid = hist[npad];
#pragma unroll
    for (uint32_t e = 0; e < kMaxPer; ++e)
        if (kreg[e] != kEmptyKey) keys[hist[(uint32_t)(kreg[e] >> 34) >> bshift] + rreg[e]] = kreg[e];
    for (uint32_t i = nvalid + threadIdx.x; i < npad; i += 256) keys[i] = kEmptyKey;
    __syncthreads();
    if (s_max_bucket == 0) {
        // ---- the usual case: a handful of keys per bucket, sorted where they lie -----------------
        for (uint32_t e = 0; e < per; ++e) {
            const uint32_t bI = threadIdx.x * per + e;
            const uint32_t lo = hist[bI], hi = hist[bI + 1];
            for (uint32_t i = lo + 1; i < hi; ++i) {                // insertion sort, at most 16 keys
                const uint64_t x = keys[i];
                uint32_t j = i;
                while (j > lo && keys[j - 1] > x) { keys[j] = keys[j - 1]; --j; }
                keys[j] = x;
            }
        }
        __syncthreads();
    } else {
        // ---- a crowded bucket (repetitive query): bitonic sort of everything -------------------
        for (uint32_t kk = 2; kk <= npad; kk <<= 1)
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t i = threadIdx.x; i < npad; i += 256) {
                    const uint32_t l = i ^ j;
                    if (l > i) {
                        const uint64_t a = keys[i], b = keys[l];
                        const bool up = (i & kk) == 0;
                        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                    }
                }
                __syncthreads();
            }
    }

    // each thread owns a contiguous run so that the compaction keeps partition order
    const uint32_t b0 = threadIdx.x * per;
    uint64_t out[kShortMax / 256];
    uint32_t cnt = 0;
    for (uint32_t e = 0; e < per; ++e) {
        const uint32_t i = b0 + e;
        const uint64_t key = keys[i];
        if (key == kEmptyKey) continue;
        const uint32_t bucket = (uint32_t)(key >> 34);
        if (i > 0 && (uint32_t)(keys[i - 1] >> 34) == bucket) continue;   // not the run's first
        if (key & 1u) out[cnt++] = make_entry(bucket, (uint32_t)(key >> 18) & 0xffffu);
    }
    // exclusive scan of cnt over the workgroup
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave_tot[w];
    const uint32_t total = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
    uint64_t *__restrict__ dst = entries + ent_off[q] + base + (incl - cnt);
    for (uint32_t e = 0; e < cnt; ++e) dst[e] = out[e];
    if (threadIdx.x == 0) nent[q] = total;
}

int launch_query_sketch_short(mk_ctx *c, mk_qset *qs)
{
    if (!qs->nq || qs->short_max_nk == 0) {
        // queries without any k-mer still need a zero count
        if (qs->nq) MK_HIP(hipMemsetAsync(qs->d_nent, 0, (size_t)qs->nq * sizeof(uint32_t), c->stream));
        return MK_OK;
    }
    uint32_t npad = 256;
    while (npad < qs->short_max_nk) npad <<= 1;
    MK_HIP(hipMemsetAsync(qs->d_nent, 0, (size_t)qs->nq * sizeof(uint32_t), c->stream));
    const size_t lds = (size_t)npad * (sizeof(uint64_t) + sizeof(uint32_t)) + 16 + npad + 64;   // keys, histogram, codes
    hipLaunchKernelGGL(query_sketch_kernel, dim3(qs->nq), dim3(256), lds, c->stream, qs->d_seq, qs->d_off,
                       qs->d_ent_off, qs->d_entries, qs->d_nent, c->d_bloom, c->bloom_dev_bytes, c->d_bloom_full, npad,
                       make_sp(c));
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- K4' query sketch (long)
// Whole-genome and other long queries reuse the genome path (one table in HBM),
// then gate and compact the table.  Entry order is irrelevant to the scan, so the
// compaction is a wave-aggregated append.
__global__ __launch_bounds__(256) void long_compact_kernel(const uint64_t *__restrict__ table,
                                                           const char *__restrict__ s,
                                                           const uint32_t *__restrict__ seed_valid,
                                                           const uint8_t *__restrict__ bloom,
                                                           uint64_t bloom_dev_bytes,
                                                           uint64_t *__restrict__ dst,
                                                           uint32_t *__restrict__ count, SketchParams sp)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    uint32_t fp = 0;
    if (p < sp.P) {
        const uint64_t key = table[p];
        if (key != kEmptyKey) {
            fp = (uint32_t)(key >> kPosBits);
            const uint64_t canon = canon_at(s, key & ((1ULL << kPosBits) - 1), sp.k, seed_valid[0] != 0);
            keep = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, revhash64(canon), sp.bloom_log2);
        }
    }
    const uint64_t mask = __ballot(keep);
    if (!mask) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t slot0 = 0;
    if (lane == leader) slot0 = atomicAdd(count, (uint32_t)__popcll(mask));
    slot0 = __shfl(slot0, (int)leader);
    if (keep) dst[slot0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = make_entry(p, fp);
}

int launch_query_sketch_long(mk_ctx *c, mk_qset *qs, uint32_t q)
{
    const uint64_t one_off[2] = {qs->h_off[q], qs->h_off[q + 1]};
    // table build on the single sequence: offsets are read from the set's own array
    uint32_t *d_valid = c->d_seed_valid;
    hipLaunchKernelGGL(seed_valid_kernel, dim3(1), dim3(64), 0, c->stream, qs->d_seq, qs->d_off + q, 1u,
                       c->p.k, d_valid);
    MK_TRY(launch_genome_sketch(c, qs->d_seq, qs->d_off + q, one_off, d_valid, 1, c->d_long_table));
    hipLaunchKernelGGL(long_compact_kernel, dim3((c->P + 255) / 256), dim3(256), 0, c->stream,
                       c->d_long_table, qs->d_seq + qs->h_off[q], d_valid, c->d_bloom, c->bloom_dev_bytes,
                       qs->d_entries + qs->h_ent_off[q], qs->d_nent + q, make_sp(c));
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// The same for a run of neighbouring long queries: one binned K1 run builds their tables,
// one launch (grid = (P / 256, n)) gates and appends.  A workgroup adds its keepers up first
// and reserves their places with ONE atomic, on a counter that has a cache line to itself:
// per-wave atomics on the n adjacent d_nent words would all queue on one L2 channel.
constexpr uint32_t kCountStride = 32;             // u32 words between two queries' counters
__global__ __launch_bounds__(256) void long_compact_batch_kernel(const uint64_t *__restrict__ tables,
                                                                 const uint8_t *__restrict__ codes,
                                                                 const uint8_t *__restrict__ except,
                                                                 const uint64_t *__restrict__ code_off,
                                                                 const uint32_t *__restrict__ dirty,
                                                                 const uint8_t *__restrict__ bloom,
                                                                 uint64_t bloom_dev_bytes,
                                                                 const uint32_t *__restrict__ full,
                                                                 uint64_t *__restrict__ entries,
                                                                 const uint64_t *__restrict__ ent_off,
                                                                 uint32_t *__restrict__ counters, SketchParams sp)
{
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_base;
    const uint32_t g = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    bool keep = false;
    uint32_t fp = 0;
    if (p < sp.P) {
        const uint64_t key = __builtin_nontemporal_load(tables + (uint64_t)g * sp.P + p);
        if (key != kEmptyKey) {
            fp = (uint32_t)(key >> kPosBits);
            // (the queries' packed form, as the build's kernels left it: launch_query_tables)
            const uint64_t canon = canon_from_packed(reinterpret_cast<const uint64_t *>(codes + code_off[g]),
                                                     reinterpret_cast<const uint64_t *>(except + code_off[g] / 2), dirty[g] != 0,
                                                     key & ((1ULL << kPosBits) - 1), sp.k);
            keep = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, revhash64(canon), sp.bloom_log2, full);
        }
    }
    const uint64_t mask = __ballot(keep);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (total == 0) return;                                        // uniform over the workgroup
    if (threadIdx.x == 0) s_base = atomicAdd(counters + (uint64_t)g * kCountStride, total);
    __syncthreads();
    if (!keep) return;
    uint32_t at = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w];
    entries[ent_off[g] + at] = make_entry(p, fp);
}

__global__ void spread_counts_kernel(const uint32_t *__restrict__ counters, uint32_t n, uint32_t *__restrict__ nent)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) nent[g] = counters[(uint64_t)g * kCountStride];
}

// long queries q0 .. q0+n-1 (neighbours in the set); *done = false: nothing was written, sketch them one by one
int launch_query_sketch_long_batch(mk_ctx *c, mk_qset *qs, uint32_t q0, uint32_t n, bool *done)
{
    *done = false;
    bool used = false;
    MK_TRY(launch_query_tables(c, qs->d_seq, qs->d_off + q0, qs->h_off.data() + q0, n, c->d_tables, &used));
    if (!used) return MK_OK;
    const uint8_t *pk_codes = c->d_pk[0], *pk_except = c->d_pk[0] + c->pk_cap[0];
    const uint32_t *pk_dirty = c->side[0].d_counters->dirty;
    // (no build is in flight: a side's overflow list is free scratch)
    uint32_t *counters = reinterpret_cast<uint32_t *>(c->d_ovf);
    MK_HIP(hipMemsetAsync(counters, 0, (size_t)n * kCountStride * 4, c->stream));
    hipLaunchKernelGGL(long_compact_batch_kernel, dim3((c->P + 255) / 256, n), dim3(256), 0, c->stream, c->d_tables,
                       pk_codes, pk_except, c->d_pk_off[0], pk_dirty, c->d_bloom, c->bloom_dev_bytes, c->d_bloom_full, qs->d_entries,
                       qs->d_ent_off + q0, counters, make_sp(c));
    hipLaunchKernelGGL(spread_counts_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, counters, n, qs->d_nent + q0);
    MK_HIP(hipGetLastError());
    *done = true;
    return MK_OK;
}

// ---------------------------------------------------------------- K4' query sketch (mid-length): O(length)
// Long reads and contigs -- more k-mers than the in-LDS sketch takes (kShortMax), fewer than a quarter of the 2^h
// partitions -- cost the table path O(2^h) each whatever their length: a 2^h-entry table filled by the build's kernels and
// swept by the gate-and-append launch (13 us per 20 kb read at -h 20; the reference's sketch loop is O(length),
// Miekki.cpp:162-183, and its O(2^h) gate, 214-224, is the defect SURVEY.md row A8 names).  Here a query gets an
// open-addressing HASH TABLE of 2 x (its k-mers) 64-bit slots in device memory instead: a workgroup rolls 4096 k-mers of
// one query out of LDS, asks the Bloom gate while the k-mer is at hand (as the short kernel does: the answer rides in the
// key's lowest bit, below the position) and puts  partition << 35 | fingerprint << 19 | position << 1 | gate  into the
// partition's slot -- claimed with a compare-and-swap, lowered with an atomic minimum: the smallest fingerprint, the
// earliest position among equals (Miekki.cpp:172), whatever the order the k-mers arrive in.  A second launch sweeps the
// slots (2 per k-mer, not 2^h per query) and appends the winners that pass the gate.  Entry order within a query is that
// of the slots: such queries take the plain scan schedule, which does not need sorted lists.
constexpr uint32_t kMidSeg = 4096;                 // k-mers per workgroup
constexpr uint32_t kMidPer = kMidSeg / 256;
constexpr uint32_t kMidPosBits = 18;               // positions a key can hold: queries up to 2^18 k-mers
struct MidWork { uint32_t q_local, chunk; };

// A partition's home slot rises with the partition (partitions are hash bits: they spread evenly as they are), so that
// the sweep over the slots meets the winners in NEARLY partition order -- neighbours displaced by a few places, 256-slot
// blocks appended in the order their workgroups arrive -- and the scan walks the matrix rows nearly in order.
__device__ __forceinline__ uint32_t mid_home(uint32_t bucket, uint32_t nslots, uint32_t h)
{
    return (uint32_t)(((uint64_t)bucket * nslots) >> h);
}

__global__ __launch_bounds__(256) void mid_insert_kernel(const char *__restrict__ seq, const uint64_t *__restrict__ off,
                                                         const uint32_t *__restrict__ qidx, const uint64_t *__restrict__ tab_off,
                                                         const MidWork *__restrict__ work, unsigned long long *__restrict__ table,
                                                         const uint8_t *__restrict__ bloom, uint64_t bloom_dev_bytes,
                                                         const uint32_t *__restrict__ bloom_full, SketchParams sp)
{
    __shared__ uint8_t codes[kMidSeg + 32];
    __shared__ uint32_t s_seed_bad;
    const MidWork w = work[blockIdx.x];
    const uint32_t q = qidx[w.q_local];
    const uint64_t len = off[q + 1] - off[q];
    const uint32_t nk = (uint32_t)(len - sp.k);                        // (the host only sends queries with kShortMax < nk <= 2^18)
    const uint32_t i_begin = w.chunk * kMidSeg;
    const uint32_t cnt = min(kMidSeg, nk - i_begin);
    const char *__restrict__ s = seq + off[q];
    if (threadIdx.x == 0) s_seed_bad = 0;
    __syncthreads();
    if (threadIdx.x + 1 < sp.k && seed_code((uint8_t)s[threadIdx.x]) == 4u) atomicOr(&s_seed_bad, 1u);
    __syncthreads();
    const bool sv = s_seed_bad == 0;
    const uint32_t nchar = cnt + sp.k - 1;
    for (uint32_t j = threadIdx.x; j < nchar; j += 256) codes[j] = (uint8_t)pos_codes((uint8_t)s[i_begin + j], (uint64_t)i_begin + j, sp.k, sv);
    __syncthreads();
    const uint32_t i0 = threadIdx.x * kMidPer;
    if (i0 >= cnt) return;
    unsigned long long *__restrict__ tab = table + tab_off[w.q_local];
    const uint32_t nslots = (uint32_t)(tab_off[w.q_local + 1] - tab_off[w.q_local]);
    uint64_t S = 0, RC = 0;
    for (uint32_t j = 0; j + 1 < sp.k; ++j) {                         // first k-1 digits of k-mer i0
        const uint32_t cd = codes[i0 + j];
        S = (S << 2) | (cd & 3u);
        RC |= (uint64_t)(cd >> 2) << (2 * (j + 1));
    }
    const uint32_t topshift = 2 * sp.k - 2;
    for (uint32_t e = 0; e < kMidPer && i0 + e < cnt; ++e) {
        const uint32_t cd = codes[i0 + e + sp.k - 1];
        S = ((S << 2) | (cd & 3u)) & sp.kmask;
        RC = (RC >> 2) | ((uint64_t)(cd >> 2) << topshift);
        const uint64_t canon = S < RC ? S : RC;
        const uint64_t anc = revhash64(canon);
        uint32_t bucket, fp;
        bucket_fp(anc, sp.h, sp.f, sp.empty, bucket, fp);
        if (fp == sp.empty) continue;
        const uint32_t pass = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, anc, sp.bloom_log2, bloom_full);
        const unsigned long long key = ((unsigned long long)bucket << (kMidPosBits + 17)) | ((unsigned long long)fp << (kMidPosBits + 1)) |
                                       ((unsigned long long)(i_begin + i0 + e) << 1) | pass;
        uint32_t at = mid_home(bucket, nslots, sp.h);
        for (;;) {                                                    // fewer distinct partitions than slots: the walk ends
            const unsigned long long old = atomicCAS(&tab[at], (unsigned long long)kEmptyKey, key);
            if (old == kEmptyKey) break;
            if ((uint32_t)(old >> (kMidPosBits + 17)) == bucket) { atomicMin(&tab[at], key); break; }
            if (++at == nslots) at = 0;
        }
    }
}

// 256 slots of one query per workgroup (a query's slots are a multiple of 256): winners that pass the gate -> entries
__global__ __launch_bounds__(256) void mid_compact_kernel(const unsigned long long *__restrict__ table, const uint64_t *__restrict__ tab_off,
                                                          uint32_t n, const uint32_t *__restrict__ qidx, uint64_t *__restrict__ entries,
                                                          const uint64_t *__restrict__ ent_off, uint32_t *__restrict__ counters)
{
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_base, s_q;
    const uint64_t slot0 = (uint64_t)blockIdx.x * 256;
    if (threadIdx.x == 0) {                                           // the query whose slots these are
        uint32_t lo = 0, hi = n;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tab_off[mid] <= slot0) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    const uint32_t ql = s_q, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long key = __builtin_nontemporal_load(table + slot0 + threadIdx.x);
    const bool keep = key != kEmptyKey && (key & 1ull);
    const uint64_t mask = __ballot(keep);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (total == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(counters + (uint64_t)ql * kCountStride, total);
    __syncthreads();
    if (!keep) return;
    uint32_t at = s_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wave; ++w) at += s_cnt[w];
    entries[ent_off[qidx[ql]] + at] = make_entry((uint32_t)(key >> (kMidPosBits + 17)), (uint32_t)(key >> (kMidPosBits + 1)) & 0xffffu);
}

__global__ void mid_counts_kernel(const uint32_t *__restrict__ counters, uint32_t n, const uint32_t *__restrict__ qidx, uint32_t *__restrict__ nent)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) nent[qidx[g]] = counters[(uint64_t)g * kCountStride];
}

bool query_is_mid_length(const mk_ctx *c, uint64_t nk) { return nk > kShortMax && nk <= (1ull << kMidPosBits) && nk < c->P / 4; }

// the mid-length queries `which` (set indices): hash tables carved out of `d_scratch` (scratch_slots 64-bit slots), as many
// queries per round as fit
int launch_query_sketch_mid(mk_ctx *c, mk_qset *qs, const std::vector<uint32_t> &which, unsigned long long *d_scratch, uint64_t scratch_slots)
{
    if (which.empty()) return MK_OK;
    const SketchParams sp = make_sp(c);
    // per round: qidx[n] u32 | tab_off[n + 1] u64 | work[...] -- one upload
    std::vector<uint32_t> qidx;
    std::vector<uint64_t> tab_off;
    std::vector<MidWork> work;
    uint8_t *d_meta = nullptr;
    uint64_t meta_cap = 0;
    int rc = MK_OK;
    size_t i = 0;
    while (i < which.size() && rc == MK_OK) {
        qidx.clear(); work.clear(); tab_off.assign(1, 0);
        while (i < which.size() && qidx.size() < kOvfCap / kCountStride) {
            const uint32_t q = which[i];
            const uint64_t nk = qs->h_off[q + 1] - qs->h_off[q] - c->p.k;
            const uint64_t slots = (2 * nk + 255) / 256 * 256;
            if (tab_off.back() + slots > scratch_slots) break;
            for (uint32_t ch = 0; ch * (uint64_t)kMidSeg < nk; ++ch) work.push_back(MidWork{(uint32_t)qidx.size(), ch});
            qidx.push_back(q);
            tab_off.push_back(tab_off.back() + slots);
            ++i;
        }
        const uint32_t n = (uint32_t)qidx.size();
        if (!n) { set_error("a mid-length query does not fit the sketch scratch"); rc = MK_ERR_NOMEM; break; }
        const uint64_t o_tab = ((uint64_t)n * 4 + 7) / 8 * 8, o_work = o_tab + (uint64_t)(n + 1) * 8, bytes = o_work + work.size() * sizeof(MidWork);
        if (bytes > meta_cap) {
            if (d_meta) { (void)hipStreamSynchronize(c->stream); (void)hipFree(d_meta); d_meta = nullptr; }
            meta_cap = bytes + bytes / 2;
            if (hipMalloc((void **)&d_meta, meta_cap) != hipSuccess) { set_error("out of device memory"); rc = MK_ERR_NOMEM; break; }
        }
        std::vector<uint8_t> img(bytes);
        memcpy(img.data(), qidx.data(), (size_t)n * 4);
        memcpy(img.data() + o_tab, tab_off.data(), (size_t)(n + 1) * 8);
        memcpy(img.data() + o_work, work.data(), work.size() * sizeof(MidWork));
        bool ok = hipMemcpyAsync(d_meta, img.data(), bytes, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;              // (img is pageable and about to go)
        uint32_t *counters = reinterpret_cast<uint32_t *>(c->d_ovf);         // (no build is in flight: a side's list is free scratch)
        ok = ok && hipMemsetAsync(counters, 0, (size_t)n * kCountStride * 4, c->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(d_scratch, 0xFF, tab_off.back() * 8, c->stream) == hipSuccess;
        if (!ok) { set_error("mid-length sketch setup failed: %s", hipGetErrorString(hipGetLastError())); rc = MK_ERR_DEVICE; break; }
        const uint32_t *d_qidx = reinterpret_cast<const uint32_t *>(d_meta);
        const uint64_t *d_tab_off = reinterpret_cast<const uint64_t *>(d_meta + o_tab);
        hipLaunchKernelGGL(mid_insert_kernel, dim3((uint32_t)work.size()), dim3(256), 0, c->stream, qs->d_seq, qs->d_off, d_qidx, d_tab_off,
                           reinterpret_cast<const MidWork *>(d_meta + o_work), d_scratch, c->d_bloom, c->bloom_dev_bytes, c->d_bloom_full, sp);
        hipLaunchKernelGGL(mid_compact_kernel, dim3((uint32_t)(tab_off.back() / 256)), dim3(256), 0, c->stream, d_scratch, d_tab_off, n, d_qidx,
                           qs->d_entries, qs->d_ent_off, counters);
        hipLaunchKernelGGL(mid_counts_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, counters, n, d_qidx, qs->d_nent);
        if (hipGetLastError() != hipSuccess) { set_error("mid-length sketch launch failed"); rc = MK_ERR_DEVICE; }
    }
    if (d_meta) { (void)hipStreamSynchronize(c->stream); (void)hipFree(d_meta); }
    return rc;
}

// ---------------------------------------------------------------- K4'' query sketch (dense)
// table -> this query's lane of the group's interleaved fingerprint vector
template <int W>
__global__ __launch_bounds__(256) void long_dense_kernel(const uint64_t *__restrict__ table,
                                                         const char *__restrict__ s,
                                                         const uint32_t *__restrict__ seed_valid,
                                                         const uint8_t *__restrict__ bloom,
                                                         uint64_t bloom_dev_bytes, uint8_t *__restrict__ dense_group,
                                                         uint32_t slot_in_group, uint32_t *__restrict__ count,
                                                         SketchParams sp)
{
    using fp_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    uint32_t fp = sp.empty;
    if (p < sp.P) {
        const uint64_t key = table[p];
        if (key != kEmptyKey) {
            const uint64_t canon = canon_at(s, key & ((1ULL << kPosBits) - 1), sp.k, seed_valid[0] != 0);
            keep = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, revhash64(canon), sp.bloom_log2);
            if (keep) fp = (uint32_t)(key >> kPosBits);
        }
        reinterpret_cast<fp_t *>(dense_group)[(uint64_t)p * 4 + slot_in_group] = (fp_t)fp;
    }
    const uint64_t mask = __ballot(keep);
    if (mask && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)mask) - 1u) atomicAdd(count, (uint32_t)__popcll(mask));
}

int launch_query_sketch_dense(mk_ctx *c, mk_qset *qs, uint32_t slot)
{
    const uint32_t q = qs->dense_q[slot];
    const uint64_t one_off[2] = {qs->h_off[q], qs->h_off[q + 1]};
    uint32_t *d_valid = c->d_seed_valid;
    hipLaunchKernelGGL(seed_valid_kernel, dim3(1), dim3(64), 0, c->stream, qs->d_seq, qs->d_off + q, 1u,
                       c->p.k, d_valid);
    MK_TRY(launch_genome_sketch(c, qs->d_seq, qs->d_off + q, one_off, d_valid, 1, c->d_long_table));
    uint8_t *group = qs->d_dense + (uint64_t)(slot / 4) * c->P * 4 * c->W;
    if (c->W == 1)
        hipLaunchKernelGGL(long_dense_kernel<1>, dim3((c->P + 255) / 256), dim3(256), 0, c->stream, c->d_long_table,
                           qs->d_seq + qs->h_off[q], d_valid, c->d_bloom, c->bloom_dev_bytes, group, slot % 4,
                           qs->d_nent + q, make_sp(c));
    else
        hipLaunchKernelGGL(long_dense_kernel<2>, dim3((c->P + 255) / 256), dim3(256), 0, c->stream, c->d_long_table,
                           qs->d_seq + qs->h_off[q], d_valid, c->d_bloom, c->bloom_dev_bytes, group, slot % 4,
                           qs->d_nent + q, make_sp(c));
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// The same for a batch: tables of n queries (one binned K1 run), k-mers read from the packed
// codes that run wrote, Bloom gate through the summary -- the access pattern of the build's
// Bloom pass A, and the same remedies.  grid = (P / 256, n).  The count of active partitions
// goes through per-workgroup partial sums: one atomic per wave on the n adjacent counters
// (two cache lines) serialises a million atomics on one L2 channel -- 10 ms per batch.
template <int W>
__global__ __launch_bounds__(256) void dense_batch_kernel(const uint64_t *__restrict__ tables,
                                                          const uint8_t *__restrict__ codes,
                                                          const uint8_t *__restrict__ except,
                                                          const uint64_t *__restrict__ code_off,
                                                          const uint32_t *__restrict__ dirty,
                                                          const uint8_t *__restrict__ bloom, uint64_t bloom_dev_bytes,
                                                          const uint32_t *__restrict__ full,
                                                          uint8_t *__restrict__ dense, uint32_t slot0,
                                                          uint32_t *__restrict__ partial, SketchParams sp)
{
    using fp_t = typename std::conditional<W == 1, uint8_t, uint16_t>::type;
    __shared__ uint32_t s_cnt[4];
    const uint32_t g = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    const uint32_t slot = slot0 + g;
    bool keep = false;
    uint32_t fp = sp.empty;
    if (p < sp.P) {
        const uint64_t key = __builtin_nontemporal_load(tables + (uint64_t)g * sp.P + p);
        if (key != kEmptyKey) {
            const uint64_t canon = canon_from_packed(reinterpret_cast<const uint64_t *>(codes + code_off[g]),
                                                     reinterpret_cast<const uint64_t *>(except + code_off[g] / 2), dirty[g] != 0,
                                                     key & ((1ULL << kPosBits) - 1), sp.k);
            keep = !bloom || bloom_check(bloom, bloom_dev_bytes, canon, revhash64(canon), sp.bloom_log2, full);
            if (keep) fp = (uint32_t)(key >> kPosBits);
        }
        fp_t *group = reinterpret_cast<fp_t *>(dense + (uint64_t)(slot / 4) * sp.P * 4 * W);
        group[(uint64_t)p * 4 + (slot & 3u)] = (fp_t)fp;
    }
    const uint64_t mask = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) partial[(uint64_t)g * gridDim.x + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(256) void dense_count_kernel(const uint32_t *__restrict__ partial, uint32_t nblk,
                                                          uint32_t *__restrict__ nent)
{
    __shared__ uint32_t s_sum[4];
    uint32_t v = 0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 256) v += partial[(uint64_t)blockIdx.x * nblk + i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) nent[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// Dense queries dense_q[slot .. slot+n) that are neighbours in the set (q, q+1, ...): their
// tables come from ONE run of the binned genome sketch (K1) instead of n runs of the atomic
// kernel.  *done = false leaves the queries untouched (shape does not fit the bins, or the
// overflow list ran over): the caller then sketches them one by one.
int launch_query_sketch_dense_batch(mk_ctx *c, mk_qset *qs, uint32_t slot, uint32_t n, bool *done)
{
    *done = false;
    const uint32_t q0 = qs->dense_q[slot];
    bool used = false;
    MK_TRY(launch_query_tables(c, qs->d_seq, qs->d_off + q0, qs->h_off.data() + q0, n, c->d_tables, &used));
    if (!used) return MK_OK;
    const uint8_t *pk_codes = c->d_pk[0], *pk_except = c->d_pk[0] + c->pk_cap[0];
    const uint32_t *pk_dirty = c->side[0].d_counters->dirty;
    const dim3 grid((c->P + 255) / 256, n);
    // (no build is in flight: a side's overflow list is free scratch)
    uint32_t *partial = reinterpret_cast<uint32_t *>(c->d_ovf);
    if ((uint64_t)grid.x * n * 4 > (uint64_t)kOvfCap * 16) return MK_OK;
    if (c->W == 1)
        hipLaunchKernelGGL(dense_batch_kernel<1>, grid, dim3(256), 0, c->stream, c->d_tables, pk_codes, pk_except, c->d_pk_off[0],
                           pk_dirty, c->d_bloom, c->bloom_dev_bytes, c->d_bloom_full, qs->d_dense, slot, partial, make_sp(c));
    else
        hipLaunchKernelGGL(dense_batch_kernel<2>, grid, dim3(256), 0, c->stream, c->d_tables, pk_codes, pk_except, c->d_pk_off[0],
                           pk_dirty, c->d_bloom, c->bloom_dev_bytes, c->d_bloom_full, qs->d_dense, slot, partial, make_sp(c));
    hipLaunchKernelGGL(dense_count_kernel, dim3(n), dim3(256), 0, c->stream, partial, grid.x, qs->d_nent + q0);
    MK_HIP(hipGetLastError());
    *done = true;
    return MK_OK;
}

// scan_n = nent, except 0 for the queries the dense kernel handles
__global__ void scan_counts_kernel(const uint32_t *__restrict__ nent, uint32_t nq, const uint32_t *__restrict__ dense_q,
                                   uint32_t ndense, uint32_t *__restrict__ scan_n, int pass)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (pass == 0) { if (i < nq) scan_n[i] = nent[i]; }
    else if (i < ndense && dense_q[i] != 0xffffffffu) scan_n[dense_q[i]] = 0;
}

int launch_scan_counts(mk_ctx *c, mk_qset *qs)
{
    if (!qs->nq) return MK_OK;
    hipLaunchKernelGGL(scan_counts_kernel, dim3((qs->nq + 255) / 256), dim3(256), 0, c->stream, qs->d_nent, qs->nq,
                       qs->d_dense_q, (uint32_t)qs->dense_q.size(), qs->d_scan_n, 0);
    if (!qs->dense_q.empty())
        hipLaunchKernelGGL(scan_counts_kernel, dim3(((uint32_t)qs->dense_q.size() + 255) / 256), dim3(256), 0, c->stream,
                           qs->d_nent, qs->nq, qs->d_dense_q, (uint32_t)qs->dense_q.size(), qs->d_scan_n, 1);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- range boundaries (slab schedule)
// entries of a short query are ascending by partition: split[q][r] = first entry
// whose partition is >= r * P / S.
__global__ void split_kernel(const uint64_t *__restrict__ entries, const uint64_t *__restrict__ ent_off,
                             const uint32_t *__restrict__ nent, uint32_t nq, uint32_t S, uint32_t P, uint32_t limit,
                             uint32_t *__restrict__ split, uint32_t *__restrict__ flag)
{
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint64_t)nq * S) return;
    const uint32_t q = (uint32_t)(id / S), r = (uint32_t)(id - (uint64_t)q * S);
    const uint64_t *__restrict__ e = entries + ent_off[q];
    const uint32_t n = nent[q];
    uint32_t bound[2];
    for (uint32_t k = 0; k < 2; ++k) {
        const uint64_t target = (uint64_t)(r + k) * (P / S);
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint32_t)e[mid] < target) lo = mid + 1; else hi = mid;
        }
        bound[k] = (r + k == S) ? n : lo;
    }
    split[(uint64_t)q * (S + 1) + r] = bound[0];
    if (r + 1 == S) split[(uint64_t)q * (S + 1) + S] = n;
    if (bound[1] - bound[0] > limit) atomicOr(flag, 1u);
}

int launch_query_split(mk_ctx *c, mk_qset *qs, uint32_t S, uint32_t limit, uint32_t *d_flag)
{
    if (!qs->nq) return MK_OK;
    const uint64_t n = (uint64_t)qs->nq * S;
    hipLaunchKernelGGL(split_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, qs->d_entries,
                       qs->d_ent_off, qs->d_nent, qs->nq, S, c->P, limit, qs->d_split, d_flag);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- synthetic inputs
__global__ void synth_genomes_kernel(uint64_t first_id, uint32_t n, uint64_t len, char *__restrict__ out)
{
    const uint64_t words = (len + 31) / 32;
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = blockIdx.y;
    if (w >= words || g >= n) return;
    const uint64_t bits = genome_word(first_id + g, w);
    char *dst = out + (uint64_t)g * len + w * 32;
    const uint32_t m = (uint32_t)min((uint64_t)32, len - w * 32);
    if (m == 32 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        // 32 characters as two 16-byte stores: 2-bit code c -> 'A' 'C' 'G' 'T' = 0x41 + (0x13060200 >> 8c & 0xff)
        uint32_t wd[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t cd = (uint32_t)(bits >> (62 - 2 * (4 * j + e))) & 3u;
                v |= (0x41u + ((0x13060200u >> (8 * cd)) & 0xffu)) << (8 * e);
            }
            wd[j] = v;
        }
        reinterpret_cast<uint4 *>(dst)[0] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        reinterpret_cast<uint4 *>(dst)[1] = make_uint4(wd[4], wd[5], wd[6], wd[7]);
        return;
    }
    for (uint32_t i = 0; i < m; ++i) dst[i] = "ACGT"[(bits >> (62 - 2 * i)) & 3];
}

int launch_synth_genomes(mk_ctx *c, uint64_t first_id, uint32_t n, uint64_t len, char *d_out)
{
    if (!n || !len) return MK_OK;
    const uint64_t words = (len + 31) / 32;
    hipLaunchKernelGGL(synth_genomes_kernel, dim3((uint32_t)((words + 255) / 256), n), dim3(256), 0, c->stream,
                       first_id, n, len, d_out);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

__global__ void synth_queries_kernel(uint64_t first_id, uint32_t nq, uint64_t G, uint64_t L, uint64_t qlen,
                                     char *__restrict__ out)
{
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const uint64_t id = first_id + q;
    const uint64_t g = id % G;
    const uint64_t o = splitmix64(kSeedQ ^ id) % (L - qlen);
    for (uint64_t i = threadIdx.x; i < qlen; i += blockDim.x) {
        const uint64_t pos = o + i;
        const uint64_t bits = genome_word(g, pos >> 5);
        out[(uint64_t)q * qlen + i] = "ACGT"[(bits >> (62 - 2 * (pos & 31))) & 3];
    }
}

int launch_synth_queries(mk_ctx *c, uint64_t first_id, uint32_t nq, uint64_t G, uint64_t L, uint64_t qlen,
                         char *d_out)
{
    if (!nq) return MK_OK;
    hipLaunchKernelGGL(synth_queries_kernel, dim3(nq), dim3(256), 0, c->stream, first_id, nq, G, L, qlen, d_out);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---------------------------------------------------------------- column import / export
// dense staging [rows][G*W] (reference byte order: 16-bit big-endian) <-> M rows
template <int W, bool TO_DEVICE>
__global__ void convert_columns_kernel(MatRef M, uint64_t ld, uint32_t G, uint32_t p_begin,
                                       uint32_t rows, uint8_t *__restrict__ staging)
{
    const uint64_t total = (uint64_t)rows * G;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(idx / G), g = (uint32_t)(idx - (uint64_t)r * G);
        uint8_t *m = mat_row(M, p_begin + r, ld) + (uint64_t)g * W;
        uint8_t *s = staging + idx * W;
        if (W == 1) {
            if (TO_DEVICE) *m = *s; else *s = *m;
        } else {
            if (TO_DEVICE) { m[0] = s[1]; m[1] = s[0]; } else { s[0] = m[1]; s[1] = m[0]; }
        }
    }
}

int launch_convert_columns(mk_ctx *c, bool to_device, uint32_t p_begin, uint32_t p_end, uint8_t *d_staging)
{
    const uint32_t rows = p_end - p_begin;
    if (!rows || !c->G) return MK_OK;
    const uint64_t total = (uint64_t)rows * c->G;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 8192);
#define MK_CONV(Wv, TD)                                                                                 \
    hipLaunchKernelGGL((convert_columns_kernel<Wv, TD>), dim3(blocks), dim3(256), 0, c->stream, mat_ref(c), \
                       c->ld, c->G, p_begin, rows, d_staging)
    if (c->W == 1) { if (to_device) MK_CONV(1, true); else MK_CONV(1, false); }
    else           { if (to_device) MK_CONV(2, true); else MK_CONV(2, false); }
#undef MK_CONV
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// The columns of n chosen genomes, every partition: dst[p][j] = fingerprint of genome ids[j] in partition p, in the
// byte order of dump_disk (16-bit values big-endian) -- i.e. the column block an index of just those genomes would
// dump (Miekki.cpp:665-668).  A sample of a collection's columns then costs n x 2^h x W bytes instead of an
// export of the whole matrix.
template <int W>
__global__ void export_genomes_kernel(MatRef M, uint64_t ld, uint32_t P, const uint32_t *__restrict__ ids, uint32_t n,
                                      uint8_t *__restrict__ dst)
{
    const uint64_t total = (uint64_t)P * n;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)(idx / n), j = (uint32_t)(idx - (uint64_t)p * n);
        const uint8_t *m = mat_row(M, p, ld) + (uint64_t)ids[j] * W;
        uint8_t *d = dst + idx * W;
        if (W == 1) d[0] = m[0];
        else { d[0] = m[1]; d[1] = m[0]; }
    }
}

int launch_export_genomes(mk_ctx *c, const uint32_t *d_ids, uint32_t n, uint8_t *d_dst)
{
    if (!n) return MK_OK;
    const uint64_t total = (uint64_t)c->P * n;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 16384);
    if (c->W == 1) hipLaunchKernelGGL((export_genomes_kernel<1>), dim3(blocks), dim3(256), 0, c->stream, mat_ref(c), c->ld, c->P, d_ids, n, d_dst);
    else hipLaunchKernelGGL((export_genomes_kernel<2>), dim3(blocks), dim3(256), 0, c->stream, mat_ref(c), c->ld, c->P, d_ids, n, d_dst);
    MK_HIP(hipGetLastError());
    return MK_OK;
}


}  // namespace mk
