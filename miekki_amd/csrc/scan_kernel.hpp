// K5 scan kernel template (see scan.hip for the design notes).  Kept in a header so
// that tools/scan_tune.hip can instantiate tuning variants of exactly this code.
#pragma once
#include <type_traits>

#include "mk_internal.hpp"

namespace mk {

template <int W>
__device__ __forceinline__ uint32_t bcast_fp(uint32_t fp)
{
    return W == 1 ? fp * 0x01010101u : fp * 0x00010001u;
}

// 1 in the low bit of every byte (half-word) of d that DIFFERS from the query fingerprint
template <int W>
__device__ __forceinline__ uint32_t ne_lanes(uint32_t d, uint32_t b)
{
    const uint32_t x = d ^ b;
    if (W == 1) {
        const uint32_t y = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
        return ((y | x) >> 7) & 0x01010101u;
    } else {
        const uint32_t y = (x & 0x7fff7fffu) + 0x7fff7fffu;
        return ((y | x) >> 15) & 0x00010001u;
    }
}

// Address of the wave's tile of row p as a SCALAR base (SALU multiply-add, kept in
// SGPRs through readfirstlane) so that the load is the saddr form
// `global_load_dwordx4 v, v_off, s[base]` with one 32-bit per-lane offset.
template <bool NT>
__device__ __forceinline__ uint4 load_row16(const uint8_t *p)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (NT) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4 *>(p);
}

__device__ __forceinline__ const uint8_t *row_base(const uint8_t *base, uint32_t p, uint64_t ld)
{
    const uint64_t roff = (uint64_t)p * ld;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)roff);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(roff >> 32));
    return base + (((uint64_t)hi << 32) | lo);
}

// the same with the row's home chosen first: rows below P_hot in HBM, the rest in (device-visible)
// host memory (MatRef).  p is wave-uniform, so the choice is scalar arithmetic too.
__device__ __forceinline__ const uint8_t *row_base(const uint8_t *hot, const uint8_t *cold, uint32_t P_hot, uint32_t p,
                                                   uint64_t ld)
{
    const uint32_t pu = __builtin_amdgcn_readfirstlane(p);
    return row_base(pu < P_hot ? hot : cold, pu, ld);
}

// One work item: query `ql` of the launch against row tile `tile`.
// WINDOW: only the entries whose partition lies in [a.row_lo, a.row_hi) count -- the launch covers one window
// of rows (the part of the matrix in HBM, or a cold range staged there: api.hip, scan_windows) and ADDS its
// share to the scores when a.accumulate is set.  The entries are wave-uniform, so a step of UNROLL entries
// with none in the window is skipped by a scalar branch; a step with some loads the window's first row for
// the others (cached) and masks them out.
template <int W, int UNROLL, bool NT, bool WINDOW = false>
__device__ __forceinline__ void scan_item(const ScanArgs &a, uint32_t ql, uint32_t tile, uint32_t lane)
{
    constexpr uint32_t NCNT = 16 / W;                    // genomes per lane
    constexpr uint32_t CHUNK = W == 1 ? 255u : 65535u;   // entries before packed counters overflow
    // lanes past the end of the row have nothing to compare: retire them now so the
    // last tile of every row only fetches what it needs
    if ((uint64_t)tile * kTileBytes + lane * 16u >= (uint64_t)a.G * W) return;
    const uint32_t q = a.q_begin + ql;
    const uint64_t *__restrict__ ent = a.entries + a.ent_off[q];
    const uint32_t n = a.nent[q];
    // scalar row base + 32-bit per-lane offset -> global_load_dwordx4 v, v_off, s[base]
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * kTileBytes;
    const uint8_t *__restrict__ cbase = (a.Mc ? a.Mc : a.M) + (uint64_t)tile * kTileBytes;
    const uint32_t P_hot = a.P_hot;
    const uint32_t voff = lane * 16u;
    const uint64_t ld = a.ld;

    uint32_t ne32[NCNT];
#pragma unroll
    for (uint32_t j = 0; j < NCNT; ++j) ne32[j] = 0;
    uint32_t n_in = WINDOW ? 0u : n;                                 // entries that count (wave-uniform)
    const uint32_t wlo = a.row_lo, wspan = a.row_hi - a.row_lo;

    for (uint32_t i0 = 0; i0 < n; i0 += CHUNK) {
        const uint32_t m = min(n - i0, CHUNK);
        const uint64_t *__restrict__ e = ent + i0;
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        uint32_t j = 0;
        for (; j + UNROLL <= m; j += UNROLL) {
            uint64_t ev[UNROLL];
            uint4 d[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) ev[u] = e[j + u];
            if (WINDOW) {
                uint32_t keep[UNROLL], any = 0;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    keep[u] = ((uint32_t)ev[u] - wlo) < wspan ? 0xffffffffu : 0u;
                    any |= keep[u];
                    n_in += keep[u] & 1u;
                }
                if (!__builtin_amdgcn_readfirstlane(any)) continue;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    d[u] = load_row16<NT>(row_base(base, cbase, P_hot, keep[u] ? (uint32_t)ev[u] : wlo, ld) + voff);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const uint32_t b = bcast_fp<W>((uint32_t)(ev[u] >> 32));
                    acc0 += ne_lanes<W>(d[u].x, b) & keep[u];
                    acc1 += ne_lanes<W>(d[u].y, b) & keep[u];
                    acc2 += ne_lanes<W>(d[u].z, b) & keep[u];
                    acc3 += ne_lanes<W>(d[u].w, b) & keep[u];
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                d[u] = load_row16<NT>(row_base(base, cbase, P_hot, (uint32_t)ev[u], ld) + voff);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t b = bcast_fp<W>((uint32_t)(ev[u] >> 32));
                acc0 += ne_lanes<W>(d[u].x, b);
                acc1 += ne_lanes<W>(d[u].y, b);
                acc2 += ne_lanes<W>(d[u].z, b);
                acc3 += ne_lanes<W>(d[u].w, b);
            }
        }
        if (j < m) {                                                     // ragged last step: see scan_slab_kernel
            uint64_t ev[UNROLL];
            uint4 d[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) ev[u] = e[min(j + u, m - 1)];
            uint32_t keep[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                keep[u] = j + u < m ? 0xffffffffu : 0u;
                if (WINDOW) {
                    if (((uint32_t)ev[u] - wlo) >= wspan) keep[u] = 0u;
                    n_in += keep[u] & 1u;
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                d[u] = load_row16<NT>(row_base(base, cbase, P_hot, (!WINDOW || keep[u]) ? (uint32_t)ev[u] : wlo, ld) + voff);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t b = bcast_fp<W>((uint32_t)(ev[u] >> 32));
                acc0 += ne_lanes<W>(d[u].x, b) & keep[u];
                acc1 += ne_lanes<W>(d[u].y, b) & keep[u];
                acc2 += ne_lanes<W>(d[u].z, b) & keep[u];
                acc3 += ne_lanes<W>(d[u].w, b) & keep[u];
            }
        }
        const uint32_t acc[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            if (W == 1) {
                ne32[4 * w + 0] += acc[w] & 0xffu;
                ne32[4 * w + 1] += (acc[w] >> 8) & 0xffu;
                ne32[4 * w + 2] += (acc[w] >> 16) & 0xffu;
                ne32[4 * w + 3] += acc[w] >> 24;
            } else {
                ne32[2 * w + 0] += acc[w] & 0xffffu;
                ne32[2 * w + 1] += acc[w] >> 16;
            }
        }
    }

    // shared fingerprints = active entries - differing ones
    const uint32_t g0 = tile * (kTileBytes / W) + lane * NCNT;
    uint32_t score[NCNT];
#pragma unroll
    for (uint32_t j = 0; j < NCNT; ++j) score[j] = n_in - ne32[j];

    // scores[tile * score_tile_stride + ql * score_q_stride + genome-in-tile]: row-major
    // ([query][genome], q_stride = row pitch, tile_stride = genomes per tile) for callers
    // that want dense rows, tile-major ([tile][query][genomes per tile]) for the
    // pipeline -- there consecutive waves store consecutive 4 KiB pieces, a pure stream
    if (!a.scores) {                                                   // store-less timing runs (tools/scan_tune)
#pragma unroll
        for (uint32_t j = 0; j < NCNT; ++j) asm volatile("" ::"v"(score[j]));   // keep the compare loop live
        return;
    }
    uint32_t *__restrict__ row = a.scores + (uint64_t)tile * a.score_tile_stride + (uint64_t)ql * a.score_q_stride;
    const uint32_t i0 = lane * NCNT;                                   // genome index inside the tile
    if (WINDOW && a.accumulate) {                                      // a later window of the same launch sequence: this
        if (n_in == 0) return;                                         // wave owns the same words in every window, in stream order
        if (a.score_vec) {
#pragma unroll
            for (uint32_t j = 0; j < NCNT; j += 4) {
                uint4 v = *reinterpret_cast<uint4 *>(row + i0 + j);
                v.x += score[j]; v.y += score[j + 1]; v.z += score[j + 2]; v.w += score[j + 3];
                *reinterpret_cast<uint4 *>(row + i0 + j) = v;
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < NCNT; ++j)
                if (g0 + j < a.G) row[i0 + j] += score[j];
        }
        return;
    }
    if (a.score_vec) {
#pragma unroll
        for (uint32_t j = 0; j < NCNT; j += 4)
            *reinterpret_cast<uint4 *>(row + i0 + j) = make_uint4(score[j], score[j + 1], score[j + 2], score[j + 3]);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < NCNT; ++j)
            if (g0 + j < a.G) row[i0 + j] = score[j];
    }
}

// ORDER 0: consecutive waves take adjacent tiles of one query (query-major);
// ORDER 1: consecutive waves take the same tile of consecutive queries (tile-major).
// NT: non-temporal row loads.
template <int W, int UNROLL, int ORDER = 1, bool NT = false, bool WINDOW = false>
__global__ __launch_bounds__(256) void scan_kernel(const ScanArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t work = blockIdx.x * 4u + wave;
    if (work >= a.nq * a.ntiles) return;                 // wave-uniform exit
    uint32_t ql, tile;
    if (ORDER == 0) { ql = work / a.ntiles; tile = work - ql * a.ntiles; }
    else            { tile = work / a.nq;   ql = work - tile * a.nq; }
    scan_item<W, UNROLL, NT, WINDOW>(a, ql, tile, lane);
}

// ---------------------------------------------------------------- slab schedule
// The partitions are cut into S equal ranges and a work item is (tile, range,
// query), ordered tile-major then range-major: the waves in flight share one
// (2^h / S) x 1 KiB slab of the matrix, sized by the host to fit the Infinity
// Cache, so a row piece is fetched from HBM about once and then re-read on die by
// the other queries that need it (measured: 6.65 -> 8.0 TB/s at 1/8 of 2^20 rows).
// A (query, range) has at most 255 entries (65,535 at 2 bytes) -- the host checks --
// so the packed per-genome mismatch counters never overflow and are simply stored,
// 16 bytes per lane, as this range's PARTIAL: no read-modify-write, no ordering
// between ranges.  select.hip sums the S partials of a genome.
template <int W, int UNROLL>
__global__ __launch_bounds__(256) void scan_slab_kernel(const SlabArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t work = blockIdx.x * 4u + wave;
    if (work >= a.ntiles * a.r_count * a.nq) return;     // wave-uniform exit
    const uint32_t trl = work / a.nq, ql = work - trl * a.nq;
    const uint32_t tile = trl / a.r_count, r = a.r_begin + (trl - tile * a.r_count);   // this launch walks ranges [r_begin, r_begin + r_count)
    const uint32_t tr = tile * a.S + r;
    if ((uint64_t)tile * kTileBytes + lane * 16u >= (uint64_t)a.G * W) return;
    const uint32_t q = a.q_begin + ql;
    uint32_t lo, hi;
    if (a.chunk) {                                      // small sets: ranges by entry COUNT (no range table, no eligibility check)
        const uint32_t n = a.nent[q];
        lo = min(r * a.chunk, n);
        hi = min(lo + a.chunk, n);
    } else {
        lo = a.split[(uint64_t)q * (a.S + 1) + r];
        hi = a.split[(uint64_t)q * (a.S + 1) + r + 1];
    }
    const uint64_t *__restrict__ e = a.entries + a.ent_off[q] + lo;
    const uint32_t m = hi - lo;
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * kTileBytes;
    const uint8_t *__restrict__ cbase = (a.Mc ? a.Mc : a.M) + (uint64_t)tile * kTileBytes;
    const uint32_t P_hot = a.P_hot;
    const uint32_t voff = lane * 16u;
    const uint64_t ld = a.ld;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    uint32_t j = 0;
    for (; j + UNROLL <= m; j += UNROLL) {
        uint64_t ev[UNROLL];
        uint4 d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) ev[u] = e[j + u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) d[u] = load_row16<false>(row_base(base, cbase, P_hot, (uint32_t)ev[u], ld) + voff);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t b = bcast_fp<W>((uint32_t)(ev[u] >> 32));
            acc0 += ne_lanes<W>(d[u].x, b);
            acc1 += ne_lanes<W>(d[u].y, b);
            acc2 += ne_lanes<W>(d[u].z, b);
            acc3 += ne_lanes<W>(d[u].w, b);
        }
    }
    if (j < m) {
        // ragged last step, still UNROLL loads in flight: the spare slots re-read the
        // last entry (cached) and are masked out of the sums -- branch-free, so the
        // tail costs one memory latency instead of one per entry
        uint64_t ev[UNROLL];
        uint4 d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) ev[u] = e[min(j + u, m - 1)];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) d[u] = load_row16<false>(row_base(base, cbase, P_hot, (uint32_t)ev[u], ld) + voff);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t keep = j + u < m ? 0xffffffffu : 0u;           // wave-uniform
            const uint32_t b = bcast_fp<W>((uint32_t)(ev[u] >> 32));
            acc0 += ne_lanes<W>(d[u].x, b) & keep;
            acc1 += ne_lanes<W>(d[u].y, b) & keep;
            acc2 += ne_lanes<W>(d[u].z, b) & keep;
            acc3 += ne_lanes<W>(d[u].w, b) & keep;
        }
    }
    uint8_t *__restrict__ out = a.partials + ((uint64_t)tr * a.nq + ql) * kTileBytes + voff;
    *reinterpret_cast<uint4 *>(out) = make_uint4(acc0, acc1, acc2, acc3);
}

// ---------------------------------------------------------------- dense queries
// Whole-genome queries (-A) have nearly every partition active: walking a sparse
// list buys nothing and every query would stream the whole matrix on its own.
// Here a wave owns (GB groups of four queries, row tile, chunk of rows): each row piece
// is loaded ONCE and compared against the 4 * GB queries' fingerprints of that
// partition (one scalar dword per group), which divides the bytes per comparison by
// 4 * GB.  Mismatch counters are packed as in the sparse kernel (four byte counters per
// word, spilled every 248 rows into two 16-bit counters per word: a chunk has at most
// 8,192 rows); the wave's share of each score is added to the score row with integer
// atomics (order-independent, exact).
template <int W, int GB>
__global__ __launch_bounds__(256) void scan_dense_kernel(const DenseArgs a)
{
    constexpr uint32_t NCNT = 16 / W, FLUSH = W == 1 ? 248u : 65528u, QB = 4 * GB;
    constexpr uint32_t NWORD = 8;                                    // counter words per query (two counters each)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t work = blockIdx.x * 4u + wave;
    const uint32_t nsets = (a.ngroups + GB - 1) / GB;
    if (work >= nsets * a.ntiles * a.nchunks) return;
    const uint32_t chunk = work % a.nchunks, gt = work / a.nchunks;
    const uint32_t tile = gt % a.ntiles, set = gt / a.ntiles;
    if ((uint64_t)tile * kTileBytes + lane * 16u >= (uint64_t)a.G * W) return;
    uint32_t qidx[QB];
    bool any = false, grp_on[GB];
#pragma unroll
    for (uint32_t gb = 0; gb < (uint32_t)GB; ++gb) {
        const uint32_t group = set * GB + gb;
        grp_on[gb] = false;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t qi = group < a.ngroups ? a.dense_q[group * 4 + j] : 0xffffffffu;
            qidx[gb * 4 + j] = qi;
            grp_on[gb] |= qi >= a.q0 && qi < a.q1;
        }
        any |= grp_on[gb];
    }
    if (!any) return;
    // this launch walks rows [row_lo, row_hi) (the whole matrix, or one window of it: api.hip, scan_windows)
    const uint32_t row0 = a.row_lo + chunk * a.rows_per_item, row1 = min(a.row_hi, row0 + a.rows_per_item);
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * kTileBytes;
    const uint8_t *__restrict__ cbase = (a.Mc ? a.Mc : a.M) + (uint64_t)tile * kTileBytes;
    const uint32_t voff = lane * 16u;
    using fp4_t = typename std::conditional<W == 1, uint32_t, uint2>::type;
    const fp4_t *__restrict__ dv[GB];
#pragma unroll
    for (uint32_t gb = 0; gb < (uint32_t)GB; ++gb)
        dv[gb] = reinterpret_cast<const fp4_t *>(a.dense) + (uint64_t)min(set * GB + gb, a.ngroups - 1) * a.P;
    // W == 1: cnt[j][2w] / [2w+1] hold the even / odd byte counters of word w, 16 bits each;
    // W == 2: cnt[j][w] is acc's word w itself (two 16-bit counters), no spill ever needed
    uint32_t cnt[QB][NWORD], nact[QB];
#pragma unroll
    for (uint32_t j = 0; j < QB; ++j) {
        nact[j] = 0;
#pragma unroll
        for (uint32_t k = 0; k < NWORD; ++k) cnt[j][k] = 0;
    }
    for (uint32_t r0 = row0; r0 < row1; r0 += FLUSH) {
        const uint32_t r1 = min(row1, r0 + FLUSH);
        uint32_t acc[QB][4];
#pragma unroll
        for (uint32_t j = 0; j < QB; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0;
        for (uint32_t r = r0; r < r1; r += 4) {
            uint4 d[4];
            fp4_t f[4][GB];
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t rr = min(r + u, r1 - 1);                   // tail rows repeat the last (masked below)
#pragma unroll
                for (uint32_t gb = 0; gb < (uint32_t)GB; ++gb) f[u][gb] = dv[gb][rr];
                d[u] = load_row16<false>(row_base(base, cbase, a.P_hot, rr, a.ld) + voff);
            }
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                if (r + u >= r1) break;
#pragma unroll
                for (uint32_t j = 0; j < QB; ++j) {
                    if (!grp_on[j / 4]) continue;                         // wave-uniform
                    const fp4_t &fw = f[u][j / 4];
                    const uint32_t jj = j & 3u;
                    uint32_t fp;
                    if (W == 1) fp = (reinterpret_cast<const uint32_t &>(fw) >> (8 * jj)) & 0xffu;
                    else { const uint2 &ff = reinterpret_cast<const uint2 &>(fw); fp = ((jj < 2 ? ff.x : ff.y) >> (16 * (jj & 1))) & 0xffffu; }
                    if (fp == a.empty) continue;                          // wave-uniform
                    ++nact[j];
                    const uint32_t b = bcast_fp<W>(fp);
                    acc[j][0] += ne_lanes<W>(d[u].x, b);
                    acc[j][1] += ne_lanes<W>(d[u].y, b);
                    acc[j][2] += ne_lanes<W>(d[u].z, b);
                    acc[j][3] += ne_lanes<W>(d[u].w, b);
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < QB; ++j)
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                if (W == 1) {
                    cnt[j][2 * w + 0] += acc[j][w] & 0x00ff00ffu;
                    cnt[j][2 * w + 1] += (acc[j][w] >> 8) & 0x00ff00ffu;
                } else {
                    cnt[j][w] += acc[j][w];
                }
            }
    }
    const uint32_t g0 = tile * (kTileBytes / W) + lane * NCNT;
#pragma unroll
    for (uint32_t j = 0; j < QB; ++j) {
        if (qidx[j] < a.q0 || qidx[j] >= a.q1 || nact[j] == 0) continue;
        uint32_t *__restrict__ row = a.scores + (uint64_t)tile * a.score_tile_stride +
                                     (uint64_t)(qidx[j] - a.q0) * a.score_q_stride + lane * NCNT;
#pragma unroll
        for (uint32_t k = 0; k < NCNT; ++k) {
            uint32_t ne;
            if (W == 1) ne = (cnt[j][2 * (k / 4) + (k & 1u)] >> (16 * ((k & 3u) / 2))) & 0xffffu;   // byte k & 3 of word k / 4
            else ne = (cnt[j][k / 2] >> (16 * (k & 1u))) & 0xffffu;
            if (g0 + k < a.G && nact[j] != ne) atomicAdd(row + k, nact[j] - ne);
        }
    }
}

// ---------------------------------------------------------------- dense queries by table (described for one-byte fingerprints;
// two bytes: the last paragraph)
// scan_dense_kernel above tests a row piece against every query on its own -- six vector instructions per four
// comparisons, 48 per data word for its eight queries -- and is bound by exactly those (2.97e13 comparisons/s = 3.85 TB/s x
// 8 queries, half of what HBM would carry).  Here a data word costs about a dozen for EIGHT queries:
//  * "which of the eight queries have this byte at this row" is a function of the byte: mask(v) = T0[v & 3] & T1[(v >> 2) & 3]
//    & T2[(v >> 4) & 3] & T3[v >> 6], where Tf holds, per value of the 2-bit field f, the queries whose fingerprint has that
//    value there (a byte equals a fingerprint iff all four fields do).  The four tables of a row -- four bytes each, made
//    once per set by dense_lut_kernel -- are looked up for the four bytes of a word at once by v_perm_b32 (a four-entry
//    byte table IS one source word, and it comes straight from a scalar register: tables of eight entries need two
//    sources, of which only one may be scalar -- their copies into vector registers cost more than the fourth lookup);
//    the field selectors are shared by every group of eight queries the wave carries.
//  * the masks of successive rows are COUNTED bit by bit: every bit of the mask word (query j of genome byte b) has a
//    counter spread over bit planes, fed by carry-save adders -- sixteen rows go in with fifteen adders of two
//    instructions each (v_bitop3: sum and majority) and one ripple through the upper planes: three instructions per row
//    and word for all eight queries, where the byte counters above took three per query.
// Matches are counted directly (no "active rows minus mismatches"): a query without a fingerprint at a row is in no table.
// Two-byte fingerprints: a fingerprint matches where its low byte matches a query's low byte AND its high byte that query's
// high byte -- two sets of tables per row and octet; a word's high bytes (1 and 3) select from the upper half of an eight-entry
// table (the second source word of v_perm_b32) whose lower half is the low byte's; the byte-wise sets meet by m & (m >> 8).
struct DenseLut { uint32_t t[4]; };   // four 4-entry byte tables, one per 2-bit field of a byte: 16 bytes per (octet of queries, row) at one
                                      // byte per fingerprint; two of them at two bytes (the low byte's tables, then the high byte's)

template <int W>
__global__ __launch_bounds__(256) void dense_lut_kernel(const uint8_t *__restrict__ dense, uint32_t ngroups, uint32_t P, uint32_t empty,
                                                        DenseLut *__restrict__ lut)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x, octet = blockIdx.y;
    if (p >= P) return;
    DenseLut t[2];                                                    // (one byte: the three fields' tables; two bytes: the low byte's, the high byte's)
#pragma unroll
    for (uint32_t x = 0; x < 2u; ++x) t[x].t[0] = t[x].t[1] = t[x].t[2] = t[x].t[3] = 0;
#pragma unroll
    for (uint32_t half = 0; half < 2; ++half) {
        const uint32_t group = octet * 2 + half;
        if (group >= ngroups) break;
        // four queries' fingerprints of this row
        uint32_t f4[W];
        if (W == 1) f4[0] = reinterpret_cast<const uint32_t *>(dense)[(uint64_t)group * P + p];
        else { const uint2 v2 = reinterpret_cast<const uint2 *>(dense)[(uint64_t)group * P + p]; f4[0] = v2.x; f4[W - 1] = v2.y; }
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t v = W == 1 ? (f4[0] >> (8 * j)) & 0xffu : (f4[j / 2] >> (16 * (j & 1u))) & 0xffffu, bit = 1u << (half * 4 + j);
            if (v == empty) continue;
            if (W == 1) {
                // fields of three, three and two bits: two eight-entry byte tables (a pair of words each) and a four-entry one
                const uint32_t f0 = v & 7u, f1 = (v >> 3) & 7u, f2 = v >> 6;
                t[0].t[f0 >> 2] |= bit << (8 * (f0 & 3u));
                t[0].t[2 + (f1 >> 2)] |= bit << (8 * (f1 & 3u));
                t[1].t[0] |= bit << (8 * f2);
            } else {
#pragma unroll
                for (uint32_t x = 0; x < 2u; ++x)
#pragma unroll
                    for (uint32_t f = 0; f < 4; ++f) t[x].t[f] |= bit << (8 * ((v >> (8 * x + 2 * f)) & 3u));
            }
        }
    }
#pragma unroll
    for (uint32_t x = 0; x < 2u; ++x) lut[((uint64_t)octet * P + p) * 2 + x] = t[x];
}

// sum and carry of three one-bit inputs per bit position
__device__ __forceinline__ void csa(uint32_t &carry, uint32_t &sum, uint32_t a, uint32_t b, uint32_t c)
{
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);            // majority: one instruction (v_bitop3_b32)
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);              // parity: one more
}

// GS: the sets (of NO octets) whose waves SHARE the rows of one (tile, chunk of rows): 1 -- a wave loads its rows itself --, or
// 2 or 4 waves of the workgroup: each loads a share of a batch of eight rows, leaves it in LDS, and all of them work from
// there.  (Round 4 let the sets of a tile walk the same rows as neighbouring waves and trusted the caches: the counters saw
// every row come through the fabric 1.8 times at two sets, and a wave that carries sixteen queries has one neighbour on its
// SIMD to hide a load's latency behind -- vector issue stood at 0.66.)
// (a row's tables are the wave's -- every lane the same address -- and the compiler loads them with VECTOR loads: it cannot know
// that the score rows' atomics do not touch them.  Forced through the constant address space they become scalar loads, and the
// kernel twice as slow at 64 queries (162 against 80 ms: 268 MB of tables stream through a 16 KiB scalar cache): left as it is.)
__device__ __forceinline__ uint4 load_tables16(const DenseLut *p) { return *reinterpret_cast<const uint4 *>(p); }

template <int W, int NO, int GS>
__global__ __launch_bounds__(256) void scan_dense_lut_kernel(const DenseArgs a)
{
    // counts up to 16,383: the host cuts chunks of at most 16,368 rows (dense_chunk_rows)
    constexpr uint32_t kPlanes = 14;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nsets = (a.noctets + NO - 1) / NO;
    // a sharing group: GS waves, one (tile, chunk) and GS sets; a workgroup holds 4 / GS of them.  A wave without work of its own
    // (a set beyond the last, a group beyond the grid's end, no query of this pass) still loads its share and keeps the barriers
    const uint32_t nsg = (nsets + GS - 1) / GS, ngroups_all = nsg * a.ntiles * a.nchunks;
    const uint32_t grp_in_wg = wave / GS, w_in_grp = wave % GS;
    // Workgroups are dealt to the eight XCDs round robin: XCD x walks the groups x * n / 8 ... in order, and a group's number has
    // the TILE running fastest -- so the workgroups in flight on one XCD walk the same chunk of rows of neighbouring tiles and
    // find that chunk's tables (32 bytes per row and octet, the same for every tile) in their L2 instead of fetching them
    // once per tile (98 x 268 MB at 64 queries: a quarter on top of the matrix at the fabric, profiles/r6_pmc_dense.txt).
    // (The grid is a multiple of eight workgroups: launch_scan_dense.)
    const uint32_t per_xcd = gridDim.x / 8u, wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const uint32_t group_id = wg * (4u / GS) + grp_in_wg;
    const bool group_live = group_id < ngroups_all;
    if (GS == 1 && !group_live) return;
    const uint32_t gid = min(group_id, ngroups_all - 1u);
    const uint32_t tile = gid % a.ntiles, cs = gid / a.ntiles, set = (cs % nsg) * GS + w_in_grp, chunk = cs / nsg;
    const bool lane_live = (uint64_t)tile * kTileBytes + lane * 16u < (uint64_t)a.G * W;
    // (a lane beyond the tile's last genome stays: it carries its share of the rows' tables, below; its rows lie in the tile's padding)
    uint32_t qidx[NO][8];
    bool any = false;
#pragma unroll
    for (uint32_t o = 0; o < (uint32_t)NO; ++o)
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t slot = (set * NO + o) * 8 + j;
            const uint32_t qi = group_live && set < nsets && slot < a.ngroups * 4 ? a.dense_q[slot] : 0xffffffffu;
            qidx[o][j] = qi;
            any |= qi >= a.q0 && qi < a.q1;
        }
    if (GS == 1 && !any) return;
    const bool working = any && lane_live;                              // (wave-uniform but for the tile's last lanes)
    const uint32_t row0 = a.row_lo + chunk * a.rows_per_item, row1 = min(a.row_hi, row0 + a.rows_per_item);
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * kTileBytes;
    const uint8_t *__restrict__ cbase = (a.Mc ? a.Mc : a.M) + (uint64_t)tile * kTileBytes;
    const uint32_t voff = lane * 16u;
    const DenseLut *__restrict__ lut[NO];
#pragma unroll
    for (uint32_t o = 0; o < (uint32_t)NO; ++o) lut[o] = a.lut + (uint64_t)min(set * NO + o, a.noctets - 1) * a.P * 2;
    // plane[o][w][k]: bit k of the counters of mask word w (bit 8 b + j = query j of octet o against genome byte b of the word).
    // The low EIGHT planes live in registers; a carry out of them -- a counter passes a multiple of 256: at most once per counter in
    // 256 rows -- is OR-ed into a pending word and rippled through the five upper planes, which live in LDS (10 KiB per wave,
    // lane-major: no bank conflicts), once per 256 rows.  (Round 4 rippled every carry of weight sixteen through all nine upper
    // planes, one octet's in registers, the other's in LDS: eighteen instructions per sixteen rows, word and octet; nine now.)
    constexpr uint32_t kRegPlanes = 8, kUp = kPlanes - kRegPlanes;
    __shared__ uint32_t s_up[4][NO * kUp * 4][64];
    uint32_t plane[NO][4][kRegPlanes], pend[NO][4];
#pragma unroll
    for (uint32_t o = 0; o < (uint32_t)NO; ++o)
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            pend[o][w] = 0;
#pragma unroll
            for (uint32_t k = 0; k < kPlanes; ++k) {
                if (k < kRegPlanes) plane[o][w][k] = 0;
                else s_up[wave][(o * kUp + (k - kRegPlanes)) * 4 + w][lane] = 0;
            }
        }
    auto flush_pending = [&]() {                                         // the pending carries of weight 256 into the upper planes
#pragma unroll
        for (uint32_t o = 0; o < (uint32_t)NO; ++o)
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                uint32_t c = pend[o][w];
                pend[o][w] = 0;
#pragma unroll
                for (uint32_t k = 0; k < kUp; ++k) {
                    const uint32_t t = s_up[wave][(o * kUp + k) * 4 + w][lane];
                    s_up[wave][(o * kUp + k) * 4 + w][lane] = t ^ c;
                    c &= t;
                }
            }
    };
    constexpr uint32_t RB = GS == 2 ? 4 : 8;                             // rows per batch of loads (two sharing groups in a workgroup: half, for their row buffers' LDS)
    // shared rows: two batches of eight rows (1 KiB each) per sharing group -- one being worked from, one being filled
    constexpr uint32_t kMine = RB / GS;                                  // rows of a batch this wave loads
    __shared__ uint4 s_rows[GS == 1 ? 1 : 4 / GS][GS == 1 ? 1 : 2][GS == 1 ? 1 : RB][GS == 1 ? 1 : 64];
    // (every group of a workgroup makes the same number of passes -- a chunk's rows, clamped at its end -- so that the barriers match)
    const uint32_t row_span = a.rows_per_item;
    auto load_mine = [&](uint32_t rb, uint4 (&v)[kMine]) {               // this wave's share of the batch whose first row is rb
#pragma unroll
        for (uint32_t u = 0; u < kMine; ++u) {
            const uint32_t rr = min(rb + u * GS + w_in_grp, row1 - 1u);
            v[u] = load_row16<false>(row_base(base, cbase, a.P_hot, rr, a.ld) + voff);
        }
    };
    auto store_mine = [&](uint32_t buf, const uint4 (&v)[kMine]) {
#pragma unroll
        for (uint32_t u = 0; u < kMine; ++u) s_rows[grp_in_wg][buf][u * GS + w_in_grp][lane] = v[u];
    };
    uint32_t nbuf = 0;
    if (GS > 1) {
        uint4 v0[kMine];
        load_mine(row0, v0);
        store_mine(0, v0);
        __syncthreads();
    }
    // The tables of sixteen rows -- 32 bytes per row and octet, 512 bytes per octet: contiguous -- come with ONE load of the wave
    // (a lane: 16 bytes; lanes 0-31 the first octet's, 32-63 the second's, with one octet lanes 32-63 idle along), a block of
    // sixteen rows ahead, and wait in LDS, where a pair of rows' tables are read as broadcasts.  (Before: 68 vector loads of 64
    // equal addresses per sixteen rows and wave -- a third of a wave's cycles were waits for them, profiles/r6_pmc_dense.txt.)
    __shared__ uint4 s_tab[4][2][64];
    auto load_tab = [&](uint32_t rb) {                                   // this lane's piece of the tables of rows [rb, rb + 16)
        const uint32_t o = NO == 2 ? lane >> 5 : 0u, piece = lane & 31u;
        const uint32_t rr = min(rb + (piece >> 1), row1 - 1u);
        return load_tables16(lut[NO == 2 ? o : 0] + (uint64_t)rr * 2 + (piece & 1u));
    };
    uint32_t tbuf = 0;
    uint4 tab_next = load_tab(row0);
    for (uint32_t r = row0; r < (GS == 1 ? row1 : row0 + row_span); r += 16) {
        uint32_t twosP[NO][4], foursP[NO][4], eightsP[NO][4];            // carries waiting for their partner
        s_tab[wave][tbuf][lane] = tab_next;                              // (this block's tables: in LDS before its first pair asks)
        tab_next = load_tab(r + 16);
#pragma unroll
        for (uint32_t bt = 0; bt < 16 / RB; ++bt) {                       // a batch: its loads first (all in flight), then the work
            uint4 d[RB];
            uint4 vnext[kMine];
            if (GS == 1) {
#pragma unroll
                for (uint32_t u = 0; u < RB; ++u) d[u] = load_row16<false>(row_base(base, cbase, a.P_hot, r + bt * RB + u, a.ld) + voff);
            } else {
                // the NEXT batch's share from memory (in flight under this batch's work), this batch's rows from LDS
                load_mine(r + bt * RB + RB, vnext);
#pragma unroll
                for (uint32_t u = 0; u < RB; ++u) d[u] = s_rows[grp_in_wg][nbuf][u][lane];
            }
            if (GS == 1 || (working && r < row1)) {
            // Two rows at a time (their tables: four scalar words per row and octet, asked for when their turn comes -- all
            // rows' tables at once do not fit the scalar registers and end up as vector copies): masks of the pair's four words,
            // into the ones; a carry waits for its partner of the same weight.  The fences keep the compiler from pooling the pairs.
#pragma unroll
            for (uint32_t pp = 0; pp < RB / 2; ++pp) {
                const uint32_t pr = bt * (RB / 2) + pp;                   // pair 0 .. 7 of the sixteen rows
                __builtin_amdgcn_sched_barrier(0);
                // this pair's tables: four (eight) broadcast reads
                uint4 tb[2][NO][2];
#pragma unroll
                for (uint32_t v = 0; v < 2; ++v)
#pragma unroll
                    for (uint32_t o = 0; o < (uint32_t)NO; ++o)
#pragma unroll
                        for (uint32_t x = 0; x < 2u; ++x) tb[v][o][x] = s_tab[wave][tbuf][o * 32u + (pr * 2u + v) * 2u + x];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (uint32_t w = 0; w < 4; ++w) {
                    uint32_t sel[2][4];
#pragma unroll
                    for (uint32_t v = 0; v < 2; ++v) {
                        const uint4 &dd = d[pp * 2 + v];
                        const uint32_t dw = w == 0 ? dd.x : w == 1 ? dd.y : w == 2 ? dd.z : dd.w;
                        if (W == 1) { sel[v][0] = dw & 0x07070707u; sel[v][1] = (dw >> 3) & 0x07070707u; sel[v][2] = (dw >> 6) & 0x03030303u; sel[v][3] = 0; }
                        else { sel[v][0] = dw & 0x03030303u; sel[v][1] = (dw >> 2) & 0x03030303u; sel[v][2] = (dw >> 4) & 0x03030303u; sel[v][3] = (dw >> 6) & 0x03030303u; }
                        // two-byte fingerprints: a word's high bytes (1 and 3) look their fields up in the high byte's tables -- the
                        // upper half of an eight-entry table whose lower half is the low byte's
                        if (W == 2) { sel[v][0] |= 0x04000400u; sel[v][1] |= 0x04000400u; sel[v][2] |= 0x04000400u; sel[v][3] |= 0x04000400u; }
                    }
#pragma unroll
                    for (uint32_t o = 0; o < (uint32_t)NO; ++o) {
                        uint32_t m[2];
#pragma unroll
                        for (uint32_t v = 0; v < 2; ++v) {
                            const uint4 lo = tb[v][o][0], hi = tb[v][o][1];
                            if (W == 1) {
                                // three lookups -- two in eight-entry tables (a pair of words: v_perm_b32's two sources, of which the
                                // compiler copies one into a vector register per row: a quarter of an instruction per word) and one in
                                // a four-entry table -- and ONE three-input AND
                                m[v] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_perm(lo.y, lo.x, sel[v][0]), __builtin_amdgcn_perm(lo.w, lo.z, sel[v][1]),
                                                                   __builtin_amdgcn_perm(0u, hi.x, sel[v][2]), 0x80);
                            } else {
                                const uint32_t both = __builtin_amdgcn_perm(hi.x, lo.x, sel[v][0]) & __builtin_amdgcn_perm(hi.y, lo.y, sel[v][1]) &
                                                      __builtin_amdgcn_perm(hi.z, lo.z, sel[v][2]) & __builtin_amdgcn_perm(hi.w, lo.w, sel[v][3]);
                                // (two bytes: a fingerprint matches where its low byte's set and its high byte's meet; the result sits in the low byte's place)
                                m[v] = both & (both >> 8) & 0x00ff00ffu;
                            }
                        }
                        uint32_t twos;
                        csa(twos, plane[o][w][0], plane[o][w][0], m[0], m[1]);
                        if (!(pr & 1u)) { twosP[o][w] = twos; continue; }
                        uint32_t fours;
                        csa(fours, plane[o][w][1], plane[o][w][1], twosP[o][w], twos);
                        if (!(pr & 2u)) { foursP[o][w] = fours; continue; }
                        uint32_t eights;
                        csa(eights, plane[o][w][2], plane[o][w][2], foursP[o][w], fours);
                        if (!(pr & 4u)) { eightsP[o][w] = eights; continue; }
                        uint32_t carry;
                        csa(carry, plane[o][w][3], plane[o][w][3], eightsP[o][w], eights);
                        // one bit of weight sixteen per position: through planes 4-7, what falls out of them waits in `pend`
#pragma unroll
                        for (uint32_t k = 4; k < kRegPlanes; ++k) {
                            const uint32_t t = plane[o][w][k];
                            plane[o][w][k] = t ^ carry;
                            carry &= t;
                        }
                        pend[o][w] |= carry;
                    }
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (GS > 1) {                                                 // the next batch into the other half; everybody has read this one after the barrier
                store_mine(nbuf ^ 1u, vnext);
                nbuf ^= 1u;
                __syncthreads();
            }
        }
        tbuf ^= 1u;
        if ((((r - row0) >> 4) & 15u) == 15u) flush_pending();            // (256 rows since the last time: wave-uniform)
    }
    if (!working) return;
    flush_pending();
    // the counters' planes -> numbers, query by query: four genomes of a word at a time (low eight planes into byte counters,
    // the upper planes into a second set), added to the score rows with integer atomics
#pragma unroll
    for (uint32_t o = 0; o < (uint32_t)NO; ++o)
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t qi = qidx[o][j];
            if (qi < a.q0 || qi >= a.q1) continue;                       // wave-uniform
            constexpr uint32_t NCNT = 16 / W;                             // genomes per lane
            uint32_t *__restrict__ row = a.scores + (uint64_t)tile * a.score_tile_stride + (uint64_t)(qi - a.q0) * a.score_q_stride + lane * NCNT;
            const uint32_t g0 = tile * (kTileBytes / W) + lane * NCNT;
#pragma unroll
            for (uint32_t w = 0; w < 4; ++w) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    const uint32_t pl = plane[o][w][k];
                    lo += ((pl >> j) & 0x01010101u) << k;
                }
#pragma unroll
                for (uint32_t k = 8; k < kPlanes; ++k) {                   // (at most six planes: the upper count stays below 64)
                    const uint32_t pl = s_up[wave][(o * kUp + (k - kRegPlanes)) * 4 + w][lane];
                    hi += ((pl >> j) & 0x01010101u) << (k - 8);
                }
#pragma unroll
                for (uint32_t b = 0; b < 4; b += W) {                      // (two bytes: the counts sit in bytes 0 and 2 of the word)
                    const uint32_t n = ((lo >> (8 * b)) & 0xffu) | (((hi >> (8 * b)) & 0xffu) << 8);
                    const uint32_t gi = (w * 4 + b) / W;
                    if (n && g0 + gi < a.G) atomicAdd(row + gi, n);
                }
            }
        }
}

}  // namespace mk
