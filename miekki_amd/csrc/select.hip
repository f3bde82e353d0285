// K6: top-hit selection over a query's scores -- the device half of
// Miekki::filter_results (Miekki.cpp:376-397).
//
// The reference walks genomes in ascending id, skips those below min_score or
// min_intersection (381-384) and, once its heap holds nresults entries, skips every
// genome whose intersection is below the heap minimum (387) WITHOUT touching the
// heap.  Only the genomes that are not skipped ("entrants") shape the heap and so
// the result, ties included.  With chance matches most genomes pass the two
// thresholds at scale (≈7 % of 100,000 at -h 20), but only ≈ N·ln(m/N) of them are
// entrants.  This kernel emits exactly the entrants of each query, in genome order;
// the heap itself (pop/push with ties replacing, sort_heap) is then replayed over
// those few dozen records by mk_filter_candidates with the reference's own
// libstdc++ calls -- on the host, or on rank 0 after the multi-GPU gather: a
// shard's entrants are a superset of what the global heap can admit from that
// shard, because the global minimum is never below the shard's own.
//
// One wave per query, 256 genomes per step (four per lane): 16-byte loads of the
// scores -- or, after the slab schedule, of the S per-range mismatch counts, summed
// SWAR-wise -- and of the genomes' sizes; an f32 screen against
// max(min_intersection, current heap minimum); the decision in the reference's
// double operations; `__ballot` over the lanes that hold candidate entrants and a
// short serial loop, lane by lane and genome by genome, that re-tests each against
// the evolving minimum.  The current top-N multiset lives one value per lane; its
// minimum is a wave reduction.
#include "mk_internal.hpp"

namespace mk {

__device__ __forceinline__ double wave_min_f64(double v)
{
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, uint32_t l)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, (int)l);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), (int)l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// The selection over u32 scores (plain schedule: long queries, mk_qset_scores' layout), four genomes per lane.
__global__ __launch_bounds__(256) void select_kernel(const SelectArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= a.nq) return;
    const uint64_t tile_stride = (uint64_t)a.nq * a.tile_genomes;      // entries
    const uint32_t N = a.nresults;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double topv = inf;               // lane i < cnt: i-th value of the current top-N multiset
    uint32_t cnt = 0, emitted = 0;   // wave-uniform
    double minval = 0.0;             // minimum of the multiset, valid when cnt == N
    float screen = 0.999f * (float)a.min_inter;
    mk_hit *__restrict__ out = a.cand ? a.cand + (uint64_t)q * a.cap : nullptr;
    // compact form (multi-GPU exchange): row = [count][cap x (genome | matches << 32)]
    uint64_t *__restrict__ crow = a.rows ? a.rows + (uint64_t)q * (a.cap + 1u) : nullptr;
    for (uint32_t g0 = 0; g0 < a.G; g0 += 256) {
        const uint32_t gl = g0 + lane * 4u;                            // this lane's four genomes
        uint32_t s[4] = {0, 0, 0, 0};
        double jac[4] = {0, 0, 0, 0}, inter[4] = {0, 0, 0, 0};
        uint32_t pot = 0;
        uint32_t ss[4] = {1, 1, 1, 1};
        uint64_t gs[4] = {0, 0, 0, 0};
        bool any = false;
        if (gl < a.G) {
            const uint32_t t = gl / a.tile_genomes, wi = gl - t * a.tile_genomes;   // 256 | tile_genomes
            const uint4 v = *reinterpret_cast<const uint4 *>(a.scores + (uint64_t)t * tile_stride + (uint64_t)q * a.tile_genomes + wi);
            s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) any |= (gl + j < a.G) && s[j] >= a.min_score;     // Miekki.cpp:381
            if (any) {
                // the size arrays are padded to whole tiles: 16-byte loads stay in bounds
                const uint4 ss4 = *reinterpret_cast<const uint4 *>(a.sketch_size + gl);
                const ulonglong2 gsa = *reinterpret_cast<const ulonglong2 *>(a.genome_size + gl);
                const ulonglong2 gsb = *reinterpret_cast<const ulonglong2 *>(a.genome_size + gl + 2);
                ss[0] = ss4.x; ss[1] = ss4.y; ss[2] = ss4.z; ss[3] = ss4.w;
                gs[0] = gsa.x; gs[1] = gsa.y; gs[2] = gsb.x; gs[3] = gsb.y;
            }
        }
        if (gl < a.G) {
            if (any) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (gl + j < a.G && s[j] >= a.min_score) {
                        const float est = (float)s[j] * (float)gs[j] / (float)ss[j];
                        if (!(est < screen)) {
                            jac[j] = (double)s[j] / (double)ss[j];              // Miekki.cpp:382-383
                            inter[j] = jac[j] * (double)gs[j];
                            if (!(inter[j] < a.min_inter) && (cnt < N || !(minval > inter[j]))) pot |= 1u << j;
                        }
                    }
                }
            }
        }
        uint64_t lanes = __ballot(pot != 0);
        while (lanes) {                                                    // ascending genome order
            const uint32_t l = (uint32_t)__ffsll((long long)lanes) - 1u;
            lanes &= lanes - 1;
            const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane((int)pot, (int)l);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((pm >> j) & 1u)) continue;
                const double x = readlane_f64(inter[j], l);
                if (cnt >= N && minval > x) continue;                       // Miekki.cpp:387: skipped, heap untouched
                if (N == 0) continue;
                if (lane == l && emitted < a.cap) {
                    if (crow) {
                        crow[1u + emitted] = (uint64_t)(gl + j + a.genome_id_base) | ((uint64_t)s[j] << 32);
                    } else {
                        mk_hit h;
                        h.genome = gl + j + a.genome_id_base;
                        h.matches = s[j];
                        h.jaccard = jac[j];
                        h.intersection = inter[j];
                        out[emitted] = h;
                    }
                }
                ++emitted;
                if (cnt < N) {
                    if (lane == cnt) topv = x;
                    ++cnt;
                } else {                                                    // evict one holder of the minimum
                    const uint64_t holders = __ballot(lane < N && topv == minval);
                    const uint32_t victim = holders ? (uint32_t)__ffsll((long long)holders) - 1u : 0u;
                    if (lane == victim) topv = x;
                }
                if (cnt == N) {
                    minval = wave_min_f64(lane < N ? topv : inf);
                    const double bar = minval > a.min_inter ? minval : a.min_inter;
                    screen = 0.999f * (float)bar;
                }
            }
        }
    }
    if (lane == 0) {
        if (crow) crow[0] = emitted;
        else a.count[q] = emitted;
    }
}

// The same selection over the slab schedule's partials, EIGHT genomes per lane (round 4).  Its predecessor asked for
// four bytes per lane and range -- 256 bytes per wave-load, two half-used cache lines -- plus three 16-byte loads of the
// genomes' sizes per step: 2.75 load instructions per genome for a kernel that waits on loads three quarters of its time
// (20.7 ms per step of 100,000 queries x 100,000 genomes: 8 bytes of partials per (query, genome) = 80 GB at 3.9 TB/s).
// Here a lane takes 8 genomes: ONE 8-byte load per range (16 at 2-byte counters), the next step's words requested a step
// ahead as before, and the screen  score x genome_size / sketch_size >= bar  from ONE precomputed float per genome
// (ratio = genome_size / sketch_size, ratio_kernel) -- 1.25 load instructions per genome; the two sizes themselves are
// fetched only for the few genomes that pass the screen, where the decision is made in the reference's double operations
// (Miekki.cpp:382-383) exactly as above.  The bytes are what they are: 80 GB per step is 13 ms at the rate HBM streams.
template <int SRC>
__global__ __launch_bounds__(256) void select_ranges_kernel(const SelectArgs a, const float *__restrict__ ratio)
{
    constexpr uint32_t GPL = 8, STEP = 64 * GPL, PF = 8;
    using raw_t = typename std::conditional<SRC == 1, uint2, uint4>::type;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4u + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (q >= a.nq) return;
    const uint64_t range_stride = (uint64_t)a.nq * kTileBytes;
    const uint32_t n_active = a.nent[q];
    const uint32_t N = a.nresults;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    double topv = inf;
    uint32_t cnt = 0, emitted = 0;
    double minval = 0.0;
    float screen = 0.999f * (float)a.min_inter;
    mk_hit *__restrict__ out = a.cand ? a.cand + (uint64_t)q * a.cap : nullptr;
    uint64_t *__restrict__ crow = a.rows ? a.rows + (uint64_t)q * (a.cap + 1u) : nullptr;
    raw_t raw[PF];
    auto place = [&](uint32_t gl) -> const uint8_t * {
        const uint32_t t = gl / a.tile_genomes, wi = gl - t * a.tile_genomes;
        return a.partials + ((uint64_t)t * a.S * a.nq + q) * kTileBytes + (uint64_t)wi * SRC;
    };
    auto request = [&](uint32_t g0) {
        const uint32_t gl = g0 + lane * GPL;
#pragma unroll
        for (uint32_t r = 0; r < PF; ++r) raw[r] = raw_t{};
        if (gl >= a.G) return;
        const uint8_t *__restrict__ p = place(gl);
#pragma unroll
        for (uint32_t r = 0; r < PF; ++r)
            if (r < a.S) raw[r] = *reinterpret_cast<const raw_t *>(p + (uint64_t)r * range_stride);
    };
    request(0);
    for (uint32_t g0 = 0; g0 < a.G; g0 += STEP) {
        const uint32_t gl = g0 + lane * GPL;
        uint32_t ne[GPL];
        if constexpr (SRC == 1) {
            uint32_t e0 = 0, o0 = 0, e1 = 0, o1 = 0;               // byte-wise sums, two per 16-bit half: no carry between counters
#pragma unroll
            for (uint32_t r = 0; r < PF; ++r) {
                e0 += raw[r].x & 0x00ff00ffu; o0 += (raw[r].x >> 8) & 0x00ff00ffu;
                e1 += raw[r].y & 0x00ff00ffu; o1 += (raw[r].y >> 8) & 0x00ff00ffu;
            }
            ne[0] = e0 & 0xffffu; ne[1] = o0 & 0xffffu; ne[2] = e0 >> 16; ne[3] = o0 >> 16;
            ne[4] = e1 & 0xffffu; ne[5] = o1 & 0xffffu; ne[6] = e1 >> 16; ne[7] = o1 >> 16;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < GPL; ++j) ne[j] = 0;
#pragma unroll
            for (uint32_t r = 0; r < PF; ++r) {
                const uint32_t w[4] = {raw[r].x, raw[r].y, raw[r].z, raw[r].w};
#pragma unroll
                for (uint32_t d = 0; d < 4; ++d) { ne[2 * d] += w[d] & 0xffffu; ne[2 * d + 1] += w[d] >> 16; }
            }
        }
        if (a.S > PF && gl < a.G) {                                   // more ranges than are requested ahead (long-ish queries)
            const uint8_t *__restrict__ p = place(gl);
            for (uint32_t r = PF; r < a.S; ++r) {
                const raw_t w = *reinterpret_cast<const raw_t *>(p + (uint64_t)r * range_stride);
                if constexpr (SRC == 1) {
                    ne[0] += w.x & 0xffu; ne[1] += (w.x >> 8) & 0xffu; ne[2] += (w.x >> 16) & 0xffu; ne[3] += w.x >> 24;
                    ne[4] += w.y & 0xffu; ne[5] += (w.y >> 8) & 0xffu; ne[6] += (w.y >> 16) & 0xffu; ne[7] += w.y >> 24;
                } else {
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
                    for (uint32_t d = 0; d < 4; ++d) { ne[2 * d] += ww[d] & 0xffffu; ne[2 * d + 1] += ww[d] >> 16; }
                }
            }
        }
        uint32_t s[GPL], cand = 0;
        bool any = false;
#pragma unroll
        for (uint32_t j = 0; j < GPL; ++j) {
            s[j] = n_active - ne[j];
            any |= (gl + j < a.G) && s[j] >= a.min_score;             // Miekki.cpp:381
        }
        float rt[GPL];
        if (any) {                                                     // (the arrays are padded to whole tiles)
            const uint4 ra = *reinterpret_cast<const uint4 *>(ratio + gl), rb = *reinterpret_cast<const uint4 *>(ratio + gl + 4);
            rt[0] = __uint_as_float(ra.x); rt[1] = __uint_as_float(ra.y); rt[2] = __uint_as_float(ra.z); rt[3] = __uint_as_float(ra.w);
            rt[4] = __uint_as_float(rb.x); rt[5] = __uint_as_float(rb.y); rt[6] = __uint_as_float(rb.z); rt[7] = __uint_as_float(rb.w);
        }
        // the next step's words are requested AFTER this step's ratio loads: loads return in order
        __builtin_amdgcn_sched_barrier(0);
        if (g0 + STEP < a.G) request(g0 + STEP);
        __builtin_amdgcn_sched_barrier(0);
        if (any) {
#pragma unroll
            for (uint32_t j = 0; j < GPL; ++j)
                if (gl + j < a.G && s[j] >= a.min_score && !((float)s[j] * rt[j] < screen)) cand |= 1u << j;
        }
        double inter[GPL];
        uint32_t pot = 0;
#pragma unroll
        for (uint32_t j = 0; j < GPL; ++j) {
            inter[j] = 0.0;
            if ((cand >> j) & 1u) {                                    // past the screen: the reference's own operations
                const double jac = (double)s[j] / (double)a.sketch_size[gl + j];          // Miekki.cpp:382-383
                inter[j] = jac * (double)a.genome_size[gl + j];
                if (!(inter[j] < a.min_inter) && (cnt < N || !(minval > inter[j]))) pot |= 1u << j;
            }
        }
        uint64_t lanes = __ballot(pot != 0);
        while (lanes) {                                                // ascending genome order
            const uint32_t l = (uint32_t)__ffsll((long long)lanes) - 1u;
            lanes &= lanes - 1;
            const uint32_t pm = (uint32_t)__builtin_amdgcn_readlane((int)pot, (int)l);
#pragma unroll
            for (uint32_t j = 0; j < GPL; ++j) {
                if (!((pm >> j) & 1u)) continue;
                const double x = readlane_f64(inter[j], l);
                if (cnt >= N && minval > x) continue;                   // Miekki.cpp:387: skipped, heap untouched
                if (N == 0) continue;
                if (lane == l && emitted < a.cap) {
                    if (crow) {
                        crow[1u + emitted] = (uint64_t)(gl + j + a.genome_id_base) | ((uint64_t)s[j] << 32);
                    } else {
                        mk_hit h;
                        h.genome = gl + j + a.genome_id_base;
                        h.matches = s[j];
                        h.jaccard = (double)s[j] / (double)a.sketch_size[gl + j];
                        h.intersection = inter[j];
                        out[emitted] = h;
                    }
                }
                ++emitted;
                if (cnt < N) {
                    if (lane == cnt) topv = x;
                    ++cnt;
                } else {                                                // evict one holder of the minimum
                    const uint64_t holders = __ballot(lane < N && topv == minval);
                    const uint32_t victim = holders ? (uint32_t)__ffsll((long long)holders) - 1u : 0u;
                    if (lane == victim) topv = x;
                }
                if (cnt == N) {
                    minval = wave_min_f64(lane < N ? topv : inf);
                    const double bar = minval > a.min_inter ? minval : a.min_inter;
                    screen = 0.999f * (float)bar;
                }
            }
        }
    }
    if (lane == 0) {
        if (crow) crow[0] = emitted;
        else a.count[q] = emitted;
    }
}

// ratio[g] = genome_size / sketch_size as a float: the screen's one load per genome (padding: 0)
__global__ void ratio_kernel(const uint32_t *__restrict__ ss, const uint64_t *__restrict__ gs, uint32_t G, uint32_t padded, float *__restrict__ ratio)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < padded) ratio[g] = g < G ? (float)gs[g] / (float)ss[g] : 0.0f;
}

int launch_ratio(mk_ctx *c, float *d_ratio, uint32_t padded)
{
    if (!padded) return MK_OK;
    hipLaunchKernelGGL(ratio_kernel, dim3((padded + 255) / 256), dim3(256), 0, c->stream, c->d_sketch_size, c->d_genome_size, c->G, padded, d_ratio);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_select(mk_ctx *c, const SelectArgs &a)
{
    if (!a.nq) return MK_OK;
    if (a.nresults > kSelectMaxResults) { set_error("device selection supports nresults <= 64"); return MK_ERR_ARG; }
    const dim3 grid((a.nq + 3) / 4), block(256);
    if (a.partials) {                                                // the slab schedule's per-range counts (S <= 64 by construction)
        if (!a.ratio || a.S > 64) { set_error("selection over partial counts needs the ratio array and at most 64 ranges"); return MK_ERR_ARG; }
        if (a.W == 1) hipLaunchKernelGGL(select_ranges_kernel<1>, grid, block, 0, c->stream, a, a.ratio);
        else hipLaunchKernelGGL(select_ranges_kernel<2>, grid, block, 0, c->stream, a, a.ratio);
    } else {
        hipLaunchKernelGGL(select_kernel, grid, block, 0, c->stream, a);
    }
    MK_HIP(hipGetLastError());
    return MK_OK;
}

}  // namespace mk
