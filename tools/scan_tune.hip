// scan_tune -- A/B timing of scan_kernel variants, interleaved rounds in ONE process
// (cdna_hip_programming.md rule 24).  Synthetic matrix (random bytes) and random
// sorted entry lists of realistic shape; timing only, plus a cross-variant checksum
// of the first score rows.  Build: make -C miekki_amd/csrc tune
//   scan_tune <G> <h> <Q> <entries_per_query> <rounds>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../miekki_amd/csrc/scan_kernel.hpp"

namespace mk { void set_error(const char *, ...) {} }
using namespace mk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_kernel(uint4 *p, uint64_t n16, uint64_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = splitmix64(seed ^ (2 * i)), b = splitmix64(seed ^ (2 * i + 1));
        // skew towards a few values like real fingerprints: keep the high nibble in 0xC..0xF
        const uint64_t m = 0x3f3f3f3f3f3f3f3fULL, o = 0xc0c0c0c0c0c0c0c0ULL;
        const uint64_t x = (a & m) | o, y = (b & m) | o;
        p[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
    }
}


// ---- experiment: narrower row tiles (LB bytes per lane instead of 16) so that the
// tile-major column slab (2^h x 64*LB bytes) fits the Infinity Cache; store-less.
template <int LB, int UNROLL>
__global__ __launch_bounds__(256) void narrow_kernel(const ScanArgs a)
{
    constexpr uint32_t ND = LB / 4, TB = 64 * LB;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t ntiles = (a.G + TB - 1) / TB;
    const uint32_t work = blockIdx.x * 4u + wave;
    if (work >= a.nq * ntiles) return;
    const uint32_t tile = work / a.nq, ql = work - tile * a.nq;
    if ((uint64_t)tile * TB + lane * LB >= a.G) return;
    const uint64_t *__restrict__ ent = a.entries + a.ent_off[ql];
    const uint32_t n = a.nent[ql];
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * TB;
    const uint32_t voff = lane * LB;
    uint32_t tot[ND];
    for (uint32_t k = 0; k < ND; ++k) tot[k] = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 255) {
        const uint32_t m = min(n - i0, 255u);
        const uint64_t *__restrict__ e = ent + i0;
        uint32_t acc[ND];
        for (uint32_t k = 0; k < ND; ++k) acc[k] = 0;
        uint32_t j = 0;
        for (; j + UNROLL <= m; j += UNROLL) {
            uint64_t ev[UNROLL];
            uint32_t d[UNROLL][ND];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) ev[u] = e[j + u];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint8_t *p = row_base(base, (uint32_t)ev[u], a.ld) + voff;
                if (ND == 1) d[u][0] = *reinterpret_cast<const uint32_t *>(p);
                else if (ND == 2) { const uint2 t = *reinterpret_cast<const uint2 *>(p); d[u][0] = t.x; d[u][1] = t.y; }
                else { const uint4 t = *reinterpret_cast<const uint4 *>(p); d[u][0] = t.x; d[u][1] = t.y; d[u][2 % ND] = t.z; d[u][3 % ND] = t.w; }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t b = bcast_fp<1>((uint32_t)(ev[u] >> 32));
#pragma unroll
                for (uint32_t k = 0; k < ND; ++k) acc[k] += ne_lanes<1>(d[u][k], b);
            }
        }
        for (; j < m; ++j) {
            const uint64_t ev = e[j];
            const uint8_t *p = row_base(base, (uint32_t)ev, a.ld) + voff;
            const uint32_t b = bcast_fp<1>((uint32_t)(ev >> 32));
            for (uint32_t k = 0; k < ND; ++k) acc[k] += ne_lanes<1>(reinterpret_cast<const uint32_t *>(p)[k], b);
        }
        for (uint32_t k = 0; k < ND; ++k) tot[k] += (acc[k] & 0xff) + ((acc[k] >> 8) & 0xff) + ((acc[k] >> 16) & 0xff) + (acc[k] >> 24);
    }
    for (uint32_t k = 0; k < ND; ++k) asm volatile("" ::"v"(tot[k]));
}

// ---- experiment (rejected)
// Phase-locked sweep: the grid holds only as many waves as the chip keeps resident
// and every wave walks the tile-major work list with the grid's stride.  All items
// cost the same (one query's sorted entry list), so the resident waves start item k
// together and sweep the partitions 0 -> 2^h side by side: the rows they touch at any
// moment lie in a narrow band of the column slab, which the Infinity Cache holds.
template <int W, int UNROLL, bool NT = false>
__global__ __launch_bounds__(256) void scan_sweep_kernel(const ScanArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total = a.nq * a.ntiles, stride = gridDim.x * 4u;
    for (uint32_t work = blockIdx.x * 4u + wave; work < total; work += stride) {
        const uint32_t tile = work / a.nq, ql = work - tile * a.nq;
        scan_item<W, UNROLL, NT>(a, ql, tile, lane);
    }
}


// ---- experiment: the query's entry list staged in LDS (the north-star's suggestion)
// instead of SGPRs.  A workgroup = one query x four adjacent tiles (query-major), the
// list is loaded cooperatively (coalesced) into LDS once per workgroup, every wave
// then reads entry i as a broadcast ds_read_b64 and lifts it to SGPRs.
__global__ __launch_bounds__(256) void lds_kernel(const ScanArgs a)
{
    extern __shared__ uint64_t s_ent[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tiles4 = (a.ntiles + 3) / 4;
    const uint32_t ql = blockIdx.x / tiles4, tile = (blockIdx.x - ql * tiles4) * 4 + wave;
    const uint64_t *__restrict__ ent = a.entries + a.ent_off[ql];
    const uint32_t n = a.nent[ql];
    for (uint32_t i = threadIdx.x; i < n; i += 256) s_ent[i] = ent[i];
    __syncthreads();
    if (tile >= a.ntiles || (uint64_t)tile * kTileBytes + lane * 16u >= a.G) return;
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile * kTileBytes;
    const uint32_t voff = lane * 16u;
    uint32_t tot[4] = {0, 0, 0, 0};
    for (uint32_t i0 = 0; i0 < n; i0 += 248) {
        const uint32_t m = min(n - i0, 248u);
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, j = 0;
        for (; j + 8 <= m; j += 8) {
            uint64_t ev[8]; uint4 d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint64_t v = s_ent[i0 + j + u];                      // same address in every lane: broadcast
                ev[u] = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = load_row16<false>(row_base(base, (uint32_t)ev[u], a.ld) + voff);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t b = bcast_fp<1>((uint32_t)(ev[u] >> 32));
                acc0 += ne_lanes<1>(d[u].x, b); acc1 += ne_lanes<1>(d[u].y, b);
                acc2 += ne_lanes<1>(d[u].z, b); acc3 += ne_lanes<1>(d[u].w, b);
            }
        }
        for (; j < m; ++j) {
            const uint64_t v = s_ent[i0 + j];
            const uint64_t ev = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)v);
            const uint4 d = load_row16<false>(row_base(base, (uint32_t)ev, a.ld) + voff);
            const uint32_t b = bcast_fp<1>((uint32_t)(ev >> 32));
            acc0 += ne_lanes<1>(d.x, b); acc1 += ne_lanes<1>(d.y, b); acc2 += ne_lanes<1>(d.z, b); acc3 += ne_lanes<1>(d.w, b);
        }
        tot[0] += acc0; tot[1] += acc1; tot[2] += acc2; tot[3] += acc3;   // timing build: no widening
    }
    for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(tot[k]));
}

// ---- the box's read-only stream ceiling (SURVEY.md 8d asks for it): every lane sums
// 16-byte loads over the whole matrix buffer, grid-stride, 8 loads in flight
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4 *__restrict__ p, uint64_t n16, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;                       // keeps the loads live
}

// ---- experiment: one wave owns TWO adjacent row tiles of a (query, range): the entry list is walked once per
// 2 KiB of row instead of once per 1 KiB (half the scalar work and half the waves per byte), 2 x UNROLL loads
// in flight per wave
template <int UNROLL>
__global__ __launch_bounds__(256) void scan_slab2_kernel(const SlabArgs a)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t work = blockIdx.x * 4u + wave;
    const uint32_t npairs = (a.ntiles + 1) / 2;
    if (work >= npairs * a.S * a.nq) return;
    const uint32_t tr = work / a.nq, ql = work - tr * a.nq;
    const uint32_t pair = tr / a.S, r = tr - pair * a.S;
    const uint32_t tile0 = 2 * pair, tile1 = 2 * pair + 1;
    const bool on0 = (uint64_t)tile0 * kTileBytes + lane * 16u < (uint64_t)a.G;
    const bool on1 = tile1 < a.ntiles && (uint64_t)tile1 * kTileBytes + lane * 16u < (uint64_t)a.G;
    if (!on0) return;
    const uint32_t q = a.q_begin + ql;
    const uint32_t lo = a.split[(uint64_t)q * (a.S + 1) + r], hi = a.split[(uint64_t)q * (a.S + 1) + r + 1];
    const uint64_t *__restrict__ e = a.entries + a.ent_off[q] + lo;
    const uint32_t m = hi - lo;
    const uint8_t *__restrict__ base = a.M + (uint64_t)tile0 * kTileBytes;
    const uint32_t voff = lane * 16u, voff1 = on1 ? voff + kTileBytes : voff;
    const uint64_t ld = a.ld;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t j = 0; j < m; j += UNROLL) {
        uint64_t ev[UNROLL];
        uint4 d0[UNROLL], d1[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) ev[u] = e[min(j + u, m - 1)];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint8_t *rb = row_base(base, (uint32_t)ev[u], ld);
            d0[u] = load_row16<false>(rb + voff);
            d1[u] = load_row16<false>(rb + voff1);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t keep = j + u < m ? 0xffffffffu : 0u;
            const uint32_t b = bcast_fp<1>((uint32_t)(ev[u] >> 32));
            acc[0] += ne_lanes<1>(d0[u].x, b) & keep; acc[1] += ne_lanes<1>(d0[u].y, b) & keep;
            acc[2] += ne_lanes<1>(d0[u].z, b) & keep; acc[3] += ne_lanes<1>(d0[u].w, b) & keep;
            acc[4] += ne_lanes<1>(d1[u].x, b) & keep; acc[5] += ne_lanes<1>(d1[u].y, b) & keep;
            acc[6] += ne_lanes<1>(d1[u].z, b) & keep; acc[7] += ne_lanes<1>(d1[u].w, b) & keep;
        }
    }
    uint8_t *__restrict__ out0 = a.partials + (((uint64_t)tile0 * a.S + r) * a.nq + ql) * kTileBytes + voff;
    *reinterpret_cast<uint4 *>(out0) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    if (on1) {
        uint8_t *__restrict__ out1 = a.partials + (((uint64_t)tile1 * a.S + r) * a.nq + ql) * kTileBytes + voff;
        *reinterpret_cast<uint4 *>(out1) = make_uint4(acc[4], acc[5], acc[6], acc[7]);
    }
}

struct Variant { const char *name; void (*fn)(const ScanArgs); uint32_t blocks_per_cu; uint32_t lane_bytes; };   // blocks_per_cu > 0: sweep grid

#define V(U, O, N) {"u" #U "_o" #O "_nt" #N, scan_kernel<1, U, O, N>, 0, 16}
#define S(B) {"sweep_" #B "perCU", scan_sweep_kernel<1, 8, false>, B, 16}
#define N(LB, U) {"narrow_" #LB "B_u" #U, narrow_kernel<LB, U>, 0, LB}

int main(int argc, char **argv)
{
    const uint32_t G = argc > 1 ? atoi(argv[1]) : 12500;
    const uint32_t h = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t Q = argc > 3 ? atoi(argv[3]) : 20000;
    const uint32_t NE = argc > 4 ? atoi(argv[4]) : 908;
    const int rounds = argc > 5 ? atoi(argv[5]) : 5;
    const uint32_t range_div = argc > 6 ? atoi(argv[6]) : 1;      // draw partitions from [0, P / range_div): slab experiment
    const uint32_t slabS = argc > 7 ? atoi(argv[7]) : 8;           // ranges of the slab schedule
    const uint32_t P = 1u << h;
    const uint64_t ld = ((uint64_t)G + kTileBytes - 1) / kTileBytes * kTileBytes;
    uint8_t *M; CK(hipMalloc((void **)&M, (uint64_t)P * ld));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, (uint4 *)M, (uint64_t)P * ld / 16, 0x1234567ULL);
    std::vector<uint64_t> ent((size_t)Q * NE), off(Q + 1);
    std::vector<uint32_t> nent(Q, NE);
    std::mt19937_64 rng(42);
    for (uint32_t q = 0; q < Q; ++q) {
        off[q] = (uint64_t)q * NE;
        std::vector<uint32_t> ps(NE);
        for (auto &p : ps) p = (uint32_t)(rng() % (P / range_div));
        std::sort(ps.begin(), ps.end());
        for (uint32_t i = 0; i < NE; ++i) ent[(size_t)q * NE + i] = make_entry(ps[i], 0xC0u | (uint32_t)(rng() & 0x3f));
    }
    off[Q] = (uint64_t)Q * NE;
    uint64_t *d_ent, *d_off; uint32_t *d_nent, *d_scores;
    const uint32_t sld = (G + kTileBytes - 1) / kTileBytes * kTileBytes;
    CK(hipMalloc((void **)&d_ent, ent.size() * 8)); CK(hipMalloc((void **)&d_off, off.size() * 8));
    CK(hipMalloc((void **)&d_nent, Q * 4)); CK(hipMalloc((void **)&d_scores, (uint64_t)Q * sld * 4));
    CK(hipMemcpy(d_ent, ent.data(), ent.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nent, nent.data(), Q * 4, hipMemcpyHostToDevice));
    ScanArgs a{};
    a.M = M; a.Mc = nullptr; a.P_hot = 0xffffffffu; a.ld = ld; a.G = G; a.ntiles = (uint32_t)(ld / kTileBytes); a.nq = Q; a.q_begin = 0;
    a.ntiles = (G + kTileBytes - 1) / kTileBytes;
    a.entries = d_ent; a.ent_off = d_off; a.nent = d_nent; a.scores = d_scores;
    const bool tile_major_scores = getenv("SCORES_ROWS") == nullptr;
    if (getenv("NO_STORE")) a.scores = nullptr;
    const ScoreLayout lay = tile_major_scores ? score_layout_tiles(1, Q) : score_layout_rows(1, sld, G);
    a.score_tile_stride = lay.tile_stride; a.score_q_stride = lay.q_stride; a.score_vec = lay.vec;
    const Variant vars[] = {V(8, 0, false), V(8, 1, false), {"lds_staged_o0", lds_kernel, 0, 16}, N(8, 8), N(4, 8)};
    const int nv = sizeof vars / sizeof vars[0];
    const uint64_t work = (uint64_t)Q * a.ntiles;
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    const double algo = (double)Q * NE * G + 4.0 * Q * G;
    std::vector<std::vector<float>> ms(nv);
    std::vector<uint64_t> chk(nv, 0);
    // slab schedule inputs: per query the entry index of every range boundary
    std::vector<uint32_t> split((size_t)Q * (slabS + 1));
    for (uint32_t q = 0; q < Q; ++q)
        for (uint32_t r = 0; r <= slabS; ++r) {
            const uint64_t bound = (uint64_t)r * (P / slabS);
            uint32_t i = 0;
            while (i < NE && (uint32_t)ent[(size_t)q * NE + i] < bound) ++i;
            split[(size_t)q * (slabS + 1) + r] = r == slabS ? NE : i;
        }
    uint32_t *d_split; uint8_t *d_part;
    CK(hipMalloc((void **)&d_split, split.size() * 4));
    CK(hipMemcpy(d_split, split.data(), split.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_part, (uint64_t)a.ntiles * slabS * Q * kTileBytes));
    SlabArgs sa;
    sa.chunk = 0; sa.nent = nullptr; sa.Mc = nullptr; sa.P_hot = 0xffffffffu; sa.r_begin = 0;
    sa.M = M; sa.ld = ld; sa.G = G; sa.ntiles = a.ntiles; sa.nq = Q; sa.q_begin = 0; sa.S = slabS; sa.r_count = slabS;
    sa.entries = d_ent; sa.ent_off = d_off; sa.split = d_split; sa.partials = d_part;
    std::vector<float> slab_ms;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < nv; ++v) {
            CK(hipEventRecord(e0, 0));
            uint32_t grid = vars[v].blocks_per_cu ? std::min(blocks, 256u * vars[v].blocks_per_cu) : blocks;
            if (vars[v].lane_bytes != 16 || vars[v].fn == (void (*)(const ScanArgs))narrow_kernel<16, 8>) {
                const uint32_t tb = 64 * vars[v].lane_bytes;
                grid = (uint32_t)(((uint64_t)Q * ((G + tb - 1) / tb) + 3) / 4);
            }
            size_t lds = 0;
            if (vars[v].fn == lds_kernel) { grid = Q * ((a.ntiles + 3) / 4); lds = (size_t)NE * 8; }
            hipLaunchKernelGGL(vars[v].fn, dim3(grid), dim3(256), lds, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r) ms[v].push_back(t);                     // round 0 = warm-up
            else {
                if (!a.scores) continue;
                std::vector<uint32_t> c((size_t)Q * sld);                 // checksum of all scores
                CK(hipMemcpy(c.data(), d_scores, c.size() * 4, hipMemcpyDeviceToHost));
                for (uint32_t r = 0; r < Q; ++r)
                    for (uint32_t g = 0; g < G; ++g) {
                        const uint32_t t = g / kTileBytes, w = g % kTileBytes;
                        chk[v] += c[(size_t)t * lay.tile_stride + (size_t)r * lay.q_stride + w];
                    }
            }
        }
    for (int r = 0; r < rounds + 1; ++r) {
        const uint64_t sw = (uint64_t)a.ntiles * slabS * Q;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_slab_kernel<1, 8>), dim3((uint32_t)((sw + 3) / 4)), dim3(256), 0, 0, sa);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r) slab_ms.push_back(t);
    }
    std::vector<float> slab2_ms[2];
    for (int r = 0; r < rounds + 1; ++r)
        for (int v = 0; v < 2; ++v) {
            const uint64_t sw = (uint64_t)((a.ntiles + 1) / 2) * slabS * Q;
            CK(hipEventRecord(e0, 0));
            if (v == 0) hipLaunchKernelGGL((scan_slab2_kernel<4>), dim3((uint32_t)((sw + 3) / 4)), dim3(256), 0, 0, sa);
            else        hipLaunchKernelGGL((scan_slab2_kernel<8>), dim3((uint32_t)((sw + 3) / 4)), dim3(256), 0, 0, sa);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r) slab2_ms[v].push_back(t);
        }
    for (int v = 0; v < 2; ++v) {
        std::sort(slab2_ms[v].begin(), slab2_ms[v].end());
        printf("slab2_u%d       median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)  [two tiles per wave]\n", v ? 8 : 4,
               slab2_ms[v][slab2_ms[v].size() / 2], slab2_ms[v][0], algo / slab2_ms[v][slab2_ms[v].size() / 2] / 1e6);
    }
    // check the slab partials against the plain kernel's scores for the first queries
    if (a.scores) {
        hipLaunchKernelGGL((scan_kernel<1, 8, 1, false>), dim3(blocks), dim3(256), 0, 0, a);
        std::vector<uint32_t> sc((size_t)Q * sld);
        CK(hipMemcpy(sc.data(), d_scores, sc.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> pt((size_t)a.ntiles * slabS * Q * kTileBytes);
        CK(hipMemcpy(pt.data(), d_part, pt.size(), hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (uint32_t q = 0; q < std::min<uint32_t>(Q, 50); ++q)
            for (uint32_t g = 0; g < G; ++g) {
                const uint32_t t = g / kTileBytes, w = g % kTileBytes;
                uint32_t ne = 0;
                for (uint32_t r = 0; r < slabS; ++r) ne += pt[(((size_t)t * slabS + r) * Q + q) * kTileBytes + w];
                bad += (NE - ne) != sc[(size_t)t * lay.tile_stride + (size_t)q * lay.q_stride + w];
            }
        printf("slab vs plain mismatches (first 50 queries): %llu\n", (unsigned long long)bad);
    }
    printf("range 1/%u  scores %s  G=%u h=%u Q=%u entries=%u  M=%.1f GB  algorithmic %.1f GB per launch\n", range_div, tile_major_scores ? "tile-major" : "row-major", G, h, Q, NE, P * (double)ld / 1e9, algo / 1e9);
    for (int v = 0; v < nv; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const float med = ms[v][ms[v].size() / 2], mn = ms[v][0];
        printf("%-14s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)  %7.1f (min)  chk %llu\n", vars[v].name, med, mn,
               algo / med / 1e6, algo / mn / 1e6, (unsigned long long)chk[v]);
    }
    {   // read-only stream over the same buffer
        uint32_t *d_sink; CK(hipMalloc((void **)&d_sink, 4));
        const uint64_t n16 = (uint64_t)P * ld / 16;
        std::vector<float> t;
        for (int r = 0; r < rounds + 1; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, 0, (const uint4 *)M, n16, d_sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("%-14s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)  [read-only stream of the %.1f GB matrix]\n", "stream_read",
               t[t.size() / 2], t[0], (double)P * ld / t[t.size() / 2] / 1e6, (double)P * ld / 1e9);
    }
    std::sort(slab_ms.begin(), slab_ms.end());
    printf("%-14s median %8.3f ms  min %8.3f ms  -> %7.1f GB/s (median)  [S=%u, partials %.1f GB]\n", "slab", slab_ms[slab_ms.size() / 2],
           slab_ms[0], algo / slab_ms[slab_ms.size() / 2] / 1e6, slabS, (double)a.ntiles * slabS * Q * kTileBytes / 1e9);
    return 0;
}
