// K5: query-vs-all-genomes fingerprint scan -- launchers.  The kernels are in
// scan_kernel.hpp (shared with tools/scan_tune.hip); DESIGN.md section 4.1 has the
// measurements behind each choice.
//
// Replaces the loop nest of Miekki::query_sequences (Miekki.cpp:355-369); the
// filter of Miekki::filter_results runs over the scores in select.hip (K6).
//
// One WAVE owns one (query, 1 KiB row tile) pair -- 1024 genomes at 1-byte
// fingerprints, 512 at 2-byte -- so that tiny collections (G = 1000 is one tile) and
// huge ones (G = 100,000 is 98 tiles) both fill the 256 CUs.  The query's sparse
// sketch is wave-uniform: it is staged in SGPRs through the scalar cache, the row base
// is scalar arithmetic, and every lane issues one 16-byte load per entry from
// M[p][tile*1024 + lane*16].  Fingerprints are compared four (two) at a time with
// SWAR zero-byte detection on the XOR against the broadcast query fingerprint, into
// packed 8-bit (16-bit) per-genome counters.  No atomics, no LDS: a pure row stream
// whose speed is decided by the ORDER of the work items:
//   scan_slab_kernel   (tile, partition range, query): the waves in flight share a
//                      128 MiB slab that the Infinity Cache holds -- default pipeline;
//   scan_kernel        tile-major over whole entry lists, u32 scores -- long queries,
//                      score export;
//   scan_dense_kernel  four whole-genome queries per pass over the matrix (-A).
#include "scan_kernel.hpp"

namespace mk {

template <int W>
static int launch_scan_t(mk_ctx *c, const ScanArgs &a)
{
    const uint64_t work = (uint64_t)a.nq * a.ntiles;
    if (work == 0) return MK_OK;
    if (work >= (1ull << 31)) { set_error("scan launch too large"); return MK_ERR_ARG; }
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    if (a.windowed) hipLaunchKernelGGL((scan_kernel<W, 8, 1, false, true>), dim3(blocks), dim3(256), 0, c->stream, a);
    else            hipLaunchKernelGGL((scan_kernel<W, 8, 1, false, false>), dim3(blocks), dim3(256), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_scan(mk_ctx *c, const ScanArgs &a)
{
    return c->W == 1 ? launch_scan_t<1>(c, a) : launch_scan_t<2>(c, a);
}

int launch_scan_slab(mk_ctx *c, const SlabArgs &a)
{
    const uint64_t work = (uint64_t)a.ntiles * a.r_count * a.nq;
    if (work == 0) return MK_OK;
    if (work >= (1ull << 31)) { set_error("scan launch too large"); return MK_ERR_ARG; }
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    if (c->W == 1) hipLaunchKernelGGL((scan_slab_kernel<1, 8>), dim3(blocks), dim3(256), 0, c->stream, a);
    else           hipLaunchKernelGGL((scan_slab_kernel<2, 8>), dim3(blocks), dim3(256), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_dense_lut(mk_ctx *c, const uint8_t *d_dense, uint32_t ngroups, DenseLut *d_lut)
{
    if (!ngroups) return MK_OK;
    if (c->W == 1) hipLaunchKernelGGL(dense_lut_kernel<1>, dim3((c->P + 255) / 256, (ngroups + 1) / 2), dim3(256), 0, c->stream, d_dense, ngroups, c->P, c->empty, d_lut);
    else           hipLaunchKernelGGL(dense_lut_kernel<2>, dim3((c->P + 255) / 256, (ngroups + 1) / 2), dim3(256), 0, c->stream, d_dense, ngroups, c->P, c->empty, d_lut);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_scan_dense(mk_ctx *c, const DenseArgs &a)
{
    // by table, sixteen queries per pass over the matrix (eight when there are no more)
    // (the table kernel walks rows in groups of sixteen: every window and chunk of rows is a multiple of that for P >= 16;
    // sketches of fewer partitions take the compare kernel below)
    if (a.lut && a.rows_per_item % 16 == 0 && (a.row_hi - a.row_lo) % 16 == 0) {
        // (sixteen queries per wave: eight -- a third more instructions per query for a third wave per SIMD -- measured 97 against
        // 79 ms at 64 queries, profiles/r6_pmc_dense.txt)
        const uint32_t no = a.noctets >= 2 ? 2 : 1;
        // the sets of sixteen (eight) queries share a tile's rows through LDS, two or four waves of a workgroup (scan_kernel.hpp)
        const uint32_t nsets = (a.noctets + no - 1) / no, gs = dense_share(a.noctets);
        const uint64_t groups = (uint64_t)((nsets + gs - 1) / gs) * a.ntiles * a.nchunks;
        if (groups == 0) return MK_OK;
        if (groups >= (1ull << 31)) { set_error("dense scan launch too large"); return MK_ERR_ARG; }
        const uint32_t per_wg = 4 / gs, blocks = ((uint32_t)((groups + per_wg - 1) / per_wg) + 7u) / 8u * 8u;   // (a multiple of eight: the kernel deals them to the XCDs)
#define MK_DENSE_LUT(W_, NO_, GS_) hipLaunchKernelGGL((scan_dense_lut_kernel<W_, NO_, GS_>), dim3(blocks), dim3(256), 0, c->stream, a)
        if (c->W == 1) {
            if (no == 1 && gs == 1) MK_DENSE_LUT(1, 1, 1);
            else if (no == 1 && gs == 2) MK_DENSE_LUT(1, 1, 2);
            else if (no == 1) MK_DENSE_LUT(1, 1, 4);
            else if (gs == 1) MK_DENSE_LUT(1, 2, 1);
            else if (gs == 2) MK_DENSE_LUT(1, 2, 2);
            else MK_DENSE_LUT(1, 2, 4);
        } else {
            if (no == 1 && gs == 1) MK_DENSE_LUT(2, 1, 1);
            else if (no == 1 && gs == 2) MK_DENSE_LUT(2, 1, 2);
            else if (no == 1) MK_DENSE_LUT(2, 1, 4);
            else if (gs == 1) MK_DENSE_LUT(2, 2, 1);
            else if (gs == 2) MK_DENSE_LUT(2, 2, 2);
            else MK_DENSE_LUT(2, 2, 4);
        }
#undef MK_DENSE_LUT
        MK_HIP(hipGetLastError());
        return MK_OK;
    }
    // eight queries per pass over the matrix when there are that many, else four
    const uint32_t gb = a.ngroups >= 2 ? 2 : 1;
    const uint64_t work = (uint64_t)((a.ngroups + gb - 1) / gb) * a.ntiles * a.nchunks;
    if (work == 0) return MK_OK;
    if (work >= (1ull << 31)) { set_error("dense scan launch too large"); return MK_ERR_ARG; }
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    if (c->W == 1) {
        if (gb == 2) hipLaunchKernelGGL((scan_dense_kernel<1, 2>), dim3(blocks), dim3(256), 0, c->stream, a);
        else         hipLaunchKernelGGL((scan_dense_kernel<1, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
    } else {
        if (gb == 2) hipLaunchKernelGGL((scan_dense_kernel<2, 2>), dim3(blocks), dim3(256), 0, c->stream, a);
        else         hipLaunchKernelGGL((scan_dense_kernel<2, 1>), dim3(blocks), dim3(256), 0, c->stream, a);
    }
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// ---- the box's read-only stream ceiling (SURVEY.md 8d asks for it next to the spec peak):
// every lane sums 16-byte loads over the resident matrix buffer, grid-stride, eight loads in
// flight.  Measurement aid behind mk_probe_stream_read; nothing on the query path calls it.
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

int probe_stream_read(mk_ctx *c, uint32_t rounds, double *gbps, uint64_t *bytes)
{
    const uint64_t total = (uint64_t)c->P_hot * c->ld;             // the rows resident in HBM
    *gbps = 0; *bytes = total;
    if (!c->d_M || total < (1ull << 20)) return MK_OK;
    if (!c->d_flag) MK_HIP(hipMalloc((void **)&c->d_flag, 4));
    hipEvent_t e0, e1;
    MK_HIP(hipEventCreate(&e0)); MK_HIP(hipEventCreate(&e1));
    float best = 0;
    for (uint32_t r = 0; r <= rounds; ++r) {                    // first round warms up
        MK_HIP(hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, c->stream, (const uint4 *)c->d_M, total / 16,
                           c->d_flag);
        MK_HIP(hipEventRecord(e1, c->stream));
        MK_HIP(hipEventSynchronize(e1));
        float ms = 0;
        MK_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r && (best == 0 || ms < best)) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (best > 0) *gbps = (double)total / best / 1e6;
    return MK_OK;
}

}  // namespace mk
