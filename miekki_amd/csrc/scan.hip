// K5: query-vs-all-genomes fingerprint scan, with the threshold filter of
// filter_results fused into its epilogue (K6).
//
// Replaces the loop nest of Miekki::query_sequences (Miekki.cpp:355-369) and the
// per-genome tests of Miekki::filter_results (Miekki.cpp:379-384).
//
// Work decomposition (gfx950): one WAVE owns one (query, 1 KiB row tile) pair --
// 1024 genomes at 1-byte fingerprints, 512 at 2-byte -- so that tiny collections
// (G = 1000 is one tile) and huge ones (G = 100,000 is 98 tiles) both fill the
// 256 CUs.  The query's sparse sketch (partition, fingerprint) list is wave-uniform:
// it is staged in SGPRs through the scalar cache, the per-row base address is
// scalar arithmetic, and every lane issues one 16-byte load per entry from
// M[p][tile*1024 + lane*16] -- 1 KiB contiguous per wave-instruction.  Work items
// are ordered TILE-major (all queries of tile 0, then tile 1, ...): the waves in
// flight at any moment then share one 2^h x 1 KiB column slab of the matrix, a
// quarter of which stays resident in the 256 MiB Infinity Cache at h = 20, which
// measured +16 % over query-major order (tools/scan_tune, profiles/).  Fingerprints are compared
// four (two) at a time with SWAR zero-byte detection on the XOR against the
// broadcast query fingerprint; per-genome counters live in packed 8-bit (16-bit)
// register lanes and are widened every 255 (65535) entries.  No atomics in the
// main loop, no LDS: the kernel is a pure HBM row-stream.
#include "scan_kernel.hpp"

namespace mk {

template <int W, bool FILTER>
static int launch_scan_t(mk_ctx *c, const ScanArgs &a)
{
    const uint64_t work = (uint64_t)a.nq * a.ntiles;
    if (work == 0) return MK_OK;
    if (work >= (1ull << 31)) { set_error("scan launch too large"); return MK_ERR_ARG; }
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    hipLaunchKernelGGL((scan_kernel<W, 8, FILTER, 1, false>), dim3(blocks), dim3(256), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_scan(mk_ctx *c, const ScanArgs &a, bool filter)
{
    if (c->W == 1) return filter ? launch_scan_t<1, true>(c, a) : launch_scan_t<1, false>(c, a);
    return filter ? launch_scan_t<2, true>(c, a) : launch_scan_t<2, false>(c, a);
}

// Put every query's candidate row in ascending genome order (the order in which
// filter_results meets them, Miekki.cpp:379).  One wave per query; rank sort.
__global__ __launch_bounds__(64) void sort_candidates_kernel(uint32_t nq, uint32_t cap,
                                                             const uint32_t *__restrict__ count,
                                                             mk_hit *__restrict__ cand)
{
    extern __shared__ __align__(16) unsigned char smem[];
    mk_hit *tmp = reinterpret_cast<mk_hit *>(smem);
    const uint32_t q = blockIdx.x;
    if (q >= nq) return;
    const uint32_t n = min(count[q], cap);
    if (n < 2) return;
    mk_hit *row = cand + (uint64_t)q * cap;
    for (uint32_t i = threadIdx.x; i < n; i += 64) tmp[i] = row[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        const uint32_t g = tmp[i].genome;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += tmp[j].genome < g;
        row[rank] = tmp[i];
    }
}

int launch_sort_candidates(mk_ctx *c, uint32_t nq, uint32_t cap, const uint32_t *d_count, mk_hit *d_cand)
{
    if (nq == 0 || cap < 2) return MK_OK;
    const size_t lds = (size_t)cap * sizeof(mk_hit);
    if (lds > 64 * 1024) { set_error("candidate cap too large"); return MK_ERR_ARG; }
    hipLaunchKernelGGL(sort_candidates_kernel, dim3(nq), dim3(64), lds, c->stream, nq, cap, d_count, d_cand);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

}  // namespace mk
