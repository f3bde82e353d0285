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
    hipLaunchKernelGGL((scan_kernel<W, 8, 1, false>), dim3(blocks), dim3(256), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_scan(mk_ctx *c, const ScanArgs &a)
{
    return c->W == 1 ? launch_scan_t<1>(c, a) : launch_scan_t<2>(c, a);
}

int launch_scan_slab(mk_ctx *c, const SlabArgs &a)
{
    const uint64_t work = (uint64_t)a.ntiles * a.S * a.nq;
    if (work == 0) return MK_OK;
    if (work >= (1ull << 31)) { set_error("scan launch too large"); return MK_ERR_ARG; }
    const uint32_t blocks = (uint32_t)((work + 3) / 4);
    if (c->W == 1) hipLaunchKernelGGL((scan_slab_kernel<1, 8>), dim3(blocks), dim3(256), 0, c->stream, a);
    else           hipLaunchKernelGGL((scan_slab_kernel<2, 8>), dim3(blocks), dim3(256), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

int launch_scan_dense(mk_ctx *c, const DenseArgs &a)
{
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

}  // namespace mk
