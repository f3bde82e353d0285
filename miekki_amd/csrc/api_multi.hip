// C ABI of libmiekki_hip.so, what several GPUs need of each other: the heap over gathered entrant rows (mk_filter_candidates,
// mk_merge_*), id bases, device memory for callers without a GPU runtime (mk_dev_*), peer copies, the Bloom filter as
// device bytes (export / import / first-writer fold).  The collectives themselves are in comm.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>

#include "mk_internal.hpp"

using namespace mk;

extern "C" {

// Miekki::filter_results (Miekki.cpp:376-397) on pre-thresholded candidates in
// ascending genome order; same libstdc++ heap calls as the reference.
uint32_t mk_filter_candidates(const mk_hit *cand, uint32_t ncand, uint32_t nresults, mk_hit *out)
{
    const auto compare = [](const mk_hit &a, const mk_hit &b) { return a.intersection > b.intersection; };
    std::vector<mk_hit> heap;
    heap.reserve((size_t)nresults + 1);
    for (uint32_t i = 0; i < ncand; ++i) {
        if (heap.size() >= nresults) {
            if (heap.empty()) continue;
            if (heap.front().intersection > cand[i].intersection) continue;   // ties replace
            std::pop_heap(heap.begin(), heap.end(), compare);
            heap.pop_back();
        }
        heap.push_back(cand[i]);
        std::push_heap(heap.begin(), heap.end(), compare);
    }
    std::sort_heap(heap.begin(), heap.end(), compare);
    std::copy(heap.begin(), heap.end(), out);
    return (uint32_t)heap.size();
}

int mk_merge_entrants(mk_ctx *c, const uint32_t *d_count, const mk_hit *d_cand, uint32_t world, uint32_t nq,
                      uint32_t cap, uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits)
{
    if (!c || (nq && (!d_count || !d_cand || !d_nhits || (nresults && !d_hits)))) { set_error("null argument"); return MK_ERR_ARG; }
    if (!world || !cap) { set_error("world and cap must be positive"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MergeArgs ma{d_count, d_cand, world, nq, cap, nresults, d_hits, d_nhits};
    return launch_merge(c, ma);
}

int mk_merge_set_sizes(mk_ctx *c, const uint64_t *genome_size, const uint32_t *sketch_size, uint32_t n,
                       uint32_t id_base)
{
    if (!c || (n && (!genome_size || !sketch_size))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MK_HIP(hipStreamSynchronize(c->stream));
    dev_free(c->d_all_ss); dev_free(c->d_all_gs);
    c->all_n = 0; c->all_base = id_base;
    if (!n) return MK_OK;
    MK_TRY(dev_alloc(&c->d_all_ss, n));
    MK_TRY(dev_alloc(&c->d_all_gs, n));
    MK_HIP(hipMemcpy(c->d_all_ss, sketch_size, (size_t)n * 4, hipMemcpyHostToDevice));
    MK_HIP(hipMemcpy(c->d_all_gs, genome_size, (size_t)n * 8, hipMemcpyHostToDevice));
    c->all_n = n;
    return MK_OK;
}

int mk_merge_get_sizes(mk_ctx *c, uint64_t *genome_size, uint32_t *sketch_size, uint32_t n)
{
    if (!c || (n && (!genome_size || !sketch_size))) { set_error("null argument"); return MK_ERR_ARG; }
    if (n != c->all_n) { set_error("%u sizes were set (mk_merge_set_sizes), %u asked for", c->all_n, n); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    if (!n) return MK_OK;
    MK_HIP(hipMemcpy(sketch_size, c->d_all_ss, (size_t)n * 4, hipMemcpyDeviceToHost));
    MK_HIP(hipMemcpy(genome_size, c->d_all_gs, (size_t)n * 8, hipMemcpyDeviceToHost));
    return MK_OK;
}

int mk_merge_compact(mk_ctx *c, const uint64_t *d_rows, uint32_t world, uint32_t nq, uint32_t cap,
                     uint32_t nresults, mk_hit *d_hits, uint32_t *d_nhits)
{
    if (!c || (nq && (!d_rows || !d_nhits || (nresults && !d_hits)))) { set_error("null argument"); return MK_ERR_ARG; }
    if (!world || !cap) { set_error("world and cap must be positive"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    MergeArgs ma{nullptr, nullptr, world, nq, cap, nresults, d_hits, d_nhits, d_rows, nullptr, nullptr, 0};
    if (c->all_n) { ma.ss = c->d_all_ss; ma.gs = c->d_all_gs; ma.id_base = c->all_base; }
    else if (world == 1) { ma.ss = c->d_sketch_size; ma.gs = c->d_genome_size; ma.id_base = c->p.genome_id_base; }
    else { set_error("mk_merge_compact over several shards needs mk_merge_set_sizes first"); return MK_ERR_STATE; }
    return launch_merge(c, ma);
}

int mk_set_genome_id_base(mk_ctx *c, uint32_t base)
{
    if (!c) { set_error("null context"); return MK_ERR_ARG; }
    c->p.genome_id_base = base;
    return MK_OK;
}

// ---- device buffers for callers that have no GPU runtime of their own
int mk_dev_alloc(mk_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    MK_TRY(use_device(c, false));
    MK_HIP(hipMalloc(out, bytes ? bytes : 1));
    return MK_OK;
}

void mk_dev_free(mk_ctx *c, void *d)
{
    if (!c || !d) return;
    (void)hipSetDevice(c->p.device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d);
}

int mk_dev_upload(mk_ctx *c, void *d_dst, const void *src, uint64_t bytes)
{
    if (!c || (bytes && (!d_dst || !src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    if (bytes) MK_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

int mk_dev_download(mk_ctx *c, void *dst, const void *d_src, uint64_t bytes)
{
    if (!c || (bytes && (!dst || !d_src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(c, false));
    if (bytes) MK_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

// Device-to-device copy between two contexts' GPUs (a peer DMA over xGMI when they differ),
// queued on the SOURCE context's stream -- so behind the kernels that produced d_src -- and
// waited for before returning: afterwards d_dst is complete for any stream of dst.
// Peer access is asked for and enabled once per ordered pair of GPUs; a pair without it (a
// restricted container, say) copies through host memory instead, and every copy is counted by
// the path it took (mk_stats of the source context), so that a staged exchange is visible as
// such and not as a slow xGMI.  Reads nothing of dst but its device ordinal.
}  // extern "C"

namespace {
std::mutex g_peer_mutex;
int g_peer_state[64][64];                                        // [src][dst]: 0 not asked yet, 1 peer access on, 2 none

bool peer_access(int src, int dst)
{
    if (src == dst) return true;
    if (src < 0 || dst < 0 || src >= 64 || dst >= 64) return false;
    std::lock_guard<std::mutex> g(g_peer_mutex);
    if (g_peer_state[src][dst] == 0) {
        int can = 0;
        bool on = hipDeviceCanAccessPeer(&can, src, dst) == hipSuccess && can;
        if (on) {                                                // (the caller has bound `src`: the access is enabled FROM the current device)
            const hipError_t e = hipDeviceEnablePeerAccess(dst, 0);
            on = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();
        g_peer_state[src][dst] = on ? 1 : 2;
        if (getenv("MIEKKI_VERBOSE"))
            fprintf(stderr, "[miekki] GPU %d -> GPU %d: %s\n", src, dst, on ? "peer access (xGMI)" : "NO peer access: copies go through host memory");
    }
    return g_peer_state[src][dst] == 1;
}
}  // namespace

extern "C" {

int mk_dev_copy(mk_ctx *dst, void *d_dst, mk_ctx *src, const void *d_src, uint64_t bytes)
{
    if (!dst || !src || (bytes && (!d_dst || !d_src))) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(use_device(src, false));
    if (!bytes) { MK_HIP(hipStreamSynchronize(src->stream)); return MK_OK; }
    const int sd = src->p.device, dd = dst->p.device;
    bool direct = peer_access(sd, dd);
    if (direct) {
        const hipError_t e = sd == dd ? hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, src->stream)
                                      : hipMemcpyPeerAsync(d_dst, dd, d_src, sd, bytes, src->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (sd == dd) { set_error("device copy failed: %s", hipGetErrorString(e)); return MK_ERR_DEVICE; }
            { std::lock_guard<std::mutex> g(g_peer_mutex); g_peer_state[sd][dd] = 2; }     // the pair copies through the host from now on
            direct = false;
        }
    }
    if (direct) {
        MK_HIP(hipStreamSynchronize(src->stream));
        src->stats.peer_copies++; src->stats.peer_copy_bytes += bytes;
        return MK_OK;
    }
    // no peer path between the two GPUs: through (page-locked) host memory
    MK_HIP(hipStreamSynchronize(src->stream));
    void *tmp = nullptr;
    MK_HIP(hipHostMalloc(&tmp, bytes, hipHostMallocDefault));
    hipError_t e = hipMemcpy(tmp, d_src, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipSetDevice(dd);
    if (e == hipSuccess) e = hipMemcpy(d_dst, tmp, bytes, hipMemcpyHostToDevice);
    (void)hipSetDevice(sd);
    (void)hipHostFree(tmp);
    if (e != hipSuccess) { set_error("staged device copy failed: %s", hipGetErrorString(e)); return MK_ERR_DEVICE; }
    src->stats.staged_copies++; src->stats.staged_copy_bytes += bytes;
    return MK_OK;
}

int mk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mk_index_export_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, uint8_t *d_dst)
{
    if (!c || !d_dst) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpyAsync(d_dst, c->d_bloom + begin, dev_end - begin, hipMemcpyDeviceToDevice, c->stream));
    const uint64_t zfrom = std::max(begin, dev_end);
    if (zfrom < end) MK_HIP(hipMemsetAsync(d_dst + (zfrom - begin), 0, end - zfrom, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

int mk_index_import_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_src)
{
    if (!c || !d_src) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_HIP(hipMemcpyAsync(c->d_bloom + begin, d_src, dev_end - begin, hipMemcpyDeviceToDevice, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_TRY(forget_bloom_summary(c));
    ++c->gen;
    return MK_OK;
}

int mk_index_merge_bloom_device(mk_ctx *c, uint64_t begin, uint64_t end, const uint8_t *d_later)
{
    if (!c || !d_later) { set_error("null argument"); return MK_ERR_ARG; }
    if (begin > end || end > c->bloom_bytes) { set_error("Bloom range out of bounds"); return MK_ERR_ARG; }
    MK_TRY(use_device(c));
    const uint64_t dev_end = std::min(end, c->bloom_dev_bytes);
    if (begin < dev_end) MK_TRY(launch_bloom_merge(c, begin, dev_end, d_later));
    MK_HIP(hipStreamSynchronize(c->stream));
    c->bloom_full_stale = true;
    ++c->gen;
    return MK_OK;
}

uint64_t mk_bloom_reachable_bytes(const mk_ctx *c) { return c ? c->bloom_dev_bytes : 0; }

}  // extern "C"
