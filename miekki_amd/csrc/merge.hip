// K6b: the heap half of Miekki::filter_results (Miekki.cpp:386-396) on the device.
//
// select_kernel (K6) leaves, per query and per genome shard, the heap ENTRANTS in
// ascending genome id.  What the reference does with them is a few dozen operations
// on a binary heap of at most nresults+1 records: push while the heap is short,
// otherwise skip what is below the minimum, pop the minimum and push (ties replace),
// and at the end sort_heap.  Which of several equal records survives, and the order
// in which equal records are printed, depend on the exact sift sequence of the heap
// algorithms the reference is built with (libstdc++'s push_heap / pop_heap /
// sort_heap).  Those are restated below operation by operation -- hole-based sift
// up, sift down to a leaf followed by sift up, the even-length last-child case --
// and pinned against the host's own std:: calls (mk_filter_candidates) on rows
// full of ties in tests/test_gpu_parity.py.
//
// One lane per query: the work per query is tiny and strictly sequential, the
// parallelism is across the 10^5 queries of a batch.  The heap holds (key, row
// reference) pairs in per-lane scratch; records are gathered from the candidate
// rows once at the end.  Rows of `world` shards are consumed in shard order, which
// is genome order, so this is also the rank-0 merge after the
// multi-GPU gather, with nothing but nq x nresults records ever leaving the device.
#include "mk_internal.hpp"

namespace mk {

namespace {

struct HeapRef {
    double key[kSelectMaxResults + 1];
    uint32_t ref[kSelectMaxResults + 1];
};

// comp(a, b) of the reference's priority queue: a.intersection > b.intersection
__device__ __forceinline__ bool heap_comp(double a, double b) { return a > b; }

// std::__push_heap(first, hole, top, value, comp)
__device__ __forceinline__ void sift_up(HeapRef &h, int hole, int top, double vkey, uint32_t vref)
{
    int parent = (hole - 1) / 2;
    while (hole > top && heap_comp(h.key[parent], vkey)) {
        h.key[hole] = h.key[parent]; h.ref[hole] = h.ref[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h.key[hole] = vkey; h.ref[hole] = vref;
}

// std::__adjust_heap(first, hole, len, value, comp)
__device__ __forceinline__ void adjust(HeapRef &h, int hole, int len, double vkey, uint32_t vref)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (heap_comp(h.key[child], h.key[child - 1])) --child;
        h.key[hole] = h.key[child]; h.ref[hole] = h.ref[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h.key[hole] = h.key[child - 1]; h.ref[hole] = h.ref[child - 1];
        hole = child - 1;
    }
    sift_up(h, hole, top, vkey, vref);
}

// std::pop_heap on [0, len): the front moves to slot len-1, the rest is a heap again
__device__ __forceinline__ void pop_to_back(HeapRef &h, int len)
{
    if (len <= 1) return;
    const double vkey = h.key[len - 1];
    const uint32_t vref = h.ref[len - 1];
    h.key[len - 1] = h.key[0]; h.ref[len - 1] = h.ref[0];
    adjust(h, 0, len - 1, vkey, vref);
}

}  // namespace

// COMPACT: the rows are the 8-byte exchange form ([1 + cap] words per (shard, query): count, then
// genome | matches << 32); jaccard and intersection are recomputed here from the sizes of ALL
// genomes (a.ss / a.gs, indexed by genome id - a.id_base) in the reference's double operations
// (Miekki.cpp:382-383) -- the same two operations select_kernel ran on the owning shard, so the
// keys are bit-identical to the ones the shard's own heap test saw.
template <bool COMPACT>
__global__ __launch_bounds__(64) void merge_kernel(const MergeArgs a)
{
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    if (q >= a.nq) return;
    const uint64_t rstride = (uint64_t)a.cap + 1u;
    auto count_of = [&](uint32_t r) -> uint32_t {
        return COMPACT ? (uint32_t)a.rows[((uint64_t)r * a.nq + q) * rstride] : a.count[(uint64_t)r * a.nq + q];
    };
    for (uint32_t r = 0; r < a.world; ++r)
        if (count_of(r) > a.cap) {                                // a shard's row lost entrants: the caller replays
            a.nhits[q] = kMergeOverflow;
            return;
        }
    HeapRef h;
    int n = 0;
    const int N = (int)a.nresults;
    for (uint32_t r = 0; r < a.world; ++r) {
        const uint32_t m = count_of(r);
        const uint64_t row0 = COMPACT ? ((uint64_t)r * a.nq + q) * rstride + 1u : ((uint64_t)r * a.nq + q) * a.cap;
        for (uint32_t i = 0; i < m; ++i) {
            double v;
            if (COMPACT) {
                const uint64_t rec = a.rows[row0 + i];
                const uint32_t g = (uint32_t)rec - a.id_base;
                const double jac = (double)(uint32_t)(rec >> 32) / (double)a.ss[g];
                v = jac * (double)a.gs[g];
            } else {
                v = a.cand[row0 + i].intersection;
            }
            if (n >= N) {
                if (n == 0) continue;                             // nresults == 0
                if (h.key[0] > v) continue;                       // Miekki.cpp:387, ties replace
                pop_to_back(h, n);
                --n;
            }
            h.key[n] = v; h.ref[n] = r * a.cap + i;               // push_back + push_heap
            ++n;
            sift_up(h, n - 1, 0, v, r * a.cap + i);
        }
    }
    for (int len = n; len > 1; --len) pop_to_back(h, len);        // std::sort_heap
    mk_hit *__restrict__ out = a.hits + (uint64_t)q * a.nresults;
    for (int i = 0; i < n; ++i) {
        const uint32_t r = h.ref[i] / a.cap, j = h.ref[i] % a.cap;
        if (COMPACT) {
            const uint64_t rec = a.rows[((uint64_t)r * a.nq + q) * rstride + 1u + j];
            const uint32_t g = (uint32_t)rec - a.id_base;
            mk_hit o;
            o.genome = (uint32_t)rec;
            o.matches = (uint32_t)(rec >> 32);
            o.jaccard = (double)o.matches / (double)a.ss[g];
            o.intersection = o.jaccard * (double)a.gs[g];
            out[i] = o;
        } else {
            out[i] = a.cand[((uint64_t)r * a.nq + q) * a.cap + j];
        }
    }
    a.nhits[q] = (uint32_t)n;
}

int launch_merge(mk_ctx *c, const MergeArgs &a)
{
    if (!a.nq) return MK_OK;
    if (a.nresults > kSelectMaxResults) { set_error("device merge supports nresults <= 64"); return MK_ERR_ARG; }
    if ((uint64_t)a.world * a.cap > 0xffffffffull) { set_error("world x cap too large"); return MK_ERR_ARG; }
    if (a.rows) hipLaunchKernelGGL(merge_kernel<true>, dim3((a.nq + 63) / 64), dim3(64), 0, c->stream, a);
    else        hipLaunchKernelGGL(merge_kernel<false>, dim3((a.nq + 63) / 64), dim3(64), 0, c->stream, a);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

}  // namespace mk
