// K7: exact k-mer set intersection of queries against one genome (exact mode).
//
// Replaces the unordered_set build and probe of Miekki::ground_truth_batch
// (Miekki.cpp:803-842).  Restated as two open-addressing hash sets in HBM: the
// genome's distinct canonical k-mers (set B) and, per query, its distinct
// canonical k-mers (set A); a k-mer that is NEW to A is looked up in B once, which
// yields |A n B| and |A \ B| without sorting.  Canonical form follows str2num
// (utils.cpp:276-278): the forward word is 0 if the k-mer holds any character
// outside ACGTacgt (str2numstrand, utils.cpp:252-272) and the reverse strand maps
// every such character to 'T' (revCompChar, utils.cpp:203-215).
#include <algorithm>
#include <cstring>
#include <vector>

#include "mk_internal.hpp"

namespace mk {

constexpr uint64_t kSlotEmpty = ~0ULL;     // k <= 31: no k-mer word reaches 2^64-1

__device__ __forceinline__ uint64_t exact_canon(const char *__restrict__ s, uint32_t k)
{
    uint64_t F = 0, R = 0;
    bool valid = true;
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t sc = seed_code((uint8_t)s[j]);      // same table as str2numstrand
        valid &= sc != 4u;
        F = (F << 2) | (sc & 3u);
        // digit of the reverse complement contributed by this character
        const uint32_t rd = sc == 1u ? 2u : sc == 2u ? 1u : sc == 3u ? 0u : 3u;
        R |= (uint64_t)rd << (2 * j);
    }
    if (!valid) F = 0;
    return F < R ? F : R;
}

__device__ __forceinline__ uint32_t slot_of(uint64_t key, uint32_t log2size)
{
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - log2size));
}

// returns true when key was not in the set before
__device__ __forceinline__ bool set_insert(uint64_t *__restrict__ tab, uint32_t log2size, uint64_t key)
{
    const uint32_t mask = (1u << log2size) - 1u;
    uint32_t h = slot_of(key, log2size);
    for (uint32_t probe = 0; probe <= mask; ++probe) {        // bounded: the set is never full
        const unsigned long long prev =
            atomicCAS((unsigned long long *)&tab[h], (unsigned long long)kSlotEmpty, (unsigned long long)key);
        if (prev == kSlotEmpty) return true;
        if (prev == key) return false;
        h = (h + 1) & mask;
    }
    return false;
}

__device__ __forceinline__ bool set_contains(const uint64_t *__restrict__ tab, uint32_t log2size, uint64_t key)
{
    const uint32_t mask = (1u << log2size) - 1u;
    uint32_t h = slot_of(key, log2size);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint64_t v = tab[h];
        if (v == key) return true;
        if (v == kSlotEmpty) return false;
        h = (h + 1) & mask;
    }
    return false;
}

// grid.y = sequence; every k-mer position (len-k+1 of them, Miekki.cpp:807/818)
__global__ __launch_bounds__(256) void exact_genome_kernel(const char *__restrict__ seq,
                                                           const uint64_t *__restrict__ off, uint32_t k,
                                                           uint64_t *__restrict__ setB, uint32_t log2B,
                                                           unsigned long long *__restrict__ nB)
{
    const uint32_t s = blockIdx.y;
    const uint64_t len = off[s + 1] - off[s];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool fresh = false;
    if (len >= k && i + k <= len) fresh = set_insert(setB, log2B, exact_canon(seq + off[s] + i, k));
    const uint64_t m = __ballot(fresh);
    if (m && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(nB, (unsigned long long)__popcll(m));
}

// grid.y = query; set A of query q lives at setA + aoff[q], 2^alog[q] slots
__global__ __launch_bounds__(256) void exact_query_kernel(const char *__restrict__ seq,
                                                          const uint64_t *__restrict__ off, uint32_t k,
                                                          const uint64_t *__restrict__ setB, uint32_t log2B,
                                                          uint64_t *__restrict__ setA,
                                                          const uint64_t *__restrict__ aoff,
                                                          const uint32_t *__restrict__ alog,
                                                          unsigned long long *__restrict__ inter,
                                                          unsigned long long *__restrict__ extra)
{
    const uint32_t q = blockIdx.y;
    const uint64_t len = off[q + 1] - off[q];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    bool in_b = false, not_b = false;
    if (len >= k && i + k <= len) {
        const uint64_t key = exact_canon(seq + off[q] + i, k);
        if (set_insert(setA + aoff[q], alog[q], key)) {           // Miekki.cpp:834-840
            in_b = set_contains(setB, log2B, key);
            not_b = !in_b;
        }
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mi = __ballot(in_b), mn = __ballot(not_b);
    if (mi && lane == (uint32_t)__ffsll((long long)mi) - 1u) atomicAdd(&inter[q], (unsigned long long)__popcll(mi));
    if (mn && lane == (uint32_t)__ffsll((long long)mn) - 1u) atomicAdd(&extra[q], (unsigned long long)__popcll(mn));
}

static uint32_t log2_slots(uint64_t n_keys)
{
    uint32_t l = 4;
    while ((1ull << l) < 2 * n_keys + 16) ++l;
    return l;
}

// Scratch of the exact mode, kept on the context and only ever grown: a run verifies hundreds of
// genome files one after the other, and seven hipMalloc / hipFree pairs per file (each a device
// synchronisation) cost more than the kernels.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    mk_ctx *c; int slot;
    DevBuf(mk_ctx *ctx, int s) : c(ctx), slot(s) {}
    int alloc(uint64_t n)
    {
        const uint64_t bytes = std::max<uint64_t>(n, 1) * sizeof(T);
        if (bytes > c->exact_cap[slot]) {
            if (c->exact_buf[slot]) (void)hipFree(c->exact_buf[slot]);
            c->exact_buf[slot] = nullptr; c->exact_cap[slot] = 0;
            const uint64_t want = bytes + bytes / 4;
            MK_HIP(hipMalloc(&c->exact_buf[slot], want));
            c->exact_cap[slot] = want;
        }
        p = reinterpret_cast<T *>(c->exact_buf[slot]);
        return MK_OK;
    }
};

static int upload_seqs(mk_ctx *c, const char *const *seqs, const uint64_t *lens, uint32_t n, DevBuf<char> &d_seq,
                       DevBuf<uint64_t> &d_off, std::vector<uint64_t> &off)
{
    off.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + lens[i];
    MK_TRY(d_seq.alloc(off[n] + 64));
    MK_TRY(d_off.alloc(n + 1));
    // long sequences (contigs) go straight from the caller's memory, short ones are gathered first
    constexpr uint64_t kDirect = 256u << 10;
    std::vector<char> host;
    for (uint32_t i = 0; i < n;) {
        if (lens[i] >= kDirect) {
            MK_HIP(hipMemcpyAsync(d_seq.p + off[i], seqs[i], lens[i], hipMemcpyHostToDevice, c->stream));
            ++i;
            continue;
        }
        uint32_t e = i;
        while (e < n && lens[e] < kDirect) ++e;
        host.resize(off[e] - off[i]);
        for (uint32_t j = i; j < e; ++j) memcpy(host.data() + (off[j] - off[i]), seqs[j], lens[j]);
        if (!host.empty()) {
            MK_HIP(hipMemcpyAsync(d_seq.p + off[i], host.data(), host.size(), hipMemcpyHostToDevice, c->stream));
            MK_HIP(hipStreamSynchronize(c->stream));                // `host` is reused
        }
        i = e;
    }
    MK_HIP(hipMemcpyAsync(d_off.p, off.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    return MK_OK;
}

int exact_sets(mk_ctx *c, const char *const *contigs, const uint64_t *contig_lens, uint32_t n_contigs,
               const char *const *queries, const uint64_t *query_lens, uint32_t nq, uint64_t *inter,
               uint64_t *uni)
{
    const uint32_t k = c->p.k;
    // ---- set B
    DevBuf<char> g_seq(c, 0); DevBuf<uint64_t> g_off(c, 1);
    std::vector<uint64_t> goff;
    MK_TRY(upload_seqs(c, contigs, contig_lens, n_contigs, g_seq, g_off, goff));
    uint64_t nkB = 0, maxlen = 0;
    for (uint32_t i = 0; i < n_contigs; ++i) {
        if (contig_lens[i] >= k) nkB += contig_lens[i] - k + 1;
        maxlen = std::max(maxlen, contig_lens[i]);
    }
    const uint32_t log2B = log2_slots(nkB);
    if (log2B > 31) { set_error("genome too large for exact mode"); return MK_ERR_ARG; }
    DevBuf<uint64_t> setB(c, 2); DevBuf<unsigned long long> d_nB(c, 3);
    MK_TRY(setB.alloc(1ull << log2B));
    MK_TRY(d_nB.alloc(1));
    MK_HIP(hipMemsetAsync(setB.p, 0xFF, (8ull << log2B), c->stream));
    MK_HIP(hipMemsetAsync(d_nB.p, 0, 8, c->stream));
    if (n_contigs && maxlen >= k) {
        hipLaunchKernelGGL(exact_genome_kernel, dim3((uint32_t)((maxlen - k + 1 + 255) / 256), n_contigs), dim3(256),
                           0, c->stream, g_seq.p, g_off.p, k, setB.p, log2B, d_nB.p);
        MK_HIP(hipGetLastError());
    }
    unsigned long long nB = 0;
    MK_HIP(hipMemcpyAsync(&nB, d_nB.p, 8, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    if (!nq) return MK_OK;
    // ---- sets A, one per query
    DevBuf<char> q_seq(c, 4); DevBuf<uint64_t> q_off(c, 5);
    std::vector<uint64_t> qoff;
    MK_TRY(upload_seqs(c, queries, query_lens, nq, q_seq, q_off, qoff));
    std::vector<uint64_t> aoff(nq + 1, 0);
    std::vector<uint32_t> alog(nq);
    uint64_t qmax = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        const uint64_t nk = query_lens[q] >= k ? query_lens[q] - k + 1 : 0;
        alog[q] = log2_slots(nk);
        if (alog[q] > 31) { set_error("query too large for exact mode"); return MK_ERR_ARG; }
        aoff[q + 1] = aoff[q] + (1ull << alog[q]);
        qmax = std::max(qmax, query_lens[q]);
    }
    DevBuf<uint64_t> setA(c, 6), d_aoff(c, 7); DevBuf<uint32_t> d_alog(c, 8); DevBuf<unsigned long long> d_cnt(c, 9);
    MK_TRY(setA.alloc(aoff[nq]));
    MK_TRY(d_aoff.alloc(nq + 1));
    MK_TRY(d_alog.alloc(nq));
    MK_TRY(d_cnt.alloc(2ull * nq));
    MK_HIP(hipMemsetAsync(setA.p, 0xFF, aoff[nq] * 8, c->stream));
    MK_HIP(hipMemsetAsync(d_cnt.p, 0, 16ull * nq, c->stream));
    MK_HIP(hipMemcpyAsync(d_aoff.p, aoff.data(), (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipMemcpyAsync(d_alog.p, alog.data(), (size_t)nq * 4, hipMemcpyHostToDevice, c->stream));
    if (qmax >= k) {
        hipLaunchKernelGGL(exact_query_kernel, dim3((uint32_t)((qmax - k + 1 + 255) / 256), nq), dim3(256), 0,
                           c->stream, q_seq.p, q_off.p, k, setB.p, log2B, setA.p, d_aoff.p, d_alog.p, d_cnt.p,
                           d_cnt.p + nq);
        MK_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> cnt(2ull * nq);
    MK_HIP(hipMemcpyAsync(cnt.data(), d_cnt.p, 16ull * nq, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    for (uint32_t q = 0; q < nq; ++q) {
        inter[q] = cnt[q];
        uni[q] = nB + cnt[nq + q];                                  // nb_union = |B| + |A \ B|
    }
    return MK_OK;
}

}  // namespace mk
