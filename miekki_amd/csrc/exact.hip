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

// Both kernels walk ONE flat position space: the sequences (contigs of the genome file, or
// the queries of a call) lie end to end in `seq`, off[0..nseq] are their boundaries, and a
// position i is a k-mer start iff i + k <= the end of the sequence it lies in (every one of
// the len-k+1 positions, Miekki.cpp:807/818/832 -- unlike the sketch, the last k-mer counts).
// So the grid is ceil(total / 4096) workgroups whatever the number or shape of the
// sequences: 100,000 tiny contigs, one 1 Mb contig among 10,000 short ones, 100,000 queries.
// A workgroup stages its 4096 (+ k-1) characters once, as str2numstrand codes in LDS, and a
// thread rolls 16 consecutive k-mers from them: k-1 LDS reads to seed, one per k-mer after.
constexpr uint32_t kExTile = 4096, kExPer = 16;

// largest s < nseq with off[s] <= pos (pos < off[nseq])
__device__ __forceinline__ uint32_t seq_of(const uint64_t *__restrict__ off, uint32_t nseq, uint64_t pos)
{
    uint32_t lo = 0, hi = nseq;                               // invariant: off[lo] <= pos < off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

struct ExactArgs {
    const char *seq;               // padded with 64 readable bytes past `total`
    const uint64_t *off;           // nseq + 1
    uint32_t nseq, k;
    uint64_t total;
    uint64_t *setB;
    uint32_t log2B;
    unsigned long long *nB;        // genome pass: distinct k-mers inserted
    uint64_t *setA;                // query pass: per-query sets, aoff / alog
    const uint64_t *aoff;
    const uint32_t *alog;
    unsigned long long *inter, *extra;
};

template <bool QUERY>
__global__ __launch_bounds__(256) void exact_kernel(const ExactArgs a)
{
    __shared__ uint8_t codes[kExTile + 48];
    const uint64_t base = (uint64_t)blockIdx.x * kExTile;
    const uint32_t k = a.k;
    const uint32_t span = (uint32_t)min((uint64_t)(kExTile + k - 1), a.total - base);
    // 16 characters per lane-load (base is a multiple of 4096, the buffer is allocated aligned and padded)
    for (uint32_t i = threadIdx.x * 16u; i < span; i += 256u * 16u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(a.seq + base + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) codes[i + j] = (uint8_t)seed_code((uint8_t)(w[j >> 2] >> (8 * (j & 3u))));
    }
    __syncthreads();
    const uint32_t t0 = threadIdx.x * kExPer;
    if (base + t0 >= a.total) return;
    uint32_t s = seq_of(a.off, a.nseq, base + t0);
    uint64_t send = a.off[s + 1];
    const uint64_t fmask = (1ULL << (2 * k)) - 1ULL;
    const uint32_t top = 2 * (k - 1);
    uint64_t F = 0, R = 0;
    int last_bad = -1;                                         // tile index of the last non-ACGTacgt character seen
    auto push = [&](uint32_t idx) {
        const uint32_t sc = idx < span ? codes[idx] : 4u;
        F = ((F << 2) | (sc & 3u)) & fmask;
        // digit of the reverse complement: revCompChar maps everything outside ACGT to 'T' (utils.cpp:203-215)
        const uint32_t rd = sc == 1u ? 2u : sc == 2u ? 1u : sc == 3u ? 0u : 3u;
        R = (R >> 2) | ((uint64_t)rd << top);
        if (sc == 4u) last_bad = (int)idx;
    };
    for (uint32_t j = 0; j + 1 < k; ++j) push(t0 + j);
    uint32_t fresh = 0, cin = 0, cnot = 0;
    auto flush_counts = [&]() {
        if (QUERY) {
            if (cin) atomicAdd(&a.inter[s], (unsigned long long)cin);
            if (cnot) atomicAdd(&a.extra[s], (unsigned long long)cnot);
            cin = cnot = 0;
        }
    };
#pragma unroll 4
    for (uint32_t u = 0; u < kExPer; ++u) {
        const uint32_t i = t0 + u;
        const uint64_t p = base + i;
        if (p >= a.total) break;
        push(i + k - 1);
        if (p >= send) {
            flush_counts();
            do { ++s; send = a.off[s + 1]; } while (p >= send);  // empty sequences are stepped over
        }
        if (p + k > send) continue;                              // the window would run into the next sequence
        // str2num: min(forward word -- 0 if any character is outside ACGTacgt --, reverse strand word)
        const uint64_t fw = last_bad >= (int)i ? 0ULL : F;
        const uint64_t key = fw < R ? fw : R;
        if (QUERY) {
            if (set_insert(a.setA + a.aoff[s], a.alog[s], key)) {    // Miekki.cpp:834-840: first sight of this k-mer
                if (set_contains(a.setB, a.log2B, key)) ++cin; else ++cnot;
            }
        } else {
            fresh += set_insert(a.setB, a.log2B, key) ? 1u : 0u;
        }
    }
    flush_counts();
    if (!QUERY) {
        for (int o = 32; o > 0; o >>= 1) fresh += __shfl_xor(fresh, o);
        if ((threadIdx.x & 63u) == 0 && fresh) atomicAdd(a.nB, (unsigned long long)fresh);
    }
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

// ground_truth_batch's genome side (Miekki.cpp:803-822): set B of the contigs, left resident on the
// context until the next call.
int exact_load_genome(mk_ctx *c, const char *const *contigs, const uint64_t *contig_lens, uint32_t n_contigs)
{
    const uint32_t k = c->p.k;
    c->exact_have_B = false;
    DevBuf<char> g_seq(c, 0); DevBuf<uint64_t> g_off(c, 1);
    std::vector<uint64_t> goff;
    MK_TRY(upload_seqs(c, contigs, contig_lens, n_contigs, g_seq, g_off, goff));
    uint64_t nkB = 0;
    for (uint32_t i = 0; i < n_contigs; ++i)
        if (contig_lens[i] >= k) nkB += contig_lens[i] - k + 1;
    const uint32_t log2B = log2_slots(nkB);
    if (log2B > 31) { set_error("genome too large for exact mode"); return MK_ERR_ARG; }
    DevBuf<uint64_t> setB(c, 2); DevBuf<unsigned long long> d_nB(c, 3);
    MK_TRY(setB.alloc(1ull << log2B));
    MK_TRY(d_nB.alloc(1));
    MK_HIP(hipMemsetAsync(setB.p, 0xFF, (8ull << log2B), c->stream));
    MK_HIP(hipMemsetAsync(d_nB.p, 0, 8, c->stream));
    const uint64_t total = goff[n_contigs];
    if (nkB) {
        ExactArgs a{};
        a.seq = g_seq.p; a.off = g_off.p; a.nseq = n_contigs; a.k = k; a.total = total;
        a.setB = setB.p; a.log2B = log2B; a.nB = d_nB.p;
        const uint64_t blocks = (total + kExTile - 1) / kExTile;
        if (blocks > 0x7fffffffull) { set_error("genome too large for exact mode"); return MK_ERR_ARG; }
        hipLaunchKernelGGL(exact_kernel<false>, dim3((uint32_t)blocks), dim3(256), 0, c->stream, a);
        MK_HIP(hipGetLastError());
    }
    unsigned long long nB = 0;
    MK_HIP(hipMemcpyAsync(&nB, d_nB.p, 8, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    c->exact_nB = nB; c->exact_log2B = log2B; c->exact_have_B = true;
    return MK_OK;
}

// ground_truth_batch's query side (Miekki.cpp:826-842) against the resident set B
int exact_queries(mk_ctx *c, const char *const *queries, const uint64_t *query_lens, uint32_t nq, uint64_t *inter,
                  uint64_t *uni)
{
    if (!c->exact_have_B) { set_error("no genome loaded for exact mode"); return MK_ERR_STATE; }
    if (!nq) return MK_OK;
    const uint32_t k = c->p.k;
    DevBuf<char> q_seq(c, 4); DevBuf<uint64_t> q_off(c, 5);
    std::vector<uint64_t> qoff;
    MK_TRY(upload_seqs(c, queries, query_lens, nq, q_seq, q_off, qoff));
    std::vector<uint64_t> aoff(nq + 1, 0);
    std::vector<uint32_t> alog(nq);
    uint64_t nk_all = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        const uint64_t nk = query_lens[q] >= k ? query_lens[q] - k + 1 : 0;
        alog[q] = log2_slots(nk);
        if (alog[q] > 31) { set_error("query too large for exact mode"); return MK_ERR_ARG; }
        aoff[q + 1] = aoff[q] + (1ull << alog[q]);
        nk_all += nk;
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
    const uint64_t total = qoff[nq];
    if (nk_all) {
        ExactArgs a{};
        a.seq = q_seq.p; a.off = q_off.p; a.nseq = nq; a.k = k; a.total = total;
        a.setB = reinterpret_cast<uint64_t *>(c->exact_buf[2]); a.log2B = c->exact_log2B;
        a.setA = setA.p; a.aoff = d_aoff.p; a.alog = d_alog.p; a.inter = d_cnt.p; a.extra = d_cnt.p + nq;
        const uint64_t blocks = (total + kExTile - 1) / kExTile;
        if (blocks > 0x7fffffffull) { set_error("query batch too large for exact mode"); return MK_ERR_ARG; }
        hipLaunchKernelGGL(exact_kernel<true>, dim3((uint32_t)blocks), dim3(256), 0, c->stream, a);
        MK_HIP(hipGetLastError());
    }
    std::vector<unsigned long long> cnt(2ull * nq);
    MK_HIP(hipMemcpyAsync(cnt.data(), d_cnt.p, 16ull * nq, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    for (uint32_t q = 0; q < nq; ++q) {
        inter[q] = cnt[q];
        uni[q] = c->exact_nB + cnt[nq + q];                         // nb_union = |B| + |A \ B|
    }
    return MK_OK;
}

}  // namespace mk
