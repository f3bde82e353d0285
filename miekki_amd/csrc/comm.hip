// One process per GPU: the collectives of the genome-sharded index, straight on RCCL (xGMI inside a node).
//
// The reference scales inside ONE executable (-t threads over shared memory, main.cpp:190-196; drivers
// Miekki.cpp:546-581, 430-480) and has no distributed backend; SURVEY.md section 8e defines the multi-GPU form this
// file implements: every rank owns a contiguous genome range, the Bloom filter is made global once after the build
// (first writer in genome order wins, Miekki.cpp:125-129: a MIN all-reduce over rank-keyed cells), and a query batch
// needs ONE exchange step -- ncclGather (rccl.h:745) of the per-query heap-entrant rows to the merging rank.
//
// librccl is bound at run time (dlopen): a process that already holds one -- PyTorch brings its own copy -- keeps
// using that one, a plain C++ host gets the system's, and a single-GPU user of libmiekki_hip.so never loads it.
// Every collective is queued on a HIP stream of the context (its main stream, or the communicator's own for the
// overlapped exchange) and ordered against the kernels with events: no host wait between a scan and its gather.
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "mk_internal.hpp"

struct mk_comm {
    mk_ctx *ctx;
    int rank, world;
    ncclComm_t comm;
    hipStream_t stream;            // the overlapped exchange runs here, beside the next chunk's scan on ctx->stream
    hipEvent_t ev_chunk, ev_done, ev_enter;
    uint32_t *d_keys;              // Bloom fold: rank-keyed cells of one piece
    uint64_t keys_cap;
};

namespace mk {
namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string where;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

void load_rccl()
{
    std::vector<std::pair<std::string, int>> tries;
    if (const char *e = getenv("MIEKKI_RCCL_LIB")) tries.push_back({e, RTLD_NOW | RTLD_GLOBAL});
    // a copy the process already holds (PyTorch's, by SONAME) before any other
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_NOLOAD});
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_NOLOAD});
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
    tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL});
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_GLOBAL});
    for (const auto &t : tries) {
        g_rccl.lib = dlopen(t.first.c_str(), t.second);
        if (g_rccl.lib) { g_rccl.where = t.first + ((t.second & RTLD_NOLOAD) ? " (already loaded)" : ""); break; }
    }
    if (!g_rccl.lib) { g_rccl_error = "librccl not found (MIEKKI_RCCL_LIB names it)"; return; }
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(g_rccl.lib, name); if (!p) { ok = false; g_rccl_error = std::string("librccl lacks ") + name; } return p; };
#define MK_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(sym(name))
    MK_SYM(GetUniqueId, "ncclGetUniqueId"); MK_SYM(CommInitRank, "ncclCommInitRank"); MK_SYM(CommDestroy, "ncclCommDestroy");
    MK_SYM(GetErrorString, "ncclGetErrorString"); MK_SYM(Gather, "ncclGather"); MK_SYM(AllGather, "ncclAllGather");
    MK_SYM(AllReduce, "ncclAllReduce"); MK_SYM(Broadcast, "ncclBroadcast"); MK_SYM(Send, "ncclSend"); MK_SYM(Recv, "ncclRecv");
    MK_SYM(GroupStart, "ncclGroupStart"); MK_SYM(GroupEnd, "ncclGroupEnd");
#undef MK_SYM
    if (!ok) { dlclose(g_rccl.lib); g_rccl.lib = nullptr; }
    else if (getenv("MIEKKI_VERBOSE")) fprintf(stderr, "[miekki] RCCL: %s\n", g_rccl.where.c_str());
}

int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.lib) { set_error("%s", g_rccl_error.c_str()); return MK_ERR_DEVICE; }
    return MK_OK;
}

#define MK_NCCL(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            mk::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return MK_ERR_DEVICE;                                                                       \
        }                                                                                               \
    } while (0)

// Bloom first-writer fold (Miekki.cpp:125-129: a cell keeps the byte of its first inserter in genome order; with
// contiguous shards in rank order that is the lowest rank whose cell is non-zero): cell -> rank << 8 | byte, an empty
// cell -> all ones, MIN over the ranks, and back.
__global__ void bloom_key_kernel(const uint8_t *__restrict__ cells, uint32_t *__restrict__ keys, uint64_t n, uint32_t rank)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = cells[i];
        keys[i] = b ? (rank << 8 | b) : 0xffffffffu;
    }
}
__global__ void bloom_unkey_kernel(const uint32_t *__restrict__ keys, uint8_t *__restrict__ cells, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        cells[i] = keys[i] == 0xffffffffu ? (uint8_t)0 : (uint8_t)(keys[i] & 0xffu);
}

}  // namespace
}  // namespace mk

using namespace mk;

extern "C" {

int mk_comm_unique_id(uint8_t *id)
{
    if (!id) { set_error("null argument"); return MK_ERR_ARG; }
    MK_TRY(need_rccl());
    static_assert(sizeof(ncclUniqueId) == MK_COMM_ID_BYTES, "mk_comm id size");
    ncclUniqueId u;
    MK_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return MK_OK;
}

int mk_comm_create(mk_ctx *c, int rank, int world, const uint8_t *id, mk_comm **out)
{
    if (!c || !id || !out) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { set_error("rank %d of %d", rank, world); return MK_ERR_ARG; }
    MK_TRY(need_rccl());
    MK_HIP(hipSetDevice(c->p.device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t nc = nullptr;
    // This RCCL build prints a version banner ("RCCL version : ...", five lines) on STDOUT from rank 0's first
    // communicator -- into the middle of whatever the host program reports there (the `miekki` binary's banners are
    // compared byte for byte with the reference's; bench.py's stdout is one JSON line).  A program that owns its
    // stdout asks for it to go to stderr instead (MIEKKI_COMM_BANNER_TO_STDERR=1, set by those two before they call
    // this): descriptor 1 then points at descriptor 2 while the communicator initialises -- for the whole PROCESS and
    // for as long as the slowest rank takes to arrive, which is why a library caller has to opt in.
    const char *quiet = getenv("MIEKKI_COMM_BANNER_TO_STDERR");
    int saved = -1;
    if (quiet && *quiet && *quiet != '0') {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) (void)dup2(2, 1);
    }
    const ncclResult_t init = g_rccl.CommInitRank(&nc, world, u, rank);   // (collective: returns when every rank has called it)
    if (saved >= 0) { fflush(stdout); (void)dup2(saved, 1); (void)close(saved); }
    MK_NCCL(init);
    mk_comm *m = new mk_comm();
    m->ctx = c; m->rank = rank; m->world = world; m->comm = nc; m->d_keys = nullptr; m->keys_cap = 0;
    m->stream = nullptr; m->ev_chunk = m->ev_done = m->ev_enter = nullptr;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_chunk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_enter, hipEventDisableTiming) != hipSuccess) {
        set_error("communicator streams: %s", hipGetErrorString(hipGetLastError()));
        mk_comm_destroy(m);
        return MK_ERR_DEVICE;
    }
    *out = m;
    return MK_OK;
}

void mk_comm_destroy(mk_comm *m)
{
    if (!m) return;
    (void)hipSetDevice(m->ctx->p.device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    (void)hipStreamSynchronize(m->ctx->stream);
    if (m->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm);
    if (m->d_keys) (void)hipFree(m->d_keys);
    if (m->ev_chunk) (void)hipEventDestroy(m->ev_chunk);
    if (m->ev_done) (void)hipEventDestroy(m->ev_done);
    if (m->ev_enter) (void)hipEventDestroy(m->ev_enter);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int mk_comm_rank(const mk_comm *m) { return m ? m->rank : -1; }
int mk_comm_world(const mk_comm *m) { return m ? m->world : 0; }

// ---- plain collectives on device buffers of the context's GPU, queued on the context's stream -------------------
int mk_comm_gather(mk_comm *m, const void *d_send, uint64_t bytes, void *d_recv, int root)
{
    if (!m || (bytes && !d_send) || root < 0 || root >= m->world || (m->rank == root && bytes && !d_recv)) { set_error("bad argument"); return MK_ERR_ARG; }
    MK_HIP(hipSetDevice(m->ctx->p.device));
    if (!bytes) return MK_OK;
    if (bytes % 8 == 0) MK_NCCL(g_rccl.Gather(d_send, d_recv, bytes / 8, ncclUint64, root, m->comm, m->ctx->stream));
    else MK_NCCL(g_rccl.Gather(d_send, d_recv, bytes, ncclUint8, root, m->comm, m->ctx->stream));
    return MK_OK;
}

int mk_comm_gather_rows(mk_comm *m, const uint64_t *d_rows, uint64_t words, uint64_t *d_recv, int root)
{
    return mk_comm_gather(m, d_rows, words * 8, d_recv, root);
}

int mk_comm_allgather(mk_comm *m, const void *d_send, uint64_t bytes, void *d_recv)
{
    if (!m || (bytes && (!d_send || !d_recv))) { set_error("bad argument"); return MK_ERR_ARG; }
    MK_HIP(hipSetDevice(m->ctx->p.device));
    if (!bytes) return MK_OK;
    MK_NCCL(g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, m->comm, m->ctx->stream));
    return MK_OK;
}

int mk_comm_broadcast(mk_comm *m, void *d_buf, uint64_t bytes, int root)
{
    if (!m || (bytes && !d_buf) || root < 0 || root >= m->world) { set_error("bad argument"); return MK_ERR_ARG; }
    MK_HIP(hipSetDevice(m->ctx->p.device));
    if (!bytes) return MK_OK;
    MK_NCCL(g_rccl.Broadcast(d_buf, d_buf, bytes, ncclUint8, root, m->comm, m->ctx->stream));
    return MK_OK;
}

int mk_comm_allreduce_max_f64(mk_comm *m, double *d_values, uint32_t n)
{
    if (!m || (n && !d_values)) { set_error("bad argument"); return MK_ERR_ARG; }
    MK_HIP(hipSetDevice(m->ctx->p.device));
    if (!n) return MK_OK;
    MK_NCCL(g_rccl.AllReduce(d_values, d_values, n, ncclFloat64, ncclMax, m->comm, m->ctx->stream));
    return MK_OK;
}

int mk_comm_barrier(mk_comm *m)
{
    if (!m) { set_error("null communicator"); return MK_ERR_ARG; }
    MK_HIP(hipSetDevice(m->ctx->p.device));
    if (!m->d_keys) { MK_HIP(hipMalloc((void **)&m->d_keys, 1024)); m->keys_cap = 256; }
    MK_HIP(hipMemsetAsync(m->d_keys, 0, 4, m->ctx->stream));
    MK_NCCL(g_rccl.AllReduce(m->d_keys, m->d_keys, 1, ncclUint32, ncclSum, m->comm, m->ctx->stream));
    MK_HIP(hipStreamSynchronize(m->ctx->stream));
    return MK_OK;
}

// ---- after the build: ONE Bloom filter, and the sizes of every genome on every rank ----------------------------
int mk_comm_sync_bloom(mk_comm *m)
{
    if (!m) { set_error("null communicator"); return MK_ERR_ARG; }
    mk_ctx *c = m->ctx;
    MK_TRY(mk_sync(c));                                            // (settles a build batch still in flight)
    const uint64_t n = c->bloom_dev_bytes;                         // the cells a 2k-bit k-mer can reach: the same on every rank
    if (!n) return MK_OK;
    const uint64_t piece = 16ull << 20;                            // 64 MiB of keys at a time
    if (m->keys_cap < std::min(n, piece)) {
        if (m->d_keys) (void)hipFree(m->d_keys);
        m->d_keys = nullptr; m->keys_cap = 0;
        MK_HIP(hipMalloc((void **)&m->d_keys, std::min(n, piece) * 4));
        m->keys_cap = std::min(n, piece);
    }
    for (uint64_t o = 0; o < n; o += piece) {
        const uint64_t cnt = std::min(piece, n - o);
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((cnt + 255) / 256, 8192);
        hipLaunchKernelGGL(bloom_key_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_bloom + o, m->d_keys, cnt, (uint32_t)m->rank);
        MK_NCCL(g_rccl.AllReduce(m->d_keys, m->d_keys, cnt, ncclUint32, ncclMin, m->comm, c->stream));
        hipLaunchKernelGGL(bloom_unkey_kernel, dim3(blocks), dim3(256), 0, c->stream, m->d_keys, c->d_bloom + o, cnt);
    }
    MK_HIP(hipGetLastError());
    MK_HIP(hipStreamSynchronize(c->stream));
    MK_TRY(forget_bloom_summary(c));
    ++c->gen;
    return MK_OK;
}

int mk_comm_share_sizes(mk_comm *m, uint32_t *id_base, uint32_t *total)
{
    if (!m) { set_error("null communicator"); return MK_ERR_ARG; }
    mk_ctx *c = m->ctx;
    MK_TRY(mk_sync(c));
    const int W = m->world;
    // how many genomes every rank holds ...
    uint32_t *d_n = nullptr;
    MK_HIP(hipMalloc((void **)&d_n, (size_t)(W + 1) * 4));
    std::vector<uint32_t> counts(W, 0);
    const uint32_t mine = c->G;
    int rc = MK_OK;
    if (hipMemcpyAsync(d_n + W, &mine, 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_OK) rc = mk_comm_allgather(m, d_n + W, 4, d_n);
    if (rc == MK_OK && hipMemcpyAsync(counts.data(), d_n, (size_t)W * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    (void)hipFree(d_n);
    if (rc != MK_OK) { if (rc == MK_ERR_DEVICE) set_error("size exchange failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
    uint64_t sum = 0, base = 0, mx = 1;
    for (int r = 0; r < W; ++r) { if (r < m->rank) base += counts[r]; sum += counts[r]; mx = std::max<uint64_t>(mx, counts[r]); }
    if (sum > 0xffffff00ull) { set_error("more than 2^32 genomes in total"); return MK_ERR_ARG; }
    // ... then their sizes, padded to the largest shard: [sketch_size u32 x mx | genome_size u64 x mx] per rank
    const uint64_t rec = mx * 12;
    uint8_t *d_buf = nullptr;
    MK_HIP(hipMalloc((void **)&d_buf, rec * (W + 1)));
    std::vector<uint8_t> h(rec * W, 0), send(rec, 0);
    memcpy(send.data(), c->h_sketch_size.data(), (size_t)mine * 4);
    memcpy(send.data() + mx * 4, c->h_genome_size.data(), (size_t)mine * 8);
    if (hipMemcpyAsync(d_buf + rec * W, send.data(), rec, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_OK) rc = mk_comm_allgather(m, d_buf + rec * W, rec, d_buf);
    if (rc == MK_OK && hipMemcpyAsync(h.data(), d_buf, rec * W, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    if (rc == MK_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = MK_ERR_DEVICE;
    (void)hipFree(d_buf);
    if (rc != MK_OK) { if (rc == MK_ERR_DEVICE) set_error("size exchange failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
    std::vector<uint32_t> ss(sum ? sum : 1);
    std::vector<uint64_t> gs(sum ? sum : 1);
    uint64_t at = 0;
    for (int r = 0; r < W; ++r) {
        memcpy(ss.data() + at, h.data() + rec * r, (size_t)counts[r] * 4);
        memcpy(gs.data() + at, h.data() + rec * r + mx * 4, (size_t)counts[r] * 8);
        at += counts[r];
    }
    MK_TRY(mk_set_genome_id_base(c, (uint32_t)base));
    MK_TRY(mk_merge_set_sizes(c, gs.data(), ss.data(), (uint32_t)sum, 0));
    if (id_base) *id_base = (uint32_t)base;
    if (total) *total = (uint32_t)sum;
    return MK_OK;
}

// ---- the hot path with its exchange step --------------------------------------------------------------------
// mk_qset_run_compact, and each block of finished rows leaves for `root` on the communicator's stream while the next
// chunk of queries is still being scanned on the context's: d_recv[world][nq][1 + cap] on root fills up behind the
// scan.  The blocks are cut from nq alone (the same on every rank, whatever chunk sizes the ranks' memory budgets
// give their scans); a block is one grouped ncclSend / ncclRecv round -- a gather with the receive side strided --
// and a set that fits one block takes ncclGather itself.  On return everything is queued and the context's stream
// has been made to wait for the exchange: the next call on the context (mk_merge_compact on root) is ordered.
int mk_qset_run_compact_gather(mk_ctx *c, mk_comm *m, mk_qset *qs, uint32_t nresults, uint32_t min_score, double min_inter,
                               uint32_t cap, uint64_t *d_rows, uint64_t *d_recv, int root)
{
    if (!c || !m || !qs || !d_rows || !cap || m->ctx != c || root < 0 || root >= m->world || (m->rank == root && !d_recv)) {
        set_error("bad argument");
        return MK_ERR_ARG;
    }
    MK_HIP(hipSetDevice(c->p.device));
    const uint32_t nq = qs->nq;
    const uint64_t rstride = (uint64_t)cap + 1;
    const uint32_t nblocks = nq >= 4096 ? 4u : 1u;                 // small sets: one ncclGather
    const uint32_t block = std::max<uint32_t>(1, (nq + nblocks - 1) / nblocks);
    // the exchange may start once everything queued so far is done (the merge that read d_recv last, say)
    MK_HIP(hipEventRecord(m->ev_enter, c->stream));
    MK_HIP(hipStreamWaitEvent(m->stream, m->ev_enter, 0));
    uint32_t sent = 0;
    const std::function<int(uint32_t, uint32_t)> hook = [&](uint32_t, uint32_t q1) -> int {
        bool recorded = false;
        while (sent < nq && (sent + block <= q1 || q1 == nq)) {
            const uint32_t b0 = sent, b1 = std::min(nq, sent + block);
            if (!recorded) {                                       // rows [.., q1) are complete once the stream gets here
                MK_HIP(hipEventRecord(m->ev_chunk, c->stream));
                MK_HIP(hipStreamWaitEvent(m->stream, m->ev_chunk, 0));
                recorded = true;
            }
            const uint64_t words = (uint64_t)(b1 - b0) * rstride;
            if (b0 == 0 && b1 == nq) {
                MK_NCCL(g_rccl.Gather(d_rows, d_recv, words, ncclUint64, root, m->comm, m->stream));
            } else {
                MK_NCCL(g_rccl.GroupStart());
                ncclResult_t r = g_rccl.Send(d_rows + (uint64_t)b0 * rstride, words, ncclUint64, root, m->comm, m->stream);
                if (m->rank == root)
                    for (int s = 0; s < m->world && r == ncclSuccess; ++s)
                        r = g_rccl.Recv(d_recv + ((uint64_t)s * nq + b0) * rstride, words, ncclUint64, s, m->comm, m->stream);
                const ncclResult_t e = g_rccl.GroupEnd();
                MK_NCCL(r);
                MK_NCCL(e);
            }
            sent = b1;
        }
        return MK_OK;
    };
    MK_TRY(qset_run(c, qs, nresults, min_score, min_inter, cap, nullptr, nullptr, d_rows, &hook, nblocks));
    if (nq == 0) return MK_OK;
    MK_HIP(hipEventRecord(m->ev_done, m->stream));
    MK_HIP(hipStreamWaitEvent(c->stream, m->ev_done, 0));
    return MK_OK;
}

}  // extern "C"
