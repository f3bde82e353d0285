// FASTA text -> the sequence, on the device (behind gunzip.hip).
//
// index_file_of_file reads a genome file line by line and appends every line that does not start with '>'
// (Miekki.cpp:559-567: getline, `line[0] == '>'`): the sequence is the file's bytes without its line feeds and without
// its header lines, every other byte as it is -- carriage returns, lower case, N, junk.  Whether a byte is kept depends
// on the first byte of ITS line, which may lie any distance before it, so the text is cut into chunks of 4 KiB and the
// question travels as a small summary per piece (is there a line start in it, is the last line started a header, how
// many bytes lie before the first line start, how many kept bytes after it) under an associative combination:
//   fasta_count_kernel    the summary of every chunk (a block scan over its 256 sixteen-byte pieces);
//   fasta_offsets_kernel  one thread per file walks its chunks' summaries: where each chunk's kept bytes go, whether it
//                         starts inside a header line, the sequence's length;
//   fasta_strip_kernel    every chunk again, now with its incoming state: the kept bytes to their places.
#include "mk_internal.hpp"

namespace mk {

namespace {

constexpr uint32_t kChunk = 4096, kPiece = 16;

struct Sum { uint32_t has, st, head, body; };      // line start in it?  is the last started line a header?  bytes before the first
                                                   // line start (line feeds excluded); kept bytes from the first line start on

__device__ __forceinline__ Sum combine(const Sum &a, const Sum &b)          // a's bytes, then b's
{
    Sum r;
    if (!a.has) { r.has = b.has; r.st = b.st; r.head = a.head + b.head; r.body = b.body; }
    else { r.has = 1; r.st = b.has ? b.st : a.st; r.head = a.head; r.body = a.body + (a.st ? 0u : b.head) + b.body; }
    return r;
}

// a piece's sixteen bytes (fewer at the text's end) and the byte before them ('\n' before the text's first)
struct Piece { uint8_t b[kPiece]; uint32_t n; uint8_t prev; };

__device__ __forceinline__ Piece load_piece(const uint8_t *__restrict__ text, uint32_t len, uint32_t at)
{
    Piece p;
    p.n = at < len ? min(kPiece, len - at) : 0u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (p.n) v = *reinterpret_cast<const uint4 *>(text + at);       // (the text's room is a multiple of 16 bytes)
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (uint32_t j = 0; j < kPiece; ++j) p.b[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    p.prev = at && p.n ? text[at - 1] : (uint8_t)'\n';
    return p;
}

__device__ __forceinline__ Sum summarise(const Piece &p)
{
    Sum s{0, 0, 0, 0};
    uint8_t prev = p.prev;
#pragma unroll
    for (uint32_t j = 0; j < kPiece; ++j) {
        if (j < p.n) {
            const uint8_t c = p.b[j];
            if (prev == '\n') { s.has = 1; s.st = c == '>' ? 1u : 0u; }
            if (c != '\n') {
                if (!s.has) ++s.head;
                else if (!s.st) ++s.body;
            }
            prev = c;
        }
    }
    return s;
}

// A summary in one word (a chunk is 4 KiB: both counts fit thirteen bits): bit 0 has, bit 1 st, head from bit 2, body from
// bit 15 -- so that a scan step is ONE lane shuffle.  (Through LDS, sixteen bytes a thread and two barriers a step, the two
// kernels ran at 0.3 TB/s: 8.4 ms to count a batch of 512 genomes, 1.3 ms to strip 64.)
__device__ __forceinline__ uint32_t pack_sum(const Sum &s) { return s.has | (s.st << 1) | (s.head << 2) | (s.body << 15); }
__device__ __forceinline__ Sum unpack_sum(uint32_t w) { return Sum{w & 1u, (w >> 1) & 1u, (w >> 2) & 0x1fffu, (w >> 15) & 0x1fffu}; }
__device__ __forceinline__ uint32_t combine_packed(uint32_t a, uint32_t b) { return pack_sum(combine(unpack_sum(a), unpack_sum(b))); }

// scan of the 256 pieces' summaries: inside a wave by lane shuffles, the four waves' totals through LDS; returns the thread's
// EXCLUSIVE value (identity for thread 0)
__device__ Sum block_scan(Sum mine, uint32_t *wave_tot, Sum *total)
{
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint32_t v = pack_sum(mine);
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v = combine_packed(up, v);
    }
    if (lane == 63u) wave_tot[wave] = v;
    __syncthreads();
    uint32_t before = 0, all = 0;                                    // (the identity packs to zero)
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
        const uint32_t tw = wave_tot[w];
        if (w < wave) before = combine_packed(before, tw);
        all = combine_packed(all, tw);
    }
    *total = unpack_sum(all);
    uint32_t ex = (uint32_t)__shfl_up((int)v, 1u);
    ex = lane ? combine_packed(before, ex) : before;
    __syncthreads();                                                 // (wave_tot may be written again by the caller's next use)
    return unpack_sum(ex);
}

struct ChunkSum { uint32_t has, st, head, body; };
struct ChunkPlace { uint32_t seq_off, in_header; };

__global__ __launch_bounds__(256) void fasta_count_kernel(const uint8_t *__restrict__ text, const mk_gz_stream *__restrict__ jobs,
                                                          const uint32_t *__restrict__ chunk_first, uint32_t n, ChunkSum *__restrict__ sums)
{
    __shared__ uint32_t buf[4];
    // which stream this chunk belongs to: the last one whose first chunk is not beyond it
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (chunk_first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const mk_gz_stream job = jobs[lo];
    const uint32_t at = (blockIdx.x - chunk_first[lo]) * kChunk + threadIdx.x * kPiece;
    const uint32_t len = job.status == MK_GZ_OK ? job.out_len : 0u;
    const Piece p = load_piece(text + job.out_off, len, at);
    Sum total;
    (void)block_scan(summarise(p), buf, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = ChunkSum{total.has, total.st, total.head, total.body};
}

__global__ void fasta_offsets_kernel(const mk_gz_stream *__restrict__ jobs, const uint32_t *__restrict__ chunk_first, uint32_t n,
                                     const ChunkSum *__restrict__ sums, ChunkPlace *__restrict__ places, uint64_t *__restrict__ seq_len)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    uint32_t off = 0, in_header = 0;                                 // (a text starts at a line start: its first piece says which)
    for (uint32_t c = chunk_first[s]; c < chunk_first[s + 1]; ++c) {
        const ChunkSum cs = sums[c];
        places[c] = ChunkPlace{off, in_header};
        off += (in_header ? 0u : cs.head) + cs.body;
        if (cs.has) in_header = cs.st;
    }
    seq_len[s] = jobs[s].status == MK_GZ_OK ? off : 0u;
}

// the kept bytes of up to 64 texts to where the build wants them (its sequence buffer): a call's chunks are the chunks of
// its texts, one text after the other
struct StripCall {
    uint32_t stream[kBuildBatch];          // which texts
    uint32_t chunk0[kBuildBatch];          // each one's first chunk in the batch's numbering (sums / places)
    uint32_t first[kBuildBatch + 1];       // each one's first chunk in the call's numbering
    uint64_t dst_off[kBuildBatch];         // where each sequence starts in dst
    uint32_t n;
};

__global__ __launch_bounds__(256) void fasta_strip_kernel(const uint8_t *__restrict__ text, const mk_gz_stream *__restrict__ jobs, StripCall call,
                                                          const ChunkPlace *__restrict__ places, uint8_t *__restrict__ dst)
{
    __shared__ uint32_t buf[4];
    uint32_t lo = 0, hi = call.n;
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (call.first[mid] <= blockIdx.x) lo = mid; else hi = mid; }
    const mk_gz_stream job = jobs[call.stream[lo]];
    if (job.status != MK_GZ_OK) return;
    const uint32_t local = blockIdx.x - call.first[lo];
    const uint32_t at = local * kChunk + threadIdx.x * kPiece;
    const Piece p = load_piece(text + job.out_off, job.out_len, at);
    const ChunkPlace place = places[call.chunk0[lo] + local];
    Sum total;
    const Sum ex = block_scan(summarise(p), buf, &total);
    // what came before this piece in the chunk: the line it starts in, and how many kept bytes
    uint32_t header = ex.has ? ex.st : place.in_header;
    uint8_t *__restrict__ out = dst + call.dst_off[lo] + place.seq_off + (place.in_header ? 0u : ex.head) + ex.body;
    // a piece without a line feed inside a kept line -- four pieces in five of a text with 80-column lines -- leaves as it came,
    // one 16-byte store (to wherever it goes: no alignment is asked of it); the others byte by byte
    const Sum mine = summarise(p);
    if (p.n == kPiece && !mine.has && p.prev != '\n' && mine.head == kPiece) {
        if (!header) {
            struct __attribute__((packed, aligned(1))) Loose16 { uint32_t w[4]; } o;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) o.w[k] = (uint32_t)p.b[4 * k] | ((uint32_t)p.b[4 * k + 1] << 8) | ((uint32_t)p.b[4 * k + 2] << 16) | ((uint32_t)p.b[4 * k + 3] << 24);
            *reinterpret_cast<Loose16 *>(out) = o;
        }
        return;
    }
    uint8_t prev = p.prev;
#pragma unroll
    for (uint32_t j = 0; j < kPiece; ++j) {
        if (j < p.n) {
            const uint8_t c = p.b[j];
            if (prev == '\n') header = c == '>' ? 1u : 0u;
            if (c != '\n' && !header) *out++ = c;
            prev = c;
        }
    }
}

}  // namespace

// The sequences' lengths (streams whose status is not MK_GZ_OK: 0) and every chunk's place: d_chunk_first[n + 1] = the streams'
// first chunks (prefix sums of ceil(out_len / 4 KiB), made by the caller from the streams it has read back).  Queued on st.
int launch_fasta_count(mk_ctx *c, const uint8_t *d_text, const mk_gz_stream *d_jobs, uint32_t n, const uint32_t *d_chunk_first, uint32_t n_chunks,
                       void *d_scratch, uint64_t *d_seq_len, hipStream_t st)
{
    (void)c;
    if (!n) return MK_OK;
    ChunkSum *sums = reinterpret_cast<ChunkSum *>(d_scratch);
    ChunkPlace *places = reinterpret_cast<ChunkPlace *>(sums + n_chunks);
    if (n_chunks) {
        hipLaunchKernelGGL(fasta_count_kernel, dim3(n_chunks), dim3(256), 0, st, d_text, d_jobs, d_chunk_first, n, sums);
        MK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(fasta_offsets_kernel, dim3((n + 63u) / 64u), dim3(64), 0, st, d_jobs, d_chunk_first, n, sums, places, d_seq_len);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

// m <= 64 of the batch's sequences (streams which[0 .. m), whose first chunks are h_chunk_first[which[i]]) into d_dst at
// dst_off[i]
int launch_fasta_strip(mk_ctx *c, const uint8_t *d_text, const mk_gz_stream *d_jobs, const uint32_t *which, uint32_t m, const uint32_t *h_chunk_first,
                       uint32_t n_chunks, const void *d_scratch, uint8_t *d_dst, const uint64_t *dst_off, hipStream_t st)
{
    (void)c;
    if (!m) return MK_OK;
    if (m > kBuildBatch) { set_error("at most %u sequences per call", kBuildBatch); return MK_ERR_ARG; }
    StripCall call;
    call.n = m;
    call.first[0] = 0;
    for (uint32_t i = 0; i < m; ++i) {
        call.stream[i] = which[i];
        call.chunk0[i] = h_chunk_first[which[i]];
        call.first[i + 1] = call.first[i] + (h_chunk_first[which[i] + 1] - h_chunk_first[which[i]]);
        call.dst_off[i] = dst_off[i];
    }
    if (!call.first[m]) return MK_OK;
    const ChunkPlace *places = reinterpret_cast<const ChunkPlace *>(reinterpret_cast<const ChunkSum *>(d_scratch) + n_chunks);
    hipLaunchKernelGGL(fasta_strip_kernel, dim3(call.first[m]), dim3(256), 0, st, d_text, d_jobs, call, places, d_dst);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

uint64_t fasta_scratch_bytes(uint32_t n_chunks) { return (uint64_t)n_chunks * (sizeof(ChunkSum) + sizeof(ChunkPlace)) + 64; }

}  // namespace mk
