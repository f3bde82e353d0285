// gzip'd FASTA on the GPU (the reference's normal input: index_file_of_file reads every genome
// through zstr::ifstream, Miekki.cpp:559-567, zstr.hpp:78 inflateInit2(15 + 32)).
//
// A deflate stream is serial twice over -- a symbol's first bit is known only when the symbol before it has been decoded, and
// a match copies bytes that must have been produced -- but a collection is thousands of streams, a stream is dozens of
// blocks whose first bits can be FOUND, and decoding splits into a part that needs no history and a part that reads no bits:
//   gz_find_kernel     every bit offset of every file tested for the beginning of a dynamic block's header (type bits, counts,
//   gz_check_kernel    a complete code length code: bit-sliced over a word's 32 offsets, the Kraft sum by population counts);
//                      the few survivors validated in full (both codes by zlib's inflate_table rules).  What is left are the
//                      blocks' real starts (one per 26 KB of gzip'd DNA) and next to no impostors.
//   gz_cand_*          the starts ordered per stream ON THE DEVICE (count, scan over the streams, scatter, rank): the host
//                      never learns how many there are.
//   gz_tokens_kernel   ONE LANE PER SEGMENT -- from a stream's first byte, or from a found start, to the next start the decoding
//                      arrives at between two blocks: member headers, stored / fixed / dynamic blocks, Huffman codes through
//                      per-lane tables in LDS (six- and seven-bit roots, the longer codes' symbols beside them) into 4-byte
//                      TOKENS (a literal, length + distance, a stored run, a member's end).  One pass: a segment's tokens go
//                      to slots that follow from where it starts (a slot per byte of input); workgroups take groups of 64
//                      segments from a counter until none are left.
//   gz_chain_kernel    a thread per stream follows its segments from link to link: token count, text length, members -- the
//                      one thing the host must look at (the text block is sized from it).  A segment that outgrew its slots
//                      (literals only, say) is decoded once more with an exact room.
//   gz_resolve_kernel  ONE WAVE PER STREAM executes the tokens, 64 at a time, in a 36 KiB window in LDS (32 KiB of history +
//                      4 KiB being written): places from a prefix sum of the lengths, copies inside LDS -- a token whose
//                      source lies among the bytes the same step writes waits for a later round of the step --, distances
//                      checked against the member's first byte, finished 4 KiB blocks out in 16-byte stores, their CRC-32
//                      (each lane a 64-byte slice by table, the slices joined by multiplication with powers of x modulo the
//                      CRC polynomial) held against every member's trailer, as is ISIZE.
// The host's part (gz_open / gz_put / gz_finish below): the files' bytes come in as their readers read them -- page-locked
// pieces lent by the context, a DMA per run of files --, then the whole chain is queued back to back on the batch's own
// stream with ONE host look in the middle.  Every loop is bounded by the stream's input length or its rooms, and whatever
// is not a sequence of well-formed gzip members from the first byte to the last -- truncated input, an over-subscribed or
// incomplete code, a distance beyond the member's first byte, a stored block whose length check fails, trailing bytes --
// ends that stream with a status and nothing else: the caller (host/miekki_main.cpp) reads such a file itself, with the
// host's inflater, which then says what the file is worth.  Behind it, fasta.hip strips header lines and line ends.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <thread>
#include <vector>

#include "mk_internal.hpp"

namespace mk {

namespace {

constexpr uint32_t kLitRoot = 6, kDistRoot = 7, kLitSize = 1u << kLitRoot, kDistSize = 1u << kDistRoot;
constexpr uint32_t kTokMatch = 0x80000000u;    // | length << 16 | distance - 1
constexpr uint32_t kTokMember = 0x40000000u;   // a member ends here: the next two words are its CRC-32 and ISIZE
constexpr uint32_t kTokStored = 0x20000000u;   // | n (16 bits): n bytes of a stored block; the next word is where they lie in the stream
constexpr uint32_t kWin = 36u << 10, kHist = 32u << 10, kFlush = 4u << 10;

enum : uint32_t { ST_MEMBER = 0, ST_BLOCK = 1, ST_TOKENS = 2, ST_TRAILER = 4, ST_DONE = 5 };

// ---------------------------------------------------------------- phase 1: bits -> tokens
// LDS of a wave (64 streams), every per-lane array interleaved [index][lane]:
//   lit 64 x u16 (8 KiB), dist 128 x u16 (16 KiB): code length | kind | value, 0 = not in the table.  Six and seven bits of
//   root: in gzip'd DNA 1.6 % of the literal / length symbols and 1.6 % of the distance symbols have longer codes (counted:
//   literals 2-4 bits, lengths 4-7, distances 2-7) and take the bit-by-bit path -- and the whole structure is 78 KB, so TWO
//   waves share a CU's LDS (with nine- and eight-bit roots it was 158 KB: one wave per CU, the kernel's bound)
//   cnt_l / cnt_d 16 x u16: codes per length (the bit-by-bit path); nxt / ofs 16 x u16: scratch of the table build
//   sym_d 32 x u8: distance symbols in code order; lens 352 x u8: the block's code lengths (the code length
//   code's table borrows the lane's column of dist).  The literal / length symbols in code order live in global memory (aux: 288 x u16 per lane).
constexpr uint32_t kInWords = 32, kInPitch = 36, kTokRing = 48, kTokPitch = 52;
struct TokLds {
    uint16_t lit[kLitSize][64];
    uint16_t dist[kDistSize][64];              // (its first 128 rows also serve as the code length code's table while a header is read:
                                               // the lane's own column -- done with before the distance table is built)
    uint16_t cnt_l[16][64], cnt_d[16][64], nxt[16][64], ofs[16][64];
    uint8_t sym_d[32][64];
    // a block's code lengths while its tables are built, one byte each -- [0, 19): the code length code's own; [32, 32 + 316):
    // the two alphabets' (fixed code: [0, 320)) -- and, once they are built, the literal / length symbols whose codes are
    // longer than the root table's index, in code order (long_sym[j]: the j-th of them): a lane's bytes are the same in
    // both views (byte i of a lane = half i & 1 of its word i >> 1), so a lane that reads a header disturbs nobody's list
    uint16_t lens16[176][64];
    // The bit loop touches no global memory: a lane reads its stream from its row of `in` (the next kInWords words of it)
    // and leaves its tokens in its row of `tok`; every few dozen steps each lane refills its row with 16-byte loads and
    // writes its ring out with 16-byte stores (round(), below).  With a load or a store per lane and token in the loop --
    // 64 cache lines per instruction, and every wait for an input word a wait for the token stores before it -- and the
    // rows moved by the whole wave through lane shuffles, a token took 3,100 cycles (profiles/r5_gunzip.txt).
    uint32_t in[64][kInPitch];
    uint32_t tok[64][kTokPitch];
};

__device__ __forceinline__ uint8_t &lens_at(TokLds &L, uint32_t i, uint32_t lane) { return reinterpret_cast<uint8_t *>(&L.lens16[i >> 1][lane])[i & 1u]; }
constexpr uint32_t kLongSyms = 176;            // literal / length symbols with long codes a lane keeps in LDS (the rest: its list in memory)

struct BitReader {
    const uint32_t *in32;          // the stream's words (16-byte aligned, zero padding behind the last byte)
    const TokLds *lds;             // (its `in` row `lane` holds words [base, base + kInWords) of the stream)
    uint32_t lane;
    uint64_t bb;
    uint32_t bc, wi, wmax, base;   // words [0, wi) are in bb
    bool over;                     // asked for words beyond the padding: the stream is truncated
    __device__ __forceinline__ void start(const uint8_t *p, uint32_t in_len, const TokLds *l, uint32_t ln)
    {
        in32 = reinterpret_cast<const uint32_t *>(p);
        lds = l; lane = ln;
        wmax = (in_len + 3u) / 4u + 3u;                            // (the upload pads 16 zero bytes)
        bb = 0; bc = 0; wi = 0; over = false;
        base = 0xffffffffu - kInWords;                             // (nothing buffered yet)
    }
    __device__ __forceinline__ void refill()                       // bc <= 32 on entry
    {
        uint32_t w = 0;
        if (wi >= wmax) over = true;
        else if (wi - base < kInWords) w = lds->in[lane][wi - base];
        else w = in32[wi];                                         // beyond the buffered words (a block's header, a skip): from memory
        bb |= (uint64_t)w << bc;
        bc += 32u;
        ++wi;
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n)           // n <= 32
    {
        if (bc < n) refill();
        return (uint32_t)bb & (n >= 32u ? 0xffffffffu : ((1u << n) - 1u));
    }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint32_t get(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
    __device__ __forceinline__ uint64_t consumed_bits() const { return (uint64_t)wi * 32u - bc; }
    __device__ __forceinline__ void align_byte() { drop(bc & 7u); }
    __device__ void seek_byte(uint64_t at)                          // continue at byte `at` of the stream
    {
        bb = 0; bc = 0;
        const uint64_t w = at / 4u;
        if (w >= wmax) { over = true; wi = wmax; return; }
        wi = (uint32_t)w;
        refill();
        drop((uint32_t)(at % 4u) * 8u);
    }
};

// canonical code of `n` symbols (lengths in the block's list from `base` on) into a root table and the counts of the bit-by-bit path.
// kind: 0 literal / length alphabet, 1 distance alphabet, 2 the code length code (complete codes only).
// Returns false for what zlib's inflate_table refuses: an over-subscribed set, an incomplete one unless it is a single
// code of one bit (or, for the two data alphabets, no code at all).
__device__ bool build_code(TokLds &L, uint32_t lane, uint32_t base, uint32_t n, int kind, uint32_t *aux_sorted)
{
    const uint32_t col = ((lane & 31u) << 1) | (lane >> 5);       // (the lane's column of the two-byte tables, see the kernel)
    uint16_t (*cnt)[64] = kind == 1 ? L.cnt_d : L.cnt_l;
    for (uint32_t l = 0; l < 16; ++l) cnt[l][lane] = 0;
    uint32_t maxlen = 0;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens_at(L, base + s, lane);
        if (l) { cnt[l][lane] = (uint16_t)(cnt[l][lane] + 1u); maxlen = max(maxlen, l); }
    }
    int left = 1;
    uint32_t code = 0, off = 0;
    for (uint32_t l = 1; l < 16; ++l) {
        const uint32_t c = cnt[l][lane];
        left = (left << 1) - (int)c;
        if (left < 0) return false;                                // over-subscribed
        L.nxt[l][lane] = (uint16_t)code;                           // first code of this length
        L.ofs[l][lane] = (uint16_t)off;
        code = (code + c) << 1;
        off += c;
    }
    if (left > 0 && (kind == 2 || maxlen > 1u)) return false;      // incomplete (inftrees.c: only a lone one-bit code may be)
    if (kind == 2 && maxlen == 0) return false;
    const uint32_t root = kind == 0 ? kLitRoot : kind == 1 ? kDistRoot : 7u, size = 1u << root;
    if (kind == 0) for (uint32_t i = 0; i < size; ++i) L.lit[i][col] = 0;
    else if (kind == 1) for (uint32_t i = 0; i < size; ++i) L.dist[i][col] = 0;
    else for (uint32_t i = 0; i < size; ++i) L.dist[i][col] = 0;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens_at(L, base + s, lane);
        if (!l) continue;
        const uint32_t c = L.nxt[l][lane];
        L.nxt[l][lane] = (uint16_t)(c + 1u);
        const uint32_t o = L.ofs[l][lane];
        L.ofs[l][lane] = (uint16_t)(o + 1u);
        if (kind == 0) aux_sorted[o] = s;
        else if (kind == 1) L.sym_d[o & 31u][lane] = (uint8_t)s;
        if (l > root) continue;                                    // (the code length code has no longer ones)
        uint32_t e;
        if (kind == 0) {
            // code length | kind << 4 | value << 7: kind 7 a literal (value = the byte), 6 end of block, 0..5 a match
            // length with that many extra bits (value = its base, RFC 1951 3.2.5)
            if (s < 256u) e = l | (7u << 4) | (s << 7);
            else if (s == 256u) e = l | (6u << 4);
            else if (s > 285u) continue;                           // (286, 287: in the fixed code, never valid -- left to the slow path's check)
            else {
                const uint32_t c2 = s - 257u;
                const uint32_t ex = c2 < 8u || c2 == 28u ? 0u : (c2 >> 2) - 1u;
                const uint32_t lb = c2 < 8u ? 3u + c2 : c2 == 28u ? 258u : 3u + ((4u + (c2 & 3u)) << ex);
                e = l | (ex << 4) | (lb << 7);
            }
        } else if (kind == 1) {
            if (s > 29u) continue;
            e = l | (s << 4);
        } else {
            e = s | (l << 5);
        }
        const uint32_t r = __brev(c) >> (32u - l);
        if (kind == 0) for (uint32_t i = r; i < size; i += 1u << l) L.lit[i][col] = (uint16_t)e;
        else if (kind == 1) for (uint32_t i = r; i < size; i += 1u << l) L.dist[i][col] = (uint16_t)e;
        else for (uint32_t i = r; i < size; i += 1u << l) L.dist[i][col] = (uint16_t)e;
    }
    return true;
}

// A symbol the root table does not hold is found bit by bit against the counts of the longer lengths (the canonical walk of
// RFC 1951 3.2.2) -- from registers: a wave waits for its slowest lane, and with 64 lanes ONE of them meets a
// code longer than the root table's index in most passes (1.6 % of the symbols each: 86 % of the passes with six- and seven-bit
// roots) -- the walk above, fifteen dependent LDS reads, then cost every pass its time.  A lane keeps the counts of the
// lengths beyond the root in registers (loaded when its block's tables are built) and where the walk stands after the root's
// bits (first code and place in the sorted list); the loop is unrolled, nothing is indexed by a variable.
template <uint32_t ROOT>
struct LongCodes {
    uint32_t cnt[15u - ROOT];       // codes of length ROOT + 1 + i
    uint32_t first, index;          // the walk's state after ROOT lengths (before the shift that follows them)
    __device__ __forceinline__ void load(const uint16_t (*c)[64], uint32_t lane)
    {
        uint32_t f = 0, ix = 0;
#pragma unroll
        for (uint32_t l = 1; l <= ROOT; ++l) { const uint32_t n = c[l][lane]; ix += n; f += n; f <<= 1; }
        first = f; index = ix;
#pragma unroll
        for (uint32_t i = 0; i < 15u - ROOT; ++i) cnt[i] = c[ROOT + 1u + i][lane];
    }
    // bits: the next fifteen bits of the stream (first bit lowest); returns the symbol's place in the sorted list and its
    // length, or length 0: no such code.  `wanted`: this lane is one of those that look (the others idle along); the walk
    // ends when every lane that looks has its code -- the long codes of real streams are a bit or three longer than the
    // root, and the nine steps to fifteen bits were a quarter of the kernel's instructions.
    __device__ __forceinline__ uint32_t find(uint32_t bits, uint32_t &len, bool wanted) const
    {
        uint32_t code = (__brev(bits) >> (32u - ROOT)) << 1, f = first, ix = index, at = 0;
        len = 0;
#pragma unroll
        for (uint32_t i = 0; i < 15u - ROOT; ++i) {
            code |= (bits >> (ROOT + i)) & 1u;
            const uint32_t n = cnt[i];
            const bool hit = len == 0u && code - f < n;
            at = hit ? ix + (code - f) : at;
            len = hit ? ROOT + 1u + i : len;
            ix += n; f += n; f <<= 1; code <<= 1;
            if (!__any(wanted && len == 0u)) break;
        }
        return at;
    }
};

// One lane per SEGMENT of a stream: from the stream's first byte, or from a block's first bit that the finder (below) has
// vouched for, up to the next such bit that the decoding ARRIVES at between two blocks (or to the stream's end).  A segment
// whose start was no block's start decodes noise until an error or until it falls into step with the real blocks: nobody
// follows the chain through it.
// ONE pass: a segment's tokens go to slots that follow from where it starts -- one 4-byte slot per byte of the stream, the
// segment that starts at bit b of stream s writes from slot (in_off(s) + b / 8) & ~3 on, up to the next candidate's slot:
// gzip'd DNA spends 12-15 bits on a token, so a segment fills about half of its slots.  One that would need more (literals
// only, a run of one-bit codes; or a candidate inside it that was no block's start) keeps COUNTING without writing; the chain
// kernel sees n_tok > tok_cap, and those few segments are decoded again with exact rooms (`jobs`).
// The workgroups are a fixed number (two per CU: the LDS each needs) and take groups of 64 segments from a counter until
// there are none left -- the number of segments (streams + candidates that passed the check) lives on the device.
struct TokArgs {
    const uint8_t *gz;
    const mk_gz_stream *streams;
    uint32_t n;                      // streams: segment g < n is stream g from its first byte
    const uint64_t *sorted;          // candidates (stream << 40 | bit), ascending: segment n + j starts at sorted[j]
    const uint32_t *n_good;          // how many (on the device), at most good_cap
    uint32_t good_cap;
    mk_gz_seg *segs;                 // results, by segment
    uint32_t *slots;                 // the token slots (one per byte of the input block)
    uint32_t *aux;                   // 288 words per lane of the grid: a lane's literal / length symbols in code order
    uint32_t *queue;                 // the counter groups are taken from (zero at launch)
    const mk_gz_job *jobs;           // null, or the segments to decode (again), each with its room
    uint32_t n_jobs;
};
constexpr uint64_t kBitMask = (1ull << 40) - 1ull;
constexpr uint32_t kOutMax = 0xfff00000u;       // a stream's text and a segment's share of it stay below this

__global__ __launch_bounds__(64) void gz_tokens_kernel(TokArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TokLds &L = *reinterpret_cast<TokLds *>(smem);
    const uint32_t lane = threadIdx.x;
    // a lane's column in the two-byte tables: lanes l and l + 32 share a word -- they are served in different halves of a wave's
    // access, so 32 lanes looking 32 different rows up meet on no bank (with columns in lane order neighbours always did: 64 % of
    // the LDS cycles were conflicts)
    const uint32_t col = ((lane & 31u) << 1) | (lane >> 5);
    uint32_t *const sorted_syms = A.aux + ((uint64_t)blockIdx.x * 64u + lane) * 288u;
    const uint32_t total = A.jobs ? A.n_jobs : A.n + min(*A.n_good, A.good_cap);
    for (;;) {                                                       // (every wave leaves when the counter has passed the last group)
    uint32_t grp = 0;
    if (lane == 0) grp = atomicAdd(A.queue, 1u);
    grp = (uint32_t)__builtin_amdgcn_readfirstlane((int)grp);
    if ((uint64_t)grp * 64u >= total) break;
    const uint32_t idx = grp * 64u + lane;
    bool live = idx < total;
    const uint32_t g = live ? (A.jobs ? A.jobs[idx].seg : idx) : 0u;
    const uint64_t key = !live || g < A.n ? 0ull : A.sorted[g - A.n];
    const uint32_t sidx = !live ? 0u : g < A.n ? g : (uint32_t)(key >> 40);
    const mk_gz_stream st = A.streams[min(sidx, A.n - 1u)];
    if (live && (sidx >= A.n || st.status != MK_GZ_OK)) live = false;   // (a stream refused before anything ran has no segments)
    const uint64_t start_bit = key & kBitMask;
    const uint32_t cand_hi = st.cand_hi;
    uint32_t ci = g < A.n ? st.cand_lo : g - A.n + 1u;               // the next candidate of the stream behind this segment's start
    // the segment's slots
    uint32_t *tok_base;
    uint32_t cap;
    if (A.jobs) {
        tok_base = reinterpret_cast<uint32_t *>(live ? A.jobs[idx].tok_ptr : 0ull);
        cap = live ? A.jobs[idx].tok_cap : 0u;
    } else {
        const uint64_t s0 = (st.in_off + (start_bit >> 3)) & ~3ull;
        const uint64_t s1 = live && ci < cand_hi ? (st.in_off + ((A.sorted[ci] & kBitMask) >> 3)) & ~3ull
                                                 : st.in_off + ((uint64_t)st.in_len + 31u) / 16u * 16u;     // (the next stream's first slot)
        tok_base = A.slots + s0;
        cap = live ? (uint32_t)(s1 - s0) : 0u;
    }
    LongCodes<kLitRoot> long_l;                                   // (the lane's block: set with its tables)
    LongCodes<kDistRoot> long_d;
    long_l.first = long_l.index = long_d.first = long_d.index = 0;
    for (uint32_t i = 0; i < 15u - kLitRoot; ++i) long_l.cnt[i] = 0;
    for (uint32_t i = 0; i < 15u - kDistRoot; ++i) long_d.cnt[i] = 0;
    BitReader br;
    br.start(A.gz + st.in_off, live ? st.in_len : 0u, &L, lane);
    uint32_t state = live ? ST_MEMBER : ST_DONE, status = live ? MK_GZ_OK : MK_GZ_EMPTY;
    uint32_t n_tok = 0, n_out = 0, out_len = 0, members = 0;        // n_out: tokens that have left the ring (a multiple of four)
    uint32_t link = 0xffffffffu;                                     // the segment this one stops at
    bool final_block = false;
    if (live && start_bit == 0ull && st.in_len < 18u) { state = ST_DONE; status = MK_GZ_NOT_GZIP; }
    if (live && start_bit != 0ull) {
        br.seek_byte(start_bit >> 3);
        br.drop((uint32_t)(start_bit & 7u));
        state = br.over ? ST_DONE : ST_BLOCK;
        if (br.over) status = MK_GZ_TRUNCATED;
    }
    auto fail = [&](uint32_t why) { status = why; state = ST_DONE; };
    auto emit = [&](uint32_t t) { L.tok[lane][n_tok - n_out] = t; ++n_tok; };   // (the passes below emit at most three; round() keeps that much room)
    // The symbols with codes longer than the root's index, from the lane's list in memory (build_code has just written it)
    // into LDS, where the block's lengths lay: a wave meets such a code in two passes of three (some lane of its 64), and a
    // load from memory there -- a microsecond -- was a third of the kernel's time.
    uint32_t n_short = 0;                                           // codes no longer than the root's index: they come first in the list
    auto keep_long = [&]() {
        n_short = long_l.index;
        uint32_t n_long = 0;
#pragma unroll
        for (uint32_t i = 0; i < 15u - kLitRoot; ++i) n_long += long_l.cnt[i];
        const uint32_t m = min(n_long, kLongSyms);
        for (uint32_t j = 0; j < m; ++j)
            L.lens16[j][lane] = (uint16_t)__hip_atomic_load(&sorted_syms[n_short + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // A lane's ring out to memory, four tokens per store (what is left over moves to the ring's start) -- as far as the
    // segment's slots go: the tokens beyond them are only counted --, and its row of input refilled from the word it reads
    // next, four words per load.  Every lane its own stream: 64 cache lines per instruction, but only a few dozen
    // instructions per round and a round every few dozen tokens.
    const uint4 *__restrict__ in16 = reinterpret_cast<const uint4 *>(A.gz + st.in_off);
    auto round = [&]() {
        const uint32_t have = n_tok - n_out, whole = have & ~3u;
        for (uint32_t i = 0; __any(i < whole); i += 4u)
            if (i < whole && n_out + i + 4u <= cap) {
                uint4 v;
                v.x = L.tok[lane][i]; v.y = L.tok[lane][i + 1u]; v.z = L.tok[lane][i + 2u]; v.w = L.tok[lane][i + 3u];
                *reinterpret_cast<uint4 *>(tok_base + n_out + i) = v;
            }
        if (whole) for (uint32_t i = 0; i < have - whole; ++i) L.tok[lane][i] = L.tok[lane][whole + i];
        n_out += whole;
        br.base = br.wi & ~3u;
        uint4 v[kInWords / 4u];
#pragma unroll
        for (uint32_t i = 0; i < kInWords / 4u; ++i) {
            const uint32_t at = br.base + 4u * i;
            // (a stream's room is a multiple of 16 bytes beyond wmax words: a whole piece)
            v[i] = at < br.wmax ? in16[at >> 2] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t i = 0; i < kInWords / 4u; ++i) *reinterpret_cast<uint4 *>(&L.in[lane][4u * i]) = v[i];
    };
    round();
    // every pass of this loop consumes input bits, emits a token or ends a lane: bounded by the streams' bits
    while (__any(state != ST_DONE)) {
        // a round when some ring may not take another pass's tokens, or some lane has used most of its buffered words
        if (__any(n_tok - n_out + 4u > kTokRing || (state != ST_DONE && br.wi - br.base + 4u > kInWords))) round();
        if (__all(state == ST_TOKENS || state == ST_DONE)) {
            // ---- the loop the time goes into: one symbol (a literal, a match or a block's end) per lane and pass, without a
            // branch but for the rare ones -- a code longer than the table's index, an error, a block's end, a round.  Lanes
            // that are done idle along.
            uint32_t pre = L.in[lane][min(br.wi - br.base, kInWords - 1u)];     // the word the next refill takes
            for (;;) {
                const bool act = state == ST_TOKENS;
                uint32_t err = 0;
                auto refill = [&](bool want) {
                    const bool need = want && br.bc <= 32u;
                    if (need && br.wi >= br.wmax) err = MK_GZ_TRUNCATED;
                    br.bb |= need ? (uint64_t)pre << br.bc : 0ull;
                    br.bc += need ? 32u : 0u;
                    br.wi += need ? 1u : 0u;
                    pre = L.in[lane][min(br.wi - br.base, kInWords - 1u)];
                };
                refill(act);
                const uint32_t e = L.lit[(uint32_t)br.bb & (kLitSize - 1u)][col];
                uint32_t kind = (e >> 4) & 7u, value = e >> 7, cl = e & 15u;
                if (act && e == 0u) {                                  // a code longer than the table's index
                    uint32_t ll;
                    const uint32_t at = long_l.find((uint32_t)br.bb & 0x7fffu, ll, true);
                    // (the list was written by this lane through the L2; an L1 line of it from an earlier block would be stale)
                    const uint32_t j = at - n_short;                   // (a long code's place in the list is behind the short ones)
                    const uint32_t sym = !ll ? ~0u : j < kLongSyms ? (uint32_t)L.lens16[min(j, kLongSyms - 1u)][lane]
                                                                   : __hip_atomic_load(&sorted_syms[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    br.bb >>= ll; br.bc -= ll;
                    cl = 0;
                    if (sym < 256u) { kind = 7; value = sym; }
                    else if (sym == 256u) { kind = 6; value = 0; }
                    else if (sym <= 285u) {
                        const uint32_t c2 = sym - 257u;
                        kind = c2 < 8u || c2 == 28u ? 0u : (c2 >> 2) - 1u;
                        value = c2 < 8u ? 3u + c2 : c2 == 28u ? 258u : 3u + ((4u + (c2 & 3u)) << kind);
                    } else { kind = 7; value = 0; err = MK_GZ_BAD_CODE; }
                    pre = L.in[lane][min(br.wi - br.base, kInWords - 1u)];
                }
                br.bb >>= (act ? cl : 0u); br.bc -= (act ? cl : 0u);
                const bool is_len = act && kind < 6u, is_lit = act && kind == 7u, is_eob = act && kind == 6u;
                const uint32_t ex = is_len ? kind : 0u;
                const uint32_t len = value + ((uint32_t)br.bb & ((1u << ex) - 1u));
                br.bb >>= ex; br.bc -= ex;
                refill(is_len);
                const uint32_t d = L.dist[(uint32_t)br.bb & (kDistSize - 1u)][col];
                uint32_t dsym = d >> 4, dcl = d & 15u;
                if (is_len && d == 0u) {                               // a distance code longer than the table's index
                    uint32_t ll;
                    const uint32_t at = long_d.find((uint32_t)br.bb & 0x7fffu, ll, true);
                    dsym = ll ? (uint32_t)L.sym_d[at & 31u][lane] : ~0u;
                    br.bb >>= ll; br.bc -= ll;
                    dcl = 0;
                }
                if (is_len && dsym > 29u) { err = MK_GZ_BAD_CODE; dsym = 0; }
                br.bb >>= (is_len ? dcl : 0u); br.bc -= (is_len ? dcl : 0u);
                const uint32_t dex = !is_len || dsym < 4u ? 0u : (dsym >> 1) - 1u;
                const uint32_t dist = (dsym < 4u ? dsym + 1u : 1u + ((2u + (dsym & 1u)) << dex)) + ((uint32_t)br.bb & ((1u << dex) - 1u));
                br.bb >>= dex; br.bc -= dex;
                const uint32_t grow = is_len ? len : is_lit ? 1u : 0u;
                // (whether a distance reaches back beyond its member's first byte is the text kernel's to say: only there is it
                // known how much of the member lies before this segment)
                if (grow > kOutMax - out_len) err = err ? err : MK_GZ_OUTPUT_ROOM;
                const bool put = grow != 0u && err == 0u;
                L.tok[lane][put ? n_tok - n_out : kTokRing] = is_len ? kTokMatch | (len << 16) | (dist - 1u) : value;   // (column kTokRing: nobody reads it)
                n_tok += put ? 1u : 0u;
                out_len += put ? grow : 0u;
                if (err) { status = err; state = ST_DONE; }
                else if (is_eob) state = final_block ? ST_TRAILER : ST_BLOCK;
                // leave the loop for a block's end or an error somewhere, and for a round
                if (__any(state != (act ? ST_TOKENS : ST_DONE) || n_tok - n_out + 4u > kTokRing || (act && br.wi - br.base + 4u > kInWords))) break;
            }
            continue;
        }
        if (state == ST_TOKENS) {
            // (some other lane reads a header or a trailer in this pass: the symbol loop waits for it)
        } else if (state == ST_BLOCK && [&] {
                       // between two blocks: has the decoding arrived at a start that another segment decodes from?
                       const uint64_t here = br.consumed_bits();
                       if (here == start_bit) return false;
                       while (ci < cand_hi && (A.sorted[ci] & kBitMask) < here) ++ci;
                       return ci < cand_hi && (A.sorted[ci] & kBitMask) == here;
                   }()) {
            link = A.n + ci;
            state = ST_DONE;                                         // (status stays OK: the chain goes on in that segment)
        } else if (state == ST_BLOCK) {
            final_block = br.get(1) != 0;
            const uint32_t type = br.get(2);
            if (type == 0u) {
                br.align_byte();
                const uint32_t len = br.get(16), nlen = br.get(16);
                const uint64_t at = br.consumed_bits() >> 3;        // (byte-aligned: the block's bytes lie here as they are)
                if ((len ^ nlen) != 0xffffu) fail(MK_GZ_BAD_STORED);
                else if (br.over || at + len > st.in_len) fail(MK_GZ_TRUNCATED);
                else if (len > kOutMax - out_len) fail(MK_GZ_OUTPUT_ROOM);
                else {
                    if (len) { emit(kTokStored | len); emit((uint32_t)at); out_len += len; }
                    br.seek_byte(at + len); state = final_block ? ST_TRAILER : ST_BLOCK;
                }
            } else if (type == 1u) {
                for (uint32_t i = 0; i < 288u; ++i) lens_at(L, i, lane) = (uint8_t)(i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : 8);
                for (uint32_t i = 0; i < 32u; ++i) lens_at(L, 288u + i, lane) = 5;
                (void)build_code(L, lane, 0, 288, 0, sorted_syms);
                (void)build_code(L, lane, 288, 32, 1, sorted_syms);      // (codes 30 and 31 decode to an error, as in zlib's fixed table)
                long_l.load(L.cnt_l, lane); long_d.load(L.cnt_d, lane);
                keep_long();
                state = ST_TOKENS;
            } else if (type == 2u) {
                const uint32_t hlit = br.get(5) + 257u, hdist = br.get(5) + 1u, hclen = br.get(4) + 4u;
                bool ok = hlit <= 286u && hdist <= 30u;
                for (uint32_t i = 0; i < 19u; ++i) lens_at(L, i, lane) = 0;
                for (uint32_t i = 0; i < hclen; ++i) {
                    // whose length comes i-th (RFC 1951 3.2.7): 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                    const uint32_t j = i - 4u;
                    const uint32_t which = i < 3u ? 16u + i : i == 3u ? 0u : (j & 1u) ? 7u - (j >> 1) : 8u + (j >> 1);
                    lens_at(L, which, lane) = (uint8_t)br.get(3);
                }
                ok = ok && build_code(L, lane, 0, 19, 2, sorted_syms);
                uint32_t i = 0, prev = 0;
                const uint32_t total_lens = hlit + hdist;
                // the lengths of both alphabets, run-length coded (at most `total_lens` passes: every pass writes a length)
                while (ok && i < total_lens) {
                    const uint32_t ce = L.dist[br.peek(7)][col];
                    if (!ce) { ok = false; break; }
                    br.drop(ce >> 5);
                    const uint32_t sym = ce & 31u;
                    if (sym < 16u) { lens_at(L, 32u + i, lane) = (uint8_t)sym; prev = sym; ++i; }
                    else {
                        uint32_t rep, val = 0;
                        if (sym == 16u) { if (!i) { ok = false; break; } val = prev; rep = 3u + br.get(2); }
                        else if (sym == 17u) rep = 3u + br.get(3);
                        else rep = 11u + br.get(7);
                        if (i + rep > total_lens) { ok = false; break; }
                        for (uint32_t r = 0; r < rep; ++r) lens_at(L, 32u + i + r, lane) = (uint8_t)val;
                        i += rep; prev = val;
                    }
                    if (br.over) ok = false;
                }
                // (the lengths lie at lens[32 ...]: the code length code's own nineteen stay below them)
                ok = ok && lens_at(L, 32u + 256u, lane) != 0;             // no end-of-block code: inflate.c "missing end-of-block"
                ok = ok && build_code(L, lane, 32, hlit, 0, sorted_syms);
                ok = ok && build_code(L, lane, 32u + hlit, hdist, 1, sorted_syms);
                if (!ok) fail(br.over ? MK_GZ_TRUNCATED : MK_GZ_BAD_LENGTHS);
                else { long_l.load(L.cnt_l, lane); long_d.load(L.cnt_d, lane); keep_long(); state = ST_TOKENS; }
            } else {
                fail(MK_GZ_BAD_BLOCK);
            }
            if (br.over && state != ST_DONE) fail(MK_GZ_TRUNCATED);
        } else if (state == ST_TRAILER) {
            br.align_byte();
            const uint32_t crc = br.get(32), isize = br.get(32);
            emit(kTokMember);
            emit(crc);
            emit(isize);
            ++members;
            const uint64_t used = br.consumed_bits() >> 3;          // whole bytes: the reader is byte-aligned here
            if (br.over || used > st.in_len) fail(MK_GZ_TRUNCATED);
            else if (used == st.in_len) state = ST_DONE;              // the stream ends with this member: status stays OK
            else state = ST_MEMBER;                                   // another member must follow (anything else: a status)
        } else if (state == ST_MEMBER) {
            // RFC 1952 2.3: ID1 ID2 CM FLG MTIME(4) XFL OS [XLEN + extra] [name 0] [comment 0] [CRC16]
            const uint64_t left = (uint64_t)st.in_len - (br.consumed_bits() >> 3);
            const bool first = start_bit == 0ull && members == 0u && n_tok == 0u;
            if (left < 18u) fail(first ? MK_GZ_NOT_GZIP : MK_GZ_TRAILING);
            else {
                const uint32_t id = br.get(16), cm = br.get(8), flg = br.get(8);
                (void)br.get(32); (void)br.get(16);
                if (id != 0x8b1fu || cm != 8u || (flg & 0xe0u)) fail(first ? MK_GZ_NOT_GZIP : MK_GZ_TRAILING);
                else {
                    uint64_t budget = st.in_len;                      // bytes these loops may skip at most
                    if (flg & 4u) {
                        uint32_t xlen = br.get(16);
                        while (xlen && budget && !br.over) { (void)br.get(8); --xlen; --budget; }
                    }
                    for (uint32_t f = 8u; f <= 16u; f <<= 1)          // name, comment: zero-terminated
                        if (flg & f) while (budget && !br.over && br.get(8) != 0u) --budget;
                    if (flg & 2u) (void)br.get(16);
                    if (br.over || !budget || (br.consumed_bits() >> 3) + 8u > st.in_len) fail(MK_GZ_TRUNCATED);
                    else state = ST_BLOCK;
                }
            }
        }
    }
    round();                                                         // what is left in the rings: whole fours, then the rest
    for (uint32_t i = 0; i < n_tok - n_out; ++i) if (n_out + i < cap) tok_base[n_out + i] = L.tok[lane][i];
    if (idx < total && g < A.n + A.good_cap) {
        mk_gz_seg r;
        r.tok_ptr = (uint64_t)tok_base; r.tok_cap = cap; r.n_tok = n_tok; r.out_len = out_len;
        r.status = live ? status : (uint32_t)MK_GZ_EMPTY; r.members = members; r.link = link;
        A.segs[g] = r;
    }
    }
}

// ---------------------------------------------------------------- where blocks start
// A deflate stream says nowhere where its blocks start, but a dynamic block's header is hard to imitate: the type bits, at
// most 286 + 30 codes, a COMPLETE code length code, code lengths that decode without a repeat before the first length or
// across the end, an end-of-block code, and two complete codes (or a single distance code).  Random bits pass the first part
// (gz_find_kernel: type, counts, the code length code's Kraft sum) at one offset in a thousand and the second
// (gz_check_kernel) practically never: on gzip'd genomes exactly the blocks' real starts are left (59 of 59, no other offset
// of 12.4 million).  A start that is missed -- a fixed or stored block, a file this test is too strict for -- only means that
// the segment before it decodes on through it; a false one only costs a lane that decodes noise.
__device__ __forceinline__ uint64_t bits_at(const uint8_t *__restrict__ p, uint64_t bit)       // 57 bits from `bit` on
{
    const uint64_t b = bit >> 3;
    uint64_t v;
    __builtin_memcpy(&v, p + b, 8);                                 // (the streams have 16 bytes of padding behind them)
    return v >> (bit & 7u);
}

constexpr uint32_t kFindTiles = 16;

__global__ __launch_bounds__(256) void gz_find_kernel(const uint8_t *__restrict__ gz, const mk_gz_stream *__restrict__ streams, const uint64_t *__restrict__ word_first,
                                                      uint32_t n, uint64_t *__restrict__ hits, uint32_t *__restrict__ n_hits, uint32_t cap)
{
    // a thread: the 32 bit offsets of one 4-byte word of one stream.  Two steps, because they differ a hundredfold in cost and
    // in how many offsets take them: (1) every offset -- type bits, the two counts, room for a header: a handful of
    // instructions, a fifth of the offsets pass; (2) the code length code's Kraft sum -- a loop over up to nineteen lengths --
    // for those only, DENSELY: the wave's survivors are listed in LDS (a prefix sum over the lanes' counts) and walked 64 at a
    // time with every lane busy, each from the four words its header can touch (words w .. w + 3 of the stream, kept in LDS
    // by the lanes that own them).  (While each lane ran the loop for its own survivors the wave ran it for every offset --
    // some lane always has one -- at a fifth of its lanes: 47 ms per 1,024 genomes.)
    // The offsets that pass are collected in LDS and leave with ONE atomic per workgroup (an atomic each on one counter --
    // twelve million of them for a thousand genomes -- took 130 ms by itself).
    __shared__ uint64_t s_hits[512];
    __shared__ uint32_t s_n, s_base;
    __shared__ uint32_t s_w[4][64 + 4];                              // a wave's words: lane l's first, and three more behind the last lane's
    __shared__ uint32_t s_lo[4][64];                                 // whose stream a lane's word is
    __shared__ uint32_t s_wi[4][64];                                 // which word of its stream it is
    __shared__ uint16_t s_list[4][64 * 32];                          // the wave's survivors of step 1: lane << 5 | offset
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // (kFindTiles tiles of 256 words per workgroup: the hits of all of them leave with one atomic on the one counter -- at a
    // tile per workgroup that counter took 1.5 million atomics for a thousand genomes, 12 ns each: the kernel's whole time)
    for (uint32_t tile = 0; tile < kFindTiles; ++tile) {
    const uint64_t w = ((uint64_t)blockIdx.x * kFindTiles + tile) * 256u + threadIdx.x;
    if (((uint64_t)blockIdx.x * kFindTiles + tile) * 256u >= word_first[n]) break;                 // (the same for every thread)
    const bool live = w < word_first[n];
    // whose word: the last stream that starts at or before it.  Searched ONCE per wave, for the wave's first word, on scalar
    // registers (ten dependent loads from the scalar cache instead of ten dependent vector loads per lane -- the kernel's
    // time was that chain); a lane whose word lies in a later stream steps forward from there
    uint32_t lo = 0;
    {
        const uint64_t w_wave = ((uint64_t)blockIdx.x * kFindTiles + tile) * 256u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
        uint32_t a = 0, b = n;
        while (b - a > 1u) { const uint32_t mid = (a + b) / 2u; if (word_first[mid] <= w_wave) a = mid; else b = mid; }
        lo = a;
    }
    while (lo + 1u < n && word_first[lo + 1u] <= w) ++lo;
    const mk_gz_stream st = streams[lo];
    const uint64_t wi = live ? w - word_first[lo] : 0u;
    const uint32_t *__restrict__ p = reinterpret_cast<const uint32_t *>(gz + st.in_off) + wi;
    const uint64_t bit0 = wi * 32u, nbits = live ? (uint64_t)st.in_len * 8u : 0u;
    const bool look = bit0 + 80u <= nbits;                           // (a block and a trailer need more than that)
    const uint32_t w0 = live ? p[0] : 0u, w1 = live ? p[1] : 0u;     // (within the stream's padded room: sixteen bytes behind its end)
    s_w[wave][lane] = w0;
    if (lane == 63u) { s_w[wave][64] = w1; s_w[wave][65] = live ? p[2] : 0u; s_w[wave][66] = live ? p[3] : 0u; }
    s_lo[wave][lane] = lo;
    s_wi[wave][lane] = (uint32_t)wi;
    // ---- step 1: which of the word's 32 offsets could start a dynamic block -- all 32 at once: bit o of (w1:w0) >> k is bit
    // o + k of the stream, so "BTYPE = 2" (bit o + 1 clear, bit o + 2 set), "HLIT <= 29" (not all of bits o + 4 .. o + 7 set)
    // and "HDIST <= 29" (not all of o + 9 .. o + 12) are a dozen word operations instead of a dozen per offset
    uint32_t pass;
    {
        auto sh = [&](uint32_t k) { return __builtin_amdgcn_alignbit(w1, w0, k); };
        const uint32_t dyn = ~sh(1) & sh(2);
        const uint32_t hlit30 = sh(4) & sh(5) & sh(6) & sh(7), hdist30 = sh(9) & sh(10) & sh(11) & sh(12);
        pass = dyn & ~hlit30 & ~hdist30;
    }
    if (!look) pass = 0;
    else if (bit0 + 31u + 80u > nbits) {                             // (the stream's last words: offset by offset)
        for (uint32_t o = 0; o < 32u; ++o) if (bit0 + o + 80u > nbits) pass &= ~(1u << o);
    }
    // (a survivor's header lies in words w .. w + 3 of ITS stream: the lanes behind it hold them unless the stream ends within
    // those words -- and then `look` has already failed for it)
    const uint32_t mine = (uint32_t)__builtin_popcount(pass);
    const uint32_t incl = wave_prefix_sum(mine);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    {
        uint32_t at = incl - mine, m = pass;
        while (m) {                                                  // (at most 32 passes; seven on average)
            const uint32_t o = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            s_list[wave][at++] = (uint16_t)((lane << 5) | o);
        }
    }
    __syncthreads();
    // ---- step 2: the code length code's lengths, three bits each from bit 17 on: complete iff the sum of 2^(7 - length) is 2^7
    for (uint32_t j = lane; j < total; j += 64u) {
        const uint32_t e = s_list[wave][j], t = e >> 5, o = e & 31u;
        const uint32_t x0 = s_w[wave][t], x1 = s_w[wave][t + 1u], x2 = s_w[wave][t + 2u], x3 = s_w[wave][t + 3u];
        const uint32_t a = o ? __builtin_amdgcn_alignbit(x1, x0, o) : x0, bb = o ? __builtin_amdgcn_alignbit(x2, x1, o) : x1,
                       c = o ? __builtin_amdgcn_alignbit(x3, x2, o) : x2;                      // bits [o, ..), [o + 32, ..), [o + 64, ..)
        const uint32_t hclen = ((a >> 13) & 15u) + 4u;
        // the (up to nineteen) three-bit lengths as one 57-bit number, cut off behind the last one; its three bit planes say
        // which fields hold which length, a population count each how many: no loop over the lengths
        uint64_t x = ((((uint64_t)bb << 32) | a) >> 17) | ((uint64_t)c << 47);
        x &= (1ull << (3u * hclen)) - 1ull;
        constexpr uint64_t kEvery3rd = 0x1249249249249249ull;
        const uint64_t p0 = x & kEvery3rd, p1 = (x >> 1) & kEvery3rd, p2 = (x >> 2) & kEvery3rd;
        const uint64_t hi = p1 & p2, mid2 = p1 & ~p2, mid4 = ~p1 & p2, lo = ~p1 & ~p2;     // lengths 6-7, 2-3, 4-5, 0-1
        const uint32_t kraft = (uint32_t)__popcll(hi & p0) + 2u * (uint32_t)__popcll(hi & ~p0) + 4u * (uint32_t)__popcll(mid4 & p0) + 8u * (uint32_t)__popcll(mid4 & ~p0) +
                               16u * (uint32_t)__popcll(mid2 & p0) + 32u * (uint32_t)__popcll(mid2 & ~p0) + 64u * (uint32_t)__popcll(lo & p0);
        if (kraft != 128u) continue;
        const uint64_t hit = ((uint64_t)s_lo[wave][t] << 40) | ((uint64_t)s_wi[wave][t] * 32u + o);
        const uint32_t at = atomicAdd(&s_n, 1u);
        if (at < 512u) s_hits[at] = hit;
        else { const uint32_t g = atomicAdd(n_hits, 1u); if (g < cap) hits[g] = hit; }
    }
    __syncthreads();                                                 // (the next tile writes the words and the list again)
    }
    __syncthreads();
    const uint32_t got = min(s_n, 512u);
    if (threadIdx.x == 0 && got) s_base = atomicAdd(n_hits, got);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < got; i += 256u) if (s_base + i < cap) hits[s_base + i] = s_hits[i];
}

// the second part of the test, one thread per offset that passed the first; the offsets that pass this one too are appended
// to `good`
__global__ __launch_bounds__(64) void gz_check_kernel(const uint8_t *__restrict__ gz, const mk_gz_stream *__restrict__ streams, const uint64_t *__restrict__ hits,
                                                      const uint32_t *__restrict__ n_hits, uint32_t cap, uint64_t *__restrict__ good, uint32_t *__restrict__ n_good,
                                                      uint32_t good_cap)
{
    __shared__ uint8_t cl_tab[128][64];                              // the code length code's table, a column per lane: symbol | length << 5
    // (the number of hits lives on the device: a fixed grid walks them, every lane its own table column)
    const uint32_t lane = threadIdx.x, total = min(*n_hits, cap);
    for (uint32_t i = blockIdx.x * 64u + lane; i < total; i += gridDim.x * 64u) {
    const uint64_t h = hits[i];
    const uint32_t s = (uint32_t)(h >> 40);
    const uint64_t bit = h & ((1ull << 40) - 1ull);
    const mk_gz_stream st = streams[s];
    const uint8_t *__restrict__ p = gz + st.in_off;
    const uint64_t nbits = (uint64_t)st.in_len * 8u;
    uint64_t v = bits_at(p, bit);
    const uint32_t hlit = (uint32_t)((v >> 3) & 31u) + 257u, hdist = (uint32_t)((v >> 8) & 31u) + 1u, hclen = (uint32_t)((v >> 13) & 15u) + 4u;
    uint32_t cl[19];
#pragma unroll
    for (uint32_t j = 0; j < 19u; ++j) cl[j] = 0;
    uint64_t q = bit + 17u;
    // (cl[] is indexed by constants only: the order of RFC 1951 3.2.7 unrolled)
    uint32_t got[19];
#pragma unroll
    for (uint32_t j = 0; j < 19u; ++j) { got[j] = j < hclen ? (uint32_t)bits_at(p, q) & 7u : 0u; q += j < hclen ? 3u : 0u; }
    cl[16] = got[0]; cl[17] = got[1]; cl[18] = got[2]; cl[0] = got[3]; cl[8] = got[4]; cl[7] = got[5]; cl[9] = got[6]; cl[6] = got[7]; cl[10] = got[8];
    cl[5] = got[9]; cl[11] = got[10]; cl[4] = got[11]; cl[12] = got[12]; cl[3] = got[13]; cl[13] = got[14]; cl[2] = got[15]; cl[14] = got[16]; cl[1] = got[17];
    cl[15] = got[18];
    // canonical codes of the (complete) code length code into the table
    uint32_t cnt[8];
#pragma unroll
    for (uint32_t l = 0; l < 8u; ++l) cnt[l] = 0;
#pragma unroll
    for (uint32_t j = 0; j < 19u; ++j)
#pragma unroll
        for (uint32_t l = 1; l < 8u; ++l) cnt[l] += cl[j] == l ? 1u : 0u;
    uint32_t next[8], code = 0;
    next[0] = 0;
#pragma unroll
    for (uint32_t l = 1; l < 8u; ++l) { code = (code + (l > 1u ? cnt[l - 1u] : 0u)) << 1; next[l] = code; }
#pragma unroll
    for (uint32_t j = 0; j < 19u; ++j) {
        const uint32_t l = cl[j];
        if (!l) continue;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 1; k < 8u; ++k) if (l == k) { c = next[k]; next[k] = c + 1u; }
        const uint32_t r = __brev(c) >> (32u - l);
        for (uint32_t e = r; e < 128u; e += 1u << l) cl_tab[e][lane] = (uint8_t)(j | (l << 5));
    }
    // the lengths of both alphabets: nothing is kept but each alphabet's Kraft sum (in units of 2^-15), its longest code and
    // whether symbol 256 has one
    const uint32_t total_syms = hlit + hdist;
    uint32_t k = 0, prev = 0, kraft_l = 0, kraft_d = 0, max_l = 0, max_d = 0, eob = 0;
    bool ok = true;
    auto take = [&](uint32_t len, uint32_t rep) {                    // `rep` symbols of length `len` from symbol k on (no loop over them)
        if (len) {
            const uint32_t n_l = k < hlit ? min(rep, hlit - k) : 0u, n_d = rep - n_l, w = 32768u >> len;
            kraft_l += n_l * w; kraft_d += n_d * w;
            if (n_l) max_l = max(max_l, len);
            if (n_d) max_d = max(max_d, len);
            if (k <= 256u && 256u < k + n_l) eob = 1;
        }
        k += rep;
    };
    while (ok && k < total_syms) {                                   // (every pass takes at least one symbol's length)
        if (q + 64u > nbits + 64u) { ok = false; break; }
        v = bits_at(p, q);
        const uint32_t e = cl_tab[(uint32_t)v & 127u][lane];
        const uint32_t sym = e & 31u, l = e >> 5;                    // (the code is complete: every index holds an entry)
        q += l; v >>= l;
        if (sym < 16u) { take(sym, 1); prev = sym; }
        else if (sym == 16u) { if (!k) { ok = false; break; } const uint32_t rep = 3u + ((uint32_t)v & 3u); q += 2u; if (k + rep > total_syms) { ok = false; break; } take(prev, rep); }
        else if (sym == 17u) { const uint32_t rep = 3u + ((uint32_t)v & 7u); q += 3u; if (k + rep > total_syms) { ok = false; break; } take(0, rep); prev = 0; }
        else { const uint32_t rep = 11u + ((uint32_t)v & 127u); q += 7u; if (k + rep > total_syms) { ok = false; break; } take(0, rep); prev = 0; }
        if (q > nbits) ok = false;
        // (an over-subscribed alphabet stays one: noise is over the limit after a dozen lengths, and a wave of 64 offsets of which
        // none is a block's start -- two in three -- leaves here instead of reading three hundred lengths of each)
        if (kraft_l > 32768u || kraft_d > 32768u) ok = false;
    }
    // what inflate_table takes: not over-subscribed; incomplete only as a single one-bit code (or, the distances, none at all)
    ok = ok && eob && kraft_l <= 32768u && (kraft_l == 32768u || max_l <= 1u) && kraft_d <= 32768u && (kraft_d == 32768u || max_d <= 1u);
    if (ok) {
        const uint32_t at = atomicAdd(n_good, 1u);
        if (at < good_cap) good[at] = h;
    }
    }
}

// ---------------------------------------------------------------- the candidates in order, the chains
// The starts that passed both tests arrive in no order (an atomic counter): they are counted per stream, every stream gets
// its stretch of the list (a scan over the streams), they are dropped into it and ranked inside it -- a stream of 5 Mb has
// sixty of them, so "how many of my stream's are smaller than me" is sixty reads.  All on the device: the host never
// learns how many there are.
__global__ __launch_bounds__(256) void gz_cand_count_kernel(const uint64_t *__restrict__ good, const uint32_t *__restrict__ n_good, uint32_t good_cap, uint32_t n,
                                                            uint32_t *__restrict__ per_stream)
{
    const uint32_t total = min(*n_good, good_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t s = (uint32_t)(good[i] >> 40);
        if (s < n) atomicAdd(&per_stream[s], 1u);
    }
}

// one workgroup: every stream's stretch [cand_lo, cand_hi) of the ordered list
__global__ __launch_bounds__(1024) void gz_cand_scan_kernel(mk_gz_stream *__restrict__ streams, uint32_t n, const uint32_t *__restrict__ per_stream)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t s0 = 0; s0 < n; s0 += 1024u) {
        const uint32_t s = s0 + threadIdx.x, mine = s < n ? per_stream[s] : 0u;
        part[threadIdx.x] = mine;
        __syncthreads();
        for (uint32_t o = 1; o < 1024u; o <<= 1) {
            const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t hi = carry + part[threadIdx.x];
        if (s < n) { streams[s].cand_lo = hi - mine; streams[s].cand_hi = hi; }
        __syncthreads();
        if (threadIdx.x == 1023u) carry = hi;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gz_cand_scatter_kernel(const uint64_t *__restrict__ good, const uint32_t *__restrict__ n_good, uint32_t good_cap, uint32_t n,
                                                              const mk_gz_stream *__restrict__ streams, uint32_t *__restrict__ fill, uint64_t *__restrict__ tmp)
{
    const uint32_t total = min(*n_good, good_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint64_t key = good[i];
        const uint32_t s = (uint32_t)(key >> 40);
        if (s < n) tmp[streams[s].cand_lo + atomicAdd(&fill[s], 1u)] = key;
    }
}

// (tmp holds every stream's candidates in its stretch, unordered: a thread per entry finds its rank among its stream's)
__global__ __launch_bounds__(256) void gz_cand_rank_kernel(const uint64_t *__restrict__ tmp, const uint32_t *__restrict__ n_good, uint32_t good_cap, uint32_t n,
                                                           const mk_gz_stream *__restrict__ streams, uint64_t *__restrict__ sorted)
{
    const uint32_t total = min(*n_good, good_cap);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint64_t key = tmp[i];
        const uint32_t s = (uint32_t)(key >> 40);
        if (s >= n) continue;
        const uint32_t lo = streams[s].cand_lo, hi = streams[s].cand_hi;
        uint32_t rank = 0;
        for (uint32_t j = lo; j < hi; ++j) { const uint64_t o = tmp[j]; rank += o < key || (o == key && j < i) ? 1u : 0u; }
        sorted[lo + rank] = key;
    }
}

// The chain of every stream: from its first byte from link to link to its end -- a thread per stream.  What it passes is the
// stream's text: token count, length, members; a segment that failed, or that no link leads to, ends it with that status.
__global__ __launch_bounds__(64) void gz_chain_kernel(mk_gz_stream *__restrict__ streams, uint32_t n, const mk_gz_seg *__restrict__ segs, uint32_t *__restrict__ chain,
                                                      uint32_t *__restrict__ n_rewrite)
{
    const uint32_t s = blockIdx.x * 64u + threadIdx.x;
    if (s >= n) return;
    mk_gz_stream st = streams[s];
    if (st.status != MK_GZ_OK) return;
    const uint32_t room = st.cand_hi - st.cand_lo + 1u;               // segments of this stream
    uint32_t *__restrict__ mine = chain + st.cand_lo + s;
    uint64_t ntok = 0, nout = 0;
    uint32_t members = 0, status = MK_GZ_OK, cur = s, steps = 0, rewrite = 0;
    for (;;) {                                                       // (links only lead forward: at most `room` passes)
        const mk_gz_seg sg = segs[cur];
        if (sg.status != MK_GZ_OK) { status = sg.status; break; }
        if (steps >= room) { status = MK_GZ_INTERNAL; break; }
        mine[steps++] = cur;
        rewrite |= sg.n_tok > sg.tok_cap ? 1u : 0u;
        ntok += sg.n_tok; nout += sg.out_len; members += sg.members;
        if (nout >= kOutMax || ntok >= kOutMax) { status = MK_GZ_OUTPUT_ROOM; break; }
        if (sg.link == 0xffffffffu) break;                           // the stream's end
        if (sg.link <= cur || sg.link < n + st.cand_lo || sg.link >= n + st.cand_hi) { status = MK_GZ_INTERNAL; break; }
        cur = sg.link;
    }
    st.status = status; st.members = members; st.n_chain = status == MK_GZ_OK ? steps : 0u;
    st.n_tok = (uint32_t)ntok; st.out_len = (uint32_t)nout;
    st.rewrite = status == MK_GZ_OK ? rewrite : 0u;
    streams[s] = st;
    if (st.rewrite) atomicAdd(n_rewrite, 1u);
}

// ---------------------------------------------------------------- phase 2: tokens -> text
// x^(8 n) mod P for the reflected CRC-32 polynomial, as multiplication operands: a(x) * b(x) mod P in the reflected bit
// order (bit 31 = x^0), the shift-and-add of zlib's crc32_combine (multmodp)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
#pragma unroll 8
    for (uint32_t i = 0; i < 32; ++i) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

struct ResolveConsts { uint32_t lane_shift[64]; uint32_t block_shift; uint32_t byte_shift; };   // x^(512 (63 - lane)), x^(8 * 4096), x^8

// kResolveWaves streams per workgroup (a wave each, its own window; the CRC tables are shared).  Two, 76 KB: a workgroup then
// shares a CU with another one or with a workgroup of the token kernel (78 KB); with four (148 KB) it took a CU for itself
// -- and so did, seen from here, every single-wave workgroup of the token kernel: the two kernels of batches in flight
// side by side kept each other off the CUs.
constexpr uint32_t kResolveWaves = 2;
#ifndef GZ_WIDE_PIECES
#define GZ_WIDE_PIECES 2
#endif
constexpr uint32_t kWidePieces = GZ_WIDE_PIECES;     // eight-byte pieces a lane copies of its own match before the whole wave takes the rest
__global__ __launch_bounds__(64 * kResolveWaves) void gz_resolve_kernel(const uint8_t *__restrict__ gz, mk_gz_stream *__restrict__ jobs, uint32_t n,
                                                         const mk_gz_seg *__restrict__ segs, const uint32_t *__restrict__ chain, uint8_t *__restrict__ text,
                                                         ResolveConsts K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rsmem[];
    uint32_t (*crc_tab)[256] = reinterpret_cast<uint32_t (*)[256]>(rsmem);           // slice-by-4 tables (reflected 0xEDB88320)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, s = blockIdx.x * kResolveWaves + wave;
    uint8_t *const win = rsmem + 4096u + wave * kWin;
    for (uint32_t e = threadIdx.x; e < 256u; e += 64u * kResolveWaves) {
        uint32_t c = e;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        crc_tab[0][e] = c;
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < 256u; e += 64u * kResolveWaves) {
        uint32_t v = crc_tab[0][e];
        for (uint32_t t = 1; t < 4u; ++t) { v = crc_tab[0][v & 0xffu] ^ (v >> 8); crc_tab[t][e] = v; }
    }
    __syncthreads();
    if (s >= n) return;
    const mk_gz_stream job = jobs[s];
    if (job.status != MK_GZ_OK) return;
    uint8_t *__restrict__ out = text + job.out_off;
    // the stream's tokens lie where its segments left them: the chain says in which order
    const uint32_t *__restrict__ my_chain = chain + job.cand_lo + s;
    uint32_t pos = 0, flushed = 0;                  // pos: bytes produced so far; [0, flushed) are in `out` (flushed % kFlush == 0)
    uint32_t crc_raw = 0;                             // remainder of the member's bytes [member_pos, crc_from) (start 0, no complement)
    uint32_t member_pos = 0, crc_from = 0;
    uint32_t status = MK_GZ_OK;
    auto x_pow = [&](uint32_t nbytes) {              // x^(8 n) mod P by square-and-multiply
        uint32_t r = 0x80000000u, sq = K.byte_shift;
        for (uint32_t e = nbytes; e; e >>= 1) { if (e & 1u) r = gf_mul(r, sq); sq = gf_mul(sq, sq); }
        return r;
    };
    // bytes [a, b) of the output (inside the window, b - a <= kFlush) into crc_raw: every lane a slice of 64, four bytes per
    // table step, the slices joined by powers of x
    auto crc_span = [&](uint32_t a, uint32_t b) {
        const uint32_t len = b - a;
        if (!len) return;
        const uint32_t lo = min(a + lane * 64u, b), hi = min(lo + 64u, b);
        uint32_t r = 0, p = lo;
        if (len == kFlush && (a % kFlush) == 0u) {                    // a whole aligned block: sixteen aligned words per lane
            const uint32_t *w = reinterpret_cast<const uint32_t *>(win + (lo % kWin));
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                const uint32_t x = r ^ w[i];
                r = crc_tab[3][x & 0xffu] ^ crc_tab[2][(x >> 8) & 0xffu] ^ crc_tab[1][(x >> 16) & 0xffu] ^ crc_tab[0][x >> 24];
            }
            p = hi;
        }
        for (; p < hi; ++p) r = crc_tab[0][(r ^ win[p % kWin]) & 0xffu] ^ (r >> 8);
        const uint32_t after = b - hi;
        const uint32_t m = len == kFlush ? K.lane_shift[lane] : x_pow(after);
        r = hi > lo ? (after ? gf_mul(r, m) : r) : 0u;
        for (int o = 32; o > 0; o >>= 1) r ^= (uint32_t)__shfl_xor((int)r, o);
        crc_raw = gf_mul(crc_raw, len == kFlush ? K.block_shift : x_pow(len)) ^ r;
    };
    auto crc_to = [&](uint32_t upto) {
        while (crc_from < upto) {
            const uint32_t end = min(upto, (crc_from / kFlush + 1u) * kFlush);
            crc_span(crc_from, end);
            crc_from = end;
        }
    };
    // bytes [flushed, upto) -> out (upto a multiple of kFlush, or the very end)
    auto flush_to = [&](uint32_t upto) {
        while (flushed < upto) {
            const uint32_t end = min(upto, flushed + kFlush), len = end - flushed;
            const uint32_t w0 = flushed % kWin;                     // (a flush block never wraps: kWin is a multiple of kFlush)
            if (len == kFlush) {                                    // (out is 16-byte aligned, flushed a multiple of kFlush)
#pragma unroll
                for (uint32_t i = 0; i < kFlush / 1024u; ++i) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(&win[w0 + i * 1024u + lane * 16u]);
                    *reinterpret_cast<uint4 *>(out + flushed + i * 1024u + lane * 16u) = v;
                }
            } else {
                for (uint32_t i = lane; i < len; i += 64u) out[flushed + i] = win[w0 + i];
            }
            flushed = end;
        }
    };
    auto leave = [&]() {                                             // whole blocks leave the window, their CRC first
        const uint32_t whole = pos / kFlush * kFlush;
        if (whole > flushed) { crc_to(whole); flush_to(whole); }
    };
    for (uint32_t ch = 0; ch < job.n_chain && status == MK_GZ_OK; ++ch) {
    const mk_gz_seg sg = segs[my_chain[ch]];
    const uint32_t *__restrict__ tok = reinterpret_cast<const uint32_t *>(sg.tok_ptr);
    const uint32_t ntok = min(sg.n_tok, sg.tok_cap);                 // (more than its room: the host has had the segment decoded again)
    if (sg.n_tok > sg.tok_cap) { status = MK_GZ_INTERNAL; break; }
    uint32_t t0 = 0;
    // the tokens of this step and of the two after it are in registers, those of the third are requested now: a load from
    // memory takes a microsecond and a step a third of that (requested one step ahead, the wait for it was a quarter of the
    // kernel's time).  The usual step takes all 64; any other starts the queue again.
    // (a load is not looked at before its step: from a clamped place without a condition -- a select right behind the load
    // would wait for it there and then -- and "beyond the segment's end" is decided where the token is used)
    auto fetch = [&](uint32_t from) { return tok[min(from + lane, ntok - 1u)]; };
    uint32_t q0 = 0, q1 = 0, q2 = 0, q_at = ~0u;
    while (t0 < ntok && status == MK_GZ_OK) {                        // (every pass takes at least one token)
        if (q_at != t0) { q0 = fetch(t0); q1 = fetch(t0 + 64u); q2 = fetch(t0 + 128u); q_at = t0; }
        const uint32_t q3 = fetch(t0 + 192u);
        const uint32_t t = t0 + lane < ntok ? q0 : kTokMember;
        // a stored block or a member's end among these tokens: the tokens before it first, then the special itself
        const unsigned long long special = __ballot((t & (kTokMember | kTokStored)) && !(t & kTokMatch));
        const uint32_t upto_lane = special ? (uint32_t)__ffsll((long long)special) - 1u : 64u;
        if (upto_lane == 0u) {
            const uint32_t head = (uint32_t)__shfl((int)t, 0);
            if (head & kTokStored) {
                // the bytes of a stored block: from the stream into the window, a window's slack at a time
                if (t0 + 1u >= ntok) { status = MK_GZ_INTERNAL; break; }
                uint32_t left = head & 0xffffu;
                const uint8_t *__restrict__ from = gz + job.in_off + tok[t0 + 1u];
                while (left) {
                    const uint32_t m = min(left, kWin - kHist);
                    for (uint32_t i = lane; i < m; i += 64u) win[(pos + i) % kWin] = from[i];
                    pos += m; from += m; left -= m;
                    leave();
                }
                t0 += 2u;
            } else {
                // a member's end: its CRC-32 and length against the trailer's, then the next member starts
                if (t0 + 2u >= ntok) { status = MK_GZ_INTERNAL; break; }
                const uint32_t want_crc = tok[t0 + 1u], want_size = tok[t0 + 2u];
                crc_to(pos);
                // crc32(M) = remainder(M, start 0) ^ (0xffffffff moved up by |M| bytes) ^ 0xffffffff
                const uint32_t nbytes = pos - member_pos;
                const uint32_t crc = crc_raw ^ gf_mul(0xffffffffu, x_pow(nbytes)) ^ 0xffffffffu;
                if (crc != want_crc) status = MK_GZ_BAD_CRC;
                else if (nbytes != want_size) status = MK_GZ_BAD_SIZE;     // (ISIZE is the length mod 2^32; a stream's output is below that here)
                crc_raw = 0; member_pos = pos; crc_from = pos;
                t0 += 3u;
            }
            continue;
        }
        const bool mine = lane < upto_lane;
        const bool is_match = mine && (t & kTokMatch);
        const uint32_t len = !mine ? 0u : is_match ? (t >> 16) & 0x1ffu : 1u;
        const uint32_t incl = wave_prefix_sum(len);                  // places: inclusive prefix sum of the lengths
        // only as many tokens as fit the window's slack beyond the history
        const unsigned long long fits = __ballot(mine && incl <= kWin - kHist);
        const uint32_t take = (uint32_t)__popcll(fits);              // (prefix sums are monotone: the fitting lanes are the first `take`)
        if (take == 0u) { status = MK_GZ_INTERNAL; break; }          // (a token is at most 258 bytes: cannot happen)
        const bool act = lane < take;
        const uint32_t dst = pos + incl - len;
        const uint32_t dist = is_match ? (t & 0x7fffu) + 1u : 0u;
        const uint32_t src = dst - dist;
        // a match may reach back as far as its member's first byte and no further (checked here: the token pass does not know
        // how much of the member lies before a segment)
        if (__any(act && is_match && dist > dst - member_pos)) { status = MK_GZ_BAD_DISTANCE; break; }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(take - 1u));
        // rounds: a token is ready when its source bytes lie below the first unfinished token's place -- or when it IS that
        // token (its own bytes may overlap its source: such a copy runs byte by byte in order)
        bool done = !act;
        if (act && !is_match) { win[dst % kWin] = (uint8_t)t; done = true; }
        unsigned long long todo = __ballot(!done);
        uint8_t *const sink = rsmem + 4096u + kResolveWaves * kWin + wave * 64u + lane;     // where the bytes of a piece beyond a token's end go
        while (todo) {                                               // (each round finishes at least the lowest unfinished lane)
            const uint32_t low = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t frontier = (uint32_t)__builtin_amdgcn_readlane((int)dst, (int)low);
            const bool ready = !done && (lane == low || src + len <= frontier);
            uint32_t j = 0, d = dst % kWin, f = src % kWin;
            // eight bytes at a time where the source does not reach into what the eight writes (distance >= 8) and neither
            // run crosses the window's end: the reads of a piece are independent of each other, and a write beyond the
            // token's end goes to the lane's sink instead of under a branch (most matches of gzip'd DNA are one piece)
            const bool wide = ready && dist >= 8u;
            // (two pieces at most -- sixteen bytes: 97 % of the matches of gzip'd DNA; what is left of a longer one goes below)
            for (uint32_t piece = 0; piece < kWidePieces && __any(wide && j < len && d + 8u <= kWin && f + 8u <= kWin); ++piece) {
                const bool go = wide && j < len && d + 8u <= kWin && f + 8u <= kWin;
                // (the eight source bytes as three aligned words and two byte shifts instead of eight byte reads: the copies' LDS
                // operations are what this kernel's time is made of.  A word may reach four bytes past the window's end: the next
                // window, or the sinks -- inside the allocation, and those bytes are shifted out)
                const uint32_t fa = go ? f & ~3u : 0u, sh = f & 3u;
                const uint32_t *__restrict__ w32 = reinterpret_cast<const uint32_t *>(win + fa);
                const uint32_t a0 = w32[0], a1 = w32[1], a2 = w32[2];
                const uint32_t lo8 = __builtin_amdgcn_alignbyte(a1, a0, sh), hi8 = __builtin_amdgcn_alignbyte(a2, a1, sh);
#pragma unroll
                for (uint32_t i = 0; i < 8u; ++i) *(go && j + i < len ? win + d + i : sink) = (uint8_t)((i < 4u ? lo8 : hi8) >> (8u * (i & 3u)));
                if (go) {
                    const uint32_t adv = min(8u, len - j);
                    j += adv; d += adv; f += adv;
                    d = d == kWin ? 0u : d;
                    f = f == kWin ? 0u : f;
                }
            }
            // ... the rest -- long matches, short distances, runs across the window's end -- one token at a time by the WHOLE
            // wave, 64 bytes a pass: byte k of a token is byte k of its source when the source lies 64 or more back (an earlier
            // pass, or earlier text), and byte k mod distance of it otherwise -- the `distance` bytes before the token, which
            // are final: a run of one byte or a short period repeats them.  (Left to the token's own lane -- eight bytes a
            // pass, a byte a pass below distance 8 -- a text of long repeats ran at 50 MB/s and a run of N at a byte per 200
            // cycles: tests/test_gpu_gunzip.py's 1.1 GiB text took 23 s of this kernel.)
            unsigned long long rest = __ballot(ready && j < len);
            while (rest) {                                               // (wave-uniform: every pass takes one token off)
                const int l = __ffsll((long long)rest) - 1;
                rest &= rest - 1ull;
                const uint32_t t_dst = (uint32_t)__builtin_amdgcn_readlane((int)dst, l), t_len = (uint32_t)__builtin_amdgcn_readlane((int)len, l);
                const uint32_t t_dist = (uint32_t)__builtin_amdgcn_readlane((int)dist, l), t_j = (uint32_t)__builtin_amdgcn_readlane((int)j, l);
                const uint32_t t_src = t_dst - t_dist;
                for (uint32_t k0 = t_j; k0 < t_len; k0 += 64u) {
                    const uint32_t k = k0 + lane;
                    if (k < t_len) {
                        const uint32_t sk = t_dist >= 64u ? k : k % t_dist;
                        win[(t_dst + k) % kWin] = win[(t_src + sk) % kWin];
                    }
                }
            }
            if (ready) done = true;
            todo = __ballot(!done);
        }
        pos += total;
        t0 += take;
        if (take == 64u) { q0 = q1; q1 = q2; q2 = q3; q_at = t0; }
        leave();
    }
    }
    if (status == MK_GZ_OK) {
        if (pos != job.out_len || member_pos != pos) status = MK_GZ_INTERNAL;      // (the last token is a member's end)
        flush_to(pos);
    }
    if (lane == 0) jobs[s].status = status;
}

}  // namespace

static uint32_t gf_mul_host(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t i = 0; i < 32; ++i) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

static uint32_t x_pow_bytes(uint64_t nbytes)                          // x^(8 n) mod P
{
    uint32_t r = 0x80000000u, sq = 0x00800000u;                       // x^0; x^8 (bit 31 - 8)
    for (uint64_t e = nbytes; e; e >>= 1) { if (e & 1u) r = gf_mul_host(r, sq); sq = gf_mul_host(sq, sq); }
    return r;
}

// ---- device memory of the inflater: blocks kept by the context between batches (allocating and freeing tens of gigabytes
// per batch cost a second each)
// (what MIEKKI_VERBOSE shows of the pool: device allocations made and their time, time readers stood waiting for a piece)
static std::atomic<uint64_t> g_malloc_n{0}, g_malloc_us{0}, g_stage_wait_us{0}, g_stage_waits{0}, g_put_us{0};

// (a block is taken only for the role it was made for -- a batch's input + tokens, or its text: with one list for all, a text
// block would take the place of a token block and the next batch allocate a new one, in the middle of the run: an
// allocation of gigabytes while the device is busy took up to a second and a half)
static uint8_t *gz_block_get(mk_ctx *c, uint64_t need, uint64_t *got, int role)
{
    {
        std::lock_guard<std::mutex> g(c->gz_m);
        size_t best = c->gz_blocks.size();
        for (size_t i = 0; i < c->gz_blocks.size(); ++i)
            if (c->gz_blocks[i].role == role && c->gz_blocks[i].second >= need &&
                (best == c->gz_blocks.size() || c->gz_blocks[i].second < c->gz_blocks[best].second)) best = i;
        if (best != c->gz_blocks.size()) {
            uint8_t *p = (uint8_t *)c->gz_blocks[best].first;
            *got = c->gz_blocks[best].second;
            c->gz_blocks.erase(c->gz_blocks.begin() + (long)best);
            return p;
        }
    }
    void *p = nullptr;
    const uint64_t want = need + need / 8 + 4096;
    const auto t0 = std::chrono::steady_clock::now();
    struct Count { std::chrono::steady_clock::time_point t0; ~Count() { ++g_malloc_n; g_malloc_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } } count{t0};
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        {                                                            // what is kept and does not fit makes room
            std::lock_guard<std::mutex> g(c->gz_m);
            for (auto &blk : c->gz_blocks) (void)hipFree(blk.first);
            c->gz_blocks.clear();
        }
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    *got = want;
    return (uint8_t *)p;
}

// contexts that keep blocks (for gz_release_idle_blocks)
static std::mutex g_keepers_m;
static std::vector<mk_ctx *> g_keepers;

static void gz_block_put(mk_ctx *c, uint8_t *p, uint64_t bytes, int role)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_keepers_m);
        if (std::find(g_keepers.begin(), g_keepers.end(), c) == g_keepers.end()) g_keepers.push_back(c);
    }
    std::lock_guard<std::mutex> g(c->gz_m);
    c->gz_blocks.push_back(mk_ctx::GzBlock{p, bytes, role});
}

// every context's idle blocks back to the device; returns the bytes given back
uint64_t gz_release_idle_blocks()
{
    uint64_t freed = 0;
    int dev = 0;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess;
    std::lock_guard<std::mutex> gk(g_keepers_m);
    for (mk_ctx *c : g_keepers) {
        std::lock_guard<std::mutex> g(c->gz_m);
        if (c->gz_blocks.empty()) continue;
        (void)hipSetDevice(c->p.device);
        for (auto &blk : c->gz_blocks) { (void)hipFree(blk.first); freed += blk.second; }
        c->gz_blocks.clear();
    }
    if (have_dev) (void)hipSetDevice(dev);
    return freed;
}

static void gz_forget_keeper(mk_ctx *c)
{
    std::lock_guard<std::mutex> g(g_keepers_m);
    g_keepers.erase(std::remove(g_keepers.begin(), g_keepers.end(), c), g_keepers.end());
}

// Page-locked staging for the run's small copies (stream tables, block starts, segments): a copy to or from ordinary memory
// is a synchronous call inside the runtime -- it waits for everything queued on its stream -- and several batches unpacked
// by threads of their own then take turns instead of running side by side (measured: six batches of 128 files in flight
// finished no faster than one after the other).  Pieces of at least 1 MiB, powers of two, kept by the context.
static void *gz_pin_get(mk_ctx *c, uint64_t need, uint64_t *got)
{
    uint64_t want = 1ull << 20;
    while (want < need) want <<= 1;
    {
        std::lock_guard<std::mutex> g(c->gz_m);
        for (size_t i = 0; i < c->gz_pins.size(); ++i)
            if (c->gz_pins[i].second == want) {
                void *p = c->gz_pins[i].first;
                c->gz_pins.erase(c->gz_pins.begin() + (long)i);
                *got = want;
                return p;
            }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = want;
    return p;
}

static void gz_pin_put(mk_ctx *c, void *p, uint64_t bytes)
{
    if (!p) return;
    std::lock_guard<std::mutex> g(c->gz_m);
    c->gz_pins.emplace_back(p, bytes);
}

// One batch of streams through the inflater, three steps:
//   gz_open    the layout (every stream's place follows from the files' sizes), the device blocks, the tables up;
//   gz_put     a file's bytes (or a piece of them) to their place -- from a page-locked piece of the context's pool
//              (mk_gz_stage) a DMA on one of four upload streams, from any other memory a staged copy; any thread;
//   gz_finish  everything else, queued back to back on the run's stream: block starts (gz_find / gz_check), the candidates
//              in order, every segment decoded into its slots (gz_tokens), the chains -- then the ONE point where the host
//              must look (how long is every text: the out block is allocated from that) --, tokens to text (gz_resolve).
// The text stays on the device (blk_out) with the streams' results in `streams`; a stream whose status is not MK_GZ_OK
// has no text.
struct GzRun {
    mk_ctx *c = nullptr;
    hipStream_t st = nullptr;
    uint32_t n = 0;
    std::vector<mk_gz_stream> streams;
    uint8_t *blk_in = nullptr, *blk_tok = nullptr, *blk_out = nullptr;
    uint64_t in_bytes = 0, tok_bytes = 0, out_bytes = 0;
    uint8_t *d_gz = nullptr, *d_text = nullptr;
    mk_gz_stream *d_streams = nullptr;
    uint64_t *d_wf = nullptr, *d_hits = nullptr, *d_good = nullptr, *d_tmp = nullptr, *d_sorted = nullptr;
    uint32_t *d_per_stream = nullptr, *d_fill = nullptr, *d_cnt = nullptr;       // d_cnt: hits, good, the token kernel's queue, streams to rewrite
    uint32_t *d_slots = nullptr, *d_chain = nullptr, *d_aux = nullptr;
    mk_gz_seg *d_segs = nullptr;
    uint8_t *d_extra = nullptr;                                       // `extra` bytes of the out block for the caller (256-byte aligned)
    uint64_t in_at = 0, total_words = 0;
    uint32_t hit_cap = 0, good_cap = 0, tok_wgs = 0, lds_tok = 0, lds_text = 0;
    uint32_t n_segs = 0, n_rewritten = 0;
    hipEvent_t ev_open = nullptr;                                     // the blocks are ready for the files' bytes
    std::atomic<uint32_t> up_waited{0};                               // upload streams that wait for ev_open already
    bool uploads_joined = false;                                      // the run's stream waits for every copy queued for it
    bool loose_puts = false;                                          // some file came from memory that is not the pool's: on the run's own stream
    double t_open = 0, t_blocks = 0, t_text = 0;
    // page-locked staging: pieces in use, and the downloads that wait for the stream (staging -> the caller's memory)
    struct Pin { uint8_t *p; uint64_t size, used; };
    std::vector<Pin> pins;
    struct Down { void *dst; const void *src; size_t bytes; };
    std::vector<Down> downs;
    uint8_t *stage(uint64_t bytes)
    {
        bytes = (bytes + 63u) & ~63ull;
        if (pins.empty() || pins.back().size - pins.back().used < bytes) {
            uint64_t got = 0;
            void *p = gz_pin_get(c, bytes, &got);
            if (!p) return nullptr;
            pins.push_back(Pin{(uint8_t *)p, got, 0});
        }
        uint8_t *at = pins.back().p + pins.back().used;
        pins.back().used += bytes;
        return at;
    }
    // host -> device through staging (the caller's memory is free again at once)
    int up(void *d_dst, const void *src, size_t bytes)
    {
        if (!bytes) return MK_OK;
        uint8_t *s = stage(bytes);
        if (!s) { set_error("no page-locked memory for the inflater's tables"); return MK_ERR_NOMEM; }
        memcpy(s, src, bytes);
        MK_HIP(hipMemcpyAsync(d_dst, s, bytes, hipMemcpyHostToDevice, st));
        return MK_OK;
    }
    // device -> host: lands in dst at the next settle()
    int down(void *dst, const void *d_src, size_t bytes)
    {
        if (!bytes) return MK_OK;
        uint8_t *s = stage(bytes);
        if (!s) { set_error("no page-locked memory for the inflater's tables"); return MK_ERR_NOMEM; }
        MK_HIP(hipMemcpyAsync(s, d_src, bytes, hipMemcpyDeviceToHost, st));
        downs.push_back(Down{dst, s, bytes});
        return MK_OK;
    }
    // The waits BLOCK (an event made with hipEventBlockingSync) instead of spinning: a batch is unpacked by a thread of its
    // own beside the reader threads, which want the cores -- six spinning waits took six of the job's sixteen CPUs from the
    // readers' inflate (their rate fell from 2.2k to 1.25k files/s while the device's batches were in flight).
    hipEvent_t ev = nullptr;
    int wait_stream(hipStream_t s)
    {
        if (!ev) MK_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
        MK_HIP(hipEventRecord(ev, s));
        MK_HIP(hipEventSynchronize(ev));
        return MK_OK;
    }
    int settle()
    {
        MK_TRY(wait_stream(st));
        for (const Down &d : downs) memcpy(d.dst, d.src, d.bytes);
        downs.clear();
        for (Pin &p : pins) p.used = 0;                              // (everything staged so far has been consumed)
        while (pins.size() > 1) { gz_pin_put(c, pins.back().p, pins.back().size); pins.pop_back(); }
        return MK_OK;
    }
    ~GzRun()
    {
        if (c) {
            (void)hipSetDevice(c->p.device);
            // (copies into this run's blocks may still be queued on the upload streams -- unless gz_finish has made the run's own
            // stream wait for them: then the wait for that stream below covers them)
            if (!uploads_joined) for (hipStream_t u : c->gz_up) if (u) (void)wait_stream(u);
        }
        if (st) { (void)wait_stream(st); (void)hipStreamDestroy(st); }
        if (ev) (void)hipEventDestroy(ev);
        if (ev_open) (void)hipEventDestroy(ev_open);
        if (c) {
            gz_block_put(c, blk_in, in_bytes, 0); gz_block_put(c, blk_out, out_bytes, 1);
            for (Pin &p : pins) gz_pin_put(c, p.p, p.size);
        }
    }
};

constexpr uint64_t kStagePiece = 16ull << 20;                         // (room for a run of eight 5 Mb genomes gzip'd: a copy per run)
constexpr uint32_t kStagePieces = 24;                                 // at most 384 MB of page-locked pieces per context

static int gz_open(GzRun &r, mk_ctx *c, const uint64_t *gz_bytes, uint32_t n)
{
    r.c = c; r.n = n;
    r.streams.assign(n, mk_gz_stream{});
    if (!n) return MK_OK;
    MK_HIP(hipSetDevice(c->p.device));
    MK_HIP(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));  // (a stream of its own: the call may run beside others on the context)
    hipStream_t st = r.st;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // ---- the input block: the files, the streams, the finder's lists, the candidates
    uint64_t in_at = 0;
    std::vector<uint64_t> word_first(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        mk_gz_stream &j = r.streams[i];
        j.in_off = in_at;
        j.in_len = gz_bytes[i] < 0xfffffff0ull ? (uint32_t)gz_bytes[i] : 0u;
        j.status = gz_bytes[i] < 0xfffffff0ull ? MK_GZ_OK : MK_GZ_OUTPUT_ROOM;
        in_at += ((uint64_t)j.in_len + 16u + 15u) / 16u * 16u;
        word_first[i + 1] = word_first[i] + ((uint64_t)j.in_len + 3u) / 4u;
    }
    r.in_at = in_at;
    r.total_words = word_first[n];
    r.hit_cap = (uint32_t)std::min<uint64_t>(r.total_words / 8u + 65536u, 1u << 30);       // (0.1 % of the bit offsets pass the first test: room for 0.4 %)
    r.good_cap = (uint32_t)std::min<uint64_t>(in_at / 2048u + 64ull * n + 4096u, 1u << 28);   // (a block per 26 KB of gzip'd DNA)
    int cus = 0;
    MK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->p.device));
    // (asking for more LDS than they need, so that one workgroup of the two long-running kernels has a CU to itself and the
    // build's kernels find room beside it, was measured and left: 3.6k against 3.8k genomes/s, profiles/r6_ingest_gz.txt)
    r.lds_tok = (uint32_t)sizeof(TokLds);
    r.lds_text = 4096u + kResolveWaves * kWin + 64u * kResolveWaves;
    r.tok_wgs = (r.lds_tok <= (80u << 10) ? 2u : 1u) * (uint32_t)std::max(cus, 1);   // (workgroups of the token kernel that share a CU's LDS)
    uint64_t at = 0;
    auto carve = [&at](uint64_t bytes) { const uint64_t o = at; at += (bytes + 255u) / 256u * 256u; return o; };
    const uint64_t o_gz = carve(in_at + 16), o_streams = carve((uint64_t)n * sizeof(mk_gz_stream)), o_wf = carve(((uint64_t)n + 1) * 8),
                   o_hits = carve((uint64_t)r.hit_cap * 8), o_good = carve((uint64_t)r.good_cap * 8), o_tmp = carve((uint64_t)r.good_cap * 8),
                   o_sorted = carve((uint64_t)r.good_cap * 8), o_per = carve((uint64_t)n * 4), o_fill = carve((uint64_t)n * 4), o_cnt = carve(64);
    // ---- ... and behind them: a slot per byte of the files, the segments' results, the chains, the lanes' symbol lists
    const uint64_t n_seg_room = (uint64_t)n + r.good_cap;
    const uint64_t o_slots = carve((in_at + 16) * 4), o_segs = carve(n_seg_room * sizeof(mk_gz_seg)), o_chain = carve(n_seg_room * 4),
                   o_aux = carve((uint64_t)r.tok_wgs * 64u * 288u * 4u);
    r.blk_in = gz_block_get(c, at, &r.in_bytes, 0);
    if (!r.blk_in) { set_error("no device memory for %u gzip'd files and their tokens (%.1f GB)", n, at / 1e9); return MK_ERR_NOMEM; }
    r.blk_tok = r.blk_in;
    // (an allocation of gigabytes now and then takes a second -- the driver's doing, seen in one run of two: the blocks of the
    // NEXT batch are made on a thread of their own while this one is being filled, so that only a context's first batch can
    // stand waiting for one.  Text blocks are made for four times the files' bytes -- gzip'd DNA: 3.3; a batch that needs
    // more makes its own)
    {
        const uint64_t need0 = at, need1 = in_at * 4u + (64ull << 20);
        bool start = false;
        { std::lock_guard<std::mutex> g(c->gz_m); start = !c->gz_closing; if (start) ++c->gz_making; }
        if (start) std::thread([c, need0, need1] {
            if (hipSetDevice(c->p.device) == hipSuccess) {
                for (int role = 0; role < 2; ++role) {
                    const uint64_t need = role == 0 ? need0 : need1;
                    bool have = false;
                    {
                        std::lock_guard<std::mutex> g(c->gz_m);
                        for (auto &blk : c->gz_blocks) have = have || (blk.role == role && blk.second >= need);
                    }
                    if (have || c->gz_closing) continue;
                    uint64_t got = 0;
                    uint8_t *p = gz_block_get(c, need, &got, -1 - role);  // (a role nobody holds: always a new block)
                    if (p) gz_block_put(c, p, got, role);
                }
            } else (void)hipGetLastError();
            std::lock_guard<std::mutex> g(c->gz_m);
            --c->gz_making;
            c->gz_cv.notify_all();
        }).detach();
    }
    r.d_gz = r.blk_in + o_gz;
    r.d_streams = reinterpret_cast<mk_gz_stream *>(r.blk_in + o_streams);
    r.d_wf = reinterpret_cast<uint64_t *>(r.blk_in + o_wf); r.d_hits = reinterpret_cast<uint64_t *>(r.blk_in + o_hits);
    r.d_good = reinterpret_cast<uint64_t *>(r.blk_in + o_good); r.d_tmp = reinterpret_cast<uint64_t *>(r.blk_in + o_tmp);
    r.d_sorted = reinterpret_cast<uint64_t *>(r.blk_in + o_sorted);
    r.d_per_stream = reinterpret_cast<uint32_t *>(r.blk_in + o_per); r.d_fill = reinterpret_cast<uint32_t *>(r.blk_in + o_fill);
    r.d_cnt = reinterpret_cast<uint32_t *>(r.blk_in + o_cnt);
    r.d_slots = reinterpret_cast<uint32_t *>(r.blk_tok + o_slots);
    r.d_segs = reinterpret_cast<mk_gz_seg *>(r.blk_tok + o_segs);
    r.d_chain = reinterpret_cast<uint32_t *>(r.blk_tok + o_chain);
    r.d_aux = reinterpret_cast<uint32_t *>(r.blk_tok + o_aux);
    // (the bit reader may look 16 bytes past a stream's end, and a file that never arrives must read as nothing: zeros
    // everywhere first -- one fill instead of one per file)
    MK_HIP(hipMemsetAsync(r.d_gz, 0, in_at + 16, st));
    MK_HIP(hipMemsetAsync(r.d_per_stream, 0, (o_cnt + 64) - o_per, st));     // the per-stream counts, the fill marks, the counters
    MK_TRY(r.up(r.d_streams, r.streams.data(), (size_t)n * sizeof(mk_gz_stream)));
    MK_TRY(r.up(r.d_wf, word_first.data(), ((size_t)n + 1) * 8));
    MK_HIP(hipEventCreateWithFlags(&r.ev_open, hipEventDisableTiming));
    MK_HIP(hipEventRecord(r.ev_open, st));
    r.t_open = now() - t0;
    return MK_OK;
}

// a page-locked piece of kStagePiece bytes to read (a piece of) a file into; null: none to be had (use any memory)
// (a reader thread selects the device when it is not the thread's current one, not with every call: sixteen readers at five
// runtime calls a file stood in line for the runtime's lock)
static inline bool gz_use_device(const mk_ctx *c)
{
    int current = -1;                                               // (asked, not remembered: other calls of this thread may have chosen another)
    if (hipGetDevice(&current) == hipSuccess && current == c->p.device) return true;
    if (hipSetDevice(c->p.device) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

static void *gz_stage_get(mk_ctx *c)
{
    (void)gz_use_device(c);
    mk_ctx::GzStage got{nullptr, nullptr};
    bool wait = false;
    {
        std::lock_guard<std::mutex> g(c->gz_m);
        if (!c->gz_stage_free.empty()) { got = c->gz_stage_free.back(); c->gz_stage_free.pop_back(); }
        else if (c->gz_stage_made < kStagePieces) ++c->gz_stage_made;
        else if (!c->gz_stage_busy.empty()) { got = c->gz_stage_busy.front(); c->gz_stage_busy.pop_front(); wait = true; }
        else return nullptr;                                          // (every piece is in some caller's hands)
    }
    if (!got.p) {
        if (!(got.p = pinned_alloc(kStagePiece)) ||
            hipEventCreateWithFlags(&got.ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (got.p) pinned_free(got.p);
            std::lock_guard<std::mutex> g(c->gz_m);
            --c->gz_stage_made;
            return nullptr;
        }
    } else if (wait) {
        const auto t0 = std::chrono::steady_clock::now();
        // the oldest queued copy: done soonest -- usually long done (a blocking wait costs a sleep even then: ask first)
        if (hipEventQuery(got.ev) != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(got.ev); }
        ++g_stage_waits;
        g_stage_wait_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    std::lock_guard<std::mutex> g(c->gz_m);
    c->gz_stage_held.push_back(got);
    return got.p;
}

// bytes [at, at + bytes) of file i; staged: `data` is a piece from gz_stage_get, which takes it back -- when the copy is
// done, or at once when there is nothing to copy or the call fails
static int gz_put(GzRun &r, uint32_t i, uint64_t at, const void *data, uint64_t bytes, bool staged, uint32_t span = 1)
{
    mk_ctx *c = r.c;
    mk_ctx::GzStage piece{nullptr, nullptr};
    hipStream_t up = nullptr;
    uint32_t k_up = 0;
    const auto t_in = std::chrono::steady_clock::now();
    struct Count { std::chrono::steady_clock::time_point t0; ~Count() { g_put_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } } count{t_in};
    if (staged) {
        std::lock_guard<std::mutex> g(c->gz_m);
        for (size_t k = 0; k < c->gz_stage_held.size(); ++k)
            if (c->gz_stage_held[k].p == data) { piece = c->gz_stage_held[k]; c->gz_stage_held.erase(c->gz_stage_held.begin() + (long)k); break; }
        if (!piece.p) { set_error("not a piece of the context's staging pool"); return MK_ERR_ARG; }
        k_up = c->gz_up_next++ % 4u;
        const uint32_t k = k_up;
        if (!c->gz_up[k] && (!gz_use_device(c) || hipStreamCreateWithFlags(&c->gz_up[k], hipStreamNonBlocking) != hipSuccess)) {
            (void)hipGetLastError();
            c->gz_up[k] = nullptr;
        }
        up = c->gz_up[k];
    }
    auto give_back = [&](bool queued) {
        if (!piece.p) return;
        std::lock_guard<std::mutex> g(c->gz_m);
        if (queued) c->gz_stage_busy.push_back(piece); else c->gz_stage_free.push_back(piece);
    };
    // (a span of files: laid out by the caller as the input block is -- each file at its place, zeros up to the next)
    const bool fits = i < r.n && span >= 1u && (uint64_t)i + span <= r.n &&
                      (span == 1u ? at + bytes <= r.streams[i].in_len
                                  : at == 0 && bytes <= (i + span < r.n ? r.streams[i + span].in_off : r.in_at) - r.streams[i].in_off);
    if (!fits) { give_back(false); set_error("file %u: bytes beyond its size", i); return MK_ERR_ARG; }
    if (!gz_use_device(c)) { give_back(false); set_error("cannot use the device"); return MK_ERR_DEVICE; }
    uint8_t *dst = r.d_gz + r.streams[i].in_off + at;
    if (!staged) {
        if (bytes) MK_HIP(hipMemcpyAsync(dst, data, bytes, hipMemcpyHostToDevice, r.st));
        r.loose_puts = true;
        return MK_OK;
    }
    if (!bytes) { give_back(false); return MK_OK; }
    // (an upload stream waits for the run's blocks once, not with every piece)
    const bool first_on_stream = !(r.up_waited.fetch_or(1u << k_up) & (1u << k_up));
    if (!up || bytes > kStagePiece || (first_on_stream && hipStreamWaitEvent(up, r.ev_open, 0) != hipSuccess) || hipMemcpyAsync(dst, data, bytes, hipMemcpyHostToDevice, up) != hipSuccess ||
        hipEventRecord(piece.ev, up) != hipSuccess) {
        (void)hipGetLastError();
        if (up) (void)hipStreamSynchronize(up);                       // (whatever of it was queued is done before the piece is lent again)
        give_back(false);
        set_error("cannot queue the copy of file %u", i);
        return MK_ERR_DEVICE;
    }
    give_back(true);
    return MK_OK;
}

static int gz_finish(GzRun &r, const uint64_t *out_room, const std::function<uint64_t(const std::vector<mk_gz_stream> &)> &extra_bytes)
{
    mk_ctx *c = r.c;
    const uint32_t n = r.n;
    if (!n) return MK_OK;
    MK_HIP(hipSetDevice(c->p.device));
    hipStream_t st = r.st;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t1 = now();
    // the files' copies (queued by whoever put them, on the upload streams) before anything reads them
    for (hipStream_t u : c->gz_up)
        if (u) {
            hipEvent_t e = nullptr;
            MK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            hipError_t er = hipEventRecord(e, u);
            if (er == hipSuccess) er = hipStreamWaitEvent(st, e, 0);
            (void)hipEventDestroy(e);                                 // (released when the work queued on it is done)
            MK_HIP(er);
        }
    r.uploads_joined = true;
    // ---- where blocks start; the candidates in order; every segment's tokens; the chains
    if (r.total_words) {
        hipLaunchKernelGGL(gz_find_kernel, dim3((uint32_t)((r.total_words + 256u * kFindTiles - 1u) / (256u * kFindTiles))), dim3(256), 0, st, r.d_gz, r.d_streams, r.d_wf, n, r.d_hits, r.d_cnt, r.hit_cap);
        MK_HIP(hipGetLastError());
        hipLaunchKernelGGL(gz_check_kernel, dim3(std::min<uint32_t>((r.hit_cap + 63u) / 64u, 1u << 20)), dim3(64), 0, st, r.d_gz, r.d_streams, r.d_hits, r.d_cnt, r.hit_cap,
                           r.d_good, r.d_cnt + 1, r.good_cap);
        MK_HIP(hipGetLastError());
        const uint32_t wide = std::min<uint32_t>((r.good_cap + 255u) / 256u, 512u);
        hipLaunchKernelGGL(gz_cand_count_kernel, dim3(wide), dim3(256), 0, st, r.d_good, r.d_cnt + 1, r.good_cap, n, r.d_per_stream);
        MK_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gz_cand_scan_kernel, dim3(1), dim3(1024), 0, st, r.d_streams, n, r.d_per_stream);
    MK_HIP(hipGetLastError());
    if (r.total_words) {
        const uint32_t wide = std::min<uint32_t>((r.good_cap + 255u) / 256u, 512u);
        hipLaunchKernelGGL(gz_cand_scatter_kernel, dim3(wide), dim3(256), 0, st, r.d_good, r.d_cnt + 1, r.good_cap, n, r.d_streams, r.d_fill, r.d_tmp);
        MK_HIP(hipGetLastError());
        hipLaunchKernelGGL(gz_cand_rank_kernel, dim3(wide), dim3(256), 0, st, r.d_tmp, r.d_cnt + 1, r.good_cap, n, r.d_streams, r.d_sorted);
        MK_HIP(hipGetLastError());
    }
    static_assert(sizeof(TokLds) <= 80u << 10, "phase 1's tables and buffers: two waves per CU");
    MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gz_tokens_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r.lds_tok));
    TokArgs A;
    A.gz = r.d_gz; A.streams = r.d_streams; A.n = n; A.sorted = r.d_sorted; A.n_good = r.d_cnt + 1; A.good_cap = r.good_cap;
    A.segs = r.d_segs; A.slots = r.d_slots; A.aux = r.d_aux; A.queue = r.d_cnt + 2; A.jobs = nullptr; A.n_jobs = 0;
    hipLaunchKernelGGL(gz_tokens_kernel, dim3(r.tok_wgs), dim3(64), r.lds_tok, st, A);
    MK_HIP(hipGetLastError());
    hipLaunchKernelGGL(gz_chain_kernel, dim3((n + 63u) / 64u), dim3(64), 0, st, r.d_streams, n, r.d_segs, r.d_chain, r.d_cnt + 3);
    MK_HIP(hipGetLastError());
    // ---- the one look: every stream's length and token count (and how many candidates there were)
    uint32_t cnt[4] = {0, 0, 0, 0};
    MK_TRY(r.down(cnt, r.d_cnt, 16));
    MK_TRY(r.down(r.streams.data(), r.d_streams, (size_t)n * sizeof(mk_gz_stream)));
    MK_TRY(r.settle());
    const uint32_t ng = std::min(cnt[1], r.good_cap);
    r.n_segs = n + ng;
    uint64_t out_at = 0;
    for (uint32_t i = 0; i < n; ++i) {
        mk_gz_stream &j = r.streams[i];
        if (j.status == MK_GZ_OK && out_room && j.out_len > out_room[i]) j.status = MK_GZ_OUTPUT_ROOM;
        if (j.status != MK_GZ_OK) { j.n_chain = 0; j.rewrite = 0; j.out_len = 0; continue; }
        j.out_off = out_at;
        out_at += ((uint64_t)j.out_len + 15u) / 16u * 16u;
    }
    // the few segments whose tokens did not fit their slots, once more with exact rooms (behind the text, in the out block)
    std::vector<mk_gz_job> jobs;
    uint64_t dense_words = 0;
    if (cnt[3]) {
        std::vector<mk_gz_seg> segs(r.n_segs);
        std::vector<uint32_t> chain(r.n_segs);
        MK_TRY(r.down(segs.data(), r.d_segs, segs.size() * sizeof(mk_gz_seg)));
        MK_TRY(r.down(chain.data(), r.d_chain, chain.size() * 4));
        MK_TRY(r.settle());
        for (uint32_t i = 0; i < n; ++i) {
            const mk_gz_stream &j = r.streams[i];
            if (j.status != MK_GZ_OK || !j.rewrite) continue;
            ++r.n_rewritten;
            for (uint32_t k = 0; k < j.n_chain; ++k) {
                const uint32_t g = chain[j.cand_lo + i + k];
                if (g >= r.n_segs || segs[g].n_tok <= segs[g].tok_cap) continue;
                mk_gz_job job;
                job.seg = g; job.tok_cap = (segs[g].n_tok + 3u) & ~3u; job.tok_ptr = dense_words;   // (an offset for now)
                dense_words += job.tok_cap;
                jobs.push_back(job);
            }
        }
    }
    r.t_blocks = now() - t1;
    // ---- the out block: text, whatever the caller wants behind it, the rewritten segments' tokens
    const double t3 = now();
    const uint64_t extra = extra_bytes ? extra_bytes(r.streams) : 0;
    uint64_t at = 0;
    auto carve = [&at](uint64_t bytes) { const uint64_t o = at; at += (bytes + 255u) / 256u * 256u; return o; };
    const uint64_t o_text = carve(out_at + 32), o_extra = carve(extra), o_dense = carve(dense_words * 4 + 16), o_jobs = carve(jobs.size() * sizeof(mk_gz_job) + 16);
    r.blk_out = gz_block_get(c, at + 256, &r.out_bytes, 1);
    if (!r.blk_out) { set_error("no device memory for the text of %u files (%.1f GB)", n, at / 1e9); return MK_ERR_NOMEM; }
    r.d_text = r.blk_out + o_text;
    r.d_extra = r.blk_out + o_extra;
    MK_TRY(r.up(r.d_streams, r.streams.data(), (size_t)n * sizeof(mk_gz_stream)));
    if (!jobs.empty()) {
        for (mk_gz_job &job : jobs) job.tok_ptr = (uint64_t)(r.blk_out + o_dense) + job.tok_ptr * 4u;
        mk_gz_job *d_jobs = reinterpret_cast<mk_gz_job *>(r.blk_out + o_jobs);
        MK_TRY(r.up(d_jobs, jobs.data(), jobs.size() * sizeof(mk_gz_job)));
        MK_HIP(hipMemsetAsync(r.d_cnt + 2, 0, 4, st));
        A.jobs = d_jobs; A.n_jobs = (uint32_t)jobs.size();
        hipLaunchKernelGGL(gz_tokens_kernel, dim3(std::min<uint32_t>(r.tok_wgs, (A.n_jobs + 63u) / 64u)), dim3(64), r.lds_tok, st, A);
        MK_HIP(hipGetLastError());
    }
    // ---- tokens -> text
    ResolveConsts K;
    for (uint32_t l = 0; l < 64; ++l) K.lane_shift[l] = x_pow_bytes(64u * (63u - l));
    K.block_shift = x_pow_bytes(kFlush);
    K.byte_shift = x_pow_bytes(1);
    const size_t lds2 = r.lds_text;                                   // CRC tables, the streams' windows, the lanes' sinks (or more: gz_open)
    MK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gz_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(gz_resolve_kernel, dim3((n + kResolveWaves - 1u) / kResolveWaves), dim3(64u * kResolveWaves), lds2, st, r.d_gz, r.d_streams, n, r.d_segs, r.d_chain,
                       r.d_text, K);
    MK_HIP(hipGetLastError());
    MK_TRY(r.down(r.streams.data(), r.d_streams, (size_t)n * sizeof(mk_gz_stream)));
    r.t_text = now() - t3;                                           // (queued: the caller settles, with whatever it queues behind)
    if (getenv("MIEKKI_VERBOSE"))
        fprintf(stderr, "[gz] %u files, %.2f GB: %u segments, %u streams decoded again with exact rooms (%zu segments); blocks made ready %.3f s, block starts + tokens + chains %.3f s, "
                        "text queued after %.3f s (%.2f GB); so far %llu device allocations in %.3f s, %llu waits for a page-locked piece in %.3f s\n", n, r.in_at / 1e9, r.n_segs,
                r.n_rewritten, jobs.size(), r.t_open, r.t_blocks, r.t_text, out_at / 1e9, (unsigned long long)g_malloc_n.load(), g_malloc_us.load() / 1e6,
                (unsigned long long)g_stage_waits.load(), g_stage_wait_us.load() / 1e6);
    if (getenv("MIEKKI_VERBOSE")) fprintf(stderr, "[gz] in mk_gz_put so far %.3f s\n", g_put_us.load() / 1e6);
    return MK_OK;
}

}  // namespace mk

using namespace mk;

// A batch of genome FILES (gzip members holding FASTA text): inflated, their sequences as index_file_of_file would read
// them (Miekki.cpp:559-567) measured (fasta.hip) -- the text stays on the device until mk_gz_free; mk_index_append_gz strips
// it straight into the build's sequence buffer, sixty-four genomes a call.
struct mk_gz_batch {
    GzRun run;
    std::vector<uint32_t> chunk_first;
    std::vector<uint64_t> seq_len;
    uint32_t n_chunks = 0;
    void *d_scratch = nullptr;
    bool ran = false;
    mutable hipEvent_t last_use = nullptr;                          // behind the last strip kernel queued on the batch's text
};

extern "C" {

// Whole gzip files -> their text, inflated on the device; status[i] != MK_GZ_OK: that file is for the host's inflater
// (nothing of it is returned).
int mk_gz_inflate(mk_ctx *c, const uint8_t *const *gz, const uint64_t *gz_bytes, uint32_t n, uint8_t *const *out, const uint64_t *out_room,
                  uint64_t *out_bytes, int32_t *status)
{
    if (!c || (n && (!gz || !gz_bytes || !out || !out_room || !out_bytes || !status))) { set_error("null argument"); return MK_ERR_ARG; }
    if (!n) return MK_OK;
    GzRun r;
    MK_TRY(gz_open(r, c, gz_bytes, n));
    for (uint32_t i = 0; i < n; ++i) if (r.streams[i].in_len) MK_TRY(gz_put(r, i, 0, gz[i], r.streams[i].in_len, false));
    MK_TRY(gz_finish(r, out_room, nullptr));
    MK_TRY(r.settle());
    for (uint32_t i = 0; i < n; ++i) {
        const mk_gz_stream &j = r.streams[i];
        status[i] = (int32_t)j.status;
        out_bytes[i] = j.status == MK_GZ_OK ? j.out_len : 0;
        if (j.status == MK_GZ_OK && j.out_len) MK_HIP(hipMemcpyAsync(out[i], r.d_text + j.out_off, j.out_len, hipMemcpyDeviceToHost, r.st));
    }
    MK_TRY(r.wait_stream(r.st));
    return MK_OK;
}

// The batch in three steps, for callers that read their files themselves (host/fasta_reader.cpp: sixteen readers put the
// files of a batch side by side): mk_gz_open fixes the layout from the files' sizes, mk_gz_put moves bytes -- from a piece
// mk_gz_stage lent (page-locked: a DMA the caller does not wait for; the piece is the library's again) or from any other
// memory --, mk_gz_run does the rest and returns when the sequences' lengths are known.
int mk_gz_open(mk_ctx *c, const uint64_t *gz_bytes, uint32_t n, mk_gz_batch **out)
{
    if (!c || !out || (n && !gz_bytes)) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    std::unique_ptr<mk_gz_batch> b(new mk_gz_batch());
    b->chunk_first.assign(n + 1, 0); b->seq_len.assign(n, 0);
    MK_TRY(gz_open(b->run, c, gz_bytes, n));
    *out = b.release();
    return MK_OK;
}

void *mk_gz_stage(mk_gz_batch *b, uint64_t *cap)
{
    if (!b || !b->run.c) return nullptr;
    void *p = gz_stage_get(b->run.c);
    if (p && cap) *cap = kStagePiece;
    return p;
}

int mk_gz_put(mk_gz_batch *b, uint32_t i, uint64_t at, const void *data, uint64_t bytes, int staged)
{
    if (!b || (!data && bytes)) { set_error("null argument"); return MK_ERR_ARG; }
    if (b->ran) { set_error("the batch has run"); return MK_ERR_ARG; }
    return gz_put(b->run, i, at, data, bytes, staged != 0);
}

int mk_gz_layout(const mk_gz_batch *b, uint64_t *offsets)
{
    if (!b || !offsets) { set_error("null argument"); return MK_ERR_ARG; }
    for (uint32_t i = 0; i < b->run.n; ++i) offsets[i] = b->run.streams[i].in_off;
    offsets[b->run.n] = b->run.in_at;
    return MK_OK;
}

int mk_gz_put_span(mk_gz_batch *b, uint32_t first, uint32_t count, const void *data, uint64_t bytes, int staged)
{
    if (!b || (!data && bytes)) { set_error("null argument"); return MK_ERR_ARG; }
    if (b->ran) { set_error("the batch has run"); return MK_ERR_ARG; }
    return gz_put(b->run, first, 0, data, bytes, staged != 0, std::max(count, 1u));
}

int mk_gz_run(mk_gz_batch *b)
{
    if (!b) { set_error("null argument"); return MK_ERR_ARG; }
    if (b->ran) { set_error("the batch has run"); return MK_ERR_ARG; }
    b->ran = true;
    GzRun &r = b->run;
    const uint32_t n = r.n;
    if (!n) return MK_OK;
    uint64_t o_first = 0, o_len = 0, o_scratch = 0;
    auto extra = [&](const std::vector<mk_gz_stream> &streams) {      // behind the text: the chunks' table, the lengths, the scan's scratch
        for (uint32_t i = 0; i < n; ++i) b->chunk_first[i + 1] = b->chunk_first[i] + (streams[i].status == MK_GZ_OK ? (streams[i].out_len + 4095u) / 4096u : 0u);
        b->n_chunks = b->chunk_first[n];
        o_first = 0; o_len = (((uint64_t)n + 1) * 4 + 255u) / 256u * 256u; o_scratch = o_len + ((uint64_t)n * 8 + 255u) / 256u * 256u;
        return o_scratch + fasta_scratch_bytes(b->n_chunks);
    };
    MK_TRY(gz_finish(r, nullptr, extra));
    uint32_t *d_first = reinterpret_cast<uint32_t *>(r.d_extra + o_first);
    uint64_t *d_len = reinterpret_cast<uint64_t *>(r.d_extra + o_len);
    b->d_scratch = r.d_extra + o_scratch;
    // (a stream that failed in the text pass -- a CRC -- keeps its chunks in the table: the kernels skip them by its status)
    MK_TRY(r.up(d_first, b->chunk_first.data(), ((size_t)n + 1) * 4));
    MK_TRY(launch_fasta_count(r.c, r.d_text, r.d_streams, n, d_first, b->n_chunks, b->d_scratch, d_len, r.st));
    MK_TRY(r.down(b->seq_len.data(), d_len, (size_t)n * 8));
    MK_TRY(r.settle());
    return MK_OK;
}

int mk_gz_unpack(mk_ctx *c, const uint8_t *const *gz, const uint64_t *gz_bytes, uint32_t n, mk_gz_batch **out)
{
    if (!c || !out || (n && (!gz || !gz_bytes))) { set_error("null argument"); return MK_ERR_ARG; }
    *out = nullptr;
    mk_gz_batch *raw = nullptr;
    MK_TRY(mk_gz_open(c, gz_bytes, n, &raw));
    std::unique_ptr<mk_gz_batch> b(raw);
    for (uint32_t i = 0; i < n; ++i) if (b->run.streams[i].in_len) MK_TRY(gz_put(b->run, i, 0, gz[i], b->run.streams[i].in_len, false));
    MK_TRY(mk_gz_run(b.get()));
    *out = b.release();
    return MK_OK;
}

int mk_gz_sequence(const mk_gz_batch *b, uint32_t i, uint64_t *len, int32_t *status)
{
    if (!b || i >= b->run.n || !len || !status) { set_error("bad argument"); return MK_ERR_ARG; }
    *status = (int32_t)b->run.streams[i].status;
    *len = b->run.streams[i].status == MK_GZ_OK ? b->seq_len[i] : 0;
    return MK_OK;
}

void mk_gz_free(mk_gz_batch *b)
{
    if (!b) return;
    if (b->run.c) (void)hipSetDevice(b->run.c->p.device);
    if (b->last_use) {                                               // (the appends' strip kernels read the batch's text: the last of them)
        (void)hipEventSynchronize(b->last_use);
        (void)hipEventDestroy(b->last_use);
    }
    delete b;                                                        // (the blocks go back to the context's list)
}

// the blocks kept between batches go back to the device (after a build, before the queries' buffers are sized)
void mk_gz_trim(mk_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    gz_release_staging(c);                                           // (block makers at work finish first)
    std::lock_guard<std::mutex> g(c->gz_m);
    for (auto &blk : c->gz_blocks) (void)hipFree(blk.first);
    c->gz_blocks.clear();
    for (auto &pin : c->gz_pins) (void)hipHostFree(pin.first);
    c->gz_pins.clear();
}

}  // extern "C"

namespace mk {
// the pieces files are read into and the streams their copies run on (pieces a caller still holds stay its own)
void gz_release_staging(mk_ctx *c)
{
    gz_forget_keeper(c);                                             // (called when the context goes, or gives everything back: it registers again with its next block)
    std::unique_lock<std::mutex> g(c->gz_m);
    c->gz_closing = true;                                            // (no new block maker starts; those at work finish first)
    c->gz_cv.wait(g, [&] { return c->gz_making == 0; });
    c->gz_closing = false;
    for (hipStream_t &u : c->gz_up) if (u) { (void)hipStreamSynchronize(u); (void)hipStreamDestroy(u); u = nullptr; }
    for (auto &pc : c->gz_stage_busy) c->gz_stage_free.push_back(pc);
    c->gz_stage_busy.clear();
    for (auto &pc : c->gz_stage_free) { pinned_free(pc.p); (void)hipEventDestroy(pc.ev); --c->gz_stage_made; }
    c->gz_stage_free.clear();
}

// the batch's text for the strip kernels of mk_index_append_gz (api.hip)
int gz_batch_strip(mk_ctx *c, const mk_gz_batch *b, const uint32_t *which, uint32_t m, uint8_t *d_dst, const uint64_t *dst_off, hipStream_t st)
{
    return launch_fasta_strip(c, b->run.d_text, b->run.d_streams, which, m, b->chunk_first.data(), b->n_chunks, b->d_scratch, d_dst, dst_off, st);
}
int gz_batch_used(const mk_gz_batch *b, hipStream_t st)
{
    if (!b->last_use) MK_HIP(hipEventCreateWithFlags(&b->last_use, hipEventBlockingSync | hipEventDisableTiming));
    MK_HIP(hipEventRecord(b->last_use, st));
    return MK_OK;
}
uint32_t gz_batch_size(const mk_gz_batch *b) { return b->run.n; }
bool gz_batch_ok(const mk_gz_batch *b, uint32_t i) { return b->ran && b->run.streams[i].status == MK_GZ_OK; }
uint64_t gz_batch_len(const mk_gz_batch *b, uint32_t i) { return b->seq_len[i]; }
const mk_ctx *gz_batch_owner(const mk_gz_batch *b) { return b->run.c; }
}  // namespace mk
