// Index columns that arrive Huffman-coded are inflated on the GPU (`-i`, Miekki.cpp:699-719).
//
// The reference reads an index through zlib, one symbol at a time on one thread; this program's own index files hold the
// fingerprint columns as deflate blocks of literals only (host/fastz.cpp, deflate_huffman_only: one Huffman code of at
// most 12 bits per MiB of columns, a block of its own for every 16 KiB) and say in each gzip member's extra field where
// every block's first symbol lies and which code it uses.  Inflating such a file on the host costs 9 s of sixteen threads
// for the 106 GB of a 100,000-genome index; here the members' bytes cross PCIe as they are (a quarter fewer than the
// columns) and ONE LANE PER BLOCK decodes them: a wave takes the 64 blocks of one code, builds the code's 4,096-entry
// lookup table in LDS (canonical codes from the lengths, bit-reversed as the stream holds them), and every lane walks its
// block -- look up twelve bits, drop the code's length, keep the byte, CRC it -- into the staging rows that
// launch_convert_columns then lays out as the matrix.  Decoding is serial per block and the blocks are many: 6.8 million
// of them in that index.
//
// What leaves the device besides the rows: the CRC-32 remainder of every block's bytes (start value 0, no final
// complement), which the host folds together (crc32_shift, host/fastz.cpp) and compares with each member's trailer, and
// the number of blocks that did not end in an end-of-block code where their length says they must.
#include "mk_internal.hpp"

namespace mk {

namespace {

constexpr uint32_t kHuffBits = 12, kHuffSize = 1u << kHuffBits, kHuffSyms = 257;

__global__ __launch_bounds__(256) void huff_decode_kernel(const uint8_t *__restrict__ payload, uint64_t payload_bytes,
                                                          const mk_huff_block *__restrict__ blocks, uint32_t n_blocks,
                                                          const uint8_t *__restrict__ lens_all, uint32_t n_codes,
                                                          uint8_t *__restrict__ out, uint64_t out_bytes,
                                                          uint32_t *__restrict__ crc_out, uint32_t *__restrict__ bad)
{
    __shared__ uint16_t tab[4][kHuffSize];                         // symbol | length << 9, per wave
    __shared__ uint32_t crc_tab[256];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // the CRC-32 table (reflected polynomial 0xEDB88320), one entry per thread
    {
        uint32_t c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        crc_tab[threadIdx.x] = c;
    }
    const uint32_t group = blockIdx.x * 4u + wave;                 // 64 blocks that share a code
    const bool live = (uint64_t)group * 64u < n_blocks;
    const mk_huff_block blk = live ? blocks[(uint64_t)group * 64u + lane] : mk_huff_block{0, 0, 0, 0};
    const uint32_t code = live ? __builtin_amdgcn_readfirstlane(blk.code) : 0u;
    bool table_ok = live && code < n_codes;
    // ---- the code's table: canonical codes in symbol order within each length (RFC 1951 3.2.2)
    uint32_t my_len[5], my_code[5];
    if (table_ok) {
        const uint8_t *__restrict__ lens = lens_all + (uint64_t)code * kHuffSyms;
        uint32_t count[kHuffBits + 1];
#pragma unroll
        for (uint32_t l = 0; l <= kHuffBits; ++l) count[l] = 0;
#pragma unroll
        for (uint32_t c5 = 0; c5 < 5; ++c5) {
            const uint32_t s = c5 * 64u + lane;
            my_len[c5] = s < kHuffSyms ? lens[s] : 0u;
            if (my_len[c5] > kHuffBits) table_ok = false;
        }
        table_ok = __all(table_ok);
        // first code of each length, the same in every lane
        uint32_t first[kHuffBits + 1];
#pragma unroll
        for (uint32_t l = 1; l <= kHuffBits; ++l) {
            uint32_t n = 0;
#pragma unroll
            for (uint32_t c5 = 0; c5 < 5; ++c5) n += (uint32_t)__popcll(__ballot(my_len[c5] == l));
            count[l] = n;
        }
        uint32_t next = 0, kraft = 0;
        first[0] = 0;
#pragma unroll
        for (uint32_t l = 1; l <= kHuffBits; ++l) {
            next = (next + (l > 1 ? count[l - 1] : 0u)) << 1;
            first[l] = next;
            kraft += count[l] << (kHuffBits - l);
        }
        if (kraft != kHuffSize) table_ok = false;                  // a complete code fills the table exactly
        // a symbol's code: the first of its length + how many symbols of that length come before it
        uint32_t before[kHuffBits + 1];
#pragma unroll
        for (uint32_t l = 0; l <= kHuffBits; ++l) before[l] = 0;
#pragma unroll
        for (uint32_t c5 = 0; c5 < 5; ++c5) {
            my_code[c5] = 0;
#pragma unroll
            for (uint32_t l = 1; l <= kHuffBits; ++l) {
                const unsigned long long m = __ballot(my_len[c5] == l);
                if (my_len[c5] == l) my_code[c5] = first[l] + before[l] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                before[l] += (uint32_t)__popcll(m);
            }
        }
    }
    __syncthreads();                                               // (crc_tab; the waves' tables are their own)
    if (table_ok) {
#pragma unroll
        for (uint32_t c5 = 0; c5 < 5; ++c5) {
            const uint32_t l = my_len[c5];
            if (!l) continue;
            const uint32_t r = __brev(my_code[c5]) >> (32u - l);
            const uint16_t e = (uint16_t)((c5 * 64u + lane) | (l << 9));
            for (uint32_t i = r; i < kHuffSize; i += 1u << l) tab[wave][i] = e;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    if (!live) return;
    const uint64_t slot = (uint64_t)group * 64u + lane;
    if (!blk.out_len) { crc_out[slot] = 0; return; }
    // a block the table cannot serve, or one that points outside what was handed over: reported, nothing written
    if (!table_ok || blk.out > out_bytes || blk.out_len > out_bytes - blk.out || (blk.bit >> 3) >= payload_bytes) {
        crc_out[slot] = 0;
        atomicAdd(bad, 1u);
        return;
    }
    // ---- the lane's block
    const uint8_t *__restrict__ p = payload + (blk.bit >> 3);
    const uint8_t *const p_end = payload + payload_bytes;
    auto load32 = [&](const uint8_t *q) -> uint32_t {              // four bytes, or what is left of the payload
        if (q + 4 <= p_end) { uint32_t v; __builtin_memcpy(&v, q, 4); return v; }
        uint32_t v = 0;
        for (uint32_t i = 0; i < 4 && q + i < p_end; ++i) v |= (uint32_t)q[i] << (8 * i);
        return v;
    };
    uint64_t bb = ((uint64_t)load32(p) | ((uint64_t)load32(p + 4) << 32)) >> (blk.bit & 7u);
    uint32_t bc = 64u - (uint32_t)(blk.bit & 7u);
    p += 8;
    uint8_t *__restrict__ o = out + blk.out;
    uint32_t crc = 0;
    const uint16_t *__restrict__ t = tab[wave];
    auto next_symbol = [&]() -> uint32_t {
        if (bc < 32u) { bb |= (uint64_t)load32(p) << bc; p += 4; bc += 32u; }
        const uint32_t e = t[(uint32_t)bb & (kHuffSize - 1u)];
        const uint32_t l = e >> 9;
        bb >>= l; bc -= l;
        return e & 0x1ffu;
    };
    uint32_t done = 0, wrong = 0;
    // bytes up to the first 16-byte boundary of the output, then 16 at a time, then the rest
    const uint32_t head = min(blk.out_len, (uint32_t)((16u - ((uintptr_t)o & 15u)) & 15u));
    for (; done < head; ++done) {
        const uint32_t s = next_symbol();
        wrong |= s >> 8;
        o[done] = (uint8_t)s;
        crc = crc_tab[(crc ^ s) & 0xffu] ^ (crc >> 8);
    }
    for (; done + 16u <= blk.out_len; done += 16u) {
        uint32_t w[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t s = next_symbol();
                wrong |= s >> 8;
                v |= (s & 0xffu) << (8 * b);
                crc = crc_tab[(crc ^ s) & 0xffu] ^ (crc >> 8);
            }
            w[k] = v;
        }
        *reinterpret_cast<uint4 *>(o + done) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (; done < blk.out_len; ++done) {
        const uint32_t s = next_symbol();
        wrong |= s >> 8;
        o[done] = (uint8_t)s;
        crc = crc_tab[(crc ^ s) & 0xffu] ^ (crc >> 8);
    }
    if (next_symbol() != 256u) wrong = 1;                          // the block ends where its length says
    crc_out[slot] = crc;
    if (wrong) atomicAdd(bad, 1u);
}

}  // namespace

int launch_huff_decode(mk_ctx *c, const uint8_t *d_payload, uint64_t payload_bytes, const mk_huff_block *d_blocks, uint32_t n_blocks,
                       const uint8_t *d_lens, uint32_t n_codes, uint8_t *d_out, uint64_t out_bytes, uint32_t *d_crc, uint32_t *d_bad)
{
    hipLaunchKernelGGL(huff_decode_kernel, dim3((n_blocks / 64u + 3u) / 4u), dim3(256), 0, c->stream, d_payload, payload_bytes, d_blocks,
                       n_blocks, d_lens, n_codes, d_out, out_bytes, d_crc, d_bad);
    MK_HIP(hipGetLastError());
    return MK_OK;
}

}  // namespace mk
