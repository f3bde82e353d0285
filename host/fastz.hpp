// Whole-buffer DEFLATE pieces for the host side of the index file and of gzip'd FASTA input (RFC 1951 / 1952), written for
// the two shapes of data this program moves by the hundred gigabytes:
//   * fingerprint columns -- bytes with ~5.8 bits of order-0 entropy and nothing for LZ77 to find: the writer codes them
//     with Huffman codes only (literals + end-of-block, a fresh code every 256 KiB), the reader decodes such members;
//   * gzip'd FASTA -- ordinary deflate streams of any make, inflated in one go into a buffer of known size.
// zlib does both at 80 - 250 MB/s per thread (its inflate takes one symbol per turn of a state machine that can stop
// anywhere, its deflate runs every byte through the match finder's bookkeeping even when told to find nothing); with whole
// buffers on both sides neither is needed.  Every stream written here is read by any inflater (zlib's, the reference's
// zstr); every stream zlib writes is read here -- and whatever this reader does not take (an error of any kind) the caller
// gives to zlib, which then reports what is wrong with it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mkhost {

// zlib's crc32(crc, p, n) by carry-less multiplication (PCLMULQDQ) where the CPU has it
uint32_t crc32_fast(uint32_t crc, const void *p, size_t n);

// CRC-32 remainders of pieces put together (the GPU inflater hands back one per block, start value 0 and no final
// complement -- the linear part of the CRC): crc32_shift(r, n) = the remainder r after n more zero bytes, so that
// raw(A || B) = crc32_shift(raw(A), |B|) ^ raw(B); crc32_from_raw turns the raw remainder of a whole buffer of n bytes
// into zlib's crc32() of it.
uint32_t crc32_shift(uint32_t raw, uint64_t n_bytes);
uint32_t crc32_shift_factor(uint64_t n_bytes);                      // for many shifts by one length: crc32_shift_by(factor, raw)
uint32_t crc32_shift_by(uint32_t factor, uint32_t raw);
uint32_t crc32_from_raw(uint32_t raw, uint64_t n_bytes);
uint32_t crc32_raw(const void *p, size_t n);                         // (the remainder itself: tests, small pieces)

enum { FZ_OK = 0, FZ_BAD = -1, FZ_OUT_FULL = -2, FZ_IN_SHORT = -3 };
// One raw deflate stream, from its first block to the end of its final block.  `in_used` = bytes of input the stream
// occupied (the final block's last partial byte included), `out_len` = bytes produced.
int inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_len);

// One gzip member at in[0 .. in_len): header, stream, CRC-32 and ISIZE checked.  `in_used` = the member's length.
int gunzip_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_len);

// Where the blocks of a deflate_huffman_only stream lie, for a decoder that takes them all at once (the GPU: huff.hip): the
// stream codes every kSuper bytes of input with one Huffman code (lengths of at most kMaxLen bits) and cuts them into
// blocks of kSub bytes, each a deflate block of its own with that code's header.
struct HuffIndex {
    static constexpr size_t kSuper = 1u << 20, kSub = 16u << 10;
    static constexpr uint32_t kMaxLen = 12;
    std::vector<uint8_t> lens;           // 257 code lengths (literals 0 .. 255, end-of-block) per stretch of kSuper bytes
    std::vector<uint64_t> sym_bit;       // per block: the bit of the stream at which its first symbol's code starts
    bool all_coded = true;               // false: some stretch went in as stored blocks (the two lists then say nothing)
};

// n bytes as a raw deflate stream of dynamic-Huffman blocks without matches (or stored blocks where that is smaller).
// `out` must hold huffman_only_bound(n) bytes; returns the stream's length.
size_t huffman_only_bound(size_t n);
size_t deflate_huffman_only(const uint8_t *in, size_t n, uint8_t *out, HuffIndex *index = nullptr);

}  // namespace mkhost
