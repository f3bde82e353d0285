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

namespace mkhost {

// zlib's crc32(crc, p, n) by carry-less multiplication (PCLMULQDQ) where the CPU has it
uint32_t crc32_fast(uint32_t crc, const void *p, size_t n);

enum { FZ_OK = 0, FZ_BAD = -1, FZ_OUT_FULL = -2, FZ_IN_SHORT = -3 };
// One raw deflate stream, from its first block to the end of its final block.  `in_used` = bytes of input the stream
// occupied (the final block's last partial byte included), `out_len` = bytes produced.
int inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_len);

// One gzip member at in[0 .. in_len): header, stream, CRC-32 and ISIZE checked.  `in_used` = the member's length.
int gunzip_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_len);

// n bytes as a raw deflate stream of dynamic-Huffman blocks without matches (or stored blocks where that is smaller).
// `out` must hold huffman_only_bound(n) bytes; returns the stream's length.
size_t huffman_only_bound(size_t n);
size_t deflate_huffman_only(const uint8_t *in, size_t n, uint8_t *out);

}  // namespace mkhost
