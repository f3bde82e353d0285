// Parallel gzip for the index file (host only, plain zlib + std::thread).
//
// Writer: the byte stream is cut into 32 MiB blocks, each deflated (level 1, like
// zstr::ofstream, zstr.hpp:82) by its own thread into its own gzip MEMBER, written in order by one
// output thread.  Every member's header carries an FEXTRA subfield "MK" with the
// compressed payload size, the way BGZF does, so that a reader can find member
// boundaries without inflating; members coded with Huffman codes only (the index's
// fingerprint columns) carry a second subfield "MH": the code lengths of every MiB and
// the bit at which every 16 KiB block's first symbol starts (fastz.hpp, HuffIndex) --
// what a decoder needs to take all blocks at once (the GPU inflates them: huff.hip).  Any gzip reader -- zlib's gzread, the reference's
// zstr::ifstream (which restarts its inflator at each member end, zstr.hpp:186-190),
// gunzip -- reads such a file as one stream.
//
// Reader: members with the "MK" subfield are read (pread) and inflated concurrently,
// each by its own thread -- straight into the caller's buffer when a request covers
// whole members (the index loader asks for a few hundred megabytes of columns at a
// time, into page-locked memory), through a buffer of their own otherwise; anything
// else (the reference's own dumps, plain files) goes through gzread.
#pragma once
#include <zlib.h>

#include "fastz.hpp"

#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mkhost {

class ParallelGzipWriter {
public:
    ParallelGzipWriter(const std::string &path, unsigned threads);
    ~ParallelGzipWriter();
    bool ok() const { return fd_ >= 0 && !failed_; }
    void write(const void *p, size_t n);
    // n zero bytes.  Whole blocks of zeros -- the reference's index always carries its full 2^(b-3)-byte Bloom filter,
    // of which a 2k-bit k-mer reaches only the first part: 960 MiB of zeros at k = 31, b = 33 -- are not deflated again
    // and again: the member of an all-zero block is made once and written as often as needed.  Same stream, same format.
    void write_zeros(size_t n);
    // deflate strategy of the blocks submitted from now on (zlib: Z_DEFAULT_STRATEGY, Z_HUFFMAN_ONLY, ...).  Fingerprint
    // columns have no repeats for LZ77 to find (profiles/r3_column_entropy.txt: level 1 gets 1.35:1, an order-0 entropy
    // coder 1.39:1): Huffman-only codes them as small, several times faster.  Any inflater reads either.
    void set_strategy(int strategy) { strategy_ = strategy; }
    // level 0 = stored blocks (no deflate work at all, the file grows by what level 1 would have saved), 1 = the reference's
    void set_level(int level) { level_ = level; }
    void flush_block() { if (!cur_.empty()) submit(); }   // what is buffered becomes a (short) member of its own
    // A whole block handed over without a copy: n <= kBlock bytes at p, which stay the CALLER's (page-locked memory the
    // GPU exported into, say) until on_done runs on the output thread, after the member has been written.  Only at a block
    // boundary (nothing buffered by write()): false otherwise, and nothing is taken.
    bool write_block(const uint8_t *p, size_t n, std::function<void()> on_done);
    bool finish();                       // flushes, closes; false on any error
    static constexpr size_t kBlock = 32u << 20;
private:
    struct Job {
        std::vector<uint8_t> in;
        std::unique_ptr<uint8_t[]> out; size_t out_at = 0, out_n = 0;   // the finished member: out[out_at .. out_at + out_n)
        const uint8_t *ext = nullptr; size_t ext_n = 0;          // write_block: the caller's bytes instead of `in`
        const std::vector<uint8_t> *ready = nullptr;             // a member made earlier (a run of zeros): nothing to deflate
        std::function<void()> on_done;
        std::thread th;
        bool bad = false;
        int strategy = 0, level = 1;
    };
    static void deflate_block(Job *j);
    void submit();
    void enqueue(std::unique_ptr<Job> j);
    void output_loop();
    int fd_ = -1;
    unsigned nthreads_;
    std::vector<uint8_t> cur_;
    std::deque<std::unique_ptr<Job>> jobs_;                      // handed to the output thread, in stream order
    std::mutex m_;
    std::condition_variable cv_;
    std::thread out_thread_;
    unsigned in_flight_ = 0;
    bool closing_ = false, finished_ = false;
    bool failed_ = false, wrote_any_ = false;
    int strategy_ = 0;                   // Z_DEFAULT_STRATEGY
    int level_ = 1;
    uint64_t file_off_ = 0;              // where the output thread writes next
    std::vector<uint8_t> zero_member_;   // the gzip member of kBlock zero bytes, made on first use
};

class ParallelGzipReader {
public:
    ParallelGzipReader(const std::string &path, unsigned threads);
    ~ParallelGzipReader();
    bool ok() const { return (fd_ >= 0 || gz_) && !failed_; }
    // up to n bytes; 0 at the end of the stream (or on error: check ok())
    size_t read_some(void *dst, size_t n);
    // exactly n bytes or false (truncated / corrupt)
    bool read(void *dst, size_t n) { return read_some(dst, n) == n; }
    bool parallel() const { return fd_ >= 0; }
    unsigned threads() const { return nthreads_; }
    // One member as the file holds it: where its deflate stream lies, what it inflates to, and -- for members with the
    // "MH" subfield -- the block index of that stream.
    struct Member {
        uint64_t begin = 0, at = 0, payload = 0;                 // the member's first byte, its deflate stream's first byte and length
        uint32_t crc = 0, isize = 0;
        bool indexed = false;
        std::vector<uint8_t> lens;                               // 257 code lengths per MiB of output
        std::vector<uint32_t> sym_bit;                           // per 16 KiB block: bit of the deflate stream where its first symbol starts
    };
    // The members from the reader's position to the end of the file (headers and trailers only; nothing is inflated, the
    // position stays).  Only at a member boundary -- nothing read ahead, nothing half handed out: false otherwise.
    bool list_members(std::vector<Member> &out);
    // continue reading at the member that starts at file offset `begin` (one of list_members')
    bool seek_member(uint64_t begin);
    // `n` bytes of the file as they are (a member's deflate stream, say)
    bool read_raw(uint64_t at, void *dst, size_t n) const;
private:
    struct Job {
        Member m;
        uint8_t *dst = nullptr;                                  // the caller's memory, or `own`
        std::unique_ptr<uint8_t[]> own;
        std::thread th;
        bool bad = false;
    };
    static void inflate_member(Job *j, int fd);
    bool next_member(Member &m);         // the member at scan_; false at the clean end of the file or on error
    void prefetch();
    void start(std::unique_ptr<Job> j);
    int fd_ = -1;                        // "MK" member mode
    gzFile gz_ = nullptr;                // generic mode
    unsigned nthreads_;
    uint64_t scan_ = 0, file_size_ = 0;
    bool have_peek_ = false;
    Member peek_;
    std::deque<std::unique_ptr<Job>> jobs_;                      // members on their way into buffers of their own, in order
    std::unique_ptr<uint8_t[]> cur_;     // the member being handed out piecewise
    size_t cur_n_ = 0, cur_pos_ = 0;
    unsigned depth_ = 1;                 // how far small reads look ahead: grows while they keep coming
    bool eof_ = false, failed_ = false;
};

}  // namespace mkhost
