#include "gzpar.hpp"

#include "fastz.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mkhost {

namespace {

constexpr size_t kHeader = 24;     // fixed gzip header (10) + XLEN (2) + "MK" subfield (4 + 8); an "MH" subfield may follow
constexpr size_t kMaxHeader = kHeader + 65535;

void put_le(uint8_t *p, uint64_t v, int n) { for (int i = 0; i < n; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
uint64_t get_le(const uint8_t *p, int n) { uint64_t v = 0; for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

bool is_mk_header(const uint8_t *h)
{
    return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && h[3] == 4 && get_le(h + 10, 2) >= 12 && h[12] == 'M' &&
           h[13] == 'K' && get_le(h + 14, 2) == 8;
}

bool pwrite_all(int fd, const uint8_t *p, size_t n, uint64_t at)
{
    while (n) {
        const ssize_t w = pwrite(fd, p, std::min<size_t>(n, 1u << 30), (off_t)at);
        if (w <= 0) return false;
        p += w; n -= (size_t)w; at += (uint64_t)w;
    }
    return true;
}

bool pread_all(int fd, uint8_t *p, size_t n, uint64_t at)
{
    while (n) {
        const ssize_t r = pread(fd, p, std::min<size_t>(n, 1u << 30), (off_t)at);
        if (r <= 0) return false;
        p += r; n -= (size_t)r; at += (uint64_t)r;
    }
    return true;
}

}  // namespace

// ------------------------------------------------------------------ writer
// Three kinds of threads: the caller's (fills blocks), one deflate thread per block in flight, and ONE output thread that
// takes the finished members in order and writes them -- so that the caller neither deflates nor waits for the file.
// (One writer on purpose.  Members written by the threads that made them -- each at its own offset, pwrite or a copy into
// a mapping of the file -- were measured against it on the GPU boxes' memory file system, profiles/r4_tmpfs_io.txt: one
// thread writes 7.2 GB/s, four 3.8 - 5.6, sixteen 3.7 - 3.9: allocating a file's pages does not scale across threads there,
// and a 76 GB dump took 13.7 / 22.9 s that way.)
ParallelGzipWriter::ParallelGzipWriter(const std::string &path, unsigned threads)
    : fd_(open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644)), nthreads_(std::max(1u, threads))
{
    if (fd_ >= 0) out_thread_ = std::thread([this] { output_loop(); });
}

ParallelGzipWriter::~ParallelGzipWriter()
{
    (void)finish();
}

void ParallelGzipWriter::output_loop()
{
    for (;;) {
        std::unique_ptr<Job> j;
        {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [this] { return !jobs_.empty() || closing_; });
            if (jobs_.empty()) return;
            j = std::move(jobs_.front());
            jobs_.pop_front();
        }
        j->th.join();
        const uint8_t *data = j->ready ? j->ready->data() : j->out.get() + j->out_at;
        const size_t n = j->ready ? j->ready->size() : j->out_n;
        if (j->bad || !pwrite_all(fd_, data, n, file_off_)) failed_ = true; else file_off_ += n;
        j->out.reset();
        if (j->on_done) j->on_done();
        {
            std::lock_guard<std::mutex> g(m_);
            --in_flight_;
        }
        cv_.notify_all();
    }
}

// hand a job to its thread and to the output thread (in order); waits while as many blocks are in flight as there are threads
void ParallelGzipWriter::enqueue(std::unique_ptr<Job> j)
{
    wrote_any_ = true;
    if (fd_ < 0) { failed_ = true; if (j->on_done) j->on_done(); return; }
    Job *raw = j.get();
    raw->th = std::thread([raw] { if (!raw->ready) deflate_block(raw); });
    {
        std::unique_lock<std::mutex> g(m_);
        jobs_.push_back(std::move(j));
        ++in_flight_;
        cv_.notify_all();
        cv_.wait(g, [this] { return in_flight_ < nthreads_ + 2; });
    }
}

void ParallelGzipWriter::write(const void *p, size_t n)
{
    const uint8_t *c = (const uint8_t *)p;
    while (n) {
        const size_t take = std::min(n, kBlock - cur_.size());
        cur_.insert(cur_.end(), c, c + take);
        c += take; n -= take;
        if (cur_.size() == kBlock) submit();
    }
}

bool ParallelGzipWriter::finish()
{
    if (finished_) return !failed_;
    finished_ = true;
    if (fd_ >= 0 && (!cur_.empty() || !wrote_any_)) submit();    // an empty stream is still one (empty) member
    {
        std::lock_guard<std::mutex> g(m_);
        closing_ = true;
    }
    cv_.notify_all();
    if (out_thread_.joinable()) out_thread_.join();
    if (fd_ >= 0) { if (close(fd_) != 0) failed_ = true; fd_ = -1; }
    return !failed_;
}

void ParallelGzipWriter::deflate_block(Job *j)
{
    const uint8_t *in = j->ext ? j->ext : j->in.data();
    const size_t n = j->ext ? j->ext_n : j->in.size();
    uint64_t payload = 0;
    // the member is built behind room for the largest header; the header is then written right in front of the stream
    std::vector<uint8_t> mh;                                     // the "MH" subfield's data, when there is one
    if (j->strategy == Z_HUFFMAN_ONLY && j->level != 0) {
        // literals and Huffman codes only: fastz's coder (one pass for the histogram, one for the bits), not zlib's deflate
        // with its match finder idling (80 -> 500 MB/s per thread on fingerprint columns, the same stream size)
        j->out.reset(new uint8_t[kMaxHeader + huffman_only_bound(n) + 8]);
        HuffIndex hx;
        payload = deflate_huffman_only(in, n, j->out.get() + kMaxHeader, &hx);
        const size_t want = 8 + hx.lens.size() + 4 * hx.sym_bit.size();
        if (hx.all_coded && n && want + 4 + 12 <= 65535 && payload < (1ull << 29)) {
            mh.resize(want);
            put_le(mh.data(), hx.lens.size() / 257, 4);
            put_le(mh.data() + 4, hx.sym_bit.size(), 4);
            memcpy(mh.data() + 8, hx.lens.data(), hx.lens.size());
            for (size_t i = 0; i < hx.sym_bit.size(); ++i) put_le(mh.data() + 8 + hx.lens.size() + 4 * i, hx.sym_bit[i], 4);
        }
    } else {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, j->level, Z_DEFLATED, -15, 8, j->strategy) != Z_OK) { j->bad = true; return; }   // raw deflate
        const size_t bound = deflateBound(&zs, (uLong)n) + 64;
        j->out.reset(new uint8_t[kMaxHeader + bound + 8]);
        zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n;
        zs.next_out = j->out.get() + kMaxHeader; zs.avail_out = (uInt)bound;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) j->bad = true;
        payload = zs.total_out;
        deflateEnd(&zs);
    }
    const size_t xlen = 12 + (mh.empty() ? 0 : 4 + mh.size()), hsize = 12 + xlen;
    uint8_t *h = j->out.get() + kMaxHeader - hsize;
    h[0] = 0x1f; h[1] = 0x8b; h[2] = 8; h[3] = 4;                // FLG.FEXTRA
    put_le(h + 4, 0, 4); h[8] = 4; h[9] = 3;                     // mtime 0, XFL = fastest, OS = unix
    put_le(h + 10, xlen, 2); h[12] = 'M'; h[13] = 'K'; put_le(h + 14, 8, 2);
    put_le(h + 16, payload, 8);
    if (!mh.empty()) { h[24] = 'M'; h[25] = 'H'; put_le(h + 26, mh.size(), 2); memcpy(h + 28, mh.data(), mh.size()); }
    uint8_t *t = j->out.get() + kMaxHeader + payload;
    put_le(t, crc32_fast(0, in, n), 4);
    put_le(t + 4, n & 0xffffffffu, 4);
    j->out_at = kMaxHeader - hsize;
    j->out_n = hsize + payload + 8;
    std::vector<uint8_t>().swap(j->in);
}

void ParallelGzipWriter::submit()
{
    std::unique_ptr<Job> j(new Job());
    j->in.swap(cur_);
    cur_.reserve(kBlock);
    j->strategy = strategy_; j->level = level_;
    enqueue(std::move(j));
}

bool ParallelGzipWriter::write_block(const uint8_t *p, size_t n, std::function<void()> on_done)
{
    if (!cur_.empty() || n > kBlock) return false;
    if (!n) { if (on_done) on_done(); return true; }
    std::unique_ptr<Job> j(new Job());
    j->ext = p; j->ext_n = n; j->on_done = std::move(on_done);
    j->strategy = strategy_; j->level = level_;
    enqueue(std::move(j));
    return true;
}

void ParallelGzipWriter::write_zeros(size_t n)
{
    static const std::vector<uint8_t> zeros(1u << 20, 0);
    // up to the next block boundary (and whatever is less than a block at the end) as ordinary bytes
    while (n && !cur_.empty()) { const size_t take = std::min({n, zeros.size(), kBlock - cur_.size()}); write(zeros.data(), take); n -= take; }
    if (n >= kBlock && zero_member_.empty()) {
        Job j;
        j.in.assign(kBlock, 0);
        j.level = 1;
        deflate_block(&j);
        if (j.bad) failed_ = true; else zero_member_.assign(j.out.get() + j.out_at, j.out.get() + j.out_at + j.out_n);
    }
    while (n >= kBlock && !failed_) {                             // whole blocks: the ready-made member, in order
        std::unique_ptr<Job> j(new Job());
        j->ready = &zero_member_;
        enqueue(std::move(j));
        n -= kBlock;
    }
    while (n) { const size_t take = std::min(n, zeros.size()); write(zeros.data(), take); n -= take; }
}

// ------------------------------------------------------------------ reader
ParallelGzipReader::ParallelGzipReader(const std::string &path, unsigned threads) : nthreads_(std::max(1u, threads))
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { failed_ = true; return; }
    uint8_t h[kHeader];
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && pread(fd, h, kHeader, 0) == (ssize_t)kHeader && is_mk_header(h)) {
        fd_ = fd;
        file_size_ = (uint64_t)st.st_size;
        // (members are read with pread into buffers; inflating them out of a mapping of the file was level while it ran and
        // cost 2.5 s to take a 76 GB mapping down again: profiles/r4_cli_c3.txt)
    } else {
        close(fd);
        gz_ = gzopen(path.c_str(), "rb");                        // gzip of any make, or plain bytes
        if (!gz_) failed_ = true; else gzbuffer(gz_, 1 << 20);
    }
}

ParallelGzipReader::~ParallelGzipReader()
{
    for (auto &j : jobs_) if (j->th.joinable()) j->th.join();
    if (fd_ >= 0) close(fd_);
    if (gz_) gzclose(gz_);
}

// header and trailer of the member at scan_: a few bytes each, so that a member's size is known before anything is inflated
bool ParallelGzipReader::next_member(Member &m)
{
    if (have_peek_) { m = peek_; have_peek_ = false; return true; }
    if (eof_ || failed_) return false;
    if (scan_ == file_size_) { eof_ = true; return false; }
    uint8_t h[kHeader], t[8];
    if (scan_ + kHeader + 8 > file_size_ || !pread_all(fd_, h, kHeader, scan_) || !is_mk_header(h)) { failed_ = true; return false; }
    const uint64_t xlen = get_le(h + 10, 2);
    m = Member();
    m.begin = scan_;
    m.payload = get_le(h + 16, 8);
    m.at = scan_ + 12 + xlen;
    if (m.payload > (1ull << 31) || m.at + m.payload + 8 > file_size_ || !pread_all(fd_, t, 8, m.at + m.payload)) { failed_ = true; return false; }
    m.crc = (uint32_t)get_le(t, 4);
    m.isize = (uint32_t)get_le(t + 4, 4);
    if (m.isize > (1u << 30)) { failed_ = true; return false; }   // (the writer's members hold kBlock bytes at most)
    if (xlen > 12 + 4) {                                          // a second subfield: the block index of a Huffman-only stream
        std::vector<uint8_t> x((size_t)xlen - 12);
        if (!pread_all(fd_, x.data(), x.size(), scan_ + kHeader)) { failed_ = true; return false; }
        const size_t len = (size_t)get_le(x.data() + 2, 2);
        if (x[0] == 'M' && x[1] == 'H' && len + 4 <= x.size() && len >= 8) {
            const uint8_t *d = x.data() + 4;
            const uint64_t nsuper = get_le(d, 4), nsub = get_le(d + 4, 4);
            const uint64_t want_super = ((uint64_t)m.isize + HuffIndex::kSuper - 1) / HuffIndex::kSuper;
            uint64_t want_sub = 0;
            for (uint64_t sp = 0; sp < want_super; ++sp)
                want_sub += (std::min<uint64_t>(HuffIndex::kSuper, m.isize - sp * HuffIndex::kSuper) + HuffIndex::kSub - 1) / HuffIndex::kSub;
            if (nsuper == want_super && nsub == want_sub && len == 8 + nsuper * 257 + nsub * 4) {   // (anything else: not an index this reader knows)
                m.lens.assign(d + 8, d + 8 + nsuper * 257);
                m.sym_bit.resize((size_t)nsub);
                for (size_t i = 0; i < nsub; ++i) m.sym_bit[i] = (uint32_t)get_le(d + 8 + nsuper * 257 + 4 * i, 4);
                m.indexed = true;
            }
        }
    }
    scan_ = m.at + m.payload + 8;
    return true;
}

bool ParallelGzipReader::list_members(std::vector<Member> &out)
{
    out.clear();
    if (fd_ < 0 || failed_ || cur_pos_ < cur_n_ || !jobs_.empty()) return false;
    const uint64_t keep_scan = have_peek_ ? peek_.begin : scan_;
    const bool keep_eof = eof_;
    Member m;
    while (next_member(m)) out.push_back(m);
    const bool ok = !failed_;
    have_peek_ = false; eof_ = keep_eof; scan_ = keep_scan;
    return ok;
}

bool ParallelGzipReader::seek_member(uint64_t begin)
{
    if (fd_ < 0 || failed_ || cur_pos_ < cur_n_ || !jobs_.empty() || begin > file_size_) return false;
    have_peek_ = false; eof_ = false; scan_ = begin;
    return true;
}

bool ParallelGzipReader::read_raw(uint64_t at, void *dst, size_t n) const
{
    if (fd_ < 0 || at + n > file_size_) return false;
    return pread_all(fd_, (uint8_t *)dst, n, at);
}

void ParallelGzipReader::inflate_member(Job *j, int fd)
{
    const size_t payload = (size_t)j->m.payload, isize = j->m.isize;
    std::unique_ptr<uint8_t[]> buf(new uint8_t[payload + 16]);
    if (!pread_all(fd, buf.get(), payload, j->m.at)) { j->bad = true; return; }
    const uint8_t *src = buf.get();
    size_t used = 0, got = 0;
    if (inflate_raw(src, payload, j->dst, isize, &used, &got) != FZ_OK || got != isize || used != payload) {
        // whatever fastz does not take goes to zlib, which then decides what the member is worth (one spare byte of room:
        // an empty member still needs some to finish)
        std::unique_ptr<uint8_t[]> tmp(new uint8_t[isize + 1]);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { j->bad = true; return; }
        zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = (uInt)payload;
        zs.next_out = tmp.get(); zs.avail_out = (uInt)isize + 1;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.total_out != isize) j->bad = true;
        inflateEnd(&zs);
        if (!j->bad && isize) memcpy(j->dst, tmp.get(), isize);
    }
    if (!j->bad && crc32_fast(0, j->dst, isize) != j->m.crc) j->bad = true;
}

void ParallelGzipReader::start(std::unique_ptr<Job> j)
{
    Job *raw = j.get();
    const int fd = fd_;
    raw->th = std::thread([raw, fd] { inflate_member(raw, fd); });
    jobs_.push_back(std::move(j));
}

// members ahead of the reader, into buffers of their own: for callers that read in small pieces
void ParallelGzipReader::prefetch()
{
    while (!failed_ && jobs_.size() < depth_) {
        Member m;
        if (!next_member(m)) break;
        std::unique_ptr<Job> j(new Job());
        j->m = m;
        j->own.reset(new uint8_t[(size_t)m.isize + 1]);
        j->dst = j->own.get();
        start(std::move(j));
    }
}

size_t ParallelGzipReader::read_some(void *dst, size_t n)
{
    size_t done = 0;
    if (gz_) {
        char *c = (char *)dst;
        while (done < n) {
            const unsigned chunk = (unsigned)std::min<size_t>(n - done, 1u << 30);
            const int got = gzread(gz_, c + done, chunk);
            if (got < 0) { failed_ = true; break; }
            done += (size_t)got;
            if ((unsigned)got < chunk) break;
        }
        return done;
    }
    if (fd_ < 0) return 0;
    uint8_t *c = (uint8_t *)dst;
    while (done < n && !failed_) {
        if (cur_pos_ < cur_n_) {                                  // what is left of the member being handed out
            const size_t take = std::min(n - done, cur_n_ - cur_pos_);
            memcpy(c + done, cur_.get() + cur_pos_, take);
            cur_pos_ += take; done += take;
            continue;
        }
        if (!jobs_.empty()) {                                     // the next member that went into a buffer of its own
            std::unique_ptr<Job> j = std::move(jobs_.front());
            jobs_.pop_front();
            j->th.join();
            if (j->bad) { failed_ = true; break; }
            cur_ = std::move(j->own);
            cur_n_ = j->m.isize; cur_pos_ = 0;
            // a caller that keeps coming back for pieces gets more members ahead of it each time
            depth_ = std::min(nthreads_ + 1, depth_ * 2);
            prefetch();
            continue;
        }
        Member m;
        if (!next_member(m)) break;                               // end of the stream (or an error: failed_)
        if (m.isize <= n - done) {
            // whole members that fit what is asked for: inflated where they are wanted, nthreads_ at a time
            std::deque<std::unique_ptr<Job>> direct;
            bool more = true;
            while (more) {
                std::unique_ptr<Job> j(new Job());
                j->m = m;
                j->dst = c + done;
                done += m.isize;
                Job *raw = j.get();
                const int fd = fd_;
                raw->th = std::thread([raw, fd] { inflate_member(raw, fd); });
                direct.push_back(std::move(j));
                if (direct.size() >= nthreads_) {
                    direct.front()->th.join();
                    if (direct.front()->bad) failed_ = true;
                    direct.pop_front();
                }
                more = !failed_ && next_member(m);
                if (more && m.isize > n - done) { peek_ = m; have_peek_ = true; more = false; }
            }
            for (auto &j : direct) { j->th.join(); if (j->bad) failed_ = true; }
            depth_ = 1;
            continue;
        }
        peek_ = m; have_peek_ = true;                             // larger than what is asked for: through a buffer of its own
        prefetch();
    }
    return failed_ ? 0 : done;
}

}  // namespace mkhost
