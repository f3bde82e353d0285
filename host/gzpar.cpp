#include "gzpar.hpp"

#include "fastz.hpp"

#include <algorithm>
#include <cstring>

namespace mkhost {

namespace {

constexpr size_t kHeader = 24;     // fixed gzip header (10) + XLEN (2) + "MK" subfield (4 + 8)

void put_le(uint8_t *p, uint64_t v, int n) { for (int i = 0; i < n; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
uint64_t get_le(const uint8_t *p, int n) { uint64_t v = 0; for (int i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i); return v; }

bool is_mk_header(const uint8_t *h)
{
    return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && h[3] == 4 && get_le(h + 10, 2) == 12 && h[12] == 'M' &&
           h[13] == 'K' && get_le(h + 14, 2) == 8;
}

}  // namespace

// ------------------------------------------------------------------ writer
// Three kinds of threads: the caller's (fills blocks), one deflate thread per block in flight, and ONE output thread that
// takes the finished members in order and writes them -- so that the caller neither deflates nor waits for the file.
ParallelGzipWriter::ParallelGzipWriter(const std::string &path, unsigned threads)
    : f_(fopen(path.c_str(), "wb")), nthreads_(std::max(1u, threads))
{
    if (f_) out_thread_ = std::thread([this] { output_loop(); });
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
        if (!j->ready) j->th.join();
        if (j->bad || fwrite(j->out.data(), 1, j->out.size(), f_) != j->out.size()) failed_ = true;
        if (j->on_done) j->on_done();
        {
            std::lock_guard<std::mutex> g(m_);
            --in_flight_;
        }
        cv_.notify_all();
    }
}

// hand a job to the output thread (in order); waits while as many blocks are in flight as there are threads
void ParallelGzipWriter::enqueue(std::unique_ptr<Job> j)
{
    wrote_any_ = true;
    if (!f_) { failed_ = true; if (j->on_done) j->on_done(); return; }
    Job *raw = j.get();
    if (!raw->ready) raw->th = std::thread(deflate_block, raw);
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
    if (f_ && (!cur_.empty() || !wrote_any_)) submit();          // an empty stream is still one (empty) member
    {
        std::lock_guard<std::mutex> g(m_);
        closing_ = true;
    }
    cv_.notify_all();
    if (out_thread_.joinable()) out_thread_.join();
    if (f_) { if (fclose(f_) != 0) failed_ = true; f_ = nullptr; }
    return !failed_;
}

void ParallelGzipWriter::deflate_block(Job *j)
{
    const uint8_t *in = j->ext ? j->ext : j->in.data();
    const size_t n = j->ext ? j->ext_n : j->in.size();
    uint64_t payload = 0;
    if (j->strategy == Z_HUFFMAN_ONLY && j->level != 0) {
        // literals and Huffman codes only: fastz's coder (one pass for the histogram, one for the bits), not zlib's deflate
        // with its match finder idling (80 -> 500 MB/s per thread on fingerprint columns, the same stream size)
        j->out.resize(kHeader + huffman_only_bound(n) + 8);
        payload = deflate_huffman_only(in, n, j->out.data() + kHeader);
    } else {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, j->level, Z_DEFLATED, -15, 8, j->strategy) != Z_OK) { j->bad = true; return; }   // raw deflate
        const size_t bound = deflateBound(&zs, (uLong)n) + 64;
        j->out.resize(kHeader + bound + 8);
        zs.next_in = const_cast<uint8_t *>(in); zs.avail_in = (uInt)n;
        zs.next_out = j->out.data() + kHeader; zs.avail_out = (uInt)bound;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) j->bad = true;
        payload = zs.total_out;
        deflateEnd(&zs);
    }
    uint8_t *h = j->out.data();
    h[0] = 0x1f; h[1] = 0x8b; h[2] = 8; h[3] = 4;                // FLG.FEXTRA
    put_le(h + 4, 0, 4); h[8] = 4; h[9] = 3;                     // mtime 0, XFL = fastest, OS = unix
    put_le(h + 10, 12, 2); h[12] = 'M'; h[13] = 'K'; put_le(h + 14, 8, 2);
    put_le(h + 16, payload, 8);
    uint8_t *t = h + kHeader + payload;
    put_le(t, crc32_fast(0, in, n), 4);
    put_le(t + 4, n & 0xffffffffu, 4);
    j->out.resize(kHeader + payload + 8);
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
        if (j.bad) failed_ = true; else zero_member_.swap(j.out);
    }
    while (n >= kBlock && !failed_) {                             // whole blocks: the ready-made member, in order
        std::unique_ptr<Job> j(new Job());
        j->out = zero_member_;
        j->ready = true;
        enqueue(std::move(j));
        n -= kBlock;
    }
    while (n) { const size_t take = std::min(n, zeros.size()); write(zeros.data(), take); n -= take; }
}

// ------------------------------------------------------------------ reader
ParallelGzipReader::ParallelGzipReader(const std::string &path, unsigned threads) : nthreads_(std::max(1u, threads))
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { failed_ = true; return; }
    uint8_t h[kHeader];
    const size_t got = fread(h, 1, kHeader, f);
    if (got == kHeader && is_mk_header(h)) {
        fseek(f, 0, SEEK_SET);
        f_ = f;
        prefetch();
    } else {
        fclose(f);
        gz_ = gzopen(path.c_str(), "rb");                        // gzip of any make, or plain bytes
        if (!gz_) failed_ = true; else gzbuffer(gz_, 1 << 20);
    }
}

ParallelGzipReader::~ParallelGzipReader()
{
    for (auto &j : jobs_) if (j->th.joinable()) j->th.join();
    if (f_) fclose(f_);
    if (gz_) gzclose(gz_);
}

bool ParallelGzipReader::read_member_header(uint64_t &payload)
{
    uint8_t h[kHeader];
    const size_t got = fread(h, 1, kHeader, f_);
    if (got == 0) { eof_ = true; return false; }
    if (got != kHeader || !is_mk_header(h)) { failed_ = true; return false; }
    payload = get_le(h + 16, 8);
    return true;
}

void ParallelGzipReader::inflate_block(Job *j)
{
    const size_t payload = j->in.size() - 8;
    const uint32_t crc = (uint32_t)get_le(j->in.data() + payload, 4), isize = (uint32_t)get_le(j->in.data() + payload + 4, 4);
    j->out.resize((size_t)isize + 1);                            // one spare byte: an empty member still needs room to finish
    size_t used = 0, got = 0;
    if (inflate_raw(j->in.data(), payload, j->out.data(), isize, &used, &got) != FZ_OK || got != isize || used != payload) {
        // whatever fastz does not take goes to zlib, which then decides what the member is worth
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { j->bad = true; return; }
        zs.next_in = j->in.data(); zs.avail_in = (uInt)payload;
        zs.next_out = j->out.data(); zs.avail_out = isize + 1;
        const int rc = inflate(&zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs.total_out != isize) j->bad = true;
        inflateEnd(&zs);
    }
    j->out.resize(isize);
    if (!j->bad && crc32_fast(0, j->out.data(), isize) != crc) j->bad = true;
    std::vector<uint8_t>().swap(j->in);
}

void ParallelGzipReader::prefetch()
{
    while (!eof_ && !failed_ && jobs_.size() < nthreads_ + 1) {
        uint64_t payload = 0;
        if (!read_member_header(payload)) break;
        if (payload > (1ull << 31)) { failed_ = true; break; }
        std::unique_ptr<Job> j(new Job());
        j->in.resize((size_t)payload + 8);
        if (fread(j->in.data(), 1, j->in.size(), f_) != j->in.size()) { failed_ = true; break; }
        Job *raw = j.get();
        j->th = std::thread(inflate_block, raw);
        jobs_.push_back(std::move(j));
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
    if (!f_) return 0;
    uint8_t *c = (uint8_t *)dst;
    while (done < n) {
        if (cur_pos_ == cur_.size()) {
            if (jobs_.empty()) prefetch();
            if (jobs_.empty()) break;                              // end of stream
            std::unique_ptr<Job> j = std::move(jobs_.front());
            jobs_.pop_front();
            j->th.join();
            if (j->bad) { failed_ = true; break; }
            cur_.swap(j->out);
            cur_pos_ = 0;
            prefetch();
            continue;
        }
        const size_t take = std::min(n - done, cur_.size() - cur_pos_);
        memcpy(c + done, cur_.data() + cur_pos_, take);
        cur_pos_ += take; done += take;
    }
    return done;
}

}  // namespace mkhost
