#include "fasta_reader.hpp"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace mkhost {

size_t strip_fasta(const char *text, size_t n, char *dst)
{
    const char *p = text, *const end = text + n;
    char *o = dst;
    while (p <= end) {                                             // a final unterminated line counts
        const char *e = p < end ? (const char *)memchr(p, '\n', (size_t)(end - p)) : nullptr;
        if (!e) e = end;
        if (e == p || *p != '>') {
            memcpy(o, p, (size_t)(e - p));
            o += e - p;
        }
        p = e + 1;
    }
    return (size_t)(o - dst);
}

size_t pack_fasta(const char *text, size_t n, PackAppendFn pack, uint64_t *codes, uint64_t *except, char head[32], bool *dirty)
{
    const char *p = text, *const end = text + n;
    size_t at = 0;
    bool any = false;
    codes[0] = 0; except[0] = 0;                                   // an empty sequence is all-zero words
    memset(head, 0, 32);
    while (p <= end) {                                             // the line rules of strip_fasta
        const char *e = p < end ? (const char *)memchr(p, '\n', (size_t)(end - p)) : nullptr;
        if (!e) e = end;
        if (e > p && *p != '>') {
            const size_t m = (size_t)(e - p);
            if (at < 32) memcpy(head + at, p, std::min<size_t>(m, 32 - at));
            if (pack(codes, except, at, p, m) > 0) any = true;
            at += m;
        }
        p = e + 1;
    }
    *dirty = any;
    return at;
}

static bool gunzip_all(const std::vector<char> &in, std::vector<char> &out)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;
    out.resize(std::max<size_t>(in.size() * 4, 1 << 16));
    zs.next_in = (Bytef *)in.data();
    size_t in_left = in.size(), produced = 0;
    bool ok = true;
    for (;;) {
        if (produced == out.size()) out.resize(out.size() * 2);
        const uInt give = (uInt)std::min<size_t>(in_left, 1u << 30);
        zs.avail_in = give;
        const uInt room = (uInt)std::min<size_t>(out.size() - produced, 1u << 30);
        zs.next_out = (Bytef *)out.data() + produced;
        zs.avail_out = room;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_left -= give - zs.avail_in;
        produced += room - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (in_left == 0) break;
            if (inflateReset(&zs) != Z_OK) { ok = false; break; }  // next member
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { ok = false; break; }
        if (rc == Z_BUF_ERROR && in_left == 0 && zs.avail_out) { ok = false; break; }   // truncated
    }
    inflateEnd(&zs);
    out.resize(produced);
    return ok;
}

bool read_file(const std::string &path, std::vector<char> &out, std::vector<char> &scratch)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    size_t guess = 1 << 20;
    if (fstat(fd, &st) == 0 && st.st_size > 0) guess = (size_t)st.st_size;
    std::vector<char> &raw = scratch;
    raw.resize(guess);
    size_t got = 0;
    for (;;) {
        if (got == raw.size()) raw.resize(raw.size() * 2);
        const ssize_t r = read(fd, raw.data() + got, raw.size() - got);
        if (r < 0) { close(fd); return false; }
        if (r == 0) break;
        got += (size_t)r;
        if (got == guess) {                                        // usually the end: one probe instead of a regrow
            char probe;
            const ssize_t r2 = read(fd, &probe, 1);
            if (r2 <= 0) break;
            raw.resize(raw.size() * 2);
            raw[got++] = probe;
        }
    }
    close(fd);
    raw.resize(got);
    if (got >= 2 && (unsigned char)raw[0] == 0x1f && (unsigned char)raw[1] == 0x8b) return gunzip_all(raw, out);
    out.swap(raw);
    return true;
}

OrderedFastaReader::OrderedFastaReader(std::vector<std::string> files, unsigned threads, HostAllocator a, size_t window,
                                       PackAppendFn pack)
    : files_(std::move(files)), items_(files_.size()), ready_(files_.size()), a_(a), pack_(pack)
{
    for (auto &r : ready_) r.store(0);
    const unsigned n = std::max(1u, std::min<unsigned>(threads, (unsigned)std::max<size_t>(files_.size(), 1)));
    window_ = std::max<size_t>(window, 2 * n + 8);
    for (unsigned t = 0; t < n; ++t) workers_.emplace_back([this] { work(); });
}

OrderedFastaReader::~OrderedFastaReader()
{
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
    for (auto &it : items_) if (it.data) recycle(it);
    for (auto &b : pool_) pool_release(b.first);
}

char *OrderedFastaReader::pool_get(size_t need, size_t &cap)
{
    {
        std::lock_guard<std::mutex> g(pool_m_);
        size_t best = pool_.size();
        for (size_t i = 0; i < pool_.size(); ++i)
            if (pool_[i].second >= need && (best == pool_.size() || pool_[i].second < pool_[best].second)) best = i;
        if (best != pool_.size()) {
            char *p = pool_[best].first;
            cap = pool_[best].second;
            pool_.erase(pool_.begin() + (long)best);
            return p;
        }
        if (!pool_.empty()) {                                      // trade the smallest free buffer for a fitting one
            size_t small = 0;
            for (size_t i = 1; i < pool_.size(); ++i) if (pool_[i].second < pool_[small].second) small = i;
            char *p = pool_[small].first;
            pool_.erase(pool_.begin() + (long)small);
            pool_release(p);
        }
    }
    cap = std::max<size_t>(need + need / 8, 1 << 16);
    char *p = a_.alloc ? (char *)a_.alloc(a_.user, cap) : nullptr;
    if (!p) {                                                      // no allocator, or it is exhausted (page-lock limit):
        p = (char *)malloc(cap);                                   // ordinary memory is slower to copy from, never wrong
        if (!p) { cap = 0; return nullptr; }
        if (a_.alloc) { std::lock_guard<std::mutex> g(pool_m_); plain_.insert(p); }
    }
    return p;
}

void OrderedFastaReader::pool_release(char *p)
{
    auto it = plain_.find(p);
    if (it != plain_.end() || !a_.alloc) {                         // came from malloc
        if (it != plain_.end()) plain_.erase(it);
        free(p);
        return;
    }
    if (a_.release) a_.release(a_.user, p);                        // allocator-owned; without a release hook not ours to free
}

void OrderedFastaReader::recycle(Item &it)
{
    if (!it.data) return;
    std::lock_guard<std::mutex> g(pool_m_);
    pool_.emplace_back(it.data, it.cap);
    it.data = nullptr; it.len = it.cap = 0;
    it.codes = it.except = nullptr;
}

OrderedFastaReader::Item OrderedFastaReader::take(size_t i)
{
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [&] { return ready_[i].load() != 0; });
    Item it = items_[i];
    items_[i] = Item();
    consumed_ = i + 1;
    ahead_bytes_ -= std::min(ahead_bytes_, it.cap);
    lk.unlock();
    cv_.notify_all();
    return it;
}

void OrderedFastaReader::work()
{
    std::vector<char> text, scratch;                               // per thread, reused from file to file
    for (;;) {
        const size_t i = next_.fetch_add(1);
        if (i >= files_.size()) return;
        {
            std::unique_lock<std::mutex> lk(m_);
            // bounded read-ahead, in files and in bytes (the file the consumer waits for always goes)
            cv_.wait(lk, [&] { return stop_ || (i < consumed_ + window_ && (ahead_bytes_ < kAheadBytes || i == consumed_)); });
            if (stop_) return;                                          // destroyed before every item was taken
        }
        Item it;
        struct stat st;
        it.exists = stat(files_[i].c_str(), &st) == 0;
        if (it.exists) {
            text.clear();
            if (!read_file(files_[i], text, scratch)) it.failed = true;
            if (pack_) {
                // no more than text.size() bases: two arrays of that many positions, 8-byte aligned in one buffer
                const size_t cw = packed_code_words(text.size()), xw = packed_except_words(text.size());
                it.data = pool_get((cw + xw) * 8, it.cap);
                if (it.data) {
                    it.packed = true;
                    it.codes = reinterpret_cast<uint64_t *>(it.data);
                    it.except = it.codes + cw;
                    it.len = pack_fasta(text.data(), text.size(), pack_, it.codes, it.except, it.head, &it.dirty);
                } else {
                    it.failed = true;
                }
            } else {
                it.data = pool_get(text.size() + 1, it.cap);
                if (it.data) it.len = strip_fasta(text.data(), text.size(), it.data);
                else it.failed = true;
            }
        }
        { std::lock_guard<std::mutex> g(m_); items_[i] = it; ahead_bytes_ += it.cap; ready_[i].store(1); }
        cv_.notify_all();
    }
}

}  // namespace mkhost
