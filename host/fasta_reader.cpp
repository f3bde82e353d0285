#include "fasta_reader.hpp"

#include "fastz.hpp"

#include <fcntl.h>
#include <sched.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
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

namespace {

// 32 characters -> 64 bits of codes (character j at bits 2j) and 32 exception bits (nuc2int / nuc2intrc,
// utils.cpp:31-49, 107-125: anything but A, C, G, T is 0 on both strands)
struct Block { uint64_t codes; uint32_t except; };

inline Block pack32_scalar(const unsigned char *c)
{
    Block b{0, 0};
    for (unsigned j = 0; j < 32; ++j) {
        const unsigned ch = c[j], t = (ch >> 1) & 3u;              // A C G T -> 0 1 3 2
        if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') b.codes |= (uint64_t)(t ^ (t >> 1)) << (2 * j);
        else b.except |= 1u << j;
    }
    return b;
}

__attribute__((target("avx2"))) inline Block pack32_avx2(const unsigned char *c)
{
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(c));
    const __m256i ok = _mm256_or_si256(
        _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('C'))),
        _mm256_or_si256(_mm256_cmpeq_epi8(x, _mm256_set1_epi8('G')), _mm256_cmpeq_epi8(x, _mm256_set1_epi8('T'))));
    const __m256i t = _mm256_and_si256(_mm256_srli_epi16(x, 1), _mm256_set1_epi8(3));
    __m256i v = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), _mm256_set1_epi8(1)));
    v = _mm256_and_si256(v, ok);
    v = _mm256_and_si256(_mm256_or_si256(v, _mm256_srli_epi64(v, 6)), _mm256_set1_epi64x(0x000F000F000F000FLL));
    v = _mm256_and_si256(_mm256_or_si256(v, _mm256_srli_epi64(v, 12)), _mm256_set1_epi64x(0x000000FF000000FFLL));
    v = _mm256_or_si256(v, _mm256_srli_epi64(v, 24));
    Block b;
    b.codes = ((uint64_t)_mm256_extract_epi64(v, 0) & 0xffffu) | (((uint64_t)_mm256_extract_epi64(v, 1) & 0xffffu) << 16) |
              (((uint64_t)_mm256_extract_epi64(v, 2) & 0xffffu) << 32) | (((uint64_t)_mm256_extract_epi64(v, 3) & 0xffffu) << 48);
    b.except = ~(uint32_t)_mm256_movemask_epi8(ok);
    return b;
}

// appends bit strings to an array of 64-bit words, least significant bit first
struct BitWriter {
    uint64_t *w;
    uint64_t acc = 0;
    unsigned fill = 0;                                             // bits waiting in acc, < 64
    explicit BitWriter(uint64_t *dst) : w(dst) {}
    inline void put(uint64_t v, unsigned nbits)                    // the low nbits (<= 64) of v; the rest of v is zero
    {
        acc |= v << fill;
        if (fill + nbits >= 64) {
            *w++ = acc;
            acc = fill ? v >> (64 - fill) : 0;
            fill = fill + nbits - 64;
        } else {
            fill += nbits;
        }
    }
    inline void finish() { *w++ = acc; *w++ = 0; }                 // (the arrays carry two words of slack)
};

template <bool AVX2>
size_t pack_fasta_t(const char *text, size_t n, uint64_t *codes, uint64_t *except, char head[32], bool *dirty)
{
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text), *const end = p + n;
    BitWriter cw(codes), xw(except);
    size_t at = 0;
    uint32_t any = 0;
    memset(head, 0, 32);
    while (p <= end) {                                             // the line rules of strip_fasta (getline, Miekki.cpp:559-567)
        const unsigned char *e = p < end ? (const unsigned char *)memchr(p, '\n', (size_t)(end - p)) : nullptr;
        if (!e) e = end;
        if (e > p && *p != '>') {
            size_t m = (size_t)(e - p);
            if (at < 32) memcpy(head + at, p, std::min<size_t>(m, 32 - at));
            at += m;
            const unsigned char *q = p;
            for (; m >= 32; m -= 32, q += 32) {
                const Block b = AVX2 ? pack32_avx2(q) : pack32_scalar(q);
                cw.put(b.codes, 64);
                xw.put(b.except, 32);
                any |= b.except;
            }
            if (m) {
                // the line's last m < 32 characters: a whole 32-byte load when the text reaches that far (what follows
                // the line is masked away), a padded copy at the very end of the text
                unsigned char tail[32];
                const unsigned char *src = q;
                if (q + 32 > end) { memset(tail, 'A', sizeof tail); memcpy(tail, q, m); src = tail; }
                const Block b = AVX2 ? pack32_avx2(src) : pack32_scalar(src);
                const uint32_t x = b.except & ((1u << m) - 1u);
                cw.put(b.codes & ((1ull << (2 * m)) - 1), 2 * (unsigned)m);
                xw.put(x, (unsigned)m);
                any |= x;
            }
        }
        p = e + 1;
    }
    cw.finish(); xw.finish();
    *dirty = any != 0;
    return at;
}

}  // namespace

size_t pack_fasta(const char *text, size_t n, uint64_t *codes, uint64_t *except, char head[32], bool *dirty)
{
    // MIEKKI_PACK_SCALAR=1 forces the portable form (tests cover both)
    static const bool avx2 = __builtin_cpu_supports("avx2") && !(getenv("MIEKKI_PACK_SCALAR") && atoi(getenv("MIEKKI_PACK_SCALAR")));
    return avx2 ? pack_fasta_t<true>(text, n, codes, except, head, dirty) : pack_fasta_t<false>(text, n, codes, except, head, dirty);
}

// the members of a gzip file one after the other through fastz (whole buffers: the file is mapped, a member's size stands
// in its last four bytes); false = something it does not take -- the caller then lets zlib read the file
static bool gunzip_members(const uint8_t *in, size_t in_size, std::vector<char> &out)
{
    if (in_size < 18) return false;
    uint32_t isize_last;
    memcpy(&isize_last, in + in_size - 4, 4);                     // right for the usual file of ONE member below 4 GiB
    // (the trailer's word is a guess until the member has been inflated: never more room up front than a text that packed
    // 64 : 1 would need -- a corrupt or many-member file must not make every reader thread clear gigabytes)
    out.resize(std::max<size_t>(std::min<size_t>((size_t)isize_last + 64, in_size * 64 + (1 << 16)), 1 << 16));
    size_t at = 0, produced = 0;
    while (at < in_size) {
        size_t used = 0, got = 0;
        int rc = mkhost::gunzip_member(in + at, in_size - at, (uint8_t *)out.data() + produced, out.size() - produced, &used, &got);
        while (rc == mkhost::FZ_OUT_FULL && out.size() < (64ull << 30)) {     // more members than one, or 4 GiB and more: grow, once more
            out.resize(out.size() * 2);
            rc = mkhost::gunzip_member(in + at, in_size - at, (uint8_t *)out.data() + produced, out.size() - produced, &used, &got);
        }
        if (rc != mkhost::FZ_OK) return false;
        at += used; produced += got;
    }
    out.resize(produced);
    return true;
}

static bool gunzip_all(const char *in, size_t in_size, std::vector<char> &out)
{
    if (gunzip_members((const uint8_t *)in, in_size, out)) return true;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return false;
    out.resize(std::max<size_t>(in_size * 4, 1 << 16));
    zs.next_in = (Bytef *)in;
    size_t in_left = in_size, produced = 0;
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

bool inflate_fasta(const char *gz, size_t n, std::vector<char> &seq)
{
    std::vector<char> text;
    if (!gunzip_all(gz, n, text)) return false;
    seq.resize(text.size() + 1);
    seq.resize(strip_fasta(text.data(), text.size(), seq.data()));
    return true;
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
    if (got >= 2 && (unsigned char)raw[0] == 0x1f && (unsigned char)raw[1] == 0x8b) return gunzip_all(raw.data(), raw.size(), out);
    out.swap(raw);
    return true;
}

unsigned usable_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
    double quota = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // cgroup v2: "<quota|max> <period>"
        char q[64];
        double period = 0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) quota = atof(q) / period;
        fclose(f);
    } else {
        double q = 0, period = 0;                                               // cgroup v1
        if (FILE *a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%lf", &q) != 1) q = 0; fclose(a); }
        if (FILE *b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%lf", &period) != 1) period = 0; fclose(b); }
        if (q > 0 && period > 0) quota = q / period;
    }
    if (quota > 0) n = std::max(1u, std::min(n, (unsigned)(quota + 0.5)));
    return n;
}

bool FileText::open(const std::string &path, std::vector<char> &store)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {              // a pipe, a directory, ...: the reading path decides
        close(fd);
        std::vector<char> scratch;
        if (!read_file(path, store, scratch)) return false;
        p = store.data(); n = store.size();
        return true;
    }
    if (st.st_size == 0) { close(fd); p = ""; n = 0; return true; }
    maplen_ = (size_t)st.st_size;
    // (no MAP_POPULATE: it would fault a whole multi-gigabyte file in before the first character is parsed, in every reader
    // thread at once, against the bounded read-ahead window; the kernel's sequential read-ahead is asked for instead.
    // Inputs must not be truncated or rewritten while they are being read: a mapped file that shrinks is a bus error.)
    map_ = mmap(nullptr, maplen_, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map_ != MAP_FAILED) (void)madvise(map_, maplen_, MADV_SEQUENTIAL);
    if (map_ == MAP_FAILED) {
        map_ = nullptr;
        std::vector<char> scratch;
        if (!read_file(path, store, scratch)) return false;
        p = store.data(); n = store.size();
        return true;
    }
    const unsigned char *m = (const unsigned char *)map_;
    if (maplen_ >= 2 && m[0] == 0x1f && m[1] == 0x8b) {             // gzip: inflate every member (zstr.hpp:186-190)
        if (!gunzip_all((const char *)map_, maplen_, store)) return false;
        p = store.data(); n = store.size();
    } else {
        p = (const char *)map_; n = maplen_;
    }
    return true;
}

FileText::~FileText()
{
    if (map_) munmap(map_, maplen_);
}

OrderedFastaReader::OrderedFastaReader(std::vector<std::string> files, unsigned threads, HostAllocator a, size_t window, bool packed,
                                       bool raw_gz, size_t raw_unit, size_t raw_units_ahead, RawSink sink, bool share)
    : files_(std::move(files)), items_(files_.size()), ready_(files_.size()), a_(a), pack_(packed), raw_gz_(raw_gz),
      raw_unit_(raw_gz ? raw_unit : 0), raw_limit_((long)(raw_unit * raw_units_ahead)), umode_(raw_unit_ ? files_.size() / raw_unit_ + 1 : 0),
      ubatch_(raw_unit_ ? files_.size() / raw_unit_ + 1 : 0, nullptr), udone_(raw_unit_ ? files_.size() / raw_unit_ + 1 : 0), sink_(sink), share_(share)
{
    for (auto &d : udone_) d.store(0);
    usize_.resize(udone_.size()); uoff_.resize(udone_.size());
    // a reader takes runs of eight files where there are plenty (a run of a device unit is read into ONE page-locked piece and
    // goes up as one copy: at a copy per file the copies' fixed cost -- a fifth of a millisecond each, whatever the runtime
    // does with them -- was what the upload ran at: 8 GB/s); a run never straddles two units
    if (sink_.open && sink_.put_span && raw_unit_ % 8 == 0 && files_.size() >= (size_t)64 * std::max(1u, threads)) run_ = 8;
    if (!raw_unit_ || !sink_.open || !sink_.put) sink_ = RawSink();     // (a sink works on units)
    for (auto &u : umode_) u.store(0);
    if (raw_gz_) ahead_limit_ = 16ull << 30;                        // (a device batch is thousands of files: their bytes wait here)
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

char *OrderedFastaReader::pool_get(size_t need, size_t &cap, bool plain)
{
    {
        std::lock_guard<std::mutex> g(pool_m_);
        // (page-locked and ordinary buffers share the pool but not their users: a packed sequence in ordinary memory would make
        // its append's copy a staged one)
        auto kind_ok = [&](char *p) { return !a_.alloc || (plain_.count(p) != 0) == plain; };
        size_t best = pool_.size();
        for (size_t i = 0; i < pool_.size(); ++i)
            if (pool_[i].second >= need && kind_ok(pool_[i].first) && (best == pool_.size() || pool_[i].second < pool_[best].second)) best = i;
        if (best != pool_.size()) {
            char *p = pool_[best].first;
            cap = pool_[best].second;
            pool_.erase(pool_.begin() + (long)best);
            return p;
        }
        size_t small = pool_.size();                               // trade the smallest free buffer of the kind for a fitting one
        for (size_t i = 0; i < pool_.size(); ++i)
            if (kind_ok(pool_[i].first) && (small == pool_.size() || pool_[i].second < pool_[small].second)) small = i;
        if (small != pool_.size()) {
            char *p = pool_[small].first;
            pool_.erase(pool_.begin() + (long)small);
            pool_release(p);
        }
    }
    cap = std::max<size_t>(need + need / 8, 1 << 16);
    char *p = a_.alloc && !plain ? (char *)a_.alloc(a_.user, cap) : nullptr;
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
    it.data = nullptr; it.len = it.cap = 0; it.raw = false;
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

// file i of a device unit, piece by piece through the sink's buffers: true when all of it went (the item is raw then, and
// holds no bytes); false: not a gzip member, changed since the unit was opened, unreadable -- the ordinary path takes it
// (what has been put of it is noise in a slot nobody will ask about)
bool OrderedFastaReader::put_raw(size_t i, const stat_view &sv, Item &it)
{
    if (!sv.regular || sv.size < 18 || sv.size >= (1ll << 31)) return false;
    const size_t unit = i / raw_unit_;
    if (!usize_[unit].empty() && usize_[unit][it.unit_index] != (uint64_t)sv.size) return false;    // (changed since the unit was opened)
    const int fd = ::open(files_[i].c_str(), O_RDONLY);
    if (fd < 0) return false;
    std::vector<char> own;                                          // (only when the sink has no buffer to lend)
    uint64_t at = 0;
    const uint64_t size = (uint64_t)sv.size;
    bool ok = true;
    while (ok && at < size) {
        uint64_t cap = 0;
        char *buf = sink_.stage ? (char *)sink_.stage(sink_.user, it.unit_batch, &cap) : nullptr;
        const bool staged = buf != nullptr;
        if (!buf) { own.resize(4u << 20); buf = own.data(); cap = own.size(); }
        const uint64_t want = std::min<uint64_t>(cap, size - at);
        uint64_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd, buf + got, (size_t)(want - got), (off_t)(at + got));
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        if (at == 0 && (got < 2 || (unsigned char)buf[0] != 0x1f || (unsigned char)buf[1] != 0x8b)) ok = false;      // not a gzip member
        if (got != want) ok = false;
        // (a piece that was lent goes back through put even when the file is given up: zero bytes of it)
        if (!sink_.put(sink_.user, it.unit_batch, it.unit_index, at, buf, ok ? got : 0, staged)) ok = false;
        at += got;
    }
    close(fd);
    if (!ok) return false;
    it.raw = true; it.len = (size_t)size; it.data = nullptr; it.cap = 0;
    return true;
}

// A reader's span: files of one unit that follow each other in the list, read into ONE lent piece at the places they have in
// the batch's input (zeros between them), put with one call.  Their items are published when the span has been put, and only
// then do they count as dealt with -- the unit must not run before their bytes are on their way.
struct OrderedFastaReader::RunStage {
    char *buf = nullptr;
    uint64_t cap = 0, used = 0;            // bytes of the piece, bytes laid out so far
    void *batch = nullptr;
    size_t unit = 0;
    uint32_t first = 0, count = 0;         // files [first, first + count) of the unit
    bool last_held = false;                // the file stage_raw took last lies in the span (not put piecewise)
    std::vector<std::pair<size_t, Item>> held;   // (list index, item): published when the span has been put
};

void OrderedFastaReader::publish(size_t i, const Item &it)
{
    { std::lock_guard<std::mutex> g(m_); items_[i] = it; ahead_bytes_ += it.cap; ready_[i].store(1); }
    cv_.notify_all();
}

// k files of the unit have been dealt with: the one that makes the unit whole says so to the sink (before its item is published)
void OrderedFastaReader::unit_file_done(size_t unit, void *batch, uint32_t k)
{
    if (!batch || !sink_.complete || !k) return;
    const size_t u0 = unit * raw_unit_;
    const uint32_t in_unit = (uint32_t)(std::min(files_.size(), u0 + raw_unit_) - u0);
    if (udone_[unit].fetch_add(k) + k == in_unit) sink_.complete(sink_.user, batch);
}

void OrderedFastaReader::flush_stage(RunStage &rs)
{
    if (!rs.buf) return;
    const bool ok = sink_.put_span(sink_.user, rs.batch, rs.first, rs.count, rs.buf, rs.used, true);
    rs.buf = nullptr;
    (void)ok;                                                        // (a copy that could not be queued leaves the files empty on the device, which says
                                                                     //  so per file: the consumer then reads them itself)
    unit_file_done(rs.unit, rs.batch, (uint32_t)rs.held.size());
    for (auto &h : rs.held) publish(h.first, h.second);
    rs.held.clear(); rs.count = 0; rs.used = 0;
}

// file i of a device unit into the reader's span (a new one when it does not follow the span's last file or does not fit);
// true: the item is raw and HELD in rs (published by flush_stage).  A file larger than a piece goes the piecewise way.
bool OrderedFastaReader::stage_raw(size_t i, size_t unit, const stat_view &sv, Item &it, RunStage &rs)
{
    rs.last_held = false;
    if (!sv.regular || sv.size < 18 || sv.size >= (1ll << 31)) return false;
    const std::vector<uint64_t> &off = uoff_[unit];
    const uint32_t j = it.unit_index;
    if (usize_[unit][j] != (uint64_t)sv.size) return false;         // (changed since the unit was opened)
    const uint64_t size = (uint64_t)sv.size, room = off[j + 1] - off[j];
    if (rs.buf && (rs.batch != it.unit_batch || rs.first + rs.count != j || off[j + 1] - off[rs.first] > rs.cap)) flush_stage(rs);
    if (!rs.buf) {
        uint64_t cap = 0;
        char *buf = sink_.stage ? (char *)sink_.stage(sink_.user, it.unit_batch, &cap) : nullptr;
        if (!buf) return put_raw(i, sv, it);                        // (none to be had)
        if (room > cap) { (void)sink_.put(sink_.user, it.unit_batch, j, 0, buf, 0, true); return put_raw(i, sv, it); }   // (a file larger than a piece)
        rs.buf = buf; rs.cap = cap; rs.used = 0; rs.batch = it.unit_batch; rs.unit = unit; rs.first = j; rs.count = 0;
    }
    char *dst = rs.buf + (off[j] - off[rs.first]);
    bool ok = false;
    const int fd = ::open(files_[i].c_str(), O_RDONLY);
    if (fd >= 0) {
        uint64_t got = 0;
        while (got < size) {
            const ssize_t r = pread(fd, dst + got, (size_t)(size - got), (off_t)got);
            if (r <= 0) break;
            got += (uint64_t)r;
        }
        close(fd);
        ok = got == size && (unsigned char)dst[0] == 0x1f && (unsigned char)dst[1] == 0x8b;
    }
    // (a file that is given up leaves zeros at its place: the device sees nothing there, and the item goes the ordinary way)
    if (!ok) memset(dst, 0, (size_t)room); else memset(dst + size, 0, (size_t)(room - size));
    rs.count = j + 1 - rs.first;
    rs.used = off[j + 1] - off[rs.first];
    if (!ok) return false;
    it.raw = true; it.len = (size_t)size; it.data = nullptr; it.cap = 0;
    rs.last_held = true;
    return true;
}

void OrderedFastaReader::work()
{
    std::vector<char> text, scratch;                               // per thread, reused from file to file
    RunStage rs;
    for (;;) {
        const size_t base = next_.fetch_add(run_);
        if (base >= files_.size()) return;
        for (size_t i = base; i < std::min(files_.size(), base + run_); ++i) {
        {
            std::unique_lock<std::mutex> lk(m_);
            // bounded read-ahead, in files and in bytes (the file the consumer waits for always goes)
            cv_.wait(lk, [&] { return stop_ || (i < consumed_ + window_ && (ahead_bytes_ < ahead_limit_ || i == consumed_)); });
            if (stop_) { lk.unlock(); flush_stage(rs); return; }        // destroyed before every item was taken
        }
        bool staged_now = false;
        Item it;
        struct stat st;
        it.exists = stat(files_[i].c_str(), &st) == 0;
        bool raw_done = false;
        // whose unit is this file's: the device's while it has room for one more (the first reader to touch a unit says)
        bool to_device = raw_gz_;
        const size_t unit = raw_unit_ ? i / raw_unit_ : 0;
        if (raw_unit_) {
            std::atomic<int> &mode = umode_[unit];
            int m = mode.load();
            if (!m) {
                const size_t u0 = unit * raw_unit_;
                const long in_unit = (long)(std::min(files_.size(), u0 + raw_unit_) - u0);
                int want = !share_ || raw_out_.load() < raw_limit_ ? 1 : 2;
                if (sink_.open) {
                    // with a sink the unit has to be OPENED (its files' sizes fix the batch's layout): by the first reader here,
                    // the others wait for it
                    if (mode.compare_exchange_strong(m, 3)) {
                        void *batch = nullptr;
                        if (want == 1) {
                            std::vector<uint64_t> sizes((size_t)in_unit, 0);
                            bool gz_first = false;
                            for (long j = 0; j < in_unit; ++j) {
                                struct stat sj;
                                if (stat(files_[u0 + (size_t)j].c_str(), &sj) == 0 && S_ISREG(sj.st_mode) && sj.st_size >= 18 && sj.st_size < (1ll << 31))
                                    sizes[(size_t)j] = (uint64_t)sj.st_size;
                            }
                            // (a unit whose first readable file is no gzip member is a list of plain files: the readers' own)
                            for (long j = 0; j < in_unit && !gz_first; ++j)
                                if (sizes[(size_t)j]) {
                                    const int fd = ::open(files_[u0 + (size_t)j].c_str(), O_RDONLY);
                                    unsigned char magic[2] = {0, 0};
                                    if (fd >= 0) { gz_first = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b; close(fd); }
                                    break;
                                }
                            std::vector<uint64_t> offs((size_t)in_unit + 1, 0);
                            if (gz_first) batch = sink_.open(sink_.user, sizes.data(), (uint32_t)in_unit, offs.data());
                            if (batch) { usize_[unit] = sizes; uoff_[unit] = offs; }
                        }
                        m = batch ? 1 : 2;
                        if (batch) raw_out_.fetch_add(in_unit);
                        { std::lock_guard<std::mutex> g(m_); ubatch_[unit] = batch; mode.store(m); }
                        cv_.notify_all();
                    }
                } else if (mode.compare_exchange_strong(m, want)) { m = want; if (want == 1) raw_out_.fetch_add(in_unit); }
            }
            if (m == 3) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || mode.load() != 3; });
                if (stop_) return;
                m = mode.load();
            }
            to_device = m == 1;
        }
        if (sink_.open && to_device) {
            it.unit_batch = ubatch_[unit];
            it.unit_index = (uint32_t)(i - unit * raw_unit_);
            it.unit_last = i + 1 == files_.size() || (i + 1) % raw_unit_ == 0;
            if (it.exists) {
                const stat_view sv{S_ISREG(st.st_mode) != 0, (long long)st.st_size};
                raw_done = run_ > 1 ? stage_raw(i, unit, sv, it, rs) : put_raw(i, sv, it);
                staged_now = raw_done && run_ > 1 && rs.last_held;
            }
        } else if (it.exists && to_device && S_ISREG(st.st_mode) && st.st_size >= 18 && st.st_size < (1ll << 31)) {
            // a gzip'd file as it is, for the device's inflater: read into a pooled buffer of ORDINARY memory (a gigabyte and
            // more of these are alive at a time: page-locking that much -- 50 ms per 256 MB, every reader waiting behind the
            // call -- cost more than the staged upload does on the batch's own thread).  Not mapped: thousands of mappings
            // made and taken down while sixteen threads fault pages in spent more time on the address space's lock than
            // on the files.
            const int fd = ::open(files_[i].c_str(), O_RDONLY);
            if (fd >= 0) {
                unsigned char magic[2] = {0, 0};
                if (pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
                    it.data = pool_get((size_t)st.st_size + 16, it.cap, true);
                    size_t got = 0;
                    while (it.data && got < (size_t)st.st_size) {
                        const ssize_t r = pread(fd, it.data + got, (size_t)st.st_size - got, (off_t)got);
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    if (it.data && got == (size_t)st.st_size) {
                        it.len = got; it.raw = true;
                        raw_done = true;
                    } else if (it.data) {                          // short read: the ordinary path says what is wrong with the file
                        std::lock_guard<std::mutex> g(pool_m_);
                        pool_.emplace_back(it.data, it.cap);
                        it.data = nullptr; it.cap = 0;
                    }
                }
                close(fd);
            }
        }
        if (it.exists && !raw_done) {
            FileText ft;                                               // mapped, or inflated into `text`
            if (!ft.open(files_[i], text)) {
                it.failed = true;
            } else if (pack_) {
                // no more than ft.n bases: two arrays of that many positions, 8-byte aligned in one buffer
                const size_t cw = packed_code_words(ft.n), xw = packed_except_words(ft.n);
                it.data = pool_get((cw + xw) * 8, it.cap);
                if (it.data) {
                    it.packed = true;
                    it.codes = reinterpret_cast<uint64_t *>(it.data);
                    it.except = it.codes + cw;
                    it.len = pack_fasta(ft.p, ft.n, it.codes, it.except, it.head, &it.dirty);
                } else {
                    it.failed = true;
                }
            } else {
                it.data = pool_get(ft.n + 1, it.cap);
                if (it.data) it.len = strip_fasta(ft.p, ft.n, it.data);
                else it.failed = true;
            }
        }
        if (raw_unit_ && to_device && !raw_done && !sink_.open) raw_out_.fetch_sub(1);     // (a file of a device unit that does not go there; with a sink the
                                                                                            // consumer counts every file of the unit back)
        if (staged_now) { rs.held.emplace_back(i, it); continue; }      // (published, and counted, when its span has been put)
        unit_file_done(unit, it.unit_batch, it.unit_batch ? 1u : 0u);
        publish(i, it);
        }
        flush_stage(rs);
    }
}

}  // namespace mkhost
