// Test helper (no GPU): host/fastz.cpp against zlib.
//   fastz_check            every check below on seeded inputs; prints "ok <n checks>" or the first difference
//   fastz_check quick      the same without the inputs above 300,000 bytes (for the sanitizer build)
//   fastz_check bench      rates of both on fingerprint-like bytes and FASTA-like text
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fastz.hpp"

using namespace mkhost;
typedef std::vector<uint8_t> Bytes;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static Bytes zlib_deflate(const Bytes &in, int level, int strategy, int wbits, int flush_every = 0)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, wbits, 8, strategy);
    Bytes out(deflateBound(&zs, in.size()) + 1024 + (flush_every ? in.size() / flush_every * 16 : 0));
    zs.next_out = out.data(); zs.avail_out = out.size();
    size_t at = 0;
    while (flush_every && at + flush_every < in.size()) {       // full-flush points: stored empty blocks, byte alignment
        zs.next_in = (Bytef *)in.data() + at; zs.avail_in = flush_every;
        deflate(&zs, at / flush_every % 2 ? Z_FULL_FLUSH : Z_SYNC_FLUSH);
        at += flush_every;
    }
    zs.next_in = (Bytef *)in.data() + at; zs.avail_in = in.size() - at;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "zlib deflate failed\n"); exit(2); }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static bool zlib_inflate(const uint8_t *in, size_t n, int wbits, Bytes &out, size_t expect)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, wbits);
    out.assign(expect + 1, 0);
    zs.next_in = (Bytef *)in; zs.avail_in = n; zs.next_out = out.data(); zs.avail_out = out.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == expect;
    inflateEnd(&zs);
    out.resize(expect);
    return ok;
}

static Bytes make_input(int kind, size_t n)
{
    Bytes b(n);
    switch (kind) {
    case 0: for (auto &x : b) x = (uint8_t)rnd(); break;                                   // incompressible
    case 1: for (auto &x : b) { uint64_t r = rnd(); int v = 0; while ((r & 1) && v < 40) { r >>= 1; ++v; } x = (uint8_t)(v * 6 + (r >> 8) % 6); } break;   // skewed, like fingerprints
    case 2: { size_t col = 0; for (auto &x : b) { if (col == 80) { x = '\n'; col = 0; } else { x = "ACGT"[rnd() & 3]; ++col; } } break; }  // FASTA
    case 3: for (size_t i = 0; i < n; ++i) b[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'A');    // long runs, short distances
        break;
    case 4: if (n) memset(b.data(), 0, n); break;                                                // zeros: distance 1, length 258
    case 5: for (size_t i = 0; i < n; ++i) b[i] = (uint8_t)((i * 2654435761u) >> 13 & (i % 1000 < 500 ? 0xff : 0x03)); break;
    case 6: { const char *w[] = {"ACGTTGCA", "GATTACA", "NNNNNNNNNN", ">contig_", "\n"}; size_t i = 0; while (i < n) { const char *s = w[rnd() % 5]; size_t l = strlen(s); for (size_t j = 0; j < l && i < n; ++j) b[i++] = s[j]; } break; }
    default: for (size_t i = 0; i < n; ++i) b[i] = (uint8_t)(rnd() % 3 == 0 ? rnd() : b[i > 300 ? i - 300 : 0]); break;   // distance 300 copies
    }
    return b;
}

static long checks = 0;
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

static void check_inflate(const Bytes &plain, const Bytes &z, const char *what)
{
    // exact room, one byte short, plenty of room; trailing garbage after the stream
    Bytes out(plain.size() + 400);
    size_t used = 0, got = 0;
    Bytes zz = z; zz.insert(zz.end(), {0xAA, 0xBB, 0xCC, 0xDD, 0xEE, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0x00, 0xFF, 0x12, 0x34});
    for (int pass = 0; pass < 3; ++pass) {
        const Bytes &src = pass == 2 ? zz : z;
        const size_t cap = pass == 1 ? plain.size() : plain.size() + 400;
        int rc = inflate_raw(src.data(), src.size(), out.data(), cap, &used, &got);
        if (rc != FZ_OK) FAIL("%s: pass %d rc %d (n %zu)", what, pass, rc, plain.size());
        if (got != plain.size() || (got && memcmp(out.data(), plain.data(), got))) FAIL("%s: pass %d output differs (n %zu got %zu)", what, pass, plain.size(), got);
        if (used != z.size()) FAIL("%s: pass %d used %zu of %zu", what, pass, used, z.size());
        ++checks;
    }
    if (!plain.empty()) {
        int rc = inflate_raw(z.data(), z.size(), out.data(), plain.size() - 1, &used, &got);
        if (rc != FZ_OUT_FULL) FAIL("%s: short output accepted (rc %d)", what, rc);
    }
    // truncated input at a few places: never success
    for (size_t cut : {z.size() - 1, z.size() / 2, (size_t)1, (size_t)0, z.size() * 3 / 4}) {
        if (cut >= z.size()) continue;
        int rc = inflate_raw(z.data(), cut, out.data(), plain.size() + 400, &used, &got);
        if (rc == FZ_OK) FAIL("%s: truncated input (%zu of %zu) accepted", what, cut, z.size());
        ++checks;
    }
    // flipped bits: whatever comes out, no crash and no write past the buffer (the sanitizer build watches)
    Bytes bad = z;
    for (int t = 0; t < 24 && !bad.empty(); ++t) {
        const size_t at = rnd() % bad.size();
        bad[at] ^= (uint8_t)(1u << (rnd() & 7));
        Bytes o2(plain.size() + 64);
        (void)inflate_raw(bad.data(), bad.size(), o2.data(), o2.size(), &used, &got);
        ++checks;
    }
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "bench")) {
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto best = [&](auto &&f) { double b = 1e30; for (int r = 0; r < 3; ++r) { auto t0 = now(); f(); b = std::min(b, std::chrono::duration<double>(now() - t0).count()); } return b; };
        for (int kind : {1, 2}) {
            const Bytes plain = make_input(kind, 64u << 20);
            const double mb = plain.size() / 1e6;
            Bytes zh;
            const double t_zdef = best([&] { zh = kind == 1 ? zlib_deflate(plain, 1, Z_HUFFMAN_ONLY, -15) : zlib_deflate(plain, 6, Z_DEFAULT_STRATEGY, -15); });
            Bytes mine(huffman_only_bound(plain.size()), 1);
            size_t mn = 0;
            const double t_hdef = best([&] { mn = deflate_huffman_only(plain.data(), plain.size(), mine.data()); });
            Bytes out(plain.size() + 400, 1);
            size_t used = 0, got = 0;
            const double t_zinf = best([&] {
                z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
                zs.next_in = zh.data(); zs.avail_in = zh.size(); zs.next_out = out.data(); zs.avail_out = out.size();
                inflate(&zs, Z_FINISH); inflateEnd(&zs);
            });
            const double t_inf = best([&] { inflate_raw(zh.data(), zh.size(), out.data(), out.size(), &used, &got); });
            const double t_inf2 = best([&] { inflate_raw(mine.data(), mn, out.data(), out.size(), &used, &got); });
            uint32_t c1 = 0, c2 = 0;
            const double t_zcrc = best([&] { c1 = crc32(0, plain.data(), plain.size()); });
            const double t_crc = best([&] { c2 = crc32_fast(0, plain.data(), plain.size()); });
            printf("%s: zlib deflate %.0f MB/s (%.3f:1), huffman-only here %.0f MB/s (%.3f:1); inflate of zlib's stream: zlib %.0f MB/s, here %.0f MB/s; "
                   "of the huffman-only stream here %.0f MB/s; crc32 zlib %.0f MB/s, here %.0f MB/s (%s)\n", kind == 1 ? "fingerprint-like" : "FASTA-like",
                   mb / t_zdef, (double)plain.size() / zh.size(), mb / t_hdef, (double)plain.size() / mn, mb / t_zinf, mb / t_inf, mb / t_inf2, mb / t_zcrc,
                   mb / t_crc, c1 == c2 ? "equal" : "DIFFERENT");
        }
        return 0;
    }
    const size_t sizes[] = {0, 1, 2, 3, 15, 16, 17, 255, 256, 257, 258, 259, 300, 319, 320, 321, 1000, 4096, 65535, 65536, 65537, 100000, 262143, 262144, 262145,
                            600001, 3u << 20};
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    for (int kind = 0; kind < 8; ++kind)
        for (size_t n : sizes) {
            if (quick && n > 300000) continue;
            const Bytes plain = make_input(kind, n);
            // crc
            for (uint32_t seed : {0u, 0xdeadbeefu}) {
                if (!n) break;                                          // (zlib answers 0 for a null buffer)
                if (crc32_fast(seed, plain.data(), n) != (uint32_t)crc32(seed, plain.data(), n)) FAIL("crc32 differs: kind %d n %zu", kind, n);
                if (n > 70 && crc32_fast(seed, plain.data() + 3, n - 5) != (uint32_t)crc32(seed, plain.data() + 3, n - 5)) FAIL("crc32 (unaligned) differs: kind %d n %zu", kind, n);
                ++checks;
            }
            // remainders of pieces put together: cut the input in three, fold, compare with zlib's value of the whole
            if (n >= 3) {
                const size_t a = 1 + rnd() % (n - 2), b2 = a + 1 + rnd() % (n - a - 1);
                uint32_t r = crc32_raw(plain.data(), a);
                r = crc32_shift(r, b2 - a) ^ crc32_raw(plain.data() + a, b2 - a);
                r = crc32_shift(r, n - b2) ^ crc32_raw(plain.data() + b2, n - b2);
                if (crc32_from_raw(r, n) != (uint32_t)crc32(0, plain.data(), n)) FAIL("folded CRC differs: kind %d n %zu", kind, n);
                // ... and a plain bytewise remainder (what the GPU computes per block) is crc32_raw
                uint32_t t = 0;
                for (size_t i = 0; i < std::min<size_t>(n, 5000); ++i) { t ^= plain[i]; for (int k = 0; k < 8; ++k) t = (t & 1) ? (t >> 1) ^ 0xEDB88320u : t >> 1; }
                if (t != crc32_raw(plain.data(), std::min<size_t>(n, 5000))) FAIL("bytewise remainder differs: kind %d n %zu", kind, n);
                ++checks;
            }
            // the block index of the writer: every listed position is where a block's first symbol starts
            if (n) {
                HuffIndex hx;
                Bytes z2(huffman_only_bound(n));
                const size_t zn = deflate_huffman_only(plain.data(), n, z2.data(), &hx);
                if (hx.all_coded) {
                    const size_t nsuper = (n + HuffIndex::kSuper - 1) / HuffIndex::kSuper;
                    size_t nsub = 0;
                    for (size_t sp = 0; sp < nsuper; ++sp) nsub += (std::min(HuffIndex::kSuper, n - sp * HuffIndex::kSuper) + HuffIndex::kSub - 1) / HuffIndex::kSub;
                    if (hx.lens.size() != nsuper * 257 || hx.sym_bit.size() != nsub) FAIL("block index sizes: kind %d n %zu", kind, n);
                    for (uint8_t l : hx.lens) if (l > HuffIndex::kMaxLen) FAIL("code longer than %u bits", HuffIndex::kMaxLen);
                    // decode the first few symbols of every block by hand with the listed lengths
                    size_t sub = 0;
                    for (size_t sp = 0; sp < nsuper; ++sp) {
                        const uint8_t *lens = hx.lens.data() + sp * 257;
                        uint32_t cnt[16] = {0}, next[16] = {0}, codes[257];
                        for (int v = 0; v < 257; ++v) ++cnt[lens[v]];
                        cnt[0] = 0;
                        uint32_t c = 0;
                        for (int l = 1; l <= 15; ++l) { c = (c + cnt[l - 1]) << 1; next[l] = c; }
                        for (int v = 0; v < 257; ++v) codes[v] = lens[v] ? next[lens[v]]++ : 0;
                        const size_t sbytes = std::min(HuffIndex::kSuper, n - sp * HuffIndex::kSuper);
                        for (size_t q = 0; q * HuffIndex::kSub < sbytes; ++q, ++sub) {
                            uint64_t bit = hx.sym_bit[sub];
                            const size_t base = sp * HuffIndex::kSuper + q * HuffIndex::kSub, m = std::min<size_t>(8, std::min(HuffIndex::kSub, sbytes - q * HuffIndex::kSub));
                            for (size_t i = 0; i < m; ++i) {
                                uint32_t code = 0, len = 0;
                                int found = -1;
                                while (found < 0 && len < 15) {
                                    if (bit / 8 >= zn) FAIL("block index points past the stream");
                                    code = (code << 1) | ((z2[bit / 8] >> (bit % 8)) & 1u); ++bit; ++len;
                                    for (int v = 0; v < 257; ++v) if (lens[v] == len && codes[v] == code) { found = v; break; }
                                }
                                if (found != plain[base + i]) FAIL("block index: kind %d n %zu block %zu symbol %zu: %d, not %d", kind, n, sub, i, found, plain[base + i]);
                            }
                        }
                    }
                    ++checks;
                }
            }
            // zlib's streams through the reader here
            struct { int level, strategy, flush; } forms[] = {{1, Z_DEFAULT_STRATEGY, 0}, {6, Z_DEFAULT_STRATEGY, 0}, {9, Z_DEFAULT_STRATEGY, 0}, {0, Z_DEFAULT_STRATEGY, 0},
                                                              {1, Z_HUFFMAN_ONLY, 0}, {6, Z_RLE, 0}, {6, Z_FIXED, 0}, {6, Z_FILTERED, 0}, {6, Z_DEFAULT_STRATEGY, 777},
                                                              {1, Z_HUFFMAN_ONLY, 5000}};
            for (auto &f : forms) {
                if (n > (1u << 20) && f.level == 9) continue;
                const Bytes z = zlib_deflate(plain, f.level, f.strategy, -15, f.flush);
                char what[96];
                snprintf(what, sizeof what, "zlib level %d strategy %d flush %d, kind %d", f.level, f.strategy, f.flush, kind);
                check_inflate(plain, z, what);
            }
            // the writer here through zlib's reader, and through the reader here
            Bytes mine(huffman_only_bound(n));
            const size_t mn = deflate_huffman_only(plain.data(), n, mine.data());
            if (mn > mine.size()) FAIL("bound exceeded: kind %d n %zu: %zu > %zu", kind, n, mn, mine.size());
            mine.resize(mn);
            Bytes back;
            if (!zlib_inflate(mine.data(), mn, -15, back, n) || back != plain) FAIL("zlib does not read the huffman-only stream: kind %d n %zu", kind, n);
            check_inflate(plain, mine, "huffman-only stream");
            if (n >= 4096 && kind == 1) {
                const Bytes zh = zlib_deflate(plain, 1, Z_HUFFMAN_ONLY, -15);
                if (mn > zh.size() + zh.size() / 100 + 64) FAIL("huffman-only stream larger than zlib's: %zu vs %zu", mn, zh.size());
            }
            // gzip members: zlib's (with a name and a header CRC) and two in a row
            {
                z_stream zs; memset(&zs, 0, sizeof zs);
                deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
                gz_header gh; memset(&gh, 0, sizeof gh);
                gh.name = (Bytef *)"genome.fa"; gh.hcrc = 1; gh.comment = (Bytef *)"c"; gh.extra = (Bytef *)"MKxx"; gh.extra_len = 4;
                deflateSetHeader(&zs, &gh);
                Bytes gz(deflateBound(&zs, n) + 256);
                zs.next_in = (Bytef *)plain.data(); zs.avail_in = n; zs.next_out = gz.data(); zs.avail_out = gz.size();
                deflate(&zs, Z_FINISH);
                gz.resize(zs.total_out);
                deflateEnd(&zs);
                Bytes two = gz; two.insert(two.end(), gz.begin(), gz.end());
                Bytes out(n + 8);
                size_t used = 0, got = 0;
                if (gunzip_member(two.data(), two.size(), out.data(), out.size(), &used, &got) != FZ_OK || used != gz.size() || got != n || (n && memcmp(out.data(), plain.data(), n)))
                    FAIL("gzip member: kind %d n %zu", kind, n);
                if (gunzip_member(two.data() + used, two.size() - used, out.data(), out.size(), &used, &got) != FZ_OK || used != gz.size() || got != n) FAIL("second gzip member");
                if (n) { two[gz.size() - 6] ^= 1; if (gunzip_member(two.data(), two.size(), out.data(), out.size(), &used, &got) == FZ_OK) FAIL("bad CRC accepted"); }
                ++checks;
            }
        }
    printf("ok %ld\n", checks);
    return 0;
}
