#include "fastz.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mkhost {

// ------------------------------------------------------------------------------------------------ CRC-32
// The CRC of a long buffer by folding: four 128-bit lanes, each multiplied (carry-less) by x^(512+-32) mod P and added to
// the next 64 bytes; then four lanes -> one, 128 -> 64 -> 32 bits by Barrett reduction (Gopal et al., "Fast CRC
// computation for generic polynomials using PCLMULQDQ", the constants of the bit-reflected polynomial 0xEDB88320).
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static inline __m128i crc_fold16(__m128i acc, __m128i next, __m128i k)
{
    const __m128i lo = _mm_clmulepi64_si128(acc, k, 0x00);
    acc = _mm_clmulepi64_si128(acc, k, 0x11);
    return _mm_xor_si128(_mm_xor_si128(acc, lo), next);
}

__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(uint32_t state, const uint8_t *p, size_t n)   // n >= 64, n % 16 == 0
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    const __m128i *q = reinterpret_cast<const __m128i *>(p);
    __m128i x1 = _mm_loadu_si128(q), x2 = _mm_loadu_si128(q + 1), x3 = _mm_loadu_si128(q + 2), x4 = _mm_loadu_si128(q + 3);
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    q += 4; n -= 64;
    while (n >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128(q));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128(q + 1));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128(q + 2));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128(q + 3));
        q += 4; n -= 64;
    }
    x1 = crc_fold16(x1, x2, k3k4); x1 = crc_fold16(x1, x3, k3k4); x1 = crc_fold16(x1, x4, k3k4);
    while (n >= 16) { x1 = crc_fold16(x1, _mm_loadu_si128(q), k3k4); ++q; n -= 16; }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), t);
    // Barrett: 64 -> 32 bits
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t crc32_fast(uint32_t crc, const void *p, size_t n)
{
    const uint8_t *c = static_cast<const uint8_t *>(p);
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 256) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_fold(~crc, c, body);
        c += body; n -= body;
    }
#endif
    while (n) {                                                   // (zlib takes a 32-bit length)
        const uInt take = (uInt)std::min<size_t>(n, 1u << 30);
        crc = (uint32_t)::crc32(crc, c, take);
        c += take; n -= take;
    }
    return crc;
}

// Polynomials over GF(2) modulo the CRC's, bit-reflected (x^0 is the top bit): a * b, and x^n by squaring.
static uint32_t gf2_mul(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if (!(a & (m - 1))) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

static uint32_t gf2_x_pow(uint64_t n)                                  // x^n
{
    static const struct Pow { uint32_t v[64]; Pow() { v[0] = 1u << 30; for (int k = 1; k < 64; ++k) v[k] = gf2_mul(v[k - 1], v[k - 1]); } } pw;   // x^(2^k)
    uint32_t r = 1u << 31;
    for (int k = 0; n; ++k, n >>= 1) if (n & 1) r = gf2_mul(r, pw.v[k]);
    return r;
}

uint32_t crc32_shift(uint32_t raw, uint64_t n_bytes) { return raw ? gf2_mul(gf2_x_pow(n_bytes * 8), raw) : 0; }
uint32_t crc32_shift_factor(uint64_t n_bytes) { return gf2_x_pow(n_bytes * 8); }
uint32_t crc32_shift_by(uint32_t factor, uint32_t raw) { return raw ? gf2_mul(factor, raw) : 0; }
uint32_t crc32_from_raw(uint32_t raw, uint64_t n_bytes) { return ~(raw ^ crc32_shift(0xffffffffu, n_bytes)); }
uint32_t crc32_raw(const void *p, size_t n)
{
    // zlib's value with its start value and complement taken out again
    return ~crc32_fast(0, p, n) ^ crc32_shift(0xffffffffu, n);
}

// ------------------------------------------------------------------------------------------------ inflate
namespace {

constexpr int kLitBits = 11, kDistBits = 8, kPreBits = 7;
constexpr uint32_t kLitFlag = 0x1000, kEobFlag = 0x2000, kSubFlag = 0x4000;
// an entry: bits to drop | extra bits << 8 | flags | value << 16 (a literal, a base length or distance, a subtable's start);
// zero = no code ends here (an invalid stream).  Subtable entries hold the WHOLE code length.
constexpr size_t kLitSize = (1u << kLitBits) + 288 * 16, kDistSize = (1u << kDistBits) + 32 * 128;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum Kind { kLitLen, kDist, kPre };

inline uint32_t entry_of(Kind kind, uint32_t sym, uint32_t len)
{
    switch (kind) {
    case kLitLen:
        if (sym < 256) return len | kLitFlag | sym << 16;
        if (sym == 256) return len | kEobFlag;
        if (sym > 285) return 0;
        return len | (uint32_t)kLenExtra[sym - 257] << 8 | (uint32_t)kLenBase[sym - 257] << 16;
    case kDist:
        if (sym > 29) return 0;
        return len | (uint32_t)kDistExtra[sym] << 8 | (uint32_t)kDistBase[sym] << 16;
    default:
        return len | sym << 16;
    }
}

inline uint32_t reverse_bits(uint32_t code, uint32_t len)
{
    code = ((code & 0x5555) << 1) | ((code >> 1) & 0x5555);
    code = ((code & 0x3333) << 2) | ((code >> 2) & 0x3333);
    code = ((code & 0x0f0f) << 4) | ((code >> 4) & 0x0f0f);
    code = ((code & 0x00ff) << 8) | ((code >> 8) & 0x00ff);
    return code >> (16 - len);
}

// Decoding table of a canonical code from its lengths.  False: over-subscribed, or incomplete in a way deflate does not
// allow (allowed: no distance code at all; one code of one bit).
bool build_table(uint32_t *tab, int tb, const uint8_t *lens, uint32_t n, Kind kind)
{
    uint32_t count[16] = {0}, next[16];
    for (uint32_t s = 0; s < n; ++s) ++count[lens[s]];
    const uint32_t used = n - count[0];
    count[0] = 0;
    int64_t left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = left * 2 - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !(kind == kDist && used == 0) && !(kind != kPre && used == 1 && count[1] == 1)) return false;
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    const uint32_t size = 1u << tb, mask = size - 1;
    memset(tab, 0, size * sizeof(uint32_t));
    bool any_long = false;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens[s];
        if (!l) continue;
        if (l > (uint32_t)tb) { any_long = true; continue; }
        const uint32_t r = reverse_bits(next[l]++, l), e = entry_of(kind, s, l);
        for (uint32_t i = r; i < size; i += 1u << l) tab[i] = e;
    }
    if (!any_long) return true;
    // codes longer than the table's index: the longest code behind each index decides its subtable's size
    uint8_t deepest[1u << kLitBits];
    memset(deepest, 0, size);
    uint32_t nx[16];
    memcpy(nx, next, sizeof nx);
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens[s];
        if (l <= (uint32_t)tb) continue;
        const uint32_t r = reverse_bits(nx[l]++, l);
        deepest[r & mask] = std::max<uint8_t>(deepest[r & mask], (uint8_t)l);
    }
    uint32_t at = size;
    for (uint32_t s = 0; s < n; ++s) {
        const uint32_t l = lens[s];
        if (l <= (uint32_t)tb) continue;
        const uint32_t r = reverse_bits(next[l]++, l), head = r & mask, sub_bits = deepest[head] - tb;
        if (!(tab[head] & kSubFlag)) {
            if (tab[head]) return false;                          // (cannot happen with a prefix code; guards the table)
            tab[head] = (uint32_t)tb | sub_bits << 8 | kSubFlag | at << 16;
            memset(tab + at, 0, sizeof(uint32_t) << sub_bits);
            at += 1u << sub_bits;
        }
        uint32_t *sub = tab + (tab[head] >> 16);
        const uint32_t e = entry_of(kind, s, l);
        for (uint32_t i = r >> tb; i < (1u << sub_bits); i += 1u << (l - tb)) sub[i] = e;
    }
    return true;
}

// Literals up to three at a time: index = the next kPairBits bits; an entry holds the one to three literals whose codes
// those bits spell out completely (bytes in bits 0-23, bits to drop in 24-27, how many in 28-29), or zero when the first
// symbol is anything else (a length, the end of the block, a code longer than the index).  Fingerprint columns are
// nothing but literals of 4 - 9 bits, FASTA text mostly literals of 2 - 3: the chain  load -> shift -> mask -> load  that
// decides a Huffman decoder's pace is walked once per two or three of them.
constexpr int kPairBits = 12;

struct Tables {
    uint32_t lit[kLitSize];
    uint32_t dist[kDistSize];
    uint32_t pair[1u << kPairBits];
};

void build_pairs(Tables &t)
{
    constexpr uint32_t lmask = (1u << kLitBits) - 1;
    for (uint32_t i = 0; i < (1u << kPairBits); ++i) {
        uint32_t v = 0, bits = 0, cnt = 0;
        while (cnt < 3) {
            const uint32_t e = t.lit[(i >> bits) & lmask], l = e & 0xff;
            if (!(e & kLitFlag) || (e & kSubFlag) || bits + l > (uint32_t)kPairBits) break;   // (a primary entry: at most kLitBits bits)
            v |= (e >> 16) << (8 * cnt);
            bits += l; ++cnt;
        }
        t.pair[i] = cnt ? v | bits << 24 | cnt << 28 : 0;
    }
}

const Tables &fixed_tables()
{
    static const Tables *t = [] {
        Tables *f = new Tables;
        uint8_t lens[288 + 32];
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        build_table(f->lit, kLitBits, lens, 288, kLitLen);
        build_table(f->dist, kDistBits, lens + 288, 32, kDist);
        build_pairs(*f);
        return f;
    }();
    return *t;
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct Stream {
    const uint8_t *in, *in_end;
    uint64_t bb = 0;             // bits not yet consumed, the next one lowest
    uint32_t bc = 0;             // how many of them are accounted for (bits above may be set after a fast refill: they are the stream's own)
    uint32_t pad = 0;            // zero bytes appended past the end of the input (consuming their bits is an error, checked at block ends)
    // byte by byte, for headers and the last stretch of the input
    inline void fill()
    {
        bb &= bc >= 64 ? ~0ull : ((1ull << bc) - 1);
        while (bc < 56) {
            if (in < in_end) bb |= (uint64_t)*in++ << bc; else ++pad;
            bc += 8;
        }
    }
    inline uint32_t take(uint32_t n) { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; }
    inline bool overran() const { return pad * 8 > bc; }
};

}  // namespace

int inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out0, size_t out_cap, size_t *in_used, size_t *out_len)
{
    Stream s;
    s.in = in; s.in_end = in + in_len;
    uint8_t *out = out0, *const out_end = out0 + out_cap;
    Tables dyn;
    uint8_t dyn_lens[320];
    uint32_t dyn_hlit = 0, dyn_hdist = 0;
    bool have_dyn = false;
    bool final_block = false;
    while (!final_block) {
        s.fill();
        final_block = s.take(1) != 0;
        const uint32_t type = s.take(2);
        const uint32_t *lit, *dist, *pair;
        if (type == 0) {                                          // stored: LEN, ~LEN, bytes, from the next byte boundary
            s.take(s.bc & 7);
            if (s.overran()) return FZ_IN_SHORT;
            s.in -= (s.bc >> 3) - s.pad;                          // bytes the bit buffer held are read again, as bytes
            s.bb = 0; s.bc = 0; s.pad = 0;
            if (s.in_end - s.in < 4) return FZ_IN_SHORT;
            const uint32_t len = s.in[0] | s.in[1] << 8, nlen = s.in[2] | s.in[3] << 8;
            if ((len ^ nlen) != 0xffff) return FZ_BAD;
            s.in += 4;
            if ((size_t)(s.in_end - s.in) < len) return FZ_IN_SHORT;
            if ((size_t)(out_end - out) < len) return FZ_OUT_FULL;
            memcpy(out, s.in, len);
            out += len; s.in += len;
            continue;
        }
        if (type == 3) return FZ_BAD;
        if (type == 1) {
            lit = fixed_tables().lit; dist = fixed_tables().dist; pair = fixed_tables().pair;
        } else {
            const uint32_t hlit = s.take(5) + 257, hdist = s.take(5) + 1, hclen = s.take(4) + 4;
            if (hlit > 286 || hdist > 30) return FZ_BAD;
            uint8_t lens[320];
            memset(lens, 0, 19);
            for (uint32_t i = 0; i < hclen; ++i) { s.fill(); lens[kPreOrder[i]] = (uint8_t)s.take(3); }
            uint32_t pre[1u << kPreBits];
            if (!build_table(pre, kPreBits, lens, 19, kPre)) return FZ_BAD;
            uint32_t got = 0;
            while (got < hlit + hdist) {
                s.fill();
                const uint32_t e = pre[s.bb & ((1u << kPreBits) - 1)];
                if (!e) return s.overran() ? FZ_IN_SHORT : FZ_BAD;
                s.take(e & 0xff);
                const uint32_t sym = e >> 16;
                if (sym < 16) { lens[got++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) { if (!got) return FZ_BAD; val = lens[got - 1]; rep = 3 + s.take(2); }
                else if (sym == 17) rep = 3 + s.take(3);
                else rep = 11 + s.take(7);
                if (got + rep > hlit + hdist) return FZ_BAD;
                memset(lens + got, (int)val, rep);
                got += rep;
            }
            if (s.overran()) return FZ_IN_SHORT;
            if (!lens[256]) return FZ_BAD;                        // a block must be able to end
            // (the same code as the block before -- deflate_huffman_only repeats one code over 64 short blocks: its tables stand)
            if (!(have_dyn && hlit == dyn_hlit && hdist == dyn_hdist && !memcmp(lens, dyn_lens, hlit + hdist))) {
                uint8_t dl[32];
                memcpy(dl, lens + hlit, hdist);
                have_dyn = false;
                if (!build_table(dyn.lit, kLitBits, lens, hlit, kLitLen) || !build_table(dyn.dist, kDistBits, dl, hdist, kDist)) return FZ_BAD;
                build_pairs(dyn);
                memcpy(dyn_lens, lens, hlit + hdist);
                dyn_hlit = hlit; dyn_hdist = hdist; have_dyn = true;
            }
            lit = dyn.lit; dist = dyn.dist; pair = dyn.pair;
        }
        // ---- the block's symbols
        constexpr uint32_t lmask = (1u << kLitBits) - 1, dmask = (1u << kDistBits) - 1, pmask = (1u << kPairBits) - 1;
        bool block_done = false;
        // fast stretch: at least 16 bytes of input ahead and room for the longest match plus the copy's overshoot
        while (!block_done && s.in_end - s.in >= 16 && out_end - out >= 320) {
#define MK_REFILL() do { s.bb |= load64(s.in) << s.bc; s.in += (63 - s.bc) >> 3; s.bc |= 56; } while (0)
#define MK_LIT_LOOKUP(e) do { e = lit[s.bb & lmask]; if (e & kSubFlag) e = lit[(e >> 16) + ((s.bb >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))]; } while (0)
            MK_REFILL();
            // up to four lookups of twelve bits on one refill; four bytes are stored whatever the count (there is room)
            uint32_t pr = pair[s.bb & pmask];
            if (pr) {
                int turns = 3;
                do {
                    memcpy(out, &pr, 4);
                    out += pr >> 28; s.bb >>= (pr >> 24) & 15; s.bc -= (pr >> 24) & 15;
                    pr = pair[s.bb & pmask];
                } while (pr && --turns);
                if (pr) {
                    memcpy(out, &pr, 4);
                    out += pr >> 28; s.bb >>= (pr >> 24) & 15; s.bc -= (pr >> 24) & 15;
                    continue;
                }
                MK_REFILL();
            }
            uint32_t e;
            MK_LIT_LOOKUP(e);
            if (e & kLitFlag) {                                   // (a literal the pairs do not hold: a code of more than twelve bits)
                *out++ = (uint8_t)(e >> 16); s.bb >>= (e & 0xff); s.bc -= (e & 0xff);
                continue;
            }
            if (e & kEobFlag) { s.bb >>= (e & 0xff); s.bc -= (e & 0xff); block_done = true; break; }
            if (!e) return FZ_BAD;
            s.bb >>= (e & 0xff); s.bc -= (e & 0xff);
            uint32_t xb = (e >> 8) & 15;
            const uint32_t len = (e >> 16) + (uint32_t)(s.bb & ((1u << xb) - 1));
            s.bb >>= xb; s.bc -= xb;
            uint32_t d = dist[s.bb & dmask];
            if (d & kSubFlag) d = dist[(d >> 16) + ((s.bb >> kDistBits) & ((1u << ((d >> 8) & 15)) - 1))];
            if (!d) return FZ_BAD;
            s.bb >>= (d & 0xff); s.bc -= (d & 0xff);
            xb = (d >> 8) & 15;
            const uint32_t back = (d >> 16) + (uint32_t)(s.bb & ((1u << xb) - 1));
            s.bb >>= xb; s.bc -= xb;
            if (back > (size_t)(out - out0)) return FZ_BAD;
            const uint8_t *from = out - back;
            uint8_t *to = out;
            out += len;
            if (back >= 16) {
                do { memcpy(to, from, 16); to += 16; from += 16; } while (to < out);
            } else if (back == 1) {
                memset(to, *from, len);
            } else if (back >= 8) {
                do { memcpy(to, from, 8); to += 8; from += 8; } while (to < out);
            } else {
                do { *to++ = *from++; } while (to < out);
            }
#undef MK_LIT_LOOKUP
#undef MK_REFILL
        }
        // careful stretch: every byte and every bit checked
        while (!block_done) {
            s.fill();
            uint32_t e = lit[s.bb & lmask];
            if (e & kSubFlag) e = lit[(e >> 16) + ((s.bb >> kLitBits) & ((1u << ((e >> 8) & 15)) - 1))];
            if (!e) return s.overran() ? FZ_IN_SHORT : FZ_BAD;
            s.take(e & 0xff);
            if (e & kLitFlag) {
                if (out == out_end) return FZ_OUT_FULL;
                *out++ = (uint8_t)(e >> 16);
                continue;
            }
            if (e & kEobFlag) { block_done = true; break; }
            const uint32_t len = (e >> 16) + s.take((e >> 8) & 15);
            s.fill();
            uint32_t d = dist[s.bb & dmask];
            if (d & kSubFlag) d = dist[(d >> 16) + ((s.bb >> kDistBits) & ((1u << ((d >> 8) & 15)) - 1))];
            if (!d) return s.overran() ? FZ_IN_SHORT : FZ_BAD;
            s.take(d & 0xff);
            const uint32_t back = (d >> 16) + s.take((d >> 8) & 15);
            if (s.overran()) return FZ_IN_SHORT;
            if (back > (size_t)(out - out0)) return FZ_BAD;
            if ((size_t)(out_end - out) < len) return FZ_OUT_FULL;
            const uint8_t *from = out - back;
            for (uint32_t i = 0; i < len; ++i) out[i] = from[i];
            out += len;
        }
        if (s.overran()) return FZ_IN_SHORT;
    }
    // whole bytes still in the bit buffer were not part of the stream
    s.bb &= s.bc >= 64 ? ~0ull : ((1ull << s.bc) - 1);
    *in_used = (size_t)(s.in - in) - ((s.bc >> 3) - s.pad);
    *out_len = (size_t)(out - out0);
    return FZ_OK;
}

int gunzip_member(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used, size_t *out_len)
{
    if (in_len < 18) return FZ_IN_SHORT;
    if (in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || (in[3] & 0xe0)) return FZ_BAD;
    const uint8_t flags = in[3];
    size_t at = 10;
    if (flags & 4) {                                              // FEXTRA
        if (at + 2 > in_len) return FZ_IN_SHORT;
        at += 2 + (in[at] | (size_t)in[at + 1] << 8);
    }
    for (int bit : {8, 16})                                       // FNAME, FCOMMENT: zero-terminated
        if (flags & bit) {
            while (at < in_len && in[at]) ++at;
            ++at;
        }
    if (flags & 2) at += 2;                                       // FHCRC
    if (at >= in_len) return FZ_IN_SHORT;
    size_t used = 0, n = 0;
    const int rc = inflate_raw(in + at, in_len - at, out, out_cap, &used, &n);
    if (rc != FZ_OK) return rc;
    at += used;
    if (at + 8 > in_len) return FZ_IN_SHORT;
    uint32_t crc, isize;
    memcpy(&crc, in + at, 4); memcpy(&isize, in + at + 4, 4);     // (little-endian hosts only, like the rest of this program)
    if (isize != (uint32_t)n || crc != crc32_fast(0, out, n)) return FZ_BAD;
    *in_used = at + 8;
    *out_len = n;
    return FZ_OK;
}

// ------------------------------------------------------------------------------------------------ Huffman-only deflate
namespace {

constexpr size_t kHuffSuper = HuffIndex::kSuper, kHuffSub = HuffIndex::kSub;   // bytes per Huffman code, bytes per block

// Code lengths of at most `limit` bits for the symbols with freq > 0 (at least two of them), optimal when the plain
// Huffman code is not deeper than the limit and made to fit otherwise: the deepest leaves move up to the limit and, until
// the code is a prefix code again, one code of the limit's length pairs up with the deepest shorter one.
void code_lengths(const uint32_t *freq, uint32_t n, uint32_t limit, uint8_t *lens)
{
    struct Node { uint64_t w; int32_t left, right; };
    uint32_t order[288];
    uint32_t m = 0;
    for (uint32_t s = 0; s < n; ++s) { lens[s] = 0; if (freq[s]) order[m++] = s; }
    if (m < 2) { if (m) lens[order[0]] = 1; return; }           // (callers bring two symbols at least)
    std::sort(order, order + m, [&](uint32_t a, uint32_t b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    Node node[2 * 288];
    for (uint32_t i = 0; i < m; ++i) node[i] = {freq[order[i]], -1, -1};
    // two queues: leaves in rising weight, inner nodes in the order they are made (rising as well)
    uint32_t leaf = 0, inner = m, made = m;
    auto pop = [&]() -> int32_t {
        if (leaf < m && (inner >= made || node[leaf].w <= node[inner].w)) return (int32_t)leaf++;
        return (int32_t)inner++;
    };
    while ((m - leaf) + (made - inner) > 1) {
        const int32_t a = pop(), b = pop();
        node[made] = {node[a].w + node[b].w, a, b};
        ++made;
    }
    // depths, top down (a node's children were made before it)
    uint8_t depth[2 * 288];
    depth[made - 1] = 0;
    uint32_t count[64] = {0};
    for (int32_t i = (int32_t)made - 1; i >= (int32_t)m; --i) {
        depth[node[i].left] = depth[node[i].right] = (uint8_t)(depth[i] + 1);
    }
    for (uint32_t i = 0; i < m; ++i) ++count[std::min<uint32_t>(depth[i], limit)];
    uint64_t total = 0;
    for (uint32_t l = 1; l <= limit; ++l) total += (uint64_t)count[l] << (limit - l);
    while (total > (1ull << limit)) {
        --count[limit];
        for (uint32_t l = limit - 1; l > 0; --l)
            if (count[l]) { --count[l]; count[l + 1] += 2; break; }
        --total;
    }
    // the rarest symbols get the longest codes
    uint32_t i = 0;
    for (uint32_t l = limit; l > 0; --l)
        for (uint32_t c = 0; c < count[l]; ++c) lens[order[i++]] = (uint8_t)l;
}

// canonical codes, bit-reversed: the form the stream holds
void codes_of(const uint8_t *lens, uint32_t n, uint16_t *codes)
{
    uint32_t count[16] = {0}, next[16];
    for (uint32_t s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    for (uint32_t s = 0; s < n; ++s) codes[s] = lens[s] ? (uint16_t)reverse_bits(next[lens[s]]++, lens[s]) : 0;
}

struct BitOut {
    uint8_t *p;
    uint64_t bb = 0;
    uint32_t bc = 0;
    inline void put(uint32_t v, uint32_t n) { bb |= (uint64_t)v << bc; bc += n; }
    inline void flush() { memcpy(p, &bb, 8); p += bc >> 3; bb >>= bc & ~7u; bc &= 7; }    // (8 bytes of room past p needed)
    inline void finish() { flush(); if (bc) { *p++ = (uint8_t)bb; bb = 0; bc = 0; } }
};

}  // namespace

size_t huffman_only_bound(size_t n)
{
    // a stretch of kHuffSuper bytes never takes more than its stored form (5 bytes per 65,535 + the bytes) plus the partial
    // byte before it
    return n + (n / 65535 + n / kHuffSuper + 2) * 6 + 24;
}

// One Huffman code per kHuffSuper bytes (a histogram, code lengths of at most 12 bits), and under it one deflate block per
// kHuffSub bytes, every one with the code's header again (~1 % of the stream): blocks that an inflater reads one after the
// other like any others -- and that a decoder which is TOLD where each block's first symbol lies (HuffIndex: the index
// file keeps it in the members' extra fields) can take all at once, one lane per block (huff.hip on the GPU).  Twelve bits
// so that such a decoder's table is one 4,096-entry lookup without a second level.
size_t deflate_huffman_only(const uint8_t *in, size_t n, uint8_t *out, HuffIndex *index)
{
    BitOut o;
    o.p = out;
    if (index) { index->lens.clear(); index->sym_bit.clear(); index->all_coded = true; }
    if (!n) { o.put(1, 1); o.put(1, 2); o.put(0, 7); o.finish(); return (size_t)(o.p - out); }   // a final fixed-code block: end-of-block
    for (size_t at = 0; at < n;) {
        const size_t take = std::min(kHuffSuper, n - at);
        const uint8_t *b = in + at;
        const bool last = at + take == n;
        // histogram in four parts: runs of one value do not wait on one counter
        uint32_t h[4][256];
        memset(h, 0, sizeof h);
        size_t i = 0;
        for (; i + 4 <= take; i += 4) { ++h[0][b[i]]; ++h[1][b[i + 1]]; ++h[2][b[i + 2]]; ++h[3][b[i + 3]]; }
        for (; i < take; ++i) ++h[0][b[i]];
        const uint32_t nsub = (uint32_t)((take + kHuffSub - 1) / kHuffSub);
        uint32_t freq[257];
        for (int v = 0; v < 256; ++v) freq[v] = h[0][v] + h[1][v] + h[2][v] + h[3][v];
        freq[256] = nsub;                                         // end-of-block, once per block
        uint8_t lens[257 + 2];
        code_lengths(freq, 257, HuffIndex::kMaxLen, lens);
        lens[257] = lens[258] = 1;                                // two distance codes of one bit: a complete code nobody uses (zlib writes the same)
        // the lengths, run-length coded with the code-length alphabet (16: repeat the last 3-6 times, 17 / 18: 3-10 / 11-138 zeros)
        uint8_t rl_sym[259 + 8];
        uint8_t rl_extra[259 + 8];
        uint32_t nrl = 0, pf[19] = {0};
        for (uint32_t s = 0; s < 259;) {
            uint32_t run = 1;
            while (s + run < 259 && lens[s + run] == lens[s]) ++run;
            const uint8_t v = lens[s];
            s += run;
            if (v == 0) {
                while (run >= 11) { const uint32_t r = std::min(run, 138u); rl_sym[nrl] = 18; rl_extra[nrl++] = (uint8_t)(r - 11); run -= r; }
                if (run >= 3) { rl_sym[nrl] = 17; rl_extra[nrl++] = (uint8_t)(run - 3); run = 0; }
            } else {
                rl_sym[nrl] = v; rl_extra[nrl++] = 0; --run;
                while (run >= 3) { const uint32_t r = std::min(run, 6u); rl_sym[nrl] = 16; rl_extra[nrl++] = (uint8_t)(r - 3); run -= r; }
            }
            while (run--) { rl_sym[nrl] = v; rl_extra[nrl++] = 0; }
        }
        for (uint32_t k = 0; k < nrl; ++k) ++pf[rl_sym[k]];
        uint32_t used = 0;
        for (int v = 0; v < 19; ++v) used += pf[v] != 0;
        if (used < 2) pf[pf[0] ? 18 : 0] = 1;                     // (a code of one symbol is not a code zlib takes here)
        uint8_t plen[19];
        code_lengths(pf, 19, 7, plen);
        uint16_t pcode[19], code[257];
        codes_of(plen, 19, pcode);
        codes_of(lens, 257, code);
        uint32_t hclen = 19;
        while (hclen > 4 && !plen[kPreOrder[hclen - 1]]) --hclen;
        uint64_t hbits = 3 + 5 + 5 + 4 + 3 * hclen;
        for (uint32_t k = 0; k < nrl; ++k) hbits += plen[rl_sym[k]] + (rl_sym[k] == 16 ? 2 : rl_sym[k] == 17 ? 3 : rl_sym[k] == 18 ? 7 : 0);
        uint64_t bits = hbits * nsub;
        for (int v = 0; v < 257; ++v) bits += (uint64_t)freq[v] * lens[v];
        if ((bits + 7) / 8 >= take + 5 * ((take + 65534) / 65535)) {
            // stored blocks of up to 65,535 bytes each
            for (size_t d = 0; d < take;) {
                const size_t m = std::min<size_t>(65535, take - d);
                o.put(last && d + m == take ? 1 : 0, 1); o.put(0, 2);
                o.finish();
                const uint16_t len = (uint16_t)m, nlen = (uint16_t)~len;
                memcpy(o.p, &len, 2); memcpy(o.p + 2, &nlen, 2); memcpy(o.p + 4, b + d, m);
                o.p += 4 + m;
                d += m;
            }
            if (index) index->all_coded = false;
            at += take;
            continue;
        }
        if (index) index->lens.insert(index->lens.end(), lens, lens + 257);
        // code and length of a value in one word
        uint32_t cl[256];
        for (int v = 0; v < 256; ++v) cl[v] = (uint32_t)code[v] | (uint32_t)lens[v] << 16;
        for (uint32_t sb = 0; sb < nsub; ++sb) {
            const uint8_t *q = b + (size_t)sb * kHuffSub;
            const size_t m = std::min(kHuffSub, take - (size_t)sb * kHuffSub);
            o.put(last && sb + 1 == nsub ? 1 : 0, 1); o.put(2, 2);
            o.put(257 - 257, 5); o.put(2 - 1, 5); o.put(hclen - 4, 4);
            o.flush();
            for (uint32_t k = 0; k < hclen; ++k) { o.put(plen[kPreOrder[k]], 3); if ((k & 7) == 7) o.flush(); }
            o.flush();
            for (uint32_t k = 0; k < nrl; ++k) {
                o.put(pcode[rl_sym[k]], plen[rl_sym[k]]);
                if (rl_sym[k] >= 16) o.put(rl_extra[k], rl_sym[k] == 16 ? 2 : rl_sym[k] == 17 ? 3 : 7);
                o.flush();
            }
            if (index) index->sym_bit.push_back((uint64_t)(o.p - out) * 8 + o.bc);
            // the bytes, four values per flush (4 x 12 bits + the 7 left over < 64)
            i = 0;
            for (; i + 4 <= m; i += 4) {
                const uint32_t a = cl[q[i]], c = cl[q[i + 1]], d = cl[q[i + 2]], e = cl[q[i + 3]];
                o.put(a & 0xffff, a >> 16); o.put(c & 0xffff, c >> 16); o.put(d & 0xffff, d >> 16); o.put(e & 0xffff, e >> 16);
                o.flush();
            }
            for (; i < m; ++i) { const uint32_t a = cl[q[i]]; o.put(a & 0xffff, a >> 16); o.flush(); }
            o.put(code[256], lens[256]);
            o.flush();
        }
        at += take;
    }
    o.finish();
    return (size_t)(o.p - out);
}

}  // namespace mkhost
