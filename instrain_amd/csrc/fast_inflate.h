// fast_inflate.h -- a raw-deflate decoder for the host side of the BAM front end (BGZF blocks: at most 64 KiB each, sizes known).
//
// What it replaces: zlib's inflate under htslib's bgzf.c, which is what the reference's pysam calls decode their BAM with
// (filter_reads.py:885-956, profile_utilities.py:150-153).  On the GPU box's lease (16 CPUs) inflating is the largest single cost of
// profile_bam's scan pass; zlib's inflate_fast delivers ~0.5 GB/s per thread there.  This decoder is table driven (RFC 1951):
//   * one 64-bit bit buffer, topped up with ONE unaligned 8-byte load per symbol group;
//   * literal / length codes through an 11-bit first-level table whose entries already carry the symbol's meaning (literal byte,
//     length base + number of extra bits, end of block) -- codes longer than 11 bits go through 16-entry second-level tables;
//     distance codes through an 8-bit first-level table (+ 128-entry second level);
//   * up to three literals per refill; the next symbol's table entry is looked up before a match is copied; matches copied sixteen
//     bytes at once and eight a step beyond (runs with a period below 8 from a pattern word);
//   * a careful byte-wise loop only for the last ~290 bytes of a block, where a wide store could reach beyond the block.
// Anything irregular (a code it cannot place, sizes that do not come out) makes it return false and the caller falls back to zlib:
// it never has to be the judge of a corrupt file.  Pinned against zlib by tests/test_inflate.py (isx_bgzf_inflate_fast).
#pragma once
#include <stdint.h>
#include <string.h>

namespace isxinf {

class FastInflater {
public:
    // one raw deflate stream of exactly n_out bytes; false = not decoded (fall back)
    bool run(const uint8_t *in, size_t n_in, uint8_t *out, size_t n_out)
    {
        in_ = in; end_ = in + n_in; buf_ = 0; cnt_ = 0; over_ = false;
        size_t o = 0;
        for (;;) {
            refill();
            const uint32_t last = bits(1), type = bits(2);
            if (type == 0) {
                drop(cnt_ & 7);
                refill();
                const uint32_t len = bits(16), nlen = bits(16);
                if (over_ || (len ^ 0xFFFFu) != nlen) return false;
                // give back the whole bytes the buffer holds beyond this point: they are read directly
                in_ -= cnt_ >> 3;
                buf_ = 0; cnt_ = 0;
                if ((size_t)(end_ - in_) < len || o + len > n_out) return false;
                memcpy(out + o, in_, len);
                in_ += len; o += len;
            } else if (type == 1 || type == 2) {
                if (type == 1) { if (!fixed_tables()) return false; }
                else if (!dynamic_tables()) return false;
                if (!codes(out, o, n_out)) return false;
            } else return false;
            if (over_) return false;
            if (last) break;
        }
        return o == n_out;
    }

private:
    enum { LB = 11, DB = 8, LSUB = 4, DSUB = 7 };                   // first-level bits; second level: 15 - first
    // entry: bits 0-3 bits to remove, 4-7 extra bits, 8-9 kind (0 literal / distance, 1 length, 2 end of block, 3 second level), 16-31 value
    enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3 };
    uint32_t lit_[(1 << LB) + 288 * (1 << LSUB)];
    uint32_t dst_[(1 << DB) + 32 * (1 << DSUB)];
    const uint8_t *in_ = nullptr, *end_ = nullptr;
    uint64_t buf_ = 0;
    int cnt_ = 0;
    bool over_ = false;
    int fixed_ready_ = 0;

    inline void refill()
    {
        if (end_ - in_ >= 8) {
            uint64_t w;
            memcpy(&w, in_, 8);
            buf_ |= w << cnt_;
            const int k = (63 - cnt_) >> 3;
            in_ += k;
            cnt_ += 8 * k;
        } else {
            while (cnt_ <= 56 && in_ < end_) { buf_ |= (uint64_t)(*in_++) << cnt_; cnt_ += 8; }
        }
    }
    inline void drop(int n)
    {
        buf_ >>= n;
        cnt_ -= n;
        if (cnt_ < 0) { cnt_ = 0; over_ = true; }
    }
    inline uint32_t bits(int n)
    {
        const uint32_t v = (uint32_t)(buf_ & (((uint64_t)1 << n) - 1));
        drop(n);
        return v;
    }

    static inline uint32_t reverse(uint32_t code, int len)
    {
        uint32_t r = 0;
        for (int k = 0; k < len; k++) r |= ((code >> k) & 1u) << (len - 1 - k);
        return r;
    }

    // canonical code of lengths[0..n) into a two-level table; meaning(sym) gives an entry's upper part (kind, extra bits, value)
    template <class Meaning>
    bool build(const uint8_t *lengths, int n, uint32_t *tab, int fb, int sb, int max_sub, Meaning meaning)
    {
        int count[16] = {0};
        for (int s = 0; s < n; s++) count[lengths[s]]++;
        if (count[0] == n) { for (int j = 0; j < (1 << fb); j++) tab[j] = 0; return true; }       // no codes: every lookup fails
        int left = 1;
        for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
        if (left > 0 && n - count[0] != 1) return false;            // incomplete only with a single code (zlib's rule, loosened like puff)
        uint32_t next[16];
        next[1] = 0;
        for (int l = 1; l < 15; l++) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
        for (int j = 0; j < (1 << fb); j++) tab[j] = 0;
        int n_sub = 0;
        for (int s = 0; s < n; s++) {
            const int l = lengths[s];
            if (!l) continue;
            const uint32_t rev = reverse(next[l]++, l);
            const uint32_t m = meaning(s);
            if (l <= fb) {
                const uint32_t e = m | (uint32_t)l;
                for (uint32_t j = rev; j < (1u << fb); j += 1u << l) tab[j] = e;
            } else {
                const uint32_t pre = rev & ((1u << fb) - 1u);
                uint32_t p = tab[pre];
                if (((p >> 8) & 3u) != K_SUB || (p & 15u) == 0) {
                    if (n_sub == max_sub) return false;
                    const uint32_t at = (1u << fb) + (uint32_t)n_sub * (1u << sb);
                    n_sub++;
                    for (uint32_t j = 0; j < (1u << sb); j++) tab[at + j] = 0;
                    p = (at << 16) | ((uint32_t)K_SUB << 8) | (uint32_t)fb;
                    tab[pre] = p;
                }
                const uint32_t at = p >> 16;
                const int rest = l - fb;
                const uint32_t e = m | (uint32_t)rest;
                for (uint32_t j = rev >> fb; j < (1u << sb); j += 1u << rest) tab[at + j] = e;
            }
        }
        return true;
    }

    static inline uint32_t lit_meaning(int s)
    {
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return ((uint32_t)s << 16) | ((uint32_t)K_LIT << 8);
        if (s == 256) return (uint32_t)K_EOB << 8;
        if (s > 285) return ((uint32_t)K_EOB << 8) | (1u << 12);                 // 286, 287: never valid in a stream (flag: bad)
        return ((uint32_t)lbase[s - 257] << 16) | ((uint32_t)K_LEN << 8) | ((uint32_t)lext[s - 257] << 4);
    }
    static inline uint32_t dist_meaning(int s)
    {
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s > 29) return ((uint32_t)K_EOB << 8) | (1u << 12);                  // 30, 31: bad
        return ((uint32_t)dbase[s] << 16) | ((uint32_t)dext[s] << 4);
    }

    bool fixed_tables()
    {
        uint8_t l[288];
        for (int s = 0; s < 144; s++) l[s] = 8;
        for (int s = 144; s < 256; s++) l[s] = 9;
        for (int s = 256; s < 280; s++) l[s] = 7;
        for (int s = 280; s < 288; s++) l[s] = 8;
        if (!build(l, 288, lit_, LB, LSUB, 288, lit_meaning)) return false;
        for (int s = 0; s < 32; s++) l[s] = 5;
        return build(l, 32, dst_, DB, DSUB, 32, dist_meaning);
    }

    bool dynamic_tables()
    {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        refill();
        const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
        if (nlen > 286 || ndist > 30) return false;
        uint8_t cl[19] = {0};
        for (int i = 0; i < ncode; i++) { if (cnt_ < 3) refill(); cl[order[i]] = (uint8_t)bits(3); }
        // the code-length code: 19 symbols, at most 7 bits -- a flat 7-bit table
        uint8_t clt[128];
        {
            int count[8] = {0};
            for (int s = 0; s < 19; s++) count[cl[s]]++;
            int left = 1;
            for (int l = 1; l <= 7; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
            if (left != 0) return false;
            uint32_t next[9];
            next[1] = 0;
            for (int l = 1; l < 8; l++) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
            memset(clt, 0xFF, sizeof clt);
            for (int s = 0; s < 19; s++) {
                const int l = cl[s];
                if (!l) continue;
                const uint32_t rev = reverse(next[l]++, l);
                for (uint32_t j = rev; j < 128; j += 1u << l) clt[j] = (uint8_t)((s << 3) | l);
            }
        }
        uint8_t lens[286 + 30];
        int idx = 0;
        while (idx < nlen + ndist) {
            if (cnt_ < 14) refill();
            const uint8_t e = clt[buf_ & 127];
            if (e == 0xFF) return false;
            drop(e & 7);
            const int sym = e >> 3;
            if (sym < 16) lens[idx++] = (uint8_t)sym;
            else {
                int rep, val = 0;
                if (sym == 16) { if (!idx) return false; val = lens[idx - 1]; rep = 3 + (int)bits(2); }
                else if (sym == 17) rep = 3 + (int)bits(3);
                else rep = 11 + (int)bits(7);
                if (idx + rep > nlen + ndist) return false;
                while (rep--) lens[idx++] = (uint8_t)val;
            }
            if (over_) return false;
        }
        if (lens[256] == 0) return false;
        return build(lens, nlen, lit_, LB, LSUB, 288, lit_meaning) && build(lens + nlen, ndist, dst_, DB, DSUB, 32, dist_meaning);
    }

    // the symbols of one block; o advances.  Two loops: the fast one runs while 8 input bytes and 272 output bytes are left (no bounds
    // checks inside: a refill always finds its 8 bytes, a wide copy never leaves the block) with the bit buffer in locals -- byte
    // stores through `out` may alias this object's members as far as the compiler knows, so members would be re-read after each
    // literal --; the careful one finishes the block.
    bool codes(uint8_t *out, size_t &o_io, size_t n_out)
    {
        size_t o = o_io;
        const size_t wide_end = n_out >= 288 ? n_out - 288 : 0;     // below this a copy may write up to 24 bytes beyond its match
        if (end_ - in_ >= 16) {
            uint64_t buf = buf_;
            int cnt = cnt_;
            const uint8_t *in = in_;
            const uint8_t *const in_fast = end_ - 16;              // two refills of 8 bytes are safe from here
            const uint32_t *const lit = lit_, *const dst = dst_;
            bool done = false, bad = false;
#define ISXINF_REFILL() do { uint64_t w_; memcpy(&w_, in, 8); buf |= w_ << cnt; const int k_ = (63 - cnt) >> 3; in += k_; cnt += 8 * k_; } while (0)
            ISXINF_REFILL();
            uint32_t e = lit[buf & ((1u << LB) - 1u)];             // the entry of the symbol about to be decoded (nothing consumed yet)
            while (in <= in_fast && o < wide_end) {
                if (__builtin_expect(((e >> 8) & 3u) == K_SUB, 0)) {
                    if (!(e & 15u)) { bad = true; break; }
                    buf >>= LB; cnt -= LB;
                    e = lit[(e >> 16) + (uint32_t)(buf & ((1u << LSUB) - 1u))];
                }
                if (__builtin_expect(!(e & 15u), 0)) { bad = true; break; }
                buf >>= (e & 15u); cnt -= (int)(e & 15u);
                const uint32_t kind = (e >> 8) & 3u;
                if (kind == K_LIT) {
                    out[o++] = (uint8_t)(e >> 16);
                    // two more literals from what the buffer still holds (>= 41 bits)
                    uint32_t f = lit[buf & ((1u << LB) - 1u)];
                    if (((f >> 8) & 3u) == K_LIT && (f & 15u)) {
                        buf >>= (f & 15u); cnt -= (int)(f & 15u);
                        out[o++] = (uint8_t)(f >> 16);
                        f = lit[buf & ((1u << LB) - 1u)];
                        if (((f >> 8) & 3u) == K_LIT && (f & 15u)) {
                            buf >>= (f & 15u); cnt -= (int)(f & 15u);
                            out[o++] = (uint8_t)(f >> 16);
                        }
                    }
                    ISXINF_REFILL();
                    e = lit[buf & ((1u << LB) - 1u)];
                    continue;
                }
                if (kind == K_EOB) { if (e & (1u << 12)) bad = true; else done = true; break; }
                const int xl = (int)((e >> 4) & 15u);
                const uint32_t len = (e >> 16) + (uint32_t)(buf & (((uint64_t)1 << xl) - 1));
                buf >>= xl; cnt -= xl;
                uint32_t d = dst[buf & ((1u << DB) - 1u)];
                if (__builtin_expect(((d >> 8) & 3u) == K_SUB, 0)) {
                    if (!(d & 15u)) { bad = true; break; }
                    buf >>= DB; cnt -= DB;
                    d = dst[(d >> 16) + (uint32_t)(buf & ((1u << DSUB) - 1u))];
                }
                if (__builtin_expect(!(d & 15u) || ((d >> 8) & 3u) != K_LIT, 0)) { bad = true; break; }
                buf >>= (d & 15u); cnt -= (int)(d & 15u);
                const int xd = (int)((d >> 4) & 15u);
                const uint32_t dist = (d >> 16) + (uint32_t)(buf & (((uint64_t)1 << xd) - 1));
                buf >>= xd; cnt -= xd;
                if (__builtin_expect(dist > o, 0)) { bad = true; break; }
                // the next symbol's entry is looked up BEFORE this match is copied: the table read's latency passes under the copy
                ISXINF_REFILL();
                e = lit[buf & ((1u << LB) - 1u)];
                uint8_t *dp = out + o;
                const uint8_t *sp = dp - dist;
                o += len;
                if (__builtin_expect(dist >= 8, 1)) {
                    // sixteen bytes whatever the length (most matches are shorter), more in steps of eight
                    uint64_t w;
                    memcpy(&w, sp, 8); memcpy(dp, &w, 8);
                    memcpy(&w, sp + 8, 8); memcpy(dp + 8, &w, 8);
                    for (uint32_t k = 16; k < len; k += 8) { memcpy(&w, sp + k, 8); memcpy(dp + k, &w, 8); }
                } else if (dist == 1) {
                    const uint64_t pat = 0x0101010101010101ull * sp[0];
                    for (uint32_t k = 0; k < len; k += 8) memcpy(dp + k, &pat, 8);
                } else {
                    uint64_t pat = 0;
                    for (uint32_t j = 0, r = 0; j < 8; j++) { pat |= (uint64_t)sp[r] << (8 * j); r = r + 1 == dist ? 0 : r + 1; }
                    const uint32_t step = (8u / dist) * dist;
                    for (uint32_t k = 0; k < len; k += step) memcpy(dp + k, &pat, 8);
                }
            }
#undef ISXINF_REFILL
            buf_ = buf; cnt_ = cnt; in_ = in;
            if (bad || cnt < 0) return false;
            if (done) { o_io = o; return true; }
        }
        for (;;) {
            refill();
            uint32_t e = lit_[buf_ & ((1u << LB) - 1u)];
            if (((e >> 8) & 3u) == K_SUB) {
                if (!(e & 15u)) return false;
                drop(LB);
                e = lit_[(e >> 16) + (uint32_t)(buf_ & ((1u << LSUB) - 1u))];
            }
            if (!(e & 15u)) return false;                           // no code here
            drop((int)(e & 15u));
            const uint32_t kind = (e >> 8) & 3u;
            if (kind == K_LIT) {
                if (o >= n_out || over_) return false;
                out[o++] = (uint8_t)(e >> 16);
                continue;
            }
            if (kind == K_EOB) { if (e & (1u << 12)) return false; break; }
            const uint32_t len = (e >> 16) + bits((int)((e >> 4) & 15u));
            uint32_t d = dst_[buf_ & ((1u << DB) - 1u)];
            if (((d >> 8) & 3u) == K_SUB) {
                if (!(d & 15u)) return false;
                drop(DB);
                d = dst_[(d >> 16) + (uint32_t)(buf_ & ((1u << DSUB) - 1u))];
            }
            if (!(d & 15u) || ((d >> 8) & 3u) != K_LIT) return false;
            drop((int)(d & 15u));
            const uint32_t dist = (d >> 16) + bits((int)((d >> 4) & 15u));
            if (over_ || dist > o || o + len > n_out) return false;
            uint8_t *dp = out + o;
            const uint8_t *sp = dp - dist;
            for (uint32_t k = 0; k < len; k++) dp[k] = sp[k];
            o += len;
        }
        o_io = o;
        return !over_;
    }
};

}  // namespace isxinf
