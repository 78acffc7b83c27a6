// isx_inflate.hip -- BGZF blocks inflated on the device.
//
// What it replaces: the block decompression under the reference's two passes over its BAM -- pysam's fetch / pileup
// (filter_reads.py:885-956, profile_utilities.py:150-153) sit on htslib's bgzf.c, which hands every 64 KiB block to zlib's inflate.
// On the host that is the front end's largest single cost (bam_front.cpp: scan = inflate + walk); the blocks are independent,
// so here each one is decoded by ONE LANE: a wave decodes 64 blocks side by side, a launch all blocks of a file.  A lane's Huffman
// tables (canonical: counts per length + symbols in code order, RFC 1951 3.2.2) live in LDS, element i of a lane at [i * 64 + lane]
// (lanes that read the same element of their own tables hit 64 different banks).  Decoding is bit-serial per symbol -- a lane
// spends ~1 us per symbol on dependent LDS reads -- and still the file is done in the time ONE block takes, a few tens of
// milliseconds, because tens of thousands of blocks are in flight; the host's 16 threads need 200 ms for the same bytes.
//
// The decoder itself (inflate_stream) is plain C++ and is compiled for the host as well: isx_bgzf_inflate_host runs it on the
// calling thread, which is how the CPU tests pin it against zlib without a GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "isx_internal.h"
#include "isx_batch.h"

namespace {

// element i of a lane's table lives at t[i * stride]
struct Tab {
    uint16_t *t;
    int stride;
    __host__ __device__ uint16_t &operator[](int i) const { return t[(size_t)i * (size_t)stride]; }
};
struct Tab8 {
    uint8_t *t;
    int stride;
    __host__ __device__ uint8_t &operator[](int i) const { return t[(size_t)i * (size_t)stride]; }
};

// the input as a stream of bits, least significant first: `cnt` valid bits in `buf` (zeros above them); bits = 8 * (end - p) + cnt
// are left, `avail` counts the same and goes negative when a caller consumes what is not there (-> over)
struct Bits {
    const uint8_t *p, *end;
    uint64_t buf;
    int cnt;
    bool over;
};

// top the buffer up to >= 56 bits: ONE unaligned 8-byte load while 8 bytes are left (a lane's loads are dependent global round trips:
// byte by byte they were most of a block's time), single bytes at the very end
__host__ __device__ inline void refill(Bits &b)
{
    if (b.end - b.p >= 8) {
        uint64_t w;
        memcpy(&w, b.p, 8);
        b.buf |= w << b.cnt;
        const int k = (63 - b.cnt) >> 3;        // whole bytes that fit
        b.p += k;
        b.cnt += 8 * k;
    } else {
        while (b.cnt <= 56 && b.p < b.end) { b.buf |= (uint64_t)(*b.p++) << b.cnt; b.cnt += 8; }
    }
}

// n <= 32 bits, least significant first
__host__ __device__ inline uint32_t take(Bits &b, int n, int64_t &avail)
{
    if (b.cnt < n) refill(b);
    avail -= n;
    if (avail < 0) b.over = true;
    const uint32_t v = (uint32_t)(b.buf & ((n == 32) ? 0xFFFFFFFFull : (((uint64_t)1 << n) - 1)));
    b.buf >>= n;
    b.cnt -= n;
    if (b.cnt < 0) b.cnt = 0;
    return v;
}

// canonical Huffman code from lengths[0..n): count[len] for len 0..15, symbols in code order; returns 0 complete, > 0 incomplete,
// < 0 over-subscribed
__host__ __device__ inline int build(const Tab8 &lengths, int first, int n, const Tab &count, const Tab &symbol, uint16_t *offs /* [16] private */)
{
    for (int l = 0; l <= 15; l++) count[l] = 0;
    for (int s = 0; s < n; s++) count[lengths[first + s]] = (uint16_t)(count[lengths[first + s]] + 1);
    if (count[0] == n) return 0;                // no codes: complete as far as the format goes, decoding anything fails
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; s++) {
        const int l = lengths[first + s];
        if (l) { symbol[offs[l]] = (uint16_t)s; offs[l]++; }
    }
    return left;
}

__host__ __device__ inline int decode(Bits &b, int64_t &avail, const Tab &count, const Tab &symbol)
{
    if (b.cnt < 15) refill(b);
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int c = count[len];
        if (code - c < first) {
            avail -= len;
            if (avail < 0) b.over = true;
            b.buf >>= len;
            b.cnt -= len;
            if (b.cnt < 0) b.cnt = 0;
            return symbol[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;              // ran out of codes
}

// First-level lookup beside the canonical tables: entry [next FB bits of the stream] = (symbol << 4) | code length for every code of at most FB
// bits, 0 for longer ones (those take the bit-serial walk above).  The lanes of a wave decode in lockstep, so the walk cost every lane the
// LONGEST code among the 64 symbols being decoded; the lookup costs one LDS read whatever the length.
constexpr int FB_LIT = 8, FB_DIST = 5;

__host__ __device__ inline void build_fast(const Tab8 &lengths, int first, int n, const Tab &count, const Tab &fast, int fb)
{
    uint16_t next[16];
    next[0] = 0; next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (uint16_t)((next[l] + count[l]) << 1);
    for (int j = 0; j < (1 << fb); j++) fast[j] = 0;
    for (int s = 0; s < n; s++) {
        const int l = lengths[first + s];
        if (!l) continue;
        const uint32_t code = next[l]++;
        if (l > fb) continue;
        uint32_t rev = 0;                       // Huffman codes are packed starting with their most significant bit (RFC 1951 3.1.1)
        for (int k = 0; k < l; k++) rev |= ((code >> k) & 1u) << (l - 1 - k);
        for (uint32_t j = rev; j < (1u << fb); j += 1u << l) fast[(int)j] = (uint16_t)((s << 4) | l);
    }
}

__host__ __device__ inline int decode_fast(Bits &b, int64_t &avail, const Tab &fast, int fb, const Tab &count, const Tab &symbol)
{
    if (b.cnt < 15) refill(b);
    const uint32_t e = fast[(int)(b.buf & ((1u << fb) - 1u))];
    if (!e) return decode(b, avail, count, symbol);
    const int len = (int)(e & 15u);
    avail -= len;
    if (avail < 0) b.over = true;
    b.buf >>= len;
    b.cnt -= len;
    if (b.cnt < 0) b.cnt = 0;
    return (int)(e >> 4);
}

enum { INF_OK = 0, INF_BAD_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_LENGTHS = 3, INF_BAD_CODE = 4, INF_BAD_DIST = 5, INF_OUT_SIZE = 6, INF_IN_SIZE = 7 };

// one raw deflate stream (RFC 1951) of exactly n_out bytes.  Tables: lc[16] ls[288] dc[16] ds[32] lf[1 << FB_LIT] df[1 << FB_DIST] (uint16),
// lengths[339] (uint8)
__host__ __device__ inline int inflate_stream(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out,
                                              const Tab &lc, const Tab &ls, const Tab &dc, const Tab &ds, const Tab &lf, const Tab &df, const Tab8 &lengths)
{
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits b{in, in + n_in, 0, 0, false};
    int64_t avail = (int64_t)n_in * 8;
    uint32_t o = 0;
    uint16_t offs[16];
    for (;;) {
        const uint32_t last = take(b, 1, avail), type = take(b, 2, avail);
        if (type == 0) {                        // stored: to the next byte boundary, LEN, NLEN, bytes
            const int drop = b.cnt & 7;
            take(b, drop, avail);
            const uint32_t len = take(b, 16, avail), nlen = take(b, 16, avail);
            if (b.over || (len ^ 0xFFFFu) != nlen) return INF_BAD_STORED;
            // whole bytes still sit in the bit buffer: hand them out first
            uint32_t k = 0;
            while (k < len && b.cnt >= 8) {
                if (o >= n_out) return INF_OUT_SIZE;
                out[o++] = (uint8_t)take(b, 8, avail);
                k++;
            }
            if ((uint32_t)(b.end - b.p) < len - k) return INF_IN_SIZE;
            if (o + (len - k) > n_out) return INF_OUT_SIZE;
            if (k < len) { b.buf = 0; b.cnt = 0; }          // (nothing counted is left in the buffer; what refill() put beyond its count goes: the bytes are read directly)
            for (; k < len; k++) out[o++] = *b.p++;
            avail = (int64_t)(b.end - b.p) * 8 + b.cnt;
        } else if (type == 1 || type == 2) {
            if (type == 1) {                    // fixed codes
                for (int s = 0; s < 144; s++) lengths[s] = 8;
                for (int s = 144; s < 256; s++) lengths[s] = 9;
                for (int s = 256; s < 280; s++) lengths[s] = 7;
                for (int s = 280; s < 288; s++) lengths[s] = 8;
                build(lengths, 0, 288, lc, ls, offs);
                build_fast(lengths, 0, 288, lc, lf, FB_LIT);
                for (int s = 0; s < 30; s++) lengths[s] = 5;
                build(lengths, 0, 30, dc, ds, offs);
                build_fast(lengths, 0, 30, dc, df, FB_DIST);
            } else {
                const int nlen = (int)take(b, 5, avail) + 257, ndist = (int)take(b, 5, avail) + 1, ncode = (int)take(b, 4, avail) + 4;
                if (nlen > 286 || ndist > 30) return INF_BAD_LENGTHS;
                for (int i = 0; i < 19; i++) lengths[i] = 0;
                for (int i = 0; i < ncode; i++) lengths[order[i]] = (uint8_t)take(b, 3, avail);
                if (build(lengths, 0, 19, lc, ls, offs) != 0) return INF_BAD_LENGTHS;     // the code-length code must be complete
                int idx = 0;
                // (the code lengths are decoded with lc/ls, then written to `lengths` from slot 19 on so that the decoding tables stay intact)
                while (idx < nlen + ndist) {
                    const int sym = decode(b, avail, lc, ls);
                    if (sym < 0 || b.over) return INF_BAD_LENGTHS;
                    if (sym < 16) lengths[19 + idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return INF_BAD_LENGTHS;
                            val = lengths[19 + idx - 1];
                            rep = 3 + (int)take(b, 2, avail);
                        } else if (sym == 17) rep = 3 + (int)take(b, 3, avail);
                        else rep = 11 + (int)take(b, 7, avail);
                        if (idx + rep > nlen + ndist) return INF_BAD_LENGTHS;
                        while (rep--) lengths[19 + idx++] = (uint8_t)val;
                    }
                }
                if (lengths[19 + 256] == 0) return INF_BAD_LENGTHS;                      // no end-of-block code
                int rc = build(lengths, 19, nlen, lc, ls, offs);
                if (rc < 0 || (rc > 0 && nlen - lc[0] != 1)) return INF_BAD_LENGTHS;     // incomplete only with a single code
                rc = build(lengths, 19 + nlen, ndist, dc, ds, offs);
                if (rc < 0 || (rc > 0 && ndist - dc[0] != 1)) return INF_BAD_LENGTHS;
                build_fast(lengths, 19, nlen, lc, lf, FB_LIT);
                build_fast(lengths, 19 + nlen, ndist, dc, df, FB_DIST);
            }
            for (;;) {
                int sym = decode_fast(b, avail, lf, FB_LIT, lc, ls);
                if (sym < 0 || b.over) return INF_BAD_CODE;
                if (sym < 256) {
                    if (o >= n_out) return INF_OUT_SIZE;
                    out[o++] = (uint8_t)sym;
                } else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return INF_BAD_CODE;
                    const uint32_t len = lbase[sym] + take(b, lext[sym], avail);
                    const int ds_ = decode_fast(b, avail, df, FB_DIST, dc, ds);
                    if (ds_ < 0 || ds_ >= 30 || b.over) return INF_BAD_DIST;
                    const uint32_t dist = dbase[ds_] + take(b, dext[ds_], avail);
                    if (dist > o) return INF_BAD_DIST;
                    if (o + len > n_out) return INF_OUT_SIZE;
                    uint32_t k = 0;
                    if (dist >= 8) {            // eight bytes a step: the source lies wholly behind what this step writes
                        for (; k + 8 <= len; k += 8, o += 8) { uint64_t w; memcpy(&w, out + o - dist, 8); memcpy(out + o, &w, 8); }
                    } else if (len >= 8) {
                        // a run with a period below 8 (BAM quality strings are full of them: distance 1, length 258): the period laid out
                        // over eight bytes once, then stored at steps of the largest multiple of the period that fits
                        uint8_t pb[8];
                        for (uint32_t j = 0; j < dist; j++) pb[j] = out[o - dist + j];
                        uint64_t pat = 0;
                        for (uint32_t j = 0, r = 0; j < 8; j++) { pat |= (uint64_t)pb[r] << (8 * j); r = r + 1 == dist ? 0 : r + 1; }
                        const uint32_t step = (8u / dist) * dist;
                        for (; k + 8 <= len; k += step, o += step) memcpy(out + o, &pat, 8);
                    }
                    for (; k < len; k++, o++) out[o] = out[o - dist];
                }
            }
        } else return INF_BAD_TYPE;
        if (b.over) return INF_IN_SIZE;
        if (last) break;
    }
    return o == n_out ? INF_OK : INF_OUT_SIZE;
}

constexpr int TAB16 = 16 + 288 + 16 + 32 + (1 << FB_LIT) + (1 << FB_DIST);       // uint16 elements of a lane's tables: 1280 bytes
constexpr int TAB8 = 320 + 19;                  // code lengths: 19 of the code-length code + up to 286 + 30 (+ slack)

// LPW lanes of a wave decode a block each, the other lanes leave at once.  A lane's decoding is a chain of dependent memory round trips
// (refill, table reads, copies of earlier output) and the lanes of a wave move in lockstep -- every step costs what the slowest lane's
// step costs -- so fewer blocks per wave and MORE WAVES per CU is what keeps a CU busy: with 64 blocks a wave a file's 26 000 blocks
// were 1.6 waves a CU (kernel 48 ms), with 8 they are 13.
template <int LPW>
__global__ void __launch_bounds__(64) k_bgzf_inflate(const uint8_t *__restrict__ comp, const isx_bgzf_block *__restrict__ blocks, int64_t n_blocks,
                                                     uint8_t *__restrict__ out, uint32_t *__restrict__ status)
{
    __shared__ uint16_t t16[TAB16 * LPW];         // 1280 bytes a lane
    uint8_t lens[TAB8];                           // the code lengths while a block's tables are built: private memory
    const int lane = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * LPW + lane;
    if (lane >= LPW || i >= n_blocks) return;
    const isx_bgzf_block bl = blocks[i];
    uint16_t *base = t16 + lane;
    const Tab lc{base, LPW}, ls{base + 16 * LPW, LPW}, dc{base + (16 + 288) * LPW, LPW}, ds{base + (16 + 288 + 16) * LPW, LPW};
    const Tab lf{base + (16 + 288 + 16 + 32) * LPW, LPW}, df{base + (16 + 288 + 16 + 32 + (1 << FB_LIT)) * LPW, LPW};
    const Tab8 lengths{lens, 1};
    const int rc = bl.out_len ? inflate_stream(comp + bl.in_off, (uint32_t)bl.in_len, out + bl.out_off, (uint32_t)bl.out_len, lc, ls, dc, ds, lf, df, lengths) : INF_OK;
    if (rc != INF_OK) atomicMax(status, (uint32_t)rc | ((uint32_t)(i & 0xFFFFFF) << 8));
}

const char *inf_text(int rc)
{
    switch (rc) {
    case INF_BAD_TYPE: return "reserved block type";
    case INF_BAD_STORED: return "stored block: LEN / NLEN mismatch";
    case INF_BAD_LENGTHS: return "bad code lengths";
    case INF_BAD_CODE: return "bad literal / length code";
    case INF_BAD_DIST: return "bad distance";
    case INF_OUT_SIZE: return "inflated size differs from ISIZE";
    case INF_IN_SIZE: return "deflate stream longer than its block";
    default: return "?";
    }
}

}  // namespace

int isx_bgzf_index(const uint8_t *file, int64_t n_bytes, int64_t cap_blocks, isx_bgzf_block *blocks, int64_t *n_blocks, int64_t *out_bytes)
{
    if (!file || n_bytes < 0 || !n_blocks || !out_bytes || (cap_blocks > 0 && !blocks)) { isx_set_error("isx_bgzf_index: bad argument"); return ISX_ERR_ARG; }
    int64_t off = 0, n = 0, total = 0;
    while (off < n_bytes) {
        if (off + 18 > n_bytes) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        const uint8_t *h = file + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { isx_set_error("not a BGZF file"); return ISX_ERR_IO; }
        const int64_t xlen = h[10] | (h[11] << 8);
        if (off + 12 + xlen > n_bytes) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        int64_t bsize = 0;
        for (int64_t x = 12; x + 4 <= 12 + xlen;) {
            const int64_t slen = h[x + 2] | (h[x + 3] << 8);
            if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (int64_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || off + bsize > n_bytes) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        const uint8_t *t = h + bsize - 4;
        const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (isize > 65536) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        if (n < cap_blocks) {
            isx_bgzf_block &b = blocks[n];
            b.in_off = off + 12 + xlen; b.in_len = (int32_t)(bsize - (12 + xlen) - 8); b.out_len = (int32_t)isize; b.out_off = total;
        }
        n++;
        total += isize;
        off += bsize;
    }
    *n_blocks = n;
    *out_bytes = total;
    if (n > cap_blocks && cap_blocks > 0) { isx_set_error("isx_bgzf_index: more blocks than cap_blocks"); return ISX_ERR_CAPACITY; }
    return ISX_OK;
}

static int check_blocks(const isx_bgzf_block *blocks, int64_t n_blocks, int64_t n_bytes, int64_t out_bytes)
{
    for (int64_t i = 0; i < n_blocks; i++) {
        const isx_bgzf_block &b = blocks[i];
        if (b.in_off < 0 || b.in_len < 0 || b.in_off + b.in_len > n_bytes || b.out_len < 0 || b.out_len > 65536 || b.out_off < 0 ||
            b.out_off + b.out_len > out_bytes) { isx_set_error("isx_bgzf_inflate: a block reaches outside its buffer"); return ISX_ERR_ARG; }
    }
    return ISX_OK;
}

int isx_bgzf_inflate_host(const uint8_t *file, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, int64_t out_bytes)
{
    if (!file || !blocks || n_blocks < 0 || (!out && out_bytes) || out_bytes < 0) { isx_set_error("isx_bgzf_inflate_host: bad argument"); return ISX_ERR_ARG; }
    int rc = check_blocks(blocks, n_blocks, n_bytes, out_bytes);
    if (rc != ISX_OK) return rc;
    std::vector<uint16_t> t16(TAB16);
    std::vector<uint8_t> t8(TAB8);
    const Tab lc{t16.data(), 1}, ls{t16.data() + 16, 1}, dc{t16.data() + 16 + 288, 1}, ds{t16.data() + 16 + 288 + 16, 1};
    const Tab lf{t16.data() + 16 + 288 + 16 + 32, 1}, df{t16.data() + 16 + 288 + 16 + 32 + (1 << FB_LIT), 1};
    const Tab8 lengths{t8.data(), 1};
    for (int64_t i = 0; i < n_blocks; i++) {
        const isx_bgzf_block &b = blocks[i];
        if (!b.out_len) continue;
        const int e = inflate_stream(file + b.in_off, (uint32_t)b.in_len, out + b.out_off, (uint32_t)b.out_len, lc, ls, dc, ds, lf, df, lengths);
        if (e != INF_OK) { isx_set_error(std::string("BGZF block ") + std::to_string(i) + ": " + inf_text(e)); return ISX_ERR_IO; }
    }
    return ISX_OK;
}

int isx_bgzf_inflate_device(isx_ctx *c, const uint8_t *file, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks, uint8_t *out,
                            int64_t out_bytes, float *kernel_ms)
{
    if (!c || !file || !blocks || n_blocks < 0 || (!out && out_bytes) || out_bytes < 0) { isx_set_error("isx_bgzf_inflate_device: bad argument"); return ISX_ERR_ARG; }
    int rc = check_blocks(blocks, n_blocks, n_bytes, out_bytes);
    if (rc != ISX_OK) return rc;
    if (kernel_ms) *kernel_ms = 0.f;
    if (!n_blocks) return ISX_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // only the span of the file that holds the blocks travels
    int64_t lo = n_bytes, hi = 0;
    for (int64_t i = 0; i < n_blocks; i++) { lo = std::min(lo, blocks[i].in_off); hi = std::max(hi, blocks[i].in_off + blocks[i].in_len); }
    if (hi < lo) { lo = hi = 0; }
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    isx_bgzf_block *d_blocks = nullptr;
    uint32_t *d_status = nullptr;
    std::vector<isx_bgzf_block> shifted(blocks, blocks + n_blocks);
    for (auto &b : shifted) b.in_off -= lo;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        if (d_comp) isx_dev_free(d_comp);
        if (d_out) isx_dev_free(d_out);
        if (d_blocks) isx_dev_free(d_blocks);
        if (d_status) isx_dev_free(d_status);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define INF_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { isx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); cleanup(); return ISX_ERR_HIP; } } while (0)
    INF_TRY(isx_dev_malloc(reinterpret_cast<void **>(&d_comp), (size_t)std::max<int64_t>(hi - lo, 16)));
    INF_TRY(isx_dev_malloc(reinterpret_cast<void **>(&d_out), (size_t)std::max<int64_t>(out_bytes, 16)));
    INF_TRY(isx_dev_malloc(reinterpret_cast<void **>(&d_blocks), (size_t)n_blocks * sizeof(isx_bgzf_block)));
    INF_TRY(isx_dev_malloc(reinterpret_cast<void **>(&d_status), 16));
    INF_TRY(hipEventCreate(&e0));
    INF_TRY(hipEventCreate(&e1));
    INF_TRY(hipMemsetAsync(d_status, 0, 16, s));
    INF_TRY(hipMemcpyAsync(d_comp, file + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, s));
    INF_TRY(hipMemcpyAsync(d_blocks, shifted.data(), (size_t)n_blocks * sizeof(isx_bgzf_block), hipMemcpyHostToDevice, s));
    INF_TRY(hipEventRecord(e0, s));
    {
        static const int lpw = [] { const char *e = getenv("ISX_INFLATE_LPW"); const int v = e ? atoi(e) : 32; return v == 64 || v == 8 || v == 16 || v == 4 ? v : 32; }();
        const unsigned g = (unsigned)((n_blocks + lpw - 1) / lpw);
        if (lpw == 64) hipLaunchKernelGGL(k_bgzf_inflate<64>, dim3(g), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
        else if (lpw == 8) hipLaunchKernelGGL(k_bgzf_inflate<8>, dim3(g), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
        else if (lpw == 16) hipLaunchKernelGGL(k_bgzf_inflate<16>, dim3(g), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
        else if (lpw == 4) hipLaunchKernelGGL(k_bgzf_inflate<4>, dim3(g), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
        else hipLaunchKernelGGL(k_bgzf_inflate<32>, dim3(g), dim3(64), 0, s, d_comp, d_blocks, n_blocks, d_out, d_status);
    }
    INF_TRY(hipGetLastError());
    INF_TRY(hipEventRecord(e1, s));
    uint32_t status = 0;
    INF_TRY(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, s));
    INF_TRY(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, s));
    INF_TRY(hipStreamSynchronize(s));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, e0, e1);
    cleanup();
#undef INF_TRY
    if (status) {
        isx_set_error(std::string("BGZF block ") + std::to_string(status >> 8) + " (index modulo 2^24): " + inf_text((int)(status & 0xFF)));
        return ISX_ERR_IO;
    }
    return ISX_OK;
}
