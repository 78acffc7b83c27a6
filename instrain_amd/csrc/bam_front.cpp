// bam_front.cpp -- host side of the hot path: BGZF/BAM decode, read-pair filter and the
// htslib-1.9 pileup rules, producing the packed observation stream the kernels consume.
//
// Replaces (host side, C++; the reference reaches all of this through pysam -> htslib, which
// is not vendored under /root/reference):
//   pysam.AlignmentFile(bam) / samfile.fetch      /root/reference/inStrain/profile/profile_utilities.py:56
//   get_paired_reads                              /root/reference/inStrain/filter_reads.py:885-956
//   paired_read_filter                            filter_reads.py:471-532 (paired_only / non_discordant / all_reads, priority reads)
//   filter_scaff2pair2info / evaluate_pair        filter_reads.py:201-260, 388-426
//   samfile.pileup(..., stepper='nofilter', ignore_overlaps=True, min_base_quality=30, ...)
//                                                 profile_utilities.py:150-153
//       = htslib 1.9 bam_plp: default flag mask UNMAP|SECONDARY|QCFAIL|DUP, overlap_push /
//         tweak_overlap_quality / cigar_iref2iseq_set/next (sam.c), and pysam's
//         `qual >= min_base_quality` test when listing PileupColumn.pileups
//   iterate_splits                                /root/reference/inStrain/profile/fasta.py:56-73
//
// Two passes over the file, like the reference (filter_reads scans the BAM, then profile piles it up), none of
// which keeps the file's reads in memory:
//   scan    (isx_bam_scan)        every BGZF block is inflated once, in waves of segments that bound the memory in
//                                 flight; per read only name, flag, NM, mapq and the reference span are kept
//                                 (~70 bytes a read, freed when the pair tables are built); per (scaffold, name) the
//                                 pair record of get_paired_reads; per read the index of its pair (4 bytes).
//   filter  (isx_bam_filter)      median insert over the whole file, evaluate_pair per pair -- or the controller's
//                                 own R2M (isx_bam_set_r2m).
//   expand  (isx_bam_expand_refs) for a SUBSET of the references (one batch, one GPU's shard): only the segments
//                                 holding them are inflated again; overlap resolution and expansion run on that
//                                 batch's reads alone.
// No device code here; it is linked into libinstrain_amd.so so that the whole path sits behind one C ABI.
#include <immintrin.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "fast_inflate.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <queue>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/instrain_amd.h"
#include "obs_encode.h"

void isx_set_error(const std::string &msg);

namespace {

enum { CM = 0, CI, CD, CN, CS, CH, CP, CEQ, CX };
constexpr uint16_t FPROPER = 0x2, FUNMAP = 0x4, FMUNMAP = 0x8, FSECONDARY = 0x100, FQCFAIL = 0x200, FDUP = 0x400;
constexpr uint16_t DEF_MASK = FUNMAP | FSECONDARY | FQCFAIL | FDUP;

// 4-bit BAM code -> inStrain base index (A,C,T,G = 0..3; everything else 4)
const uint8_t CODE2IDX[16] = {4, 0, 1, 4, 3, 4, 4, 4, 2, 4, 4, 4, 4, 4, 4, 4};

// 4-bit BAM code -> read-segment code (include/instrain_amd.h isx_segs): A,C,T,G = 0..3; anything else 5 (a base that only
// makes its mm level present, profile_utilities.py:279-285)
const uint8_t CODE2SEG[16] = {5, 0, 1, 5, 3, 5, 5, 5, 2, 5, 5, 5, 5, 5, 5, 5};

// codes of `n` (<= 150) consecutive query bases starting at query offset q0 -> out[n] (code 4 where the quality is below minq)
inline void seg_codes_scalar(const uint8_t *seq, const uint8_t *qual, int64_t q0, int n, uint8_t minq, uint8_t *out)
{
    for (int j = 0; j < n; j++) {
        const int64_t i = q0 + j;
        out[j] = qual[i] >= minq ? CODE2SEG[(seq[i >> 1] >> ((~i & 1) << 2)) & 15] : (uint8_t)4;
    }
}

__attribute__((target("avx2")))
inline void seg_codes_avx2(const uint8_t *seq, const uint8_t *qual, int64_t q0, int n, uint8_t minq, uint8_t *out)
{
    // local, padded copies: the vector loads below may run up to 31 bytes past the bases asked for
    alignas(32) uint8_t sq[96], ql[192], cd[192];
    const int64_t e0 = q0 & ~(int64_t)1;                    // even base the copy starts at
    const int lead = (int)(q0 - e0), m = n + lead;
    memcpy(sq, seq + (e0 >> 1), (size_t)((m + 1) >> 1));
    memcpy(ql, qual + e0, (size_t)m);
    const __m128i nmask = _mm_set1_epi8(0x0F);
    const __m256i lut = _mm256_broadcastsi128_si256(_mm_loadu_si128(reinterpret_cast<const __m128i *>(CODE2SEG)));
    const __m256i mq = _mm256_set1_epi8((char)minq), four = _mm256_set1_epi8(4);
    for (int j = 0; j < m; j += 32) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(sq + (j >> 1)));
        const __m128i hi = _mm_and_si128(_mm_srli_epi16(v, 4), nmask), lo = _mm_and_si128(v, nmask);
        const __m256i nib = _mm256_set_m128i(_mm_unpackhi_epi8(hi, lo), _mm_unpacklo_epi8(hi, lo));     // base order
        const __m256i code = _mm256_shuffle_epi8(lut, nib);
        const __m256i q = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(ql + j));
        const __m256i ok = _mm256_cmpeq_epi8(_mm256_max_epu8(q, mq), q);                                // q >= minq, unsigned
        _mm256_store_si256(reinterpret_cast<__m256i *>(cd + j), _mm256_blendv_epi8(four, code, ok));
    }
    memcpy(out, cd + lead, (size_t)n);
}

inline bool cpu_has_avx2_bmi2()
{
    static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    return v;
}

// ten codes per word (base j: word j / 10, bits 3 (j % 10)); slots from n on hold code 4
inline void seg_pack_scalar(const uint8_t *cd, int n, uint32_t *w)
{
    for (int k = 0; k < ISX_SEG_WORDS; k++) w[k] = ISX_SEG_SKIP_WORD;
    for (int j = 0; j < n; j++) w[j / 10] = (w[j / 10] & ~(7u << (3 * (j % 10)))) | ((uint32_t)cd[j] << (3 * (j % 10)));
}

__attribute__((target("bmi2")))
inline void seg_pack_bmi2(uint8_t *cd /* [160], writable: padded with code 4 */, int n, uint32_t *w)
{
    memset(cd + n, 4, (size_t)(160 - n));
    for (int k = 0; k < ISX_SEG_WORDS; k++) {
        uint64_t x;
        memcpy(&x, cd + 10 * k, 8);
        w[k] = (uint32_t)_pext_u64(x, 0x0707070707070707ull) | ((uint32_t)cd[10 * k + 8] << 24) | ((uint32_t)cd[10 * k + 9] << 27);
    }
}

// ---- the same segment as BIT PLANES (include/instrain_amd.h isx_read_planes): words 0-4 the 2-bit codes, words 5-7 the columns
// that are not observed (quality below minq, or a base that is not A/C/T/G) ----
inline uint64_t even_bits_of(uint64_t x)         // bits 0, 2, 4 ... 62 -> bits 0 .. 31
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

// mark_n (mm profiling on, isx_read_planes.mm): a base that is not A/C/T/G but passes the quality filter gets code 1 at its (not observed)
// column and sets the line's marker flag, bit 63 of word 7; every other column that is not observed has code 0
inline void seg_planes_scalar(const uint8_t *seq, const uint8_t *qual, int64_t q0, int n, uint8_t minq, uint64_t *P, bool mark_n = false)
{
    uint64_t b[5] = {0, 0, 0, 0, 0}, sk[3] = {0, 0, 0};
    for (int j = 0; j < n; j++) {
        const int64_t i = q0 + j;
        const uint32_t c = CODE2IDX[(seq[i >> 1] >> ((~i & 1) << 2)) & 15];
        if (c > 3 || qual[i] < minq) {
            sk[j >> 6] |= (uint64_t)1 << (j & 63);
            if (mark_n && c > 3 && qual[i] >= minq) { b[j >> 5] |= (uint64_t)1 << (2 * (j & 31)); sk[2] |= (uint64_t)1 << 63; }
        } else b[j >> 5] |= (uint64_t)c << (2 * (j & 31));
    }
    for (int k = 0; k < 5; k++) P[k] = b[k];
    for (int k = 0; k < 3; k++) P[5 + k] = sk[k];
}

inline bool cpu_has_avx512bw()
{
    static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") && !getenv("ISX_NO_AVX512");
    return v;
}

// 64 bases a step straight from the record's 4-bit seq and its qualities (masked loads: nothing beyond the read is touched):
// nibbles into base order, a 16-entry look-up gives the 2-bit code or flags "not a base", the quality compare gives the observed
// columns as a mask register; four codes are folded into a byte by two shift-or steps and a narrowing move.  The conversion starts
// at the even base below q0; a leading odd base is shifted out at the end.
__attribute__((target("avx512f,avx512bw,avx512vl,bmi2")))
inline void seg_planes_avx512(const uint8_t *seq, const uint8_t *qual, int64_t q0, int n, uint8_t minq, uint64_t *P, bool mark_n = false)
{
    const int64_t e0 = q0 & ~(int64_t)1;
    const int lead = (int)(q0 - e0), m = n + lead;                                 // m <= 151 columns from the even base e0
    const uint8_t *sq = seq + (e0 >> 1), *ql = qual + e0;
    const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8((char)0x80, 0, 1, (char)0x80, 3, (char)0x80, (char)0x80, (char)0x80, 2, (char)0x80, (char)0x80,
                                                            (char)0x80, (char)0x80, (char)0x80, (char)0x80, (char)0x80));
    const __m512i mq = _mm512_set1_epi8((char)minq);
    alignas(64) uint64_t b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t sk[3];
    uint64_t any_n = 0;
    for (int k = 0; k < 3; k++) {
        const int have = m - 64 * k;                                               // columns of this step
        if (have <= 0) { sk[k] = 0; continue; }
        const int nb = have >= 64 ? 64 : have;
        const __mmask64 cm = nb == 64 ? ~(__mmask64)0 : (((__mmask64)1 << nb) - 1);
        const __mmask32 bm = (__mmask32)((((uint64_t)1 << ((nb + 1) >> 1)) - 1));          // (nb 63 / 64: all 32 bytes)
        const __m512i w = _mm512_cvtepu8_epi16(_mm256_maskz_loadu_epi8(bm, sq + 32 * k));
        // word = byte: low byte <- high nibble (the even base), high byte <- low nibble
        const __m512i nib = _mm512_or_si512(_mm512_srli_epi16(w, 4), _mm512_slli_epi16(_mm512_and_si512(w, _mm512_set1_epi16(0x0F)), 8));
        const __m512i code = _mm512_shuffle_epi8(lut, nib);
        const __m512i q = _mm512_maskz_loadu_epi8(cm, ql + 64 * k);
        const __mmask64 qok = _mm512_mask_cmpge_epu8_mask(cm, q, mq);
        const __mmask64 ok = qok & ~_mm512_movepi8_mask(code);
        sk[k] = (uint64_t)(cm & ~ok);
        // four 2-bit codes a byte: c0 | c1 << 2 within 16 bits, then the two nibbles of a dword, then dword -> byte
        __m512i c2 = _mm512_and_si512(code, _mm512_set1_epi8(3));
        if (mark_n) {                               // codes only at observed columns, 1 at the marked ones
            const __mmask64 nm = qok & _mm512_movepi8_mask(code);
            c2 = _mm512_mask_mov_epi8(_mm512_maskz_mov_epi8(ok, c2), nm, _mm512_set1_epi8(1));
            any_n |= (uint64_t)nm;
        }
        const __m512i t = _mm512_and_si512(_mm512_or_si512(c2, _mm512_srli_epi16(c2, 6)), _mm512_set1_epi16(0x000F));
        const __m512i u = _mm512_or_si512(t, _mm512_srli_epi32(t, 12));
        _mm_store_si128(reinterpret_cast<__m128i *>(b + 2 * k), _mm512_cvtepi32_epi8(u));
    }
    if (lead) {                                                                     // drop the odd leading column
        for (int k = 0; k < 5; k++) b[k] = (b[k] >> 2) | (b[k + 1] << 62);
        sk[0] = (sk[0] >> 1) | (sk[1] << 63); sk[1] = (sk[1] >> 1) | (sk[2] << 63); sk[2] >>= 1;
    }
    if (lead && any_n) {                            // (a marked column that was the dropped leading one does not count)
        const uint64_t m0 = (even_bits_of(b[0]) | (even_bits_of(b[1]) << 32)) & sk[0], m1 = (even_bits_of(b[2]) | (even_bits_of(b[3]) << 32)) & sk[1], m2 = even_bits_of(b[4]) & sk[2];
        any_n = m0 | m1 | m2;
    }
    for (int k = 0; k < 5; k++) P[k] = b[k];
    P[5] = sk[0]; P[6] = sk[1]; P[7] = sk[2] | (any_n ? (uint64_t)1 << 63 : 0);
}

struct Read {           // a read of the batch being expanded
    int32_t tid, pos, isize, l_seq;
    uint16_t flag;
    uint32_t n_cigar;           // real operation count (a CIGAR kept in the CG tag has more than 65535)
    uint32_t pair_idx;          // index into isx_bam::pairs, 0xFFFFFFFF = not in any table
    uint64_t cigar_off;
    const uint8_t *seq;         // 4-bit codes, two per byte, where the record lies in its inflated segment (never copied)
    uint8_t *qual;              // its qualities, same place: overlap resolution rewrites them there
    int64_t ref_end;            // reference position after the last CIGAR op (bam_endpos)
};

inline uint8_t nib(const uint8_t *seq, int64_t i) { return (uint8_t)((seq[i >> 1] >> ((~i & 1) << 2)) & 15); }

struct Cursor {     // htslib sam.c cigar_iref2iseq_* state
    const uint32_t *cig; int n; int k, icig, iseq, iref;
};

int cur_set(Cursor &c, int pos)
{
    if (pos < 0) return -1;
    c.k = 0; c.icig = 0; c.iseq = 0; c.iref = 0;
    while (c.k < c.n) {
        const int op = c.cig[c.k] & 15, n = (int)(c.cig[c.k] >> 4);
        if (op == CS) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CH || op == CP) { c.k++; c.icig = 0; continue; }
        if (op == CM || op == CEQ || op == CX) {
            pos -= n;
            if (pos < 0) { c.icig = n + pos; c.iseq += c.icig; c.iref += c.icig; return 0; }
            c.k++; c.iseq += n; c.icig = 0; c.iref += n;
            continue;
        }
        if (op == CI) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CD || op == CN) {
            pos -= n;
            if (pos < 0) pos = 0;
            c.k++; c.icig = 0; c.iref += n;
            continue;
        }
        return -2;
    }
    c.iseq = -1;
    return -1;
}

int cur_next(Cursor &c)
{
    while (c.k < c.n) {
        const int op = c.cig[c.k] & 15, n = (int)(c.cig[c.k] >> 4);
        if (op == CM || op == CEQ || op == CX) {
            if (c.icig >= n - 1) { c.icig = 0; c.k++; continue; }
            c.iseq++; c.icig++; c.iref++;
            return 0;
        }
        if (op == CD || op == CN) { c.k++; c.iref += n; c.icig = 0; continue; }
        if (op == CI || op == CS) { c.k++; c.iseq += n; c.icig = 0; continue; }
        if (op == CH || op == CP) { c.k++; c.icig = 0; continue; }
        return -2;
    }
    c.iseq = -1; c.iref = -1;
    return -1;
}

struct PairInfo {       // filter_reads.py i2o order (+ what the filter decided)
    int64_t nm, insert, mapq, length, reads, start, stop;
    uint32_t name_seg, name_off;    // where the name lives (segment blob) -- valid until the scan data is dropped
    uint16_t name_len;
    bool pass, in_filter;           // in_filter: survived paired_read_filter (evaluated at all)
    uint8_t pad;
    int32_t mm;                     // R2M value when pass (nm, or the controller's)
};

// what the scan keeps of a read until the pair tables are built
struct ReadLite {
    int32_t tid, pos, isize, l_seq, nm, qlen;
    int64_t first, last;            // first / last aligned reference position
    uint64_t h64;                   // hash of the name
    uint32_t name_off;
    uint16_t name_len, flag;
    uint8_t mapq, has_nm, any, pad;
};

struct Block { uint64_t coff; uint32_t csize, hdr, isize; uint64_t ioff; };

struct Segment {
    uint32_t b0 = 0, b1 = 0;        // blocks [b0, b1)
    uint64_t ioff0 = 0, ioff1 = 0;  // inflated byte range
    uint64_t first_rec = 0;         // inflated offset of the first record that STARTS in the segment (== ioff1: none)
    uint64_t read0 = 0;             // ordinal of that record
    uint32_t n_reads = 0;
    int32_t tid_first = -1, pos_first = 0, tid_last = -1, pos_last = 0;   // first / last record (set by the scan)
};

// ---- inflate: libdeflate when the image has it (2-3x zlib), zlib otherwise ----
struct Deflate {
    void *lib = nullptr;
    void *(*alloc)() = nullptr;
    int (*decomp)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    Deflate()
    {
        for (const char *n : {"libdeflate.so.0", "libdeflate.so"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
        alloc = reinterpret_cast<void *(*)()>(dlsym(lib, "libdeflate_alloc_decompressor"));
        decomp = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(lib, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<void (*)(void *)>(dlsym(lib, "libdeflate_free_decompressor"));
        if (!alloc || !decomp || !release) { alloc = nullptr; decomp = nullptr; release = nullptr; }
    }
};
const Deflate &deflate_lib() { static Deflate d; return d; }

// ---- inflate: libdeflate when the image has it, else the table-driven decoder of fast_inflate.h (2x zlib's inflate on BAM blocks;
// ISX_BAM_ZLIB=1: zlib only), zlib for whatever that one does not decode ----
struct Inflater {       // one per thread
    void *ld = nullptr;
    std::unique_ptr<isxinf::FastInflater> fast;
    Inflater()
    {
        if (deflate_lib().alloc) ld = deflate_lib().alloc();
        static const bool zlib_only = getenv("ISX_BAM_ZLIB") != nullptr;
        if (!ld && !zlib_only) fast.reset(new isxinf::FastInflater());
    }
    ~Inflater() { if (ld) deflate_lib().release(ld); }
    bool run(const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst)
    {
        if (!n_dst) return true;
        if (ld) {
            size_t got = 0;
            return deflate_lib().decomp(ld, src, n_src, dst, n_dst, &got) == 0 && got == n_dst;
        }
        if (fast && fast->run(src, n_src, dst, n_dst)) return true;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = (uInt)n_src;
        zs.next_out = dst; zs.avail_out = (uInt)n_dst;
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = (rc == Z_STREAM_END) && zs.total_out == n_dst;
        inflateEnd(&zs);
        return ok;
    }
};

inline int32_t rd32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

// aux fields of a record: the NM tag; every read is checked against `end`
int parse_nm(const uint8_t *p, const uint8_t *end, bool &has, int32_t &nm)
{
    has = false;
    while (p + 3 <= end) {
        const bool is_nm = (p[0] == 'N' && p[1] == 'M');
        const char t = (char)p[2];
        p += 3;
        int64_t v = 0;
        bool num = true;
        size_t need = 0;
        switch (t) {
        case 'A': case 'c': case 'C': need = 1; break;
        case 's': case 'S': need = 2; break;
        case 'i': case 'I': case 'f': need = 4; break;
        case 'Z': case 'H': case 'B': break;
        default: return -1;
        }
        if ((size_t)(end - p) < need) return -1;
        switch (t) {
        case 'A': v = *p; p += 1; num = false; break;
        case 'c': v = (int8_t)*p; p += 1; break;
        case 'C': v = *p; p += 1; break;
        case 's': { int16_t x; memcpy(&x, p, 2); v = x; p += 2; break; }
        case 'S': { uint16_t x; memcpy(&x, p, 2); v = x; p += 2; break; }
        case 'i': { int32_t x; memcpy(&x, p, 4); v = x; p += 4; break; }
        case 'I': { uint32_t x; memcpy(&x, p, 4); v = x; p += 4; break; }
        case 'f': p += 4; num = false; break;
        case 'Z': case 'H':
            while (p < end && *p) p++;
            if (p >= end) return -1;
            p++; num = false; break;
        case 'B': {
            if (end - p < 5) return -1;
            const char sub = (char)p[0];
            const int32_t cnt = rd32(p + 1);
            int sz;
            switch (sub) {
            case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            default: return -1;
            }
            if (cnt < 0 || (uint64_t)cnt * sz > (uint64_t)(end - p - 5)) return -1;
            p += 5 + (size_t)cnt * sz;
            num = false;
            break;
        }
        }
        if (is_nm && num) { has = true; nm = (int32_t)v; }
    }
    return p == end ? 0 : -1;
}

// the CG:B,I tag of a record: 1 and the operations when present, 0 when absent, -1 on a malformed aux block
int find_long_cigar(const uint8_t *p, const uint8_t *end, const uint8_t *&ops, uint32_t &n_ops)
{
    while (p + 3 <= end) {
        const bool is_cg = (p[0] == 'C' && p[1] == 'G');
        const char t = (char)p[2];
        p += 3;
        size_t step = 0;
        switch (t) {
        case 'A': case 'c': case 'C': step = 1; break;
        case 's': case 'S': step = 2; break;
        case 'i': case 'I': case 'f': step = 4; break;
        case 'Z': case 'H': {
            const uint8_t *q = p;
            while (q < end && *q) q++;
            if (q >= end) return -1;
            step = (size_t)(q - p) + 1;
            break;
        }
        case 'B': {
            if (end - p < 5) return -1;
            const char sub = (char)p[0];
            const int32_t cnt = rd32(p + 1);
            const int sz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!sz || cnt < 0 || (uint64_t)cnt * sz > (uint64_t)(end - p - 5)) return -1;
            if (is_cg && sub == 'I') { ops = p + 5; n_ops = (uint32_t)cnt; return 1; }
            step = 5 + (size_t)cnt * sz;
            break;
        }
        default: return -1;
        }
        if ((size_t)(end - p) < step) return -1;
        p += step;
    }
    return p == end ? 0 : -1;
}

uint64_t hash_name(const uint8_t *s, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9E3779B97F4A7C15ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, s + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
    uint64_t w = 0;
    memcpy(&w, s + i, n - i);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return h;
}

}  // namespace

// The front end's large arrays (inflated segments, per-read tables: hundreds of MB to GB per file) are written once by
// many threads right after they are allocated.  From malloc that is one page fault per 4 KiB under the process's mm lock --
// with 32 threads faulting at once the lock, not the memory, sets the pace -- and as many PTEs to tear down on free.
// Blocks of 4 MiB and more are therefore mapped 2 MiB-aligned and advised MADV_HUGEPAGE (transparent huge pages in
// "madvise" or "always" mode: one fault per 2 MiB; elsewhere the advice is a no-op).
constexpr size_t BIG_BLOCK = (size_t)4 << 20, HUGE_PAGE = (size_t)2 << 20;
inline size_t big_len(size_t bytes) { return (bytes + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE; }

void *block_alloc(size_t bytes)
{
    if (bytes < BIG_BLOCK) { void *p = malloc(std::max<size_t>(bytes, 1)); if (!p) throw std::bad_alloc(); return p; }
    const size_t len = big_len(bytes);
    uint8_t *m = static_cast<uint8_t *>(mmap(nullptr, len + HUGE_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (m == MAP_FAILED) throw std::bad_alloc();
    uint8_t *a = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(m) + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE);
    if (a > m) (void)munmap(m, (size_t)(a - m));
    const size_t tail = (size_t)(m + len + HUGE_PAGE - (a + len));
    if (tail) (void)munmap(a + len, tail);
    (void)madvise(a, len, MADV_HUGEPAGE);
    return a;
}

void block_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes < BIG_BLOCK) free(p);
    else (void)munmap(p, big_len(bytes));
}

template <class T>
struct RawBuf {                 // sized once, written once: no value initialisation (vector::resize would memset)
    T *p = nullptr;
    size_t n = 0;
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() { block_free(p, n * sizeof(T)); }
    void resize(size_t m) { block_free(p, n * sizeof(T)); p = nullptr; n = 0; p = static_cast<T *>(block_alloc(m * sizeof(T))); n = m; }
    T *data() { return p; }
    const T *data() const { return p; }
    const T &operator[](size_t i) const { return p[i]; }
    T &operator[](size_t i) { return p[i]; }
    size_t size() const { return n; }
};

template <class T>
struct NoInitAlloc {            // vector::resize without the memset: the elements are written right after, in parallel
    using value_type = T;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) noexcept {}
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    T *allocate(size_t n) { return static_cast<T *>(block_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) noexcept { block_free(p, n * sizeof(T)); }
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const NoInitAlloc<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U> &) const noexcept { return false; }
};

template <class T> using uvec = std::vector<T, NoInitAlloc<T>>;

// v = n copies of x, written (and first touched) by the pool's threads
template <class T>
void par_assign(isxenc::HostPool &pool, uvec<T> &v, size_t n, T x)
{
    v.resize(n);
    const size_t piece = (size_t)1 << 20;
    const int n_tasks = (int)((n + piece - 1) / piece);
    T *d = v.data();
    auto body = [&](int t) { std::fill(d + (size_t)t * piece, d + std::min(n, ((size_t)t + 1) * piece), x); };
    if (n_tasks > 1) pool.run(n_tasks, body); else if (n_tasks == 1) body(0);
}

// stable bucketing of the indices [0, n) by key(i) in [0, P): order lists bucket 0's indices ascending, then bucket 1's, ...
template <class Key>
void bucket_indices(isxenc::HostPool &pool, size_t n, int P, Key key, std::vector<uint32_t> &order, std::vector<size_t> &start)
{
    const int C = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n / 65536 + 1));
    std::vector<size_t> cnt((size_t)C * (size_t)P, 0);
    pool.run(C, [&](int c) {
        size_t *k = cnt.data() + (size_t)c * (size_t)P;
        for (size_t i = n * (size_t)c / (size_t)C; i < n * ((size_t)c + 1) / (size_t)C; i++) { const int b = key(i); if (b >= 0) k[b]++; }
    });
    start.assign((size_t)P + 1, 0);
    size_t at = 0;
    for (int b = 0; b < P; b++) {
        start[(size_t)b] = at;
        for (int c = 0; c < C; c++) { size_t &k = cnt[(size_t)c * (size_t)P + (size_t)b]; const size_t m = k; k = at; at += m; }
    }
    start[(size_t)P] = at;
    order.resize(at);
    pool.run(C, [&](int c) {
        size_t *k = cnt.data() + (size_t)c * (size_t)P;
        for (size_t i = n * (size_t)c / (size_t)C; i < n * ((size_t)c + 1) / (size_t)C; i++) { const int b = key(i); if (b >= 0) order[k[b]++] = (uint32_t)i; }
    });
}

struct BamBatch;

struct isx_bam {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t map_len = 0;
    std::vector<Block> blocks;
    std::vector<Segment> segs;
    uint64_t total_inflated = 0, first_rec = 0;
    std::vector<std::string> ref_name;
    std::vector<int64_t> ref_len, ref_off;
    int threads = 0;
    std::unique_ptr<isxenc::HostPool> pool;
    // ---- scan products ----
    bool scanned = false, filtered = false;
    int32_t part = 0, n_parts = 1;                  // isx_bam_scan_part: which share of the file this handle scanned
    uint64_t n_reads = 0;
    uvec<uint32_t> read_pair;                       // per read ordinal
    std::vector<PairInfo, NoInitAlloc<PairInfo>> pairs;
    std::vector<PairInfo, NoInitAlloc<PairInfo>> pairs_scan;               // what the scan found, kept once all_reads has rewritten entries (_merge_info)
    std::vector<uint64_t> ref_pair0;                // [n_ref + 1] pairs of a reference are contiguous
    std::vector<uint32_t> ref_seg0, ref_seg1;       // segments holding records of the reference: [seg0, seg1]
    std::vector<uvec<char>> seg_names;              // name blobs, dropped after the filter (or kept for set_r2m)
    std::vector<uvec<ReadLite>> seg_reads;          // dropped after the pair tables are built
    std::vector<uint8_t> priority;                  // per pair: its name is a priority read
    std::vector<std::string> priority_names;
    isx_bam_info totals{};
    int64_t max_span = 0;                           // longest reference span of a read (region queries: how far back a read may start)
    // small files: the inflated segments and their record offsets stay (pass 2 neither inflates nor walks again)
    std::vector<uvec<uint8_t>> seg_cache;
    std::vector<std::vector<uint64_t>> seg_cache_rec;
    std::vector<int64_t> ref_filtered_pairs, ref_reads;
    // ---- results of the last expand ----
    std::unique_ptr<isx_obs[]> obs;
    std::unique_ptr<uint32_t[]> pair;
    std::vector<uint8_t> ref_wanted;                        // isx_bam_set_wanted_refs: empty = every reference counts
    // isx_bam_set_cross_names: what the other scaffolds of the FILE (other shares included) hold under the names of this
    // handle's pair entries -- the one thing non_discordant / all_reads need from beyond a share
    bool cross_set = false;
    std::vector<int64_t> cross_idx, cross_occ, cross_info;  // entry, occurrences file-wide, merged (nm, mapq, length, reads)
    std::vector<uint32_t> seg_gpos, seg_pair, seg_bases;    // isx_bam_segment_refs: the batch as read segments
    std::vector<uint64_t> seg_planes;                       // ... and as bit planes
    std::vector<uint8_t> seg_len, seg_mm;
    size_t n_obs = 0;
    std::vector<int64_t> split_bounds;
    std::vector<int32_t> split_ref;
    bool expanded = false;

    // batches a pipe is done with (bam_batch_retire): giving a gigabyte back to the system takes 100+ ms of the process' address-space
    // lock -- page faults and device calls of every other thread wait for it -- so it happens when the handle goes (isx_bam_close
    // does that on a thread of its own), or earlier only when more than RETIRE_LIMIT bytes have piled up
    // the scan's per-read records and the name blobs (dropped after the pair tables / the filter) wait here too when they are small
    // enough to keep (<= KEEP_DEAD bytes each): freed with the handle, not in the middle of the caller's run
    std::vector<uvec<ReadLite>> dead_reads;
    std::vector<uvec<char>> dead_names;
    static constexpr size_t KEEP_DEAD = (size_t)2 << 30;
    int32_t mm_cap = 0x7FFFFFFF;            // isx_bam_set_mm_cap: pairs with more mismatches are piled up at this level
    // isx_bam_set_mm_levels: the mm VALUES that occur among the kept pairs, ascending; a pair then travels with the RANK of its value
    // (the device only needs the levels' order: counts are cumulated over the levels <= mm, profile_utilities.py:297-312; the caller
    // maps ranks back to values).  Empty: a pair travels with its mm itself.
    std::vector<int32_t> mm_levels;
    int32_t level_of(int32_t mm) const
    {
        if (mm_levels.empty()) return std::min<int32_t>(mm, mm_cap);
        const int32_t k = (int32_t)(std::lower_bound(mm_levels.begin(), mm_levels.end(), mm) - mm_levels.begin());
        return std::min<int32_t>(k, mm_cap);
    }
    std::vector<uint32_t> last_dense_pair;  // of the batch prepared last: dense pair id -> index into `pairs` (kept while the names are:
                                            // isx_bam_batch_pair_names, the read_to_snvs keys of --store_everything)
    std::mutex retire_mu;
    std::condition_variable retire_cv;
    int batches_out = 0;            // batches bam_batch_prepare handed out that were neither retired nor freed yet (a pipe's finisher
                                    // may still hold them): isx_bam_close waits for them (closing the handle under an
                                    // uncollected isx_pipe_submit_bam ticket was a use-after-free)
    std::vector<std::pair<BamBatch *, size_t>> retired;
    size_t retired_bytes = 0;
    static constexpr size_t RETIRE_LIMIT = (size_t)6 << 30;

    ~isx_bam();
};

namespace {

int n_threads_default()
{
    int n = (int)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {           // a container's cpu quota, when there is one
        char q[64] = {0};
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) n = std::min<int>(n, (int)std::max<long>(1, 2 * atol(q) / period));
        fclose(f);
    }
    return std::max(1, std::min(n, 64));
}

isxenc::HostPool &pool_of(isx_bam &B)
{
    if (!B.pool) B.pool.reset(new isxenc::HostPool(B.threads > 0 ? B.threads : n_threads_default(), -1, false));
    return *B.pool;
}

// inflate blocks [b0, b1) into dst (dst[0] = inflated offset blocks[b0].ioff); false on a corrupt block
bool inflate_range(const isx_bam &B, Inflater &inf, uint32_t b0, uint32_t b1, uint8_t *dst)
{
    const uint64_t base = B.blocks[b0].ioff;
    for (uint32_t b = b0; b < b1; b++) {
        const Block &k = B.blocks[b];
        if (!inf.run(B.map + k.coff + k.hdr, k.csize - k.hdr - 8, dst + (k.ioff - base), k.isize)) return false;
    }
    return true;
}

// A segment's inflated bytes plus as much of the following blocks as its last record needs.
struct SegBuf {
    uvec<uint8_t> data;             // data[0] = inflated offset seg.ioff0 (no memset before the inflater writes it)
    uint32_t b_end = 0;             // blocks inflated so far: [seg.b0, b_end)
};

// how many inflated bytes a handle may keep between its two passes: a quarter of the memory that is available to this
// process (MemAvailable, the cgroup's limit when there is one), at most 16 GiB, at least 1 GiB
uint64_t inflate_cache_limit()
{
    uint64_t avail = (uint64_t)4 << 30;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            unsigned long long kb = 0;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (uint64_t)kb << 10; break; }
        }
        fclose(f);
    }
    if (FILE *f = fopen("/sys/fs/cgroup/memory.max", "r")) {
        unsigned long long mx = 0, cur = 0;
        if (fscanf(f, "%llu", &mx) == 1) {
            if (FILE *g = fopen("/sys/fs/cgroup/memory.current", "r")) { if (fscanf(g, "%llu", &cur) != 1) cur = 0; fclose(g); }
            if (mx > cur) avail = std::min<uint64_t>(avail, mx - cur);
        }
        fclose(f);
    }
    return std::max<uint64_t>((uint64_t)1 << 30, std::min<uint64_t>(avail / 4, (uint64_t)16 << 30));
}

bool seg_inflate(const isx_bam &B, Inflater &inf, const Segment &s, SegBuf &out)
{
    out.data.resize((size_t)(s.ioff1 - s.ioff0));
    out.b_end = s.b1;
    return inflate_range(B, inf, s.b0, s.b1, out.data.data());
}

// make sure [off, off + n) of the segment's inflated stream is there (a record may run into the next blocks)
bool seg_need(const isx_bam &B, Inflater &inf, const Segment &s, SegBuf &buf, uint64_t off, uint64_t n)
{
    while (off + n > s.ioff0 + buf.data.size()) {
        if (buf.b_end >= B.blocks.size()) return false;
        const Block &k = B.blocks[buf.b_end];
        const size_t old = buf.data.size();
        buf.data.resize(old + k.isize);
        if (!inf.run(B.map + k.coff + k.hdr, k.csize - k.hdr - 8, buf.data.data() + old, k.isize)) return false;
        buf.b_end++;
    }
    return true;
}

// record starts of a segment: walks the block_size fields from `first`; returns the offset after the last
// record that starts before s.ioff1 (= first record of the next segment).  rec_off gets inflated offsets.
int seg_hop(const isx_bam &B, Inflater &inf, const Segment &s, SegBuf &buf, uint64_t first, std::vector<uint64_t> &rec_off,
            uint64_t &next_first)
{
    uint64_t off = first;
    rec_off.clear();
    while (off < s.ioff1) {
        if (off + 4 > B.total_inflated) { isx_set_error("truncated BAM record"); return ISX_ERR_IO; }
        if (!seg_need(B, inf, s, buf, off, 4)) { isx_set_error("BGZF inflate failed"); return ISX_ERR_IO; }
        const int32_t block = rd32(buf.data.data() + (off - s.ioff0));
        if (block < 32 || off + 4 + (uint64_t)block > B.total_inflated) { isx_set_error("truncated BAM record"); return ISX_ERR_IO; }
        if (!seg_need(B, inf, s, buf, off, 4 + (uint64_t)block)) { isx_set_error("BGZF inflate failed"); return ISX_ERR_IO; }
        rec_off.push_back(off);
        off += 4 + (uint64_t)block;
    }
    next_first = off;
    return ISX_OK;
}

// Could a record start at inflated offset `off`?  (fixed fields in range, sizes consistent, a printable NUL-terminated
// name, CIGAR operators < 9)  Used to find the first record of a segment without walking the file up to it.
bool plausible_record(const isx_bam &B, const uint8_t *p, uint64_t avail, uint64_t &len)
{
    if (avail < 36) return false;
    const int32_t bs = rd32(p);
    if (bs < 32 || bs > (1 << 26)) return false;
    const int32_t tid = rd32(p + 4), pos = rd32(p + 8), l_seq = rd32(p + 20), mtid = rd32(p + 24), mpos = rd32(p + 28);
    const int n_ref = (int)B.ref_name.size();
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || pos < -1 || mpos < -1 || l_seq < 0) return false;
    if (tid >= 0 && pos > B.ref_len[(size_t)tid]) return false;
    const uint32_t l_name = p[12], n_cig = rd16(p + 16);
    if (l_name == 0) return false;
    const uint64_t need = 32 + (uint64_t)l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > (uint64_t)bs) return false;
    if (avail >= 36 + (uint64_t)l_name) {
        if (p[36 + l_name - 1] != 0) return false;
        for (uint32_t i = 0; i + 1 < l_name; i++) if (p[36 + i] < 33 || p[36 + i] > 126) return false;
    }
    if (avail >= 36 + (uint64_t)l_name + 4ull * n_cig)
        for (uint32_t k = 0; k < n_cig; k++) if ((p[36 + l_name + 4 * k] & 15) > 8) return false;
    len = 4 + (uint64_t)bs;
    return true;
}

// first offset >= s.ioff0 at which a chain of plausible records starts; ~0 when the segment seems to hold none.
// The caller verifies the guess against the walk of the previous segment (and re-walks on a mismatch).
uint64_t seg_guess(const isx_bam &B, Inflater &inf, const Segment &s, SegBuf &buf)
{
    const int CHAIN = 4;
    for (uint64_t off = s.ioff0; off < s.ioff1; off++) {
        uint64_t o = off;
        int ok = 0;
        for (; ok < CHAIN; ok++) {
            if (o >= B.total_inflated) break;                       // the chain ran off the end of the file: fine
            if (o + 36 > B.total_inflated) { ok = -1; break; }
            if (!seg_need(B, inf, s, buf, o, 36)) { ok = -1; break; }
            const uint64_t have = s.ioff0 + buf.data.size() - o;
            uint64_t len = 0;
            if (!plausible_record(B, buf.data.data() + (o - s.ioff0), std::min<uint64_t>(have, 36 + 256), len)) { ok = -1; break; }
            if (have < 36 + 256 && o + 36 + 256 <= B.total_inflated) {    // name not fully in view yet: bring it in and look again
                if (!seg_need(B, inf, s, buf, o, 36 + 256)) { ok = -1; break; }
                if (!plausible_record(B, buf.data.data() + (o - s.ioff0), 36 + 256, len)) { ok = -1; break; }
            }
            if (o + len > B.total_inflated) { ok = -1; break; }
            o += len;
        }
        if (ok >= 0) return off;
    }
    return ~0ull;
}

struct RecView {        // validated fixed part of a record
    const uint8_t *p;   // at block_size
    int32_t block, tid, pos, l_seq, isize;
    uint32_t n_cigar;   // operations at `cigar`: the record's own, or the CG tag's when the record holds the placeholder
    uint16_t flag;
    uint8_t l_name, mapq;
    const uint8_t *name, *cigar, *seq, *qual, *aux, *end;
};

int find_long_cigar(const uint8_t *p, const uint8_t *end, const uint8_t *&ops, uint32_t &n_ops);

bool rec_view(const uint8_t *p, RecView &r)
{
    r.p = p;
    r.block = rd32(p);
    r.tid = rd32(p + 4); r.pos = rd32(p + 8);
    r.l_name = p[12]; r.mapq = p[13];
    r.n_cigar = rd16(p + 16); r.flag = rd16(p + 18);
    r.l_seq = rd32(p + 20);
    r.isize = rd32(p + 32);
    if (r.l_seq < 0 || r.l_name == 0) return false;
    const uint64_t need = 32 + (uint64_t)r.l_name + (uint64_t)r.n_cigar * 4 + ((uint64_t)r.l_seq + 1) / 2 + (uint64_t)r.l_seq;
    if (need > (uint64_t)r.block) return false;
    r.name = p + 36;
    r.cigar = r.name + r.l_name;
    r.seq = r.cigar + (size_t)r.n_cigar * 4;
    r.qual = r.seq + ((size_t)r.l_seq + 1) / 2;
    r.aux = r.qual + r.l_seq;
    r.end = p + 4 + r.block;
    // a CIGAR of more than 65535 operations: the record holds the placeholder <l_seq>S<ref_len>N and the operations are the
    // CG:B,I tag (SAM spec 4.2.2).  htslib moves them back when it reads the record (bam_tag2cigar, sam.c), so pysam -- and the
    // reference with it -- never sees the placeholder; nor does anything behind this view.
    if (r.n_cigar == 2 && r.tid >= 0) {
        uint32_t c0, c1;
        memcpy(&c0, r.cigar, 4); memcpy(&c1, r.cigar + 4, 4);
        if ((c0 & 15) == CS && (int32_t)(c0 >> 4) == r.l_seq && (c1 & 15) == CN) {
            const uint8_t *ops = nullptr;
            uint32_t n_ops = 0;
            const int rc = find_long_cigar(r.aux, r.end, ops, n_ops);
            if (rc < 0) return false;
            if (rc > 0) { r.cigar = ops; r.n_cigar = n_ops; }
        }
    }
    return true;
}

struct RefSpan { int64_t first, last, end; int64_t qlen; bool any; };

RefSpan span_of(const uint8_t *cigar, int n_cigar, int32_t pos)
{
    RefSpan s{0, 0, pos, 0, false};
    int64_t ref = pos;
    for (int k = 0; k < n_cigar; k++) {
        uint32_t c;
        memcpy(&c, cigar + 4 * (size_t)k, 4);
        const int op = c & 15;
        const int64_t n = c >> 4;
        if (op == CM || op == CEQ || op == CX) {
            if (n > 0) {
                if (!s.any) { s.first = ref; s.any = true; }
                s.last = ref + n - 1;
            }
            ref += n;
        } else if (op == CD || op == CN) ref += n;
        if (op == CM || op == CI || op == CS || op == CEQ || op == CX) s.qlen += n;
    }
    s.end = ref;
    return s;
}

int n_threads_default();

// One BGZF block header at `off`: 0 = not one (or it reaches beyond the file), else the block's size; hdr / isize filled in.
inline size_t bgzf_header_at(const uint8_t *map, size_t map_len, size_t off, uint32_t &hdr, uint32_t &isize)
{
    if (off + 18 > map_len) return 0;
    const uint8_t *h = map + off;
    if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return 0;
    const size_t xlen = h[10] | (h[11] << 8);
    if (off + 12 + xlen > map_len) return 0;
    size_t bsize = 0;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
        const size_t slen = h[x + 2] | (h[x + 3] << 8);
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = (size_t)(h[x + 4] | (h[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || off + bsize > map_len) return 0;
    const uint8_t *t = h + bsize - 4;
    isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    if (isize > 65536) return 0;
    hdr = (uint32_t)(12 + xlen);
    return bsize;
}

// The block index of the whole file: the chain of block sizes, walked through the mapping.  (A parallel walk -- pieces of the file, each
// started at the first offset from which four headers chain, through pread -- was tried in round 5 and measured no faster: the walk is
// ~18 ms of page faults / system calls either way, which do not scale over threads of one process here.)
int index_blocks(isx_bam &B, uint64_t &total)
{
    const uint8_t *map = B.map;
    const size_t n = B.map_len;
    B.blocks.clear();
    size_t off = 0;
    total = 0;
    while (off < n) {
        uint32_t hdr = 0, isize = 0;
        if (off + 18 > n) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        const uint8_t *h = map + off;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { isx_set_error("not a BGZF file"); return ISX_ERR_IO; }
        const size_t bsize = bgzf_header_at(map, n, off, hdr, isize);
        if (!bsize) { isx_set_error("corrupt BGZF block"); return ISX_ERR_IO; }
        B.blocks.push_back(Block{off, (uint32_t)bsize, hdr, isize, total});
        total += isize;
        off += bsize;
    }
    return ISX_OK;
}

int open_file(const char *path, isx_bam &B)
{
    B.fd = open(path, O_RDONLY);
    if (B.fd < 0) { isx_set_error(std::string("cannot open ") + path); return ISX_ERR_IO; }
    struct stat st;
    if (fstat(B.fd, &st) != 0) { isx_set_error("fstat failed"); return ISX_ERR_IO; }
    B.map_len = (size_t)st.st_size;
    if (B.map_len < 28) { isx_set_error("not a BGZF file"); return ISX_ERR_IO; }
    void *m = mmap(nullptr, B.map_len, PROT_READ, MAP_PRIVATE, B.fd, 0);
    if (m == MAP_FAILED) { isx_set_error("mmap failed"); return ISX_ERR_IO; }
    B.map = static_cast<const uint8_t *>(m);
    (void)madvise(m, B.map_len, MADV_SEQUENTIAL);
    // ---- index of the BGZF blocks (headers only, nothing is inflated) ----
    uint64_t total = 0;
    {
        const auto t_i0 = std::chrono::steady_clock::now();
        const int rc = index_blocks(B, total);
        if (rc != ISX_OK) return rc;
        if (getenv("ISX_BAM_TIMING")) fprintf(stderr, "[isx_bam_open] block index of %zu blocks: %.1f ms\n", B.blocks.size(),
                                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_i0).count());
    }
    B.total_inflated = total;
    // ---- header: magic, text, references (inflate block by block until it is complete) ----
    Inflater inf;
    std::vector<uint8_t> head;
    uint32_t nb = 0;
    auto more = [&](size_t need) -> bool {
        while (head.size() < need) {
            if (nb >= B.blocks.size()) return false;
            const Block &k = B.blocks[nb++];
            const size_t old = head.size();
            head.resize(old + k.isize);
            if (!inf.run(B.map + k.coff + k.hdr, k.csize - k.hdr - 8, head.data() + old, k.isize)) return false;
        }
        return true;
    };
    if (!more(12) || memcmp(head.data(), "BAM\1", 4) != 0) { isx_set_error("not a BAM file"); return ISX_ERR_IO; }
    const int32_t l_text = rd32(head.data() + 4);
    if (l_text < 0 || !more(12 + (size_t)l_text)) { isx_set_error("truncated BAM header"); return ISX_ERR_IO; }
    size_t o = 8 + (size_t)l_text;
    const int32_t n_ref = rd32(head.data() + o);
    o += 4;
    if (n_ref < 0) { isx_set_error("corrupt BAM header"); return ISX_ERR_IO; }
    int64_t flat = 0;
    for (int i = 0; i < n_ref; i++) {
        if (!more(o + 4)) { isx_set_error("truncated BAM header"); return ISX_ERR_IO; }
        const int32_t l_name = rd32(head.data() + o);
        if (l_name <= 0 || !more(o + 8 + (size_t)l_name)) { isx_set_error("truncated BAM header"); return ISX_ERR_IO; }
        B.ref_name.emplace_back(reinterpret_cast<const char *>(head.data() + o + 4), (size_t)l_name - 1);
        const int32_t l_ref = rd32(head.data() + o + 4 + l_name);
        if (l_ref < 0) { isx_set_error("corrupt BAM header"); return ISX_ERR_IO; }
        B.ref_len.push_back(l_ref);
        B.ref_off.push_back(flat);
        flat += l_ref;
        o += 8 + (size_t)l_name;
    }
    B.first_rec = o;
    // ---- segments: runs of blocks, ~32 MiB inflated for big files, smaller ones when the file is small so that
    //      every thread still gets a few (segments are the unit of parallel work in both passes) ----
    const uint64_t want_segs = (uint64_t)4 * (uint64_t)(B.threads > 0 ? B.threads : n_threads_default());
    uint64_t SEG = std::min<uint64_t>((uint64_t)32 << 20, std::max<uint64_t>((uint64_t)1 << 20, total / std::max<uint64_t>(want_segs, 1)));
    if (const char *e = getenv("ISX_BAM_SEG_KIB")) SEG = std::max<uint64_t>((uint64_t)64 << 10, (uint64_t)atoll(e) << 10);        // tuning aid
    Segment cur;
    cur.b0 = 0; cur.ioff0 = 0;
    for (uint32_t b = 0; b < B.blocks.size(); b++) {
        const uint64_t end = B.blocks[b].ioff + B.blocks[b].isize;
        if (end - cur.ioff0 >= SEG || b + 1 == B.blocks.size()) {
            cur.b1 = b + 1; cur.ioff1 = end;
            if (cur.ioff1 > cur.ioff0 || B.segs.empty()) B.segs.push_back(cur);
            cur = Segment();
            cur.b0 = b + 1; cur.ioff0 = end;
        }
    }
    return ISX_OK;
}

// htslib sam.c tweak_overlap_quality on two reads of the batch
struct SegBuf;
struct Batch {
    uvec<Read> reads;
    RawBuf<uint32_t> cigars;                // copied (aligned); sequences and qualities stay in the inflated segments:
    std::vector<uvec<uint8_t>> seg_data;
};

void tweak_overlap(Batch &S, const Read &a, const Read &b)
{
    Cursor ca{S.cigars.data() + a.cigar_off, (int)a.n_cigar, 0, 0, 0, 0};
    Cursor cb{S.cigars.data() + b.cigar_off, (int)b.n_cigar, 0, 0, 0, 0};
    uint8_t *aq = a.qual, *bq = b.qual;
    const uint8_t *as = a.seq, *bs = b.seq;
    int iref = b.pos;
    int a_ret = cur_set(ca, iref - a.pos);
    if (a_ret < 0) return;
    int b_ret = cur_set(cb, iref - b.pos);
    if (b_ret < 0) return;
    for (;;) {
        while (ca.iref >= 0 && ca.iref < iref - a.pos) a_ret = cur_next(ca);
        if (a_ret < 0) break;
        if (iref < ca.iref + a.pos) iref = ca.iref + a.pos;
        while (cb.iref >= 0 && cb.iref < iref - b.pos) b_ret = cur_next(cb);
        if (b_ret < 0) break;
        if (iref < cb.iref + b.pos) iref = cb.iref + b.pos;
        iref++;
        if (ca.iref + a.pos != cb.iref + b.pos) continue;
        const int qa = aq[ca.iseq], qb = bq[cb.iseq];
        if (nib(as, ca.iseq) == nib(bs, cb.iseq)) {
            const int q = qa + qb;
            aq[ca.iseq] = (uint8_t)(q > 200 ? 200 : q);
            bq[cb.iseq] = 0;
        } else if (qa >= qb) {
            aq[ca.iseq] = (uint8_t)(0.8 * qa);
            bq[cb.iseq] = 0;
        } else {
            bq[cb.iseq] = (uint8_t)(0.8 * qb);
            aq[ca.iseq] = 0;
        }
    }
}

// ---- scan: one segment's records -> ReadLite + names ----
// what pass 1 keeps of one record: fixed fields, name (appended to `names`) + its hash, NM, reference span
inline int extract_record(const uint8_t *p, int n_ref, uvec<char> &names, ReadLite &L, const char *&err)
{
    RecView r;
    if (!rec_view(p, r)) { err = "corrupt BAM record"; return ISX_ERR_IO; }
    if (r.tid >= n_ref) { err = "corrupt BAM record (reference id)"; return ISX_ERR_IO; }
    L = ReadLite{};
    L.tid = r.tid; L.pos = r.pos; L.isize = r.isize; L.l_seq = r.l_seq; L.flag = r.flag; L.mapq = r.mapq;
    const size_t at = names.size(), nl = (size_t)r.l_name - 1;
    L.name_off = (uint32_t)at; L.name_len = (uint16_t)nl;
    names.resize(at + nl);
    memcpy(names.data() + at, r.name, nl);
    L.h64 = hash_name(r.name, nl);
    bool has = false;
    int32_t nm = 0;
    if (parse_nm(r.aux, r.end, has, nm) != 0) { err = "bad aux field"; return ISX_ERR_IO; }
    L.has_nm = has; L.nm = nm;
    const RefSpan sp = span_of(r.cigar, r.n_cigar, r.pos);
    L.first = sp.first; L.last = sp.last; L.qlen = (int32_t)sp.qlen; L.any = sp.any;
    return ISX_OK;
}

int scan_segment(isx_bam &B, uint32_t si, const SegBuf &buf, const std::vector<uint64_t> &rec_off, std::string &err)
{
    const Segment &s = B.segs[si];
    uvec<ReadLite> &out = B.seg_reads[si];
    uvec<char> &names = B.seg_names[si];
    out.resize(rec_off.size());
    size_t name_bytes = 0;
    const int n_ref = (int)B.ref_name.size();
    for (size_t i = 0; i < rec_off.size(); i++) name_bytes += buf.data[(size_t)(rec_off[i] - s.ioff0) + 12];
    names.resize(0);
    names.reserve(name_bytes);
    for (size_t i = 0; i < rec_off.size(); i++) {
        const char *e = nullptr;
        const int rc = extract_record(buf.data.data() + (rec_off[i] - s.ioff0), n_ref, names, out[i], e);
        if (rc != ISX_OK) { err = e; return rc; }
    }
    return ISX_OK;
}

// Pass 1's per-segment work in one sweep: the segment's blocks are inflated one after the other and the record chain is walked
// through every block right after it was written (while it still sits in the core's cache -- walked afterwards, over a 10-30 MiB
// buffer, every hop of the chain is a miss: 110 ns a record, as much as decoding it), from `first`, or, when that is ~0, from a
// structural guess made on the segment's first blocks (*guess_out; ~0 = the segment seems to hold no record start).  The buffer
// is reserved with room for the record that straddles the segment's end, so bringing that one in does not move 30 MiB.
// Returns false when a block does not inflate; *hop_rc = seg_hop's result for the walk (ISX_ERR_IO: the chain ran into something
// that is not a record -- the guess was wrong, or the file is corrupt: the caller's serial pass decides).
// lite / names (may be NULL): the records' fields are extracted in the same sweep (extract_record), as soon as a record lies whole in
// the inflated part; *lite_rc = ISX_OK when every record of rec_off was extracted (else the caller runs scan_segment).
bool seg_inflate_hop(const isx_bam &B, Inflater &inf, const Segment &s, SegBuf &buf, uint64_t first, bool guess, uint64_t *guess_out,
                     std::vector<uint64_t> &rec_off, uint64_t &next_first, int *hop_rc, uvec<ReadLite> *lite, uvec<char> *names, int *lite_rc)
{
    const size_t seg_bytes = (size_t)(s.ioff1 - s.ioff0);
    buf.data.reserve(seg_bytes + (size_t)4 * 65536);
    buf.data.resize(0);
    buf.b_end = s.b0;
    rec_off.clear();
    rec_off.reserve(seg_bytes / 192 + 16);
    *hop_rc = ISX_OK;
    auto inflate_next = [&]() -> bool {
        const Block &k = B.blocks[buf.b_end];
        const size_t old = buf.data.size();
        buf.data.resize(old + k.isize);
        if (!inf.run(B.map + k.coff + k.hdr, k.csize - k.hdr - 8, buf.data.data() + old, k.isize)) return false;
        buf.b_end++;
        return true;
    };
    size_t n_lite = 0;                                  // records extracted so far
    bool lite_ok = lite != nullptr;
    const int n_ref = (int)B.ref_name.size();
    if (lite) { lite->resize(0); lite->reserve(seg_bytes / 192 + 16); names->resize(0); names->reserve(seg_bytes / 12); }
    auto extract_upto = [&](uint64_t have) {            // every record that ends at or before `have`
        while (lite_ok && n_lite < rec_off.size()) {
            const uint64_t a = rec_off[n_lite], e = n_lite + 1 < rec_off.size() ? rec_off[n_lite + 1] : next_first;
            if (e > have || e == 0) break;
            ReadLite L;
            const char *msg = nullptr;
            if (extract_record(buf.data.data() + (a - s.ioff0), n_ref, *names, L, msg) != ISX_OK) { lite_ok = false; break; }
            lite->push_back(L);
            n_lite++;
        }
    };
    next_first = 0;                                     // (the end of the last record in rec_off while the walk runs: kept in `off`)
    uint64_t off = first;
    if (guess) {
        for (int k = 0; k < 4 && buf.b_end < s.b1; k++) if (!inflate_next()) return false;
        off = seg_guess(B, inf, s, buf);                // (brings further blocks in itself when a candidate chain needs them)
        *guess_out = off;
    }
    bool walking = off != ~0ull;
    for (;;) {
        if (walking) {
            const uint64_t have = s.ioff0 + buf.data.size();
            const uint8_t *d = buf.data.data();
            while (off < s.ioff1 && off + 4 <= have) {
                const int32_t block = rd32(d + (off - s.ioff0));
                if (block < 32 || off + 4 + (uint64_t)block > B.total_inflated) { walking = false; *hop_rc = ISX_ERR_IO; break; }
                rec_off.push_back(off);
                off += 4 + (uint64_t)block;
            }
            next_first = off;
            extract_upto(have);
        }
        if (buf.b_end >= s.b1) break;
        if (!inflate_next()) return false;
    }
    if (walking) {
        // what is left: a record whose size field straddles the segment's end, and the bytes of the last record beyond it
        while (off < s.ioff1) {
            if (off + 4 > B.total_inflated) { *hop_rc = ISX_ERR_IO; break; }
            if (!seg_need(B, inf, s, buf, off, 4)) return false;
            const int32_t block = rd32(buf.data.data() + (off - s.ioff0));
            if (block < 32 || off + 4 + (uint64_t)block > B.total_inflated) { *hop_rc = ISX_ERR_IO; break; }
            rec_off.push_back(off);
            off += 4 + (uint64_t)block;
        }
        if (*hop_rc == ISX_OK && !rec_off.empty() && !seg_need(B, inf, s, buf, rec_off.back(), off - rec_off.back())) return false;
        if (*hop_rc != ISX_OK) isx_set_error("truncated BAM record");
        next_first = off;
        if (*hop_rc == ISX_OK) extract_upto(s.ioff0 + buf.data.size());
    }
    next_first = off;
    if (lite_rc) *lite_rc = (lite_ok && *hop_rc == ISX_OK && n_lite == rec_off.size()) ? ISX_OK : ISX_ERR_STATE;
    return true;
}

// open-addressing (hash of name) -> local pair index
struct NameTable {
    std::vector<uint64_t> key;
    std::vector<uint32_t> val;
    uint64_t mask = 0;
    void init(size_t n)
    {
        size_t cap = 16;
        while (cap < 2 * n + 8) cap <<= 1;
        key.assign(cap, 0); val.assign(cap, 0xFFFFFFFFu);
        mask = cap - 1;
    }
};

}  // namespace

extern "C" {

int isx_bam_open(const char *path, isx_bam **out)
{
    if (!path || !out) { isx_set_error("isx_bam_open: bad argument"); return ISX_ERR_ARG; }
    *out = nullptr;
    std::unique_ptr<isx_bam> B(new isx_bam());
    const int rc = open_file(path, *B);
    if (rc != ISX_OK) return rc;
    *out = B.release();
    return ISX_OK;
}

static void await_batches(isx_bam *B)
{
    std::unique_lock<std::mutex> lk(B->retire_mu);
    B->retire_cv.wait(lk, [&] { return B->batches_out <= 0; });
}

void isx_bam_close(isx_bam *bam)
{
    if (!bam) return;
    // the tables of a large file (hundreds of MB of pair entries and names) take tens of ms to give back: not the caller's
    // ... and a handle with batches still out (an uncollected isx_pipe_submit_bam ticket) goes when the last of them is back
    bool out;
    {
        std::lock_guard<std::mutex> lk(bam->retire_mu);
        out = bam->batches_out > 0;
    }
    if (out || bam->n_reads > (1u << 20)) std::thread([](isx_bam *dead) { await_batches(dead); delete dead; }, bam).detach();
    else delete bam;
}

void isx_bam_close_wait(isx_bam *bam)
{
    if (!bam) return;
    await_batches(bam);
    delete bam;
}

int isx_bam_set_threads(isx_bam *bam, int32_t threads)
{
    if (!bam || threads < 0) { isx_set_error("isx_bam_set_threads: bad argument"); return ISX_ERR_ARG; }
    bam->threads = threads;
    bam->pool.reset();
    return ISX_OK;
}

int isx_bam_ref(const isx_bam *bam, int32_t i, const char **name, int64_t *length, int64_t *flat_offset)
{
    if (!bam || i < 0 || (size_t)i >= bam->ref_name.size()) { isx_set_error("isx_bam_ref: bad index"); return ISX_ERR_ARG; }
    if (name) *name = bam->ref_name[(size_t)i].c_str();
    if (length) *length = bam->ref_len[(size_t)i];
    if (flat_offset) *flat_offset = bam->ref_off[(size_t)i];
    return ISX_OK;
}

int isx_bam_set_priority_reads(isx_bam *bam, int64_t n, const char *names, const int64_t *offs)
{
    if (!bam || n < 0 || (n && (!names || !offs))) { isx_set_error("isx_bam_set_priority_reads: bad argument"); return ISX_ERR_ARG; }
    bam->priority_names.clear();
    for (int64_t i = 0; i < n; i++) bam->priority_names.emplace_back(names + offs[i], (size_t)(offs[i + 1] - offs[i]));
    return ISX_OK;
}

// ---- pass 1: get_paired_reads for every reference (filter_reads.py:885-956) ----
int isx_bam_scan(isx_bam *bam, isx_bam_info *info) { return isx_bam_scan_part(bam, 0, 1, info); }

// Pass 1 over ONE of n_parts shares of the file (ranks of a multi-GPU run: each scans its share, not the whole file).
// The share is a range of segments of equal compressed size, [s0, s1); the handle OWNS the references whose first read lies in
// it: their pair tables are complete (the scan runs on past s1 until the reference that straddles it has ended), every other
// reference looks empty to this handle.  A share that does not start the file begins two segments early: the chain check
// (a segment's first record must be where the previous segment's walk ended) then holds for its own first segment exactly
// as in a whole-file scan, and the last read before s0 tells whether the reference at s0 began earlier (then it belongs to the
// previous share).  What is global -- the median insert of the read filter -- is the caller's to combine
// (isx_bam_insert_sizes of every share -> isx_bam_filter(median_insert)).
int isx_bam_scan_part(isx_bam *bam, int32_t part, int32_t n_parts, isx_bam_info *info)
{
    if (!bam || n_parts < 1 || part < 0 || part >= n_parts) { isx_set_error("isx_bam_scan: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    if (B.scanned) {
        if (B.part != part || B.n_parts != n_parts) { isx_set_error("isx_bam_scan: this handle already scanned another share of the file"); return ISX_ERR_STATE; }
        if (info) *info = B.totals;
        return ISX_OK;
    }
    B.part = part; B.n_parts = n_parts;
    isxenc::HostPool &pool = pool_of(B);
    const size_t n_ref = B.ref_name.size(), n_seg = B.segs.size();
    const bool timing = getenv("ISX_BAM_TIMING") != nullptr;       // tuning aid: stage times on stderr (no effect on results)
    double t_inflate = 0, t_hop = 0, t_extract = 0;
    std::atomic<uint64_t> thr_guess_us{0};
    std::atomic<uint64_t> thr_inflate_us{0}, thr_hop_us{0};        // (timing only) summed over the pool's threads: block decode / guess + record walk
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    double t_mark = t_begin;
    B.seg_reads.assign(n_seg, {}); B.seg_names.assign(n_seg, {});
    // pass 2 needs the same bytes again: keep a file's inflated segments when they are a small part of what the host has free
    const bool keep_inflated = B.total_inflated <= inflate_cache_limit();
    if (keep_inflated) { B.seg_cache.assign(n_seg, {}); B.seg_cache_rec.assign(n_seg, {}); }
    // waves of segments: inflate (parallel) -> record boundaries (serial walk over block_size fields) -> field
    // extraction (parallel).  At most one wave of inflated data is alive.
    // (ISX_BAM_WAVE / ISX_BAM_SEG_KIB: tuning aids.  Round 5 tried several small segments per thread and wave, handed out one at a time --
    //  256 segments in waves of 64 instead of 128 in waves of 16 -- against the 15 % a wave loses to its slowest decode: 249-276 ms
    //  against 193-307 for the scan of the probe on one (noisy) box and a slower pass 2; not kept)
    size_t wave = (size_t)std::max(2, pool.size());
    if (const char *e = getenv("ISX_BAM_WAVE")) wave = (size_t)std::max(2, atoi(e));
    uint64_t first = B.first_rec, read_ord = 0;
    std::vector<std::string> errs(n_seg);
    const size_t s0 = n_seg * (size_t)part / (size_t)n_parts, s1 = n_seg * ((size_t)part + 1) / (size_t)n_parts;
    size_t sv = s0 >= 2 ? s0 - 2 : 0;           // where the walk starts (verification run-in)
    bool have_first = sv == 0;                  // the file's first record is known; any other start is a guess until the chain confirms it
    bool verified = sv == 0;                    // a guessed start counts once a later segment's own guess falls on the chain walked from it
    size_t s_end = n_seg;                       // one past the last segment scanned
    if (s0 == s1) { sv = 0; s_end = 0; }        // more shares than segments: this one is empty
    for (size_t w0 = sv, w1 = 0; w0 < s_end; w0 = w1) {
        // full waves while they start inside the share; past its end a quarter wave at a time (the scaffold that straddles
        // s1 usually ends within a segment or two)
        w1 = std::min(n_seg, w0 + ((n_parts > 1 && w0 >= s1) ? std::max<size_t>(1, wave / 4) : wave));
        std::vector<SegBuf> bufs(w1 - w0);
        std::vector<std::vector<uint64_t>> recs(w1 - w0);
        std::atomic<int> bad{0};
        // every segment: inflate, guess where its first record starts, walk its records from there (all in parallel) ...
        std::vector<uint64_t> guess(w1 - w0, ~0ull), nexts(w1 - w0, 0);
        std::vector<int> hop_rc(w1 - w0, ISX_OK), lite_rc(w1 - w0, ISX_ERR_STATE);      // lite_rc OK: the sweep extracted every record of its walk
        pool.run((int)(w1 - w0), [&](int i) {
            Inflater inf;
            const Segment &s = B.segs[w0 + (size_t)i];
            const double ta = timing ? now() : 0.0;
            const bool is_first = w0 + (size_t)i == 0;
            uint64_t g = is_first ? std::max(B.first_rec, s.ioff0) : ~0ull;
            if (is_first && B.first_rec > s.ioff1) g = ~0ull;
            const bool do_guess = !is_first;
            if (!seg_inflate_hop(B, inf, s, bufs[(size_t)i], g, do_guess, &g, recs[(size_t)i], nexts[(size_t)i], &hop_rc[(size_t)i],
                                 &B.seg_reads[w0 + (size_t)i], &B.seg_names[w0 + (size_t)i], &lite_rc[(size_t)i])) { bad.store(1); return; }
            guess[(size_t)i] = g;
            const double tb = timing ? now() : 0.0, tg = tb;
            if (timing) { const double tc = now(); thr_inflate_us.fetch_add((uint64_t)((tb - ta) * 1e3)); thr_hop_us.fetch_add((uint64_t)((tc - tb) * 1e3)); thr_guess_us.fetch_add((uint64_t)((tg - tb) * 1e3)); }
        });
        if (bad.load()) { isx_set_error("BGZF inflate failed"); return ISX_ERR_IO; }
        { const double t = now(); t_inflate += t - t_mark; t_mark = t; }
        // ... then the chain is checked serially: a segment's first record must be where the previous segment's walk
        // ended; where the guess was off (or there was none) that segment is walked again from the right place
        Inflater inf;
        for (size_t si = w0; si < w1; si++) {
            Segment &s = B.segs[si];
            const size_t k = si - w0;
            if (!have_first) {                  // a share's run-in: start from the structural guess
                if (guess[k] == ~0ull) { s.first_rec = s.ioff1; s.read0 = read_ord; s.n_reads = 0; recs[k].clear(); lite_rc[k] = ISX_ERR_STATE; continue; }
                first = guess[k];
                have_first = true;
            }
            if (first > s.ioff1) { s.first_rec = s.ioff1; s.read0 = read_ord; s.n_reads = 0; recs[k].clear(); lite_rc[k] = ISX_ERR_STATE; continue; }    // a record spans the whole segment
            s.first_rec = std::max(first, s.ioff0);
            if (!verified && s.first_rec < s.ioff1 && guess[k] != ~0ull && si > sv) {
                // the run-in of a share: the chain started from a structural guess.  The next segment's own, independent guess
                // either falls on the chain -- both are right -- or one of them is wrong and nothing here can tell which
                if (guess[k] != s.first_rec) {
                    isx_set_error("isx_bam_scan_part: the record chain walked from a guessed start disagrees with the next segment's own guess: scan the whole file instead");
                    return ISX_ERR_IO;
                }
                verified = true;
            }
            if (!verified && si >= s0 && n_parts > 1) {
                isx_set_error("isx_bam_scan_part: no second record start confirmed the guessed one in the two segments before this share: scan the whole file instead");
                return ISX_ERR_IO;
            }
            uint64_t next = first;
            if (s.first_rec >= s.ioff1) { recs[k].clear(); lite_rc[k] = ISX_ERR_STATE; next = first; }
            else if (guess[k] == s.first_rec && hop_rc[k] == ISX_OK) next = nexts[k];
            else {
                lite_rc[k] = ISX_ERR_STATE;         // (the sweep walked -- and extracted -- from a wrong start)
                const int rc = seg_hop(B, inf, s, bufs[k], s.first_rec, recs[k], next);
                if (rc != ISX_OK) return rc;
            }
            s.read0 = read_ord; s.n_reads = (uint32_t)recs[k].size();
            read_ord += s.n_reads;
            first = next;
        }
        { const double t = now(); t_hop += t - t_mark; t_mark = t; }
        std::atomic<int> rc_any{0};
        pool.run((int)(w1 - w0), [&](int i) {
            if (lite_rc[(size_t)i] == ISX_OK && B.seg_reads[w0 + (size_t)i].size() == recs[(size_t)i].size() && !getenv("ISX_BAM_NO_FUSED_EXTRACT")) return;   // done in the sweep
            const int rc = scan_segment(B, (uint32_t)(w0 + (size_t)i), bufs[(size_t)i], recs[(size_t)i], errs[w0 + (size_t)i]);
            if (rc != ISX_OK) rc_any.store(rc);
        });
        if (rc_any.load()) { for (auto &e : errs) if (!e.empty()) { isx_set_error(e); break; } return rc_any.load(); }
        if (keep_inflated)
            for (size_t si = w0; si < w1; si++) { B.seg_cache[si].swap(bufs[si - w0].data); B.seg_cache_rec[si].swap(recs[si - w0]); }
        { const double t = now(); t_extract += t - t_mark; t_mark = t; }
        if (n_parts > 1 && w1 >= s1 && w1 < n_seg) {
            // past the share: stop once the reference that straddles s1 has ended (a read of another reference, or an
            // unmapped one, was seen at or after s1)
            int32_t t_end = -2;
            for (size_t si = s1; si-- > sv;) if (!B.seg_reads[si].empty()) { t_end = B.seg_reads[si].back().tid; break; }
            bool ended = t_end < 0;
            if (!ended && s0 > 0) {             // a share inside one long scaffold owns nothing: nothing to complete either
                int32_t t_prev = -2;
                for (size_t si = s0; si-- > sv;) if (!B.seg_reads[si].empty()) { t_prev = B.seg_reads[si].back().tid; break; }
                bool owns_any = false;
                for (size_t si = s0; si < s1 && !owns_any; si++)
                    for (const ReadLite &L : B.seg_reads[si]) if (L.tid >= 0 && L.tid != t_prev) { owns_any = true; break; }
                if (!owns_any) ended = true;
            }
            for (size_t si = s1; si < w1 && !ended; si++)
                if (!B.seg_reads[si].empty() && B.seg_reads[si].back().tid != t_end) ended = true;
            if (ended) { s_end = w1; break; }
        }
    }
    if (s_end == n_seg && s0 != s1 && first != B.total_inflated) { isx_set_error("truncated BAM record"); return ISX_ERR_IO; }
    B.n_reads = read_ord;
    // which references this share owns: those whose first read lies in [s0, s1)
    std::vector<uint8_t> owned(n_ref, n_parts == 1 ? 1 : 0);
    if (n_parts > 1 && s0 != s1) {
        int32_t t_prev = -2;                    // reference of the last read before s0: its run began in an earlier share
        for (size_t si = s0; si-- > sv;) if (!B.seg_reads[si].empty()) { t_prev = B.seg_reads[si].back().tid; break; }
        if (s0 > 0 && t_prev == -2) { isx_set_error("isx_bam_scan_part: no record starts in the two segments before this share (a record longer than a segment): scan the whole file instead"); return ISX_ERR_ARG; }
        for (size_t si = s0; si < s1; si++)
            for (const ReadLite &L : B.seg_reads[si]) if (L.tid >= 0 && L.tid != t_prev) owned[(size_t)L.tid] = 1;
    }
    // per segment on the threads: the longest reference span of a read, and the runs of reads of one reference
    struct TidRun { int32_t tid; uint32_t i0, i1; };
    std::vector<std::vector<TidRun>> seg_runs(n_seg);
    std::vector<int64_t> seg_span(n_seg, 0);
    pool.run((int)n_seg, [&](int k) {
        const size_t si = (size_t)k;
        const auto &rs = B.seg_reads[si];
        if (rs.empty()) return;
        Segment &sg = B.segs[si];
        sg.tid_first = rs.front().tid; sg.pos_first = rs.front().pos; sg.tid_last = rs.back().tid; sg.pos_last = rs.back().pos;
        int64_t span = 0;
        size_t i0 = 0;
        for (size_t i = 0; i < rs.size(); i++) {
            const ReadLite &L = rs[i];
            if (L.any) span = std::max<int64_t>(span, L.last - (int64_t)L.pos + 1);
            if (L.tid != rs[i0].tid) { seg_runs[si].push_back(TidRun{rs[i0].tid, (uint32_t)i0, (uint32_t)i}); i0 = i; }
        }
        seg_runs[si].push_back(TidRun{rs[i0].tid, (uint32_t)i0, (uint32_t)rs.size()});
        seg_span[si] = span;
    });
    for (size_t si = 0; si < n_seg; si++) B.max_span = std::max(B.max_span, seg_span[si]);

    // ---- reference -> the run of reads that belongs to it (the file is sorted: a reference's reads are contiguous) ----
    struct Run { uint32_t seg; uint32_t i0, i1; };
    std::vector<std::vector<Run>> runs(n_ref);
    B.ref_seg0.assign(n_ref, 0xFFFFFFFFu); B.ref_seg1.assign(n_ref, 0);
    B.ref_reads.assign(n_ref, 0);
    {
        int32_t last_tid = -2;
        bool unsorted = false;
        std::vector<uint8_t> closed(n_ref, 0);
        for (uint32_t si = 0; si < n_seg; si++) {
            for (const TidRun &tr : seg_runs[si]) {
                const int32_t t = tr.tid;
                const size_t i = tr.i0, j = tr.i1;
                if (t >= 0 && owned[(size_t)t]) {
                    if (t != last_tid && closed[(size_t)t]) unsorted = true;
                    runs[(size_t)t].push_back(Run{si, (uint32_t)i, (uint32_t)j});
                    B.ref_seg0[(size_t)t] = std::min(B.ref_seg0[(size_t)t], si);
                    B.ref_seg1[(size_t)t] = std::max(B.ref_seg1[(size_t)t], si);
                    B.ref_reads[(size_t)t] += (int64_t)(j - i);
                }
                if (last_tid >= 0 && t != last_tid) closed[(size_t)last_tid] = 1;
                last_tid = t;
            }
        }
        if (unsorted) { isx_set_error("BAM is not sorted by reference: the reads of a reference must be contiguous"); return ISX_ERR_IO; }
    }

    // ---- pair tables: one task per (reference, name-hash partition) ----
    // The reads are first bucketed by partition (two passes over the per-segment runs: count, then fill an index list in
    // file order), so that a partition's task touches only its own reads -- not every read of the reference.
    struct Part { uint32_t ref, p, P; uint64_t r0, r1; std::vector<PairInfo> info; std::string no_nm; };
    struct ReadRef { uint32_t seg, i; };
    std::vector<Part> parts;
    std::vector<size_t> first_part(n_ref + 1, 0);
    for (size_t t = 0; t < n_ref; t++) {
        first_part[t] = parts.size();
        if (!B.ref_reads[t]) continue;
        const uint32_t P = (uint32_t)std::min<int64_t>(4096, B.ref_reads[t] / 16384 + 1);
        for (uint32_t p = 0; p < P; p++) parts.push_back(Part{(uint32_t)t, p, P, 0, 0, {}, {}});
    }
    first_part[n_ref] = parts.size();
    struct FlatRun { uint32_t ref, seg, i0, i1; size_t cur; };
    std::vector<FlatRun> flat;
    size_t n_cur = 0;
    for (size_t t = 0; t < n_ref; t++)
        for (const Run &r : runs[t]) { flat.push_back(FlatRun{(uint32_t)t, r.seg, r.i0, r.i1, n_cur}); n_cur += first_part[t + 1] - first_part[t]; }
    std::vector<uint64_t> cur(n_cur, 0);
    auto part_of = [](uint64_t h64, uint32_t P) -> uint32_t { return P == 1 ? 0u : (uint32_t)((h64 >> 40) % P); };
    pool.run((int)flat.size(), [&](int k) {
        const FlatRun &f = flat[(size_t)k];
        const uint32_t P = (uint32_t)(first_part[f.ref + 1] - first_part[f.ref]);
        const auto &rs = B.seg_reads[f.seg];
        uint64_t *c = cur.data() + f.cur;
        for (uint32_t i = f.i0; i < f.i1; i++) c[part_of(rs[i].h64, P)]++;
    });
    {   // counts -> write cursors: a partition's reads stay in file order (its runs in order, each run's reads in order)
        uint64_t at = 0;
        size_t k0 = 0;
        for (size_t t = 0; t < n_ref; t++) {
            const size_t nr = runs[t].size(), P = first_part[t + 1] - first_part[t];
            for (size_t p = 0; p < P; p++) {
                parts[first_part[t] + p].r0 = at;
                for (size_t k = 0; k < nr; k++) { uint64_t &c = cur[flat[k0 + k].cur + p]; const uint64_t n = c; c = at; at += n; }
                parts[first_part[t] + p].r1 = at;
            }
            k0 += nr;
        }
    }
    uvec<ReadRef> order((size_t)(parts.empty() ? 0 : parts.back().r1));
    pool.run((int)flat.size(), [&](int k) {
        const FlatRun &f = flat[(size_t)k];
        const uint32_t P = (uint32_t)(first_part[f.ref + 1] - first_part[f.ref]);
        const auto &rs = B.seg_reads[f.seg];
        uint64_t *c = cur.data() + f.cur;
        for (uint32_t i = f.i0; i < f.i1; i++) order[(size_t)c[part_of(rs[i].h64, P)]++] = ReadRef{f.seg, i};
    });
    std::vector<uint64_t>().swap(cur);
    const double t_bucket = now();
    par_assign(pool, B.read_pair, (size_t)B.n_reads, 0xFFFFFFFFu);
    pool.run((int)parts.size(), [&](int pi) {
        Part &pt = parts[(size_t)pi];
        NameTable tab;
        tab.init((size_t)(pt.r1 - pt.r0));
        pt.info.reserve((size_t)(pt.r1 - pt.r0) / 2 + 8);
        for (uint64_t q = pt.r0; q < pt.r1; q++) {
            // a partition's reads lie scattered over the file: what the next ones will touch is asked for ahead of time
            if (q + 16 < pt.r1) { const ReadRef r2 = order[(size_t)q + 16]; __builtin_prefetch(&B.seg_reads[r2.seg][r2.i]); }
            if (q + 8 < pt.r1) {
                const ReadRef r3 = order[(size_t)q + 8];
                const ReadLite &L3 = B.seg_reads[r3.seg][r3.i];
                __builtin_prefetch(B.seg_names[r3.seg].data() + L3.name_off);
                __builtin_prefetch(&tab.key[L3.h64 & tab.mask]);
                __builtin_prefetch(&B.read_pair[(size_t)(B.segs[r3.seg].read0 + r3.i)], 1);
            }
            const ReadRef rr = order[(size_t)q];
            const ReadLite &L = B.seg_reads[rr.seg][rr.i];
            const char *names = B.seg_names[rr.seg].data();
            // get_paired_reads skips unmapped reads and reads without aligned bases (get_reference_positions() == []);
            // the latter still take part in htslib's overlap bookkeeping by name, so they get an entry that counts nothing
            const bool counted = !(L.flag & FUNMAP) && L.any;
            if (L.flag & FUNMAP) continue;
            if (counted && !L.has_nm) { if (pt.no_nm.empty()) pt.no_nm.assign(names + L.name_off, L.name_len); continue; }
            uint64_t slot = L.h64 & tab.mask;
            const uint64_t hk = L.h64 | 1;              // 0 marks an empty slot
            uint32_t idx = 0xFFFFFFFFu;
            for (;;) {
                if (tab.key[slot] == 0) break;
                if (tab.key[slot] == hk) {
                    const PairInfo &e = pt.info[tab.val[slot]];
                    if (e.name_len == L.name_len && memcmp(B.seg_names[e.name_seg].data() + e.name_off, names + L.name_off, L.name_len) == 0) { idx = tab.val[slot]; break; }
                }
                slot = (slot + 1) & tab.mask;
            }
            if (idx == 0xFFFFFFFFu) {
                idx = (uint32_t)pt.info.size();
                tab.key[slot] = hk; tab.val[slot] = idx;
                PairInfo e{};
                e.name_seg = rr.seg; e.name_off = L.name_off; e.name_len = L.name_len;
                e.insert = -1;
                if (counted) { e.nm = L.nm; e.mapq = L.mapq; e.length = L.qlen; e.reads = 1; e.start = L.first; e.stop = L.last; }
                pt.info.push_back(e);
            } else if (counted) {
                PairInfo &e = pt.info[idx];
                if (e.reads == 0) { e.nm = L.nm; e.mapq = L.mapq; e.length = L.qlen; e.reads = 1; e.start = L.first; e.stop = L.last; e.insert = -1; }
                else {
                    e.nm += L.nm;
                    e.reads += 1;
                    e.length += L.qlen;
                    e.mapq = std::max<int64_t>(e.mapq, L.mapq);
                    if (e.reads == 2) {
                        if (L.last > e.start) e.insert = L.last - e.start;
                        else e.insert = e.stop - L.first;
                    } else e.insert = -1;
                    e.start = 0; e.stop = 0;
                }
            }
            B.read_pair[(size_t)(B.segs[rr.seg].read0 + rr.i)] = idx;       // local index for now
        }
    });
    const double t_tables = now();
    for (auto &pt : parts) if (!pt.no_nm.empty()) { isx_set_error("read without NM tag: " + pt.no_nm); return ISX_ERR_IO; }
    // one table: a reference's partitions follow each other
    std::vector<uint64_t> part_base(parts.size() + 1, 0);
    for (size_t i = 0; i < parts.size(); i++) part_base[i + 1] = part_base[i] + parts[i].info.size();
    if (part_base.back() >= 0xFFFFFFFFull) { isx_set_error("more than 2^32 read names"); return ISX_ERR_ARG; }
    B.pairs.resize((size_t)part_base.back());
    B.ref_pair0.assign(n_ref + 1, 0);
    for (size_t t = 0; t <= n_ref; t++) B.ref_pair0[t] = part_base[first_part[t]];
    pool.run((int)parts.size(), [&](int pi) {
        Part &pt = parts[(size_t)pi];
        std::copy(pt.info.begin(), pt.info.end(), B.pairs.begin() + (ptrdiff_t)part_base[(size_t)pi]);
        std::vector<PairInfo>().swap(pt.info);
    });
    // local pair index -> index into the one table, read by read in file order (a read's partition follows from its hash again)
    pool.run((int)flat.size(), [&](int k) {
        const FlatRun &f = flat[(size_t)k];
        const uint32_t P = (uint32_t)(first_part[f.ref + 1] - first_part[f.ref]);
        const auto &rs = B.seg_reads[f.seg];
        const uint64_t *pb = part_base.data() + first_part[f.ref];
        uint32_t *v = B.read_pair.data() + B.segs[f.seg].read0;
        for (uint32_t i = f.i0; i < f.i1; i++)
            if (v[i] != 0xFFFFFFFFu) v[i] += (uint32_t)pb[part_of(rs[i].h64, P)];
    });
    if (timing) fprintf(stderr, "[isx_bam_scan] share %d/%d: segments [%zu, %zu) of %zu, %d threads: inflate %.1f ms, record walk %.1f ms, field extraction %.1f ms, pair tables %.1f ms (bucketing %.1f, tables %.1f, merge %.1f); %.1f ms since the scan began\n",
                        part, n_parts, sv, s_end, n_seg, pool.size(), t_inflate, t_hop, t_extract, now() - t_mark, t_bucket - t_mark, t_tables - t_bucket, now() - t_tables, now() - t_begin);
    if (timing) fprintf(stderr, "[isx_bam_scan]   inside 'inflate', summed over the threads: block decode + buffers %.1f ms, start guess + record walk %.1f ms (guess %.1f)\n",
                        thr_inflate_us.load() / 1e3, thr_hop_us.load() / 1e3, thr_guess_us.load() / 1e3);
    {
        size_t bytes = 0;
        for (const auto &v : B.seg_reads) bytes += v.size() * sizeof(ReadLite);
        if (bytes <= isx_bam::KEEP_DEAD && B.dead_reads.empty()) B.dead_reads.swap(B.seg_reads);       // unmapped with the handle (see isx_bam::retired)
        else pool.run((int)B.seg_reads.size(), [&](int k) { uvec<ReadLite>().swap(B.seg_reads[(size_t)k]); });
        B.seg_reads.clear();
    }
    // names stay until the filter has run (set_r2m / priority reads / cross-scaffold filters)
    B.totals = isx_bam_info{};
    B.totals.n_refs = (int32_t)n_ref;
    B.totals.n_reads = (int64_t)B.n_reads;
    if (n_parts > 1) { B.totals.n_reads = 0; for (int64_t r : B.ref_reads) B.totals.n_reads += r; }     // the share's own references
    for (int64_t l : B.ref_len) B.totals.n_pos += l;
    B.scanned = true;
    if (info) *info = B.totals;
    return ISX_OK;
}

int isx_bam_insert_sizes(isx_bam *bam, int64_t *out, int64_t cap, int64_t *n)
{
    if (!bam || !n) { isx_set_error("isx_bam_insert_sizes: bad argument"); return ISX_ERR_ARG; }
    if (!bam->scanned) { isx_set_error("isx_bam_insert_sizes: scan first"); return ISX_ERR_STATE; }
    int64_t k = 0;
    const size_t n_ref = bam->ref_name.size();
    for (size_t t = 0; t < n_ref; t++) {
        if (!bam->ref_wanted.empty() && !bam->ref_wanted[t]) continue;      // the reference only loads the scaffolds of the fasta
        for (uint64_t j = bam->ref_pair0[t]; j < bam->ref_pair0[t + 1]; j++) {
            const PairInfo &i = bam->pairs[(size_t)j];
            if (i.reads == 2) { if (out && k < cap) out[k] = i.insert; k++; }
        }
    }
    *n = k;
    return ISX_OK;
}

// a second, independent hash of a name: with hash_name a 128-bit identity for names compared across shares
static uint64_t hash_name2(const uint8_t *s, size_t n)
{
    uint64_t h = 0x84222325cbf29ce4ull + n * 0xC2B2AE3D27D4EB4Full;
    for (size_t i = 0; i < n; i++) { h ^= s[i]; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 31; }
    h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33;
    return h;
}

int isx_bam_pair_keys(isx_bam *bam, uint64_t *h1, uint64_t *h2, int32_t *tid, int64_t *info4, int64_t cap, int64_t *n)
{
    if (!bam || !n) { isx_set_error("isx_bam_pair_keys: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    if (!B.scanned) { isx_set_error("isx_bam_pair_keys: scan first"); return ISX_ERR_STATE; }
    if (B.seg_names.empty() && !B.pairs.empty()) { isx_set_error("isx_bam_pair_keys: the read names were already dropped"); return ISX_ERR_STATE; }
    const std::vector<PairInfo, NoInitAlloc<PairInfo>> &tab = B.pairs_scan.empty() ? B.pairs : B.pairs_scan;     // what the scan found
    *n = (int64_t)tab.size();
    if (!h1) return ISX_OK;
    if (!h2 || !tid || !info4 || cap < *n) { isx_set_error("isx_bam_pair_keys: arrays too small"); return ISX_ERR_ARG; }
    const size_t n_ref = B.ref_name.size();
    isxenc::HostPool &pool = pool_of(B);
    pool.run((int)n_ref, [&](int t) {
        const bool gone = !B.ref_wanted.empty() && !B.ref_wanted[(size_t)t];
        for (uint64_t j = B.ref_pair0[(size_t)t]; j < B.ref_pair0[(size_t)t + 1]; j++) {
            const PairInfo &e = tab[(size_t)j];
            const uint8_t *nm = reinterpret_cast<const uint8_t *>(B.seg_names[e.name_seg].data() + e.name_off);
            h1[j] = hash_name(nm, e.name_len); h2[j] = hash_name2(nm, e.name_len);
            tid[j] = t;
            info4[4 * j] = e.nm; info4[4 * j + 1] = e.mapq; info4[4 * j + 2] = e.length;
            info4[4 * j + 3] = gone ? 0 : e.reads;          // reads == 0: the entry does not exist for the filter
        }
    });
    return ISX_OK;
}

int isx_bam_set_cross_names(isx_bam *bam, int64_t n, const int64_t *entry, const int64_t *occurrences, const int64_t *info4)
{
    if (!bam || n < 0 || (n && (!entry || !occurrences || !info4))) { isx_set_error("isx_bam_set_cross_names: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    if (!B.scanned) { isx_set_error("isx_bam_set_cross_names: scan first"); return ISX_ERR_STATE; }
    for (int64_t i = 0; i < n; i++)
        if (entry[i] < 0 || (size_t)entry[i] >= B.pairs.size() || occurrences[i] < 2) { isx_set_error("isx_bam_set_cross_names: entry out of range / fewer than two occurrences"); return ISX_ERR_ARG; }
    B.cross_idx.assign(entry, entry + n);
    B.cross_occ.assign(occurrences, occurrences + n);
    B.cross_info.assign(info4, info4 + 4 * n);
    B.cross_set = true;
    return ISX_OK;
}

int isx_bam_filter_insert_sizes(isx_bam *bam, int64_t *out, int64_t cap, int64_t *n)
{
    if (!bam || !n) { isx_set_error("isx_bam_filter_insert_sizes: bad argument"); return ISX_ERR_ARG; }
    if (!bam->filtered) { isx_set_error("isx_bam_filter_insert_sizes: filter first"); return ISX_ERR_STATE; }
    int64_t k = 0;
    for (const PairInfo &e : bam->pairs)
        if (e.in_filter && e.reads == 2) { if (out && k < cap) out[k] = e.insert; k++; }
    *n = k;
    return ISX_OK;
}

// The scaffolds that count for the read filter: the reference builds its pair table only from the scaffolds of the fasta
// (filter_reads.py:63-77, 157-178) -- the median insert (:213-217), the tallies and the cross-scaffold name look-ups of
// non_discordant / all_reads never see a read of another scaffold of the BAM.  n == 0: every reference of the file.
int isx_bam_set_wanted_refs(isx_bam *bam, const int32_t *refs, int32_t n)
{
    if (!bam || n < 0 || (n && !refs)) { isx_set_error("isx_bam_set_wanted_refs: bad argument"); return ISX_ERR_ARG; }
    const size_t n_ref = bam->ref_name.size();
    bam->ref_wanted.clear();
    if (n == 0) return ISX_OK;
    bam->ref_wanted.assign(n_ref, 0);
    for (int32_t i = 0; i < n; i++) {
        if (refs[i] < 0 || (size_t)refs[i] >= n_ref) { bam->ref_wanted.clear(); isx_set_error("isx_bam_set_wanted_refs: reference id out of range"); return ISX_ERR_ARG; }
        bam->ref_wanted[(size_t)refs[i]] = 1;
    }
    return ISX_OK;
}

// ---- paired_read_filter + filter_scaff2pair2info (filter_reads.py:471-532, 201-260) ----
int isx_bam_filter(isx_bam *bam, const isx_bam_params *p, double median_insert, isx_bam_info *info)
{
    if (!bam || !p) { isx_set_error("isx_bam_filter: bad argument"); return ISX_ERR_ARG; }
    if (!bam->scanned) { isx_set_error("isx_bam_filter: scan first"); return ISX_ERR_STATE; }
    if (p->pairing_filter < 0 || p->pairing_filter > 2) { isx_set_error("pairing_filter must be 0 (paired_only), 1 (non_discordant) or 2 (all_reads)"); return ISX_ERR_ARG; }
    if (bam->n_parts > 1 && p->pairing_filter != 0 && !bam->cross_set) { isx_set_error("isx_bam_filter: non_discordant / all_reads look read names up across all scaffolds: scan the whole file (isx_bam_scan), or tell this share what the others hold (isx_bam_pair_keys of every share -> isx_bam_set_cross_names)"); return ISX_ERR_STATE; }
    if (bam->n_parts > 1 && std::isnan(median_insert)) { isx_set_error("isx_bam_filter: a share of the file cannot know the file's median insert: combine isx_bam_insert_sizes of all shares and pass it"); return ISX_ERR_STATE; }
    isx_bam &B = *bam;
    const size_t n_ref = B.ref_name.size();
    auto name_of = [&](const PairInfo &e) { return std::string_view(B.seg_names[e.name_seg].data() + e.name_off, e.name_len); };
    if (B.seg_names.empty() && !B.pairs.empty()) { isx_set_error("isx_bam_filter: the read names were already dropped (filter runs once per scan)"); return ISX_ERR_STATE; }
    // priority reads: by name
    B.priority.assign(B.pairs.size(), 0);
    if (!B.priority_names.empty()) {
        std::unordered_map<std::string_view, int> pr;
        for (const std::string &s : B.priority_names) pr.emplace(std::string_view(s), 1);
        for (size_t i = 0; i < B.pairs.size(); i++) if (pr.count(name_of(B.pairs[i]))) B.priority[i] = 1;
    }
    // entries of scaffolds outside the wanted set do not exist for the filter
    std::vector<uint8_t> skip;
    if (!B.ref_wanted.empty()) {
        skip.assign(B.pairs.size(), 0);
        for (size_t t = 0; t < n_ref; t++)
            if (!B.ref_wanted[t]) std::fill(skip.begin() + (ptrdiff_t)B.ref_pair0[t], skip.begin() + (ptrdiff_t)B.ref_pair0[t + 1], (uint8_t)1);
    }
    auto absent = [&](size_t i) { return !skip.empty() && skip[i]; };
    isx_bam_info T = B.totals;
    T.unfiltered_pairs = T.unfiltered_singletons = T.unfiltered_reads = 0;
    T.filtered_pairs = T.filtered_singletons = T.filtered_bases = 0;
    if (!B.pairs_scan.empty()) B.pairs = B.pairs_scan;          // an earlier all_reads run merged entries in place
    else if (p->pairing_filter == 2) B.pairs_scan = B.pairs;
    isxenc::HostPool &pool = pool_of(B);
    const size_t n_pairs = B.pairs.size();
    const int C = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n_pairs / 32768 + 1));
    auto c_lo = [&](int c) { return n_pairs * (size_t)c / (size_t)C; };
    struct Tally { int64_t reads = 0, pairs = 0, single = 0, f_pairs = 0, f_single = 0, f_bases = 0, max_mm = 0; std::vector<int64_t> ins; };
    std::vector<Tally> tl((size_t)C);
    // paired_read_filter: scaffolds in header order, names in order of first appearance
    if (p->pairing_filter == 0) {
        pool.run(C, [&](int c) {
            Tally &t = tl[(size_t)c];
            for (size_t i = c_lo(c); i < c_lo(c + 1); i++) {
                PairInfo &e = B.pairs[i];
                e.pass = false; e.in_filter = false;
                if (e.reads == 0 || absent(i)) continue;
                t.reads += e.reads; t.pairs += e.reads == 2; t.single += e.reads == 1;
                e.in_filter = e.reads == 2 || B.priority[i];
            }
        });
        for (const Tally &t : tl) { T.unfiltered_reads += t.reads; T.unfiltered_pairs += t.pairs; T.unfiltered_singletons += t.single; }
    } else if (B.cross_set) {
        // the cross-scaffold look-ups were resolved by the caller over ALL shares (isx_bam_set_cross_names): a name met k times
        // in header order is, for non_discordant, kept when k == 1 or it is a priority read, dropped from both scaffolds when
        // k == 2 and a KeyError of the reference when k > 2 (paired_read_filter :497-512); for all_reads every occurrence
        // carries the merged info the caller summed (_merge_info, :514-532)
        pool.run(C, [&](int c) {
            Tally &t = tl[(size_t)c];
            for (size_t i = c_lo(c); i < c_lo(c + 1); i++) {
                PairInfo &e = B.pairs[i];
                e.pass = false; e.in_filter = false;
                if (e.reads == 0 || absent(i)) continue;
                t.reads += e.reads; t.pairs += e.reads == 2; t.single += e.reads == 1;
                e.in_filter = true;
            }
        });
        for (const Tally &t : tl) { T.unfiltered_reads += t.reads; T.unfiltered_pairs += t.pairs; T.unfiltered_singletons += t.single; }
        for (size_t j = 0; j < B.cross_idx.size(); j++) {
            const size_t i = (size_t)B.cross_idx[j];
            PairInfo &e = B.pairs[i];
            if (e.reads == 0 || absent(i)) continue;
            if (p->pairing_filter == 1) {
                if (B.priority[i]) continue;
                if (B.cross_occ[j] > 2) { isx_set_error("non_discordant: a read name occurs on three scaffolds (the reference fails with KeyError here)"); return ISX_ERR_ARG; }
                e.in_filter = false;
            } else {
                e.nm = B.cross_info[4 * j]; e.mapq = B.cross_info[4 * j + 1]; e.length = B.cross_info[4 * j + 2]; e.reads = B.cross_info[4 * j + 3];
                e.insert = -2; e.start = -1; e.stop = -1;
            }
        }
    } else {
        for (PairInfo &e : B.pairs) { e.pass = false; e.in_filter = false; }
        // names are looked up across scaffolds (pair2scaffold)
        std::unordered_map<std::string_view, uint32_t> seen;        // name -> index of the entry that holds it now
        seen.reserve(B.pairs.size());
        for (size_t i = 0; i < B.pairs.size(); i++) {
            PairInfo &e = B.pairs[i];
            if (e.reads == 0 || absent(i)) continue;
            T.unfiltered_reads += e.reads; T.unfiltered_pairs += e.reads == 2; T.unfiltered_singletons += e.reads == 1;
            auto it = seen.find(name_of(e));
            if (p->pairing_filter == 1) {               // non_discordant
                if (it == seen.end() || B.priority[i]) { e.in_filter = true; seen[name_of(e)] = (uint32_t)i; }
                else {
                    PairInfo &o = B.pairs[it->second];
                    if (!o.in_filter) { isx_set_error("non_discordant: a read name occurs on three scaffolds (the reference fails with KeyError here)"); return ISX_ERR_ARG; }
                    o.in_filter = false;                // mapped to two scaffolds: discordant, gone from the first
                }
            } else {                                    // all_reads
                if (it == seen.end()) { e.in_filter = true; seen.emplace(name_of(e), (uint32_t)i); }
                else {
                    PairInfo &o = B.pairs[it->second];
                    PairInfo m = e;                     // _merge_info(i, stored)
                    m.nm = e.nm + o.nm; m.insert = -2; m.mapq = e.mapq + o.mapq; m.length = e.length + o.length;
                    m.reads = e.reads + o.reads; m.start = -1; m.stop = -1;
                    const PairInfo keep_e = e, keep_o = o;
                    e = m; e.name_seg = keep_e.name_seg; e.name_off = keep_e.name_off; e.name_len = keep_e.name_len; e.in_filter = true;
                    o = m; o.name_seg = keep_o.name_seg; o.name_off = keep_o.name_off; o.name_len = keep_o.name_len; o.in_filter = true;
                }
            }
        }
    }
    // median insert of the pairs that went through (reads == 2)
    double median = median_insert;
    if (std::isnan(median)) {
        // np.median of the inserts: a counting selection (inserts are a few hundred, the table covers [0, 65536)) on the threads;
        // only a file with inserts outside the table takes the general route (gather + nth_element)
        constexpr int64_t HB = 65536;
        std::vector<std::vector<uint32_t>> hist((size_t)C);
        std::vector<uint8_t> odd((size_t)C, 0);
        pool.run(C, [&](int c) {
            std::vector<uint32_t> &h = hist[(size_t)c];
            h.assign((size_t)HB, 0);
            for (size_t i = c_lo(c); i < c_lo(c + 1); i++) {
                const PairInfo &e = B.pairs[i];
                if (!e.in_filter || e.reads != 2) continue;
                if (e.insert >= 0 && e.insert < HB) h[(size_t)e.insert]++; else odd[(size_t)c] = 1;
            }
        });
        bool any_odd = false;
        for (uint8_t o : odd) any_odd = any_odd || o;
        if (!any_odd) {
            std::vector<uint64_t> tot((size_t)HB, 0);
            const int HT = std::min(C, 16);
            pool.run(HT, [&](int k) {
                for (size_t v = (size_t)HB * (size_t)k / (size_t)HT; v < (size_t)HB * ((size_t)k + 1) / (size_t)HT; v++) {
                    uint64_t a = 0;
                    for (int c = 0; c < C; c++) a += hist[(size_t)c][v];
                    tot[v] = a;
                }
            });
            uint64_t n = 0;
            for (uint64_t a : tot) n += a;
            if (n) {
                auto at_rank = [&](uint64_t r) -> int64_t { uint64_t a = 0; for (int64_t v = 0; v < HB; v++) { a += tot[(size_t)v]; if (a > r) return v; } return HB - 1; };
                const double hi = (double)at_rank(n / 2);
                median = (n & 1) ? hi : ((double)at_rank(n / 2 - 1) + hi) / 2.0;
            }
        } else {
            std::vector<std::vector<uint32_t>>().swap(hist);
            pool.run(C, [&](int c) {
                std::vector<int64_t> &v = tl[(size_t)c].ins;
                for (size_t i = c_lo(c); i < c_lo(c + 1); i++) { const PairInfo &e = B.pairs[i]; if (e.in_filter && e.reads == 2) v.push_back(e.insert); }
            });
            std::vector<int64_t> ins;
            size_t n_ins = 0;
            for (const Tally &t : tl) n_ins += t.ins.size();
            ins.reserve(n_ins);
            for (Tally &t : tl) { ins.insert(ins.end(), t.ins.begin(), t.ins.end()); std::vector<int64_t>().swap(t.ins); }
            if (!ins.empty()) {                             // np.median (selection, not a full sort)
                const size_t n = ins.size();
                std::nth_element(ins.begin(), ins.begin() + (ptrdiff_t)(n / 2), ins.end());
                const double hi = (double)ins[n / 2];
                median = (n & 1) ? hi : ((double)*std::max_element(ins.begin(), ins.begin() + (ptrdiff_t)(n / 2)) + hi) / 2.0;
            }
        }
    }
    T.median_insert = median;
    const double max_insert = median * p->max_insert_relative;
    B.ref_filtered_pairs.assign(n_ref, 0);
    std::vector<std::vector<std::pair<size_t, int64_t>>> per_ref((size_t)C);       // (reference, pairs that passed) per piece
    pool.run(C, [&](int c) {
        Tally &tc = tl[(size_t)c];
        size_t i = c_lo(c);
        const size_t end = c_lo(c + 1);
        size_t t = (size_t)(std::upper_bound(B.ref_pair0.begin(), B.ref_pair0.end(), (uint64_t)i) - B.ref_pair0.begin()) - 1;
        while (i < end) {
            while (t + 1 < B.ref_pair0.size() && B.ref_pair0[t + 1] <= i) t++;
            const size_t stop = std::min<size_t>(end, (size_t)B.ref_pair0[t + 1]);
            int64_t passed = 0;
            for (; i < stop; i++) {
                PairInfo &e = B.pairs[i];
                if (!e.in_filter) continue;
                const double pid = 1 - ((double)e.nm / (double)e.length);        // evaluate_pair :406
                bool ok = pid > p->min_read_ani;
                ok = ok && (e.mapq > p->min_mapq);
                if (e.reads == 2 && e.insert != -1) ok = ok && ((double)e.insert > (double)p->min_insert) && ((double)e.insert < max_insert);
                e.pass = ok;
                if (ok) {
                    e.mm = (int32_t)e.nm;
                    tc.f_pairs++; tc.f_bases += e.length; tc.f_single += e.reads == 1;
                    passed++;
                    if (e.nm > tc.max_mm) tc.max_mm = e.nm;
                }
            }
            if (passed) per_ref[(size_t)c].push_back({t, passed});
        }
    });
    int64_t max_mm = 0;
    for (int c = 0; c < C; c++) {
        const Tally &t = tl[(size_t)c];
        T.filtered_pairs += t.f_pairs; T.filtered_bases += t.f_bases; T.filtered_singletons += t.f_single;
        max_mm = std::max(max_mm, t.max_mm);
        for (const auto &pr : per_ref[(size_t)c]) B.ref_filtered_pairs[pr.first] += pr.second;
    }
    if (max_mm > 65535) { isx_set_error("mm level > 65535"); return ISX_ERR_ARG; }
    T.max_mm = p->skip_mm ? 0 : (int32_t)max_mm;
    B.totals = T;
    B.filtered = true;
    if (info) *info = T;
    return ISX_OK;
}

// The controller's own R2M for one reference (profile_controller.py:415-433 hands sR2M[scaffold] to every split):
// exactly the named pairs pass, with the given mm.  Replaces what isx_bam_filter decided for that reference.
int isx_bam_set_r2m(isx_bam *bam, int32_t ref, int64_t n, const char *names, const int64_t *offs, const int32_t *mm)
{
    if (!bam || ref < 0 || (size_t)ref >= bam->ref_name.size() || n < 0 || (n && (!names || !offs))) { isx_set_error("isx_bam_set_r2m: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    if (!B.scanned) { isx_set_error("isx_bam_set_r2m: scan first"); return ISX_ERR_STATE; }
    if (B.seg_names.empty() && !B.pairs.empty()) { isx_set_error("isx_bam_set_r2m: the read names were already dropped (isx_bam_drop_names)"); return ISX_ERR_STATE; }
    std::unordered_map<std::string_view, int32_t> want;
    want.reserve((size_t)n * 2);
    for (int64_t i = 0; i < n; i++) {
        const int32_t v = mm ? mm[i] : 0;
        if (v < 0 || v > 65535) { isx_set_error("isx_bam_set_r2m: mm out of range"); return ISX_ERR_ARG; }
        want.emplace(std::string_view(names + offs[i], (size_t)(offs[i + 1] - offs[i])), v);
    }
    int64_t kept = 0, max_mm = B.totals.max_mm;
    for (uint64_t i = B.ref_pair0[(size_t)ref]; i < B.ref_pair0[(size_t)ref + 1]; i++) {
        PairInfo &e = B.pairs[(size_t)i];
        auto it = want.find(std::string_view(B.seg_names[e.name_seg].data() + e.name_off, e.name_len));
        e.pass = it != want.end();
        if (e.pass) { e.mm = it->second; kept++; max_mm = std::max<int64_t>(max_mm, e.mm); }
    }
    if (B.ref_filtered_pairs.size() != B.ref_name.size()) B.ref_filtered_pairs.assign(B.ref_name.size(), 0);
    B.ref_filtered_pairs[(size_t)ref] = kept;
    B.totals.max_mm = (int32_t)max_mm;
    B.filtered = true;
    return ISX_OK;
}

// What the filter decided for one reference: the reference's Rdic[scaffold] (pair name -> mm; controller.py:274-281
// stores it with the profile).  Call with names == NULL to get the sizes first.
int isx_bam_r2m(const isx_bam *bam, int32_t ref, int64_t *n, int64_t *name_bytes, char *names, int64_t *offs, int32_t *mm)
{
    if (!bam || ref < 0 || (size_t)ref >= bam->ref_name.size() || !n || !name_bytes) { isx_set_error("isx_bam_r2m: bad argument"); return ISX_ERR_ARG; }
    const isx_bam &B = *bam;
    if (!B.filtered) { isx_set_error("isx_bam_r2m: filter first"); return ISX_ERR_STATE; }
    if (B.seg_names.empty() && !B.pairs.empty()) { isx_set_error("isx_bam_r2m: the read names were already dropped"); return ISX_ERR_STATE; }
    int64_t k = 0, nb = 0;
    for (uint64_t i = B.ref_pair0[(size_t)ref]; i < B.ref_pair0[(size_t)ref + 1]; i++) {
        const PairInfo &e = B.pairs[(size_t)i];
        if (!e.pass || e.reads == 0) continue;
        if (names && offs && mm) {
            offs[k] = nb;
            memcpy(names + nb, B.seg_names[e.name_seg].data() + e.name_off, e.name_len);
            mm[k] = e.mm;
        }
        nb += e.name_len; k++;
    }
    if (names && offs) offs[k] = nb;
    *n = k; *name_bytes = nb;
    return ISX_OK;
}

// Pairs with more than `cap` mismatches are piled up at level `cap` (a device batch holds 128 mm levels; the reference bins any
// mm, profile_utilities.py:268-286): the caller warns -- the alternative is to fail the whole call.  isx_bam_r2m keeps the true values.
int isx_bam_set_mm_cap(isx_bam *bam, int32_t cap)
{
    if (!bam || cap < 0) { isx_set_error("isx_bam_set_mm_cap: bad argument"); return ISX_ERR_ARG; }
    bam->mm_cap = cap;
    return ISX_OK;
}

// Round 6: mm levels beyond 127 binned exactly.  The reference bins any mm (profile_utilities.py:268-286) and every table it makes
// depends on the levels' ORDER alone (cumulative counts over the levels <= mm); a device batch holds 128 levels.  isx_bam_mm_levels
// lists the distinct mm values of the kept pairs, isx_bam_set_mm_levels makes the pairs travel with the rank of their value in that
// list -- exact whenever no more than 128 DIFFERENT values occur, whatever their size (isx_bam_set_mm_cap then merges the ranks beyond).
int isx_bam_mm_levels(const isx_bam *bam, int32_t *levels, int32_t cap, int32_t *n)
{
    if (!bam || !n || cap < 0 || (cap && !levels)) { isx_set_error("isx_bam_mm_levels: bad argument"); return ISX_ERR_ARG; }
    std::vector<int32_t> v;
    for (const PairInfo &e : bam->pairs) if (e.pass && e.reads != 0) v.push_back(e.mm);
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    *n = (int32_t)std::min<size_t>(v.size(), 0x7FFFFFFF);
    for (int32_t i = 0; i < std::min<int32_t>(*n, cap); i++) levels[i] = v[(size_t)i];
    return ISX_OK;
}

int isx_bam_set_mm_levels(isx_bam *bam, const int32_t *levels, int32_t n)
{
    if (!bam || n < 0 || (n && !levels)) { isx_set_error("isx_bam_set_mm_levels: bad argument"); return ISX_ERR_ARG; }
    for (int32_t i = 1; i < n; i++) if (levels[i] <= levels[i - 1]) { isx_set_error("isx_bam_set_mm_levels: the values must ascend"); return ISX_ERR_ARG; }
    bam->mm_levels.assign(levels, levels + n);
    return ISX_OK;
}

// names of the read pairs of the batch prepared LAST (isx_bam_expand_refs / isx_bam_segment_refs / isx_pipe_submit_bam), by dense pair
// id -- the ids the device's tables (isx_ao.pair) carry.  Needs the names (no isx_bam_drop_names before the batch was prepared).
int isx_bam_batch_pair_names(const isx_bam *bam, int64_t *n, int64_t *name_bytes, char *names, int64_t *offs)
{
    if (!bam || !n || !name_bytes) { isx_set_error("isx_bam_batch_pair_names: bad argument"); return ISX_ERR_ARG; }
    const isx_bam &B = *bam;
    if (B.seg_names.empty() && !B.last_dense_pair.empty()) { isx_set_error("isx_bam_batch_pair_names: the read names were dropped"); return ISX_ERR_STATE; }
    int64_t nb = 0;
    const int64_t k = (int64_t)B.last_dense_pair.size();
    for (int64_t i = 0; i < k; i++) {
        const PairInfo &e = B.pairs[B.last_dense_pair[(size_t)i]];
        if (names && offs) {
            offs[i] = nb;
            memcpy(names + nb, B.seg_names[e.name_seg].data() + e.name_off, e.name_len);
        }
        nb += e.name_len;
    }
    if (names && offs) offs[k] = nb;
    *n = k; *name_bytes = nb;
    return ISX_OK;
}

// free the read names once no isx_bam_set_r2m / isx_bam_filter call will follow (they are the bulk of what the scan keeps)
int isx_bam_drop_names(isx_bam *bam)
{
    if (!bam) { isx_set_error("isx_bam_drop_names: bad argument"); return ISX_ERR_ARG; }
    size_t bytes = 0;
    for (const auto &v : bam->seg_names) bytes += v.size();
    if (bytes <= isx_bam::KEEP_DEAD && bam->dead_names.empty()) bam->dead_names.swap(bam->seg_names);
    std::vector<uvec<char>>().swap(bam->seg_names);
    return ISX_OK;
}

int isx_bam_ref_counts(const isx_bam *bam, int64_t *reads, int64_t *filtered_pairs)
{
    if (!bam || !bam->scanned) { isx_set_error("isx_bam_ref_counts: scan first"); return ISX_ERR_STATE; }
    const size_t n = bam->ref_name.size();
    for (size_t t = 0; t < n; t++) {
        if (reads) reads[t] = bam->ref_reads[t];
        if (filtered_pairs) filtered_pairs[t] = bam->ref_filtered_pairs.size() == n ? bam->ref_filtered_pairs[t] : 0;
    }
    return ISX_OK;
}

// ---- pass 2: overlap resolution + expansion of a subset of the references ----
}  // extern "C"

// A batch of references loaded, overlap-resolved and counted: any range of its observation stream can then be
// produced on demand (isx_bam_expand_refs writes all of it into the handle; isx_pipe_submit_bam lets the pipe's
// encoder pull it group by group, so the 8-byte records of a batch never exist as a whole).
struct BamBatch {
    isx_bam *B = nullptr;
    isx_bam_params prm{};
    Batch S;
    uvec<uint8_t> emit;
    uvec<uint32_t> pid;              // dense pair id per read
    uvec<uint8_t> pair_mm;           // mm profiling on: min(mm, 255) of the pairs [pair_mm_lo, ...) of the batch's references -- a 1-byte table the
    size_t pair_mm_lo = 0;           // emission looks a read's level up in (the PairInfo table is 30 x larger: a cache miss a read)
    uvec<uint64_t> out_at;           // [n_reads + 1] first observation of every read
    std::vector<int64_t> boff;              // per reference of the file: offset in the batch's flat space, -1 = not in the batch
    int64_t n_pos = 0;
    int64_t reg_lo = 0, reg_hi = -1;        // one-reference batches: only positions [reg_lo, reg_hi) are piled up (-1 = all)
    uint32_t next_pair = 0;
    uint8_t minq = 30;
    std::vector<int64_t> split_bounds;
    std::vector<int32_t> split_ref;

    // how many observations read ri contributes (bases of M/=/X blocks inside the scaffold / region with quality >= minq)
    uint64_t count_read(size_t ri) const
    {
        const Read &r = S.reads[ri];
        const int64_t ref_len = B->ref_len[(size_t)r.tid];
        const uint8_t *ql = r.qual;
        const uint8_t mq = minq;
        int64_t ref = r.pos, q = 0;
        uint64_t n_out = 0;
        for (uint32_t k = 0; k < r.n_cigar; k++) {
            const uint32_t c = S.cigars[r.cigar_off + (uint64_t)k];
            const int op = c & 15;
            const int64_t n = c >> 4;
            if (op == CM || op == CEQ || op == CX) {
                // the reference's pileups are truncated to [0, scaffold length) (profile_utilities.py:150-153)
                int64_t j0 = std::max<int64_t>(0, -ref), j1 = std::min<int64_t>(n, ref_len - ref);
                if (reg_hi >= 0) { j0 = std::max<int64_t>(j0, reg_lo - ref); j1 = std::min<int64_t>(j1, reg_hi - ref); }
                uint32_t m = 0;
                for (int64_t j = j0; j < j1; j++) m += ql[q + j] >= mq;
                n_out += m;
                q += n; ref += n;
            } else if (op == CI || op == CS) q += n;
            else if (op == CD || op == CN) ref += n;
        }
        return n_out;
    }

    // every observation of read ri -> out (room for l_seq + 1 records: the store is unconditional, the cursor advances only
    // for a base that counts); returns how many
    uint32_t walk_all(size_t ri, isx_obs *out) const
    {
        const Read &r = S.reads[ri];
        const PairInfo &pi = B->pairs[r.pair_idx];
        const uint64_t hi = (uint64_t)(prm.skip_mm ? 0 : (uint16_t)B->level_of(pi.mm)) << 32;
        const int64_t base_off = boff[(size_t)r.tid];
        const int64_t ref_len = B->ref_len[(size_t)r.tid];
        const uint8_t *ql = r.qual, *sq = r.seq;
        const uint8_t mq = minq;
        uint64_t *o64 = reinterpret_cast<uint64_t *>(out);
        int64_t ref = r.pos, q = 0;
        uint32_t n_out = 0;
        for (uint32_t k = 0; k < r.n_cigar; k++) {
            const uint32_t c = S.cigars[r.cigar_off + (uint64_t)k];
            const int op = c & 15;
            const int64_t n = c >> 4;
            if (op == CM || op == CEQ || op == CX) {
                int64_t j0 = std::max<int64_t>(0, -ref), j1 = std::min<int64_t>(n, ref_len - ref);
                if (reg_hi >= 0) { j0 = std::max<int64_t>(j0, reg_lo - ref); j1 = std::min<int64_t>(j1, reg_hi - ref); }
                const uint64_t g0 = (uint64_t)(uint32_t)(base_off + ref);
                for (int64_t j = j0; j < j1; j++) {
                    const int64_t i = q + j;
                    o64[n_out] = (g0 + (uint64_t)j) | hi | ((uint64_t)CODE2IDX[nib(sq, i)] << 48);    // gpos | mm << 32 | base << 48
                    n_out += ql[i] >= mq;
                }
                q += n; ref += n;
            } else if (op == CI || op == CS) q += n;
            else if (op == CD || op == CN) ref += n;
        }
        return n_out;
    }

    // ---- read-level hand-over: the batch as read segments (isx_segs) instead of observations ----
    uvec<uint64_t> seg_at;           // [n_reads + 1] first segment of every read
    uvec<uint32_t> seg_gpos;                // [n_segs] flat start of every segment (the staging encoder's layout pass wants them up front)
    int64_t n_seg_bases = 0;                // columns covered by the segments (>= the observations)

    // calls f(flat start, query offset, columns) for every segment of read ri: the M / = / X runs of its CIGAR, truncated to
    // the scaffold / region like the reference's pileup (profile_utilities.py:150-153), cut every ISX_SEG_BASES columns
    template <class F>
    void for_segments(size_t ri, F &&f) const
    {
        const Read &r = S.reads[ri];
        const int64_t base_off = boff[(size_t)r.tid];
        const int64_t ref_len = B->ref_len[(size_t)r.tid];
        int64_t ref = r.pos, q = 0;
        for (uint32_t k = 0; k < r.n_cigar; k++) {
            const uint32_t c = S.cigars[r.cigar_off + (uint64_t)k];
            const int op = c & 15;
            const int64_t n = c >> 4;
            if (op == CM || op == CEQ || op == CX) {
                int64_t j0 = std::max<int64_t>(0, -ref), j1 = std::min<int64_t>(n, ref_len - ref);
                if (reg_hi >= 0) { j0 = std::max<int64_t>(j0, reg_lo - ref); j1 = std::min<int64_t>(j1, reg_hi - ref); }
                for (int64_t c0 = j0; c0 < j1; c0 += ISX_SEG_BASES)
                    f(base_off + ref + c0, q + c0, std::min<int64_t>(ISX_SEG_BASES, j1 - c0));
                q += n; ref += n;
            } else if (op == CI || op == CS) q += n;
            else if (op == CD || op == CN) ref += n;
        }
    }

    // segments [first, first + count) of the batch's stream (thread safe); pair may be NULL
    void emit_segs(int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases) const
    {
        size_t ri = (size_t)(std::upper_bound(seg_at.begin(), seg_at.end(), (uint64_t)first) - seg_at.begin()) - 1;
        int64_t skip = first - (int64_t)seg_at[ri], done = 0;
        const bool fast = cpu_has_avx2_bmi2();
        const uint8_t mq = minq;
        alignas(32) uint8_t cd[192];
        for (; done < count; ri++) {
            if (seg_at[ri + 1] == seg_at[ri]) continue;
            const Read &r = S.reads[ri];
            const uint8_t m = prm.skip_mm ? (uint8_t)0 : (uint8_t)std::min<int32_t>(std::min<int32_t>(255, B->mm_cap), (int32_t)pair_mm[(size_t)r.pair_idx - pair_mm_lo]);
            const uint32_t id = pid[ri];
            for_segments(ri, [&](int64_t g, int64_t q0, int64_t cols) {
                if (skip > 0) { skip--; return; }
                if (done >= count) return;
                if (fast) { seg_codes_avx2(r.seq, r.qual, q0, (int)cols, mq, cd); seg_pack_bmi2(cd, (int)cols, bases + (size_t)done * ISX_SEG_WORDS); }
                else { seg_codes_scalar(r.seq, r.qual, q0, (int)cols, mq, cd); seg_pack_scalar(cd, (int)cols, bases + (size_t)done * ISX_SEG_WORDS); }
                gpos[done] = (uint32_t)g; len[done] = (uint8_t)cols; mm[done] = m;
                if (pair) pair[done] = id;
                done++;
            });
            skip = 0;
        }
    }

    // the same segments as bit planes (isx_read_planes: one 64-byte line each); mm != NULL (mm profiling on): the pairs' levels too, and
    // the non-ACGT bases that pass the filter are marked in the lines
    void emit_planes(int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint64_t *planes) const
    {
        size_t ri = (size_t)(std::upper_bound(seg_at.begin(), seg_at.end(), (uint64_t)first) - seg_at.begin()) - 1;
        int64_t skip = first - (int64_t)seg_at[ri], done = 0;
        const bool fast = cpu_has_avx512bw();
        const uint8_t mq = minq;
        for (; done < count; ri++) {
            if (seg_at[ri + 1] == seg_at[ri]) continue;
            const Read &r = S.reads[ri];
            const uint32_t id = pid[ri];
            const bool mark = mm != nullptr && !prm.skip_mm;
            const uint8_t m = mark ? (uint8_t)std::min<int32_t>(std::min<int32_t>(127, B->mm_cap), (int32_t)pair_mm[(size_t)r.pair_idx - pair_mm_lo]) : (uint8_t)0;
            for_segments(ri, [&](int64_t g, int64_t q0, int64_t cols) {
                if (skip > 0) { skip--; return; }
                if (done >= count) return;
                uint64_t *P = planes + (size_t)done * ISX_PLANE_WORDS;
                if (fast) seg_planes_avx512(r.seq, r.qual, q0, (int)cols, mq, P, mark);
                else seg_planes_scalar(r.seq, r.qual, q0, (int)cols, mq, P, mark);
                gpos[done] = (uint32_t)g; len[done] = (uint8_t)cols;
                if (mm) mm[done] = m;
                if (pair) pair[done] = id;
                done++;
            });
            skip = 0;
        }
    }

    // observations [first, first + count) of the batch's stream (thread safe)
    void emit_range(int64_t first, uint32_t count, isx_obs *po, uint32_t *pp) const
    {
        static_assert(sizeof(isx_obs) == 8, "isx_obs is one 64-bit word");
        size_t ri = (size_t)(std::upper_bound(out_at.begin(), out_at.end(), (uint64_t)first) - out_at.begin()) - 1;
        uint64_t skip = (uint64_t)first - out_at[ri];
        uint32_t done = 0;
        thread_local std::vector<isx_obs> tmp;
        while (done < count) {
            const uint64_t have = out_at[ri + 1] - out_at[ri];
            if (have > skip) {
                const uint32_t take = (uint32_t)std::min<uint64_t>(count - done, have - skip);
                if (skip == 0 && (uint64_t)(count - done) > have) walk_all(ri, po + done);      // whole read, room for the spare store
                else {
                    const size_t need = (size_t)std::max(S.reads[ri].l_seq, 0) + 1;
                    if (tmp.size() < need) tmp.resize(need + 256);
                    walk_all(ri, tmp.data());
                    memcpy(po + done, tmp.data() + skip, (size_t)take * sizeof(isx_obs));
                }
                if (pp) std::fill(pp + done, pp + done + take, pid[ri]);
                done += take;
            }
            skip = 0;
            ri++;
        }
    }
    int64_t n_obs() const { return (int64_t)out_at.back(); }
};

int bam_batch_prepare(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, BamBatch **out, int64_t reg_lo, int64_t reg_hi,
                      bool as_segments)
{
    *out = nullptr;
    isx_bam &B = *bam;
    if (!B.scanned || !B.filtered) { isx_set_error("isx_bam_expand_refs: scan and filter first"); return ISX_ERR_STATE; }
    const size_t n_ref_all = B.ref_name.size();
    isxenc::HostPool &pool = pool_of(B);
    const bool timing = getenv("ISX_BAM_TIMING") != nullptr;       // tuning aid: stage times on stderr (no effect on results)
    auto t_last = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[isx_bam_expand_refs] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::unique_ptr<BamBatch> Q(new BamBatch());
    Q->B = bam; Q->prm = *p;
    Q->minq = (uint8_t)std::min(255, std::max(0, p->min_base_quality));
    // flat space of the batch: its references laid end to end (file order)
    std::vector<int64_t> &boff = Q->boff;
    boff.assign(n_ref_all, -1);
    int64_t n_pos = 0;
    std::vector<uint8_t> seg_wanted(B.segs.size(), 0);
    for (int32_t i = 0; i < n_refs; i++) {
        const int32_t t = refs[i];
        if (t < 0 || (size_t)t >= n_ref_all || (i > 0 && t <= refs[i - 1])) { isx_set_error("isx_bam_expand_refs: reference ids must be valid and ascending (file order: the stream stays position-clustered)"); return ISX_ERR_ARG; }
        boff[(size_t)t] = n_pos;
        n_pos += B.ref_len[(size_t)t];
        if (B.ref_reads[(size_t)t])
            for (uint32_t s = B.ref_seg0[(size_t)t]; s <= B.ref_seg1[(size_t)t]; s++) seg_wanted[s] = 1;
    }
    if (n_pos >= (int64_t)0xFFFF0000ll) { isx_set_error("isx_bam_expand_refs: the batch's flat space must be < 2^32 - 65536 positions"); return ISX_ERR_ARG; }
    Q->n_pos = n_pos;
    const bool region = reg_hi >= 0;
    if (region) {
        if (n_refs != 1 || reg_lo < 0 || reg_hi <= reg_lo) { isx_set_error("isx_bam_expand_region: one reference and 0 <= start < stop"); return ISX_ERR_ARG; }
        Q->reg_lo = reg_lo; Q->reg_hi = std::min<int64_t>(reg_hi, B.ref_len[(size_t)refs[0]]);
        // the file is sorted: segments whose records all start at or after the region's end, or so far before its start that
        // no read reaches it, cannot contribute
        for (uint32_t s = 0; s < B.segs.size(); s++) {
            if (!seg_wanted[s]) continue;
            const Segment &sg = B.segs[s];
            if (sg.tid_first == refs[0] && sg.pos_first >= Q->reg_hi) seg_wanted[s] = 0;
            if (sg.tid_last == refs[0] && (int64_t)sg.pos_last + B.max_span <= Q->reg_lo) seg_wanted[s] = 0;
        }
    }
    std::vector<uint32_t> seg_list;
    for (uint32_t s = 0; s < B.segs.size(); s++) if (seg_wanted[s]) seg_list.push_back(s);

    // ---- load the batch's reads: inflate + walk + count per segment, then fill ----
    struct SegWork { SegBuf buf; std::vector<uint64_t> rec; const uint8_t *data = nullptr; const std::vector<uint64_t> *recp = nullptr;
                     std::vector<uint32_t> keep; uint64_t n_cig = 0, n_seq = 0; std::string err; };
    std::vector<SegWork> sw(seg_list.size());
    std::atomic<int> rc_any{0};
    pool.run((int)seg_list.size(), [&](int k) {
        const Segment &s = B.segs[seg_list[(size_t)k]];
        SegWork &w = sw[(size_t)k];
        const bool cached = !B.seg_cache.empty() && !B.seg_cache[seg_list[(size_t)k]].empty();
        if (cached) { w.data = B.seg_cache[seg_list[(size_t)k]].data(); w.recp = &B.seg_cache_rec[seg_list[(size_t)k]]; }
        else {
            Inflater inf;
            uint64_t next = 0;
            if (!seg_inflate(B, inf, s, w.buf)) { w.err = "BGZF inflate failed"; rc_any.store(ISX_ERR_IO); return; }
            if (s.n_reads && (seg_hop(B, inf, s, w.buf, s.first_rec, w.rec, next) != ISX_OK || w.rec.size() != s.n_reads)) { w.err = "BAM changed between scan and expand"; rc_any.store(ISX_ERR_IO); return; }
            w.data = w.buf.data.data(); w.recp = &w.rec;
        }
        if (s.n_reads == 0) return;
        for (uint32_t i = 0; i < s.n_reads; i++) {
            const uint8_t *q = w.data + ((*w.recp)[i] - s.ioff0);
            const int32_t tid = rd32(q + 4);
            if (tid < 0 || (size_t)tid >= n_ref_all || boff[(size_t)tid] < 0) continue;
            if (rd16(q + 18) & DEF_MASK) continue;                     // htslib's pileup never sees these
            if (region) {                                               // the fetch of a region only yields reads that overlap it
                const int32_t pos = rd32(q + 8);
                if (pos >= Q->reg_hi || (int64_t)pos + B.max_span <= Q->reg_lo) continue;
            }
            w.keep.push_back(i);
            if (rd16(q + 16) == 2) {                                    // maybe the placeholder of a CIGAR kept in the CG tag
                RecView r;
                if (!rec_view(q, r)) { w.err = "corrupt BAM record"; rc_any.store(ISX_ERR_IO); return; }
                w.n_cig += r.n_cigar;
            } else w.n_cig += rd16(q + 16);
            w.n_seq += (uint64_t)std::max(rd32(q + 20), 0);
        }
    });
    if (rc_any.load()) { for (auto &w : sw) if (!w.err.empty()) { isx_set_error(w.err); break; } return rc_any.load(); }
    std::vector<uint64_t> r_at(sw.size() + 1, 0), c_at(sw.size() + 1, 0), s_at(sw.size() + 1, 0);
    for (size_t k = 0; k < sw.size(); k++) { r_at[k + 1] = r_at[k] + sw[k].keep.size(); c_at[k + 1] = c_at[k] + sw[k].n_cig; s_at[k + 1] = s_at[k] + sw[k].n_seq; }
    Batch &S = Q->S;
    S.reads.resize((size_t)r_at.back());
    S.cigars.resize((size_t)c_at.back());
    S.seg_data.resize(sw.size());
    pool.run((int)sw.size(), [&](int k) {
        const Segment &s = B.segs[seg_list[(size_t)k]];
        SegWork &w = sw[(size_t)k];
        uint64_t ci = c_at[(size_t)k];
        if (w.keep.empty()) { SegBuf().data.swap(w.buf.data); return; }
        // the batch owns the inflated segment from here on (overlap resolution writes qualities).  A segment the handle kept
        // inflated since the scan is handed over when no reference outside this batch has reads in it (nobody will ask
        // for it again; if somebody does it is inflated anew), copied otherwise.
        if (w.buf.data.empty()) {
            uvec<uint8_t> &kept = B.seg_cache[seg_list[(size_t)k]];
            bool others = region;
            for (int32_t t = std::max(s.tid_first, 0); t <= s.tid_last && !others; t++)
                if (B.ref_reads[(size_t)t] > 0 && boff[(size_t)t] < 0) others = true;
            if (others) w.buf.data = kept;
            else w.buf.data.swap(kept);             // `kept` is now empty: a later request inflates the segment again
        }
        S.seg_data[(size_t)k].swap(w.buf.data);
        uint8_t *own = S.seg_data[(size_t)k].data();
        for (size_t j = 0; j < w.keep.size(); j++) {
            RecView r;
            if (!rec_view(own + ((*w.recp)[w.keep[j]] - s.ioff0), r)) { w.err = "corrupt BAM record"; rc_any.store(ISX_ERR_IO); return; }
            Read R{};
            R.tid = r.tid; R.pos = r.pos; R.isize = r.isize; R.l_seq = r.l_seq; R.flag = r.flag; R.n_cigar = r.n_cigar;
            R.pair_idx = B.read_pair[(size_t)(s.read0 + w.keep[j])];
            R.cigar_off = ci;
            memcpy(S.cigars.data() + ci, r.cigar, (size_t)r.n_cigar * 4);
            R.seq = r.seq; R.qual = const_cast<uint8_t *>(r.qual);
            R.ref_end = span_of(r.cigar, r.n_cigar, r.pos).end;
            S.reads[(size_t)(r_at[(size_t)k] + j)] = R;
            ci += r.n_cigar;
        }
    });
    if (rc_any.load()) { for (auto &w : sw) if (!w.err.empty()) { isx_set_error(w.err); break; } return rc_any.load(); }
    sw.clear();
    stage("load reads");

    // ---- max_depth = 100000 of the pileup call (profile_utilities.py:150, polymorpher.py:290) as htslib 1.9 applies it
    //      (bam_plp_push, sam.c): a read that starts exactly at the column the iterator stands on -- i.e. any read but the first
    //      of a run of equal starts -- is not pushed while the buffer's node pool holds more than max_depth nodes (the buffered
    //      reads, whose end lies at or beyond that start, + the list's sentinel).  Dropped reads take no part in the overlap
    //      resolution and reach no column.  Only a position with >= 100000 reads over it can drop anything: a parallel screen
    //      (reads starting within the longest read span of each start) decides whether the serial replay runs at all.
    //      PARITY UNPINNED (no reference fixture is that deep; restated from the htslib-1.9 source; the tests hold a second, pure-Python restatement,
    //      which replays every split literally: its own reads, its own buffer, its own columns).
    {
        constexpr int64_t MAX_DEPTH = 100000;
        const size_t n_all = S.reads.size();
        if (n_all > (size_t)MAX_DEPTH) {
            const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n_all / 65536 + 1));
            std::vector<int64_t> span_max((size_t)nt, 0);
            pool.run(nt, [&](int t) {
                int64_t m = 0;
                for (size_t i = n_all * (size_t)t / (size_t)nt; i < n_all * (size_t)(t + 1) / (size_t)nt; i++) m = std::max(m, S.reads[i].ref_end - (int64_t)S.reads[i].pos);
                span_max[(size_t)t] = m;
            });
            const int64_t L = std::max<int64_t>(1, *std::max_element(span_max.begin(), span_max.end()));
            std::atomic<int> deep{0};
            pool.run(nt, [&](int t) {           // reads of the same reference that start in (pos - L, pos]: an upper bound of the buffer there
                const size_t a = n_all * (size_t)t / (size_t)nt, e = n_all * (size_t)(t + 1) / (size_t)nt;
                size_t lo = a;
                while (lo > 0 && S.reads[lo - 1].tid == S.reads[a].tid && (int64_t)S.reads[lo - 1].pos > (int64_t)S.reads[a].pos - L) lo--;
                for (size_t i = a; i < e; i++) {
                    while (S.reads[lo].tid != S.reads[i].tid || (int64_t)S.reads[lo].pos <= (int64_t)S.reads[i].pos - L) lo++;
                    if ((int64_t)(i - lo + 1) >= MAX_DEPTH) { deep.store(1); return; }
                }
            });
            if (deep.load()) {
                // Round 6: the replay is restarted PER SPLIT, as the reference's iterator is (profile_utilities.py:150-153: one
                // samfile.pileup(..., max_depth=100000, start=start, stop=end+1) per split, fasta.py:56-73 iterate_splits): a split's
                // iterator is fed the reads that overlap [start, end] (pos <= end, bam_endpos > start) and starts with an empty buffer.
                // A read that overlaps two splits is replayed in both and can be dropped in one and taken in the other (a pile
                // within a read's length before a split bound: the buffer of the later split's iterator never saw the reads that
                // end before the bound).  Such a read keeps its place in the stream; its CIGAR is rewritten so that it covers no
                // column of the splits that dropped it -- M/=/X there become I + N, D becomes N -- and everything downstream
                // (overlap resolution, expansion) sees a read that is absent exactly there.  Dropped everywhere: removed as before.
                const int64_t WL = p->window_length > 0 ? p->window_length : 10000;
                std::priority_queue<int64_t, std::vector<int64_t>, std::greater<int64_t>> ends;
                std::vector<uint16_t> n_seen(n_all, 0), n_drop(n_all, 0);
                std::vector<std::pair<uint32_t, int32_t>> drops;        // (read, split ordinal on its reference)
                size_t i0 = 0;
                while (i0 < n_all) {
                    const int32_t tid = S.reads[i0].tid;
                    size_t i1 = i0;
                    while (i1 < n_all && S.reads[i1].tid == tid) i1++;
                    const int64_t sLen = tid >= 0 ? B.ref_len[(size_t)tid] : 0;
                    if (sLen > 0) {
                        const int64_t nC = sLen / WL + 1, chunk = (int64_t)((double)sLen / (double)nC);
                        size_t lo = i0;
                        for (int64_t c = 0; c < nC; c++) {
                            const int64_t s0 = c * chunk, e0 = c + 1 == nC ? sLen - 1 : (c + 1) * chunk - 1;
                            while (lo < i1 && (int64_t)S.reads[lo].pos + L <= s0) lo++;      // (cannot reach the split: ref_end <= pos + L)
                            ends = decltype(ends)();
                            int32_t prev_pos = -1;
                            for (size_t i = lo; i < i1 && (int64_t)S.reads[i].pos <= e0; i++) {
                                const Read &r = S.reads[i];
                                if (std::max<int64_t>(r.ref_end, (int64_t)r.pos + 1) <= s0) continue;       // not fetched for this split
                                while (!ends.empty() && ends.top() < (int64_t)r.pos) ends.pop();
                                if (n_seen[i] < 0xFFFF) n_seen[i]++;
                                if (prev_pos == r.pos && (int64_t)ends.size() + 1 > MAX_DEPTH) { if (n_drop[i] < 0xFFFF) n_drop[i]++; drops.emplace_back((uint32_t)i, (int32_t)c); continue; }
                                prev_pos = r.pos;
                                ends.push(r.ref_end);
                            }
                        }
                    }
                    i0 = i1;
                }
                // what the drops mean for each read
                std::vector<uint32_t> extra;                            // rewritten CIGARs, appended to S.cigars below
                const size_t old_n = S.cigars.size();
                std::sort(drops.begin(), drops.end());
                for (size_t d = 0; d < drops.size();) {
                    const uint32_t ri = drops[d].first;
                    size_t d1 = d;
                    while (d1 < drops.size() && drops[d1].first == ri) d1++;
                    Read &r = S.reads[ri];
                    if (n_drop[ri] >= n_seen[ri]) { r.pair_idx = 0xFFFFFFFFu; r.flag |= FUNMAP; d = d1; continue; }     // no iterator took it
                    const int64_t sLen = B.ref_len[(size_t)r.tid];
                    const int64_t nC = sLen / WL + 1, chunk = (int64_t)((double)sLen / (double)nC);
                    auto split_of = [&](int64_t rp) { return chunk > 0 ? std::min<int64_t>(rp / chunk, nC - 1) : 0; };
                    auto split_end = [&](int64_t c) { return c + 1 == nC ? sLen - 1 : (c + 1) * chunk - 1; };
                    auto is_dropped = [&](int64_t c) { for (size_t k = d; k < d1; k++) if (drops[k].second == (int32_t)c) return true; return false; };
                    const uint64_t new_off = (uint64_t)old_n + extra.size();
                    int64_t rp = r.pos;
                    for (uint32_t k = 0; k < r.n_cigar; k++) {
                        const uint32_t cg = S.cigars[r.cigar_off + (uint64_t)k];
                        const uint32_t op = cg & 15u;
                        int64_t n = (int64_t)(cg >> 4);
                        const bool m_like = op == CM || op == CEQ || op == CX, d_like = op == CD || op == CN;
                        if (!m_like && !d_like) { extra.push_back(cg); continue; }           // consumes no reference: as it is
                        while (n > 0) {             // the operation cut at the split bounds it crosses
                            const int64_t c = rp >= 0 && rp < sLen ? split_of(rp) : -1;
                            const int64_t run = c < 0 ? n : std::min<int64_t>(n, split_end(c) - rp + 1);
                            if (c >= 0 && is_dropped(c)) {
                                if (m_like) extra.push_back((uint32_t)(run << 4) | CI);
                                extra.push_back((uint32_t)(run << 4) | CN);
                            } else extra.push_back((uint32_t)(run << 4) | op);
                            rp += run; n -= run;
                        }
                    }
                    r.cigar_off = new_off;
                    r.n_cigar = (uint32_t)((uint64_t)old_n + extra.size() - new_off);
                    d = d1;
                }
                if (!extra.empty()) {
                    RawBuf<uint32_t> nb;
                    nb.resize(old_n + extra.size());
                    memcpy(nb.p, S.cigars.p, old_n * sizeof(uint32_t));
                    memcpy(nb.p + old_n, extra.data(), extra.size() * sizeof(uint32_t));
                    std::swap(nb.p, S.cigars.p);
                    std::swap(nb.n, S.cigars.n);
                }
            }
            stage("max_depth");
        }
    }

    // ---- overlap_push in file order (htslib-1.9 rule |isize| < 2*l_qseq), partitioned by pair: a pair's two reads
    //      only ever touch each other's qualities ----
    const size_t n_reads = S.reads.size();
    {
        // candidates only (both mates mapped, proper pair, |isize| < 2 * l_qseq), bucketed by pair so that a task touches its
        // own reads and not every read of the batch
        const int P = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n_reads / 16384 + 1));
        std::vector<uint32_t> order;
        std::vector<size_t> start;
        if (n_reads >= 0xFFFFFFFFull) { isx_set_error("isx_bam_expand_refs: more than 2^32 reads in one batch"); return ISX_ERR_ARG; }
        bucket_indices(pool, n_reads, P, [&](size_t ri) -> int {
            const Read &r = S.reads[ri];
            if (r.pair_idx == 0xFFFFFFFFu || (r.flag & FMUNMAP) || !(r.flag & FPROPER)) return -1;
            if (std::abs((int64_t)r.isize) >= 2 * (int64_t)r.l_seq) return -1;
            return (int)(r.pair_idx % (uint32_t)P);
        }, order, start);
        pool.run(P, [&](int part) {
            std::unordered_map<uint32_t, int64_t> pending;      // pair -> read waiting for its mate
            pending.reserve((start[(size_t)part + 1] - start[(size_t)part]) / 2 + 16);
            for (size_t q = start[(size_t)part]; q < start[(size_t)part + 1]; q++) {
                const size_t ri = order[q];
                const Read &r = S.reads[ri];
                auto it = pending.find(r.pair_idx);
                if (it != pending.end() && S.reads[(size_t)it->second].ref_end <= r.pos) { pending.erase(it); it = pending.end(); }   // earlier read already left the buffer
                if (it == pending.end()) pending.emplace(r.pair_idx, (int64_t)ri);
                else {
                    const size_t ai = (size_t)it->second;
                    pending.erase(it);
                    tweak_overlap(S, S.reads[ai], r);
                }
            }
        });
    }
    stage("overlap resolution");

    // ---- which reads are piled up, dense pair ids in order of first appearance, and where every read's observations
    //      start in the stream (count per read + prefix sums, all on the threads) ----
    par_assign(pool, Q->emit, n_reads, (uint8_t)0);
    std::vector<uint64_t> slot0(n_ref_all, 0);              // the batch's pair entries, reference after reference
    uint64_t n_slots = 0;
    for (int32_t i = 0; i < n_refs; i++) { slot0[(size_t)refs[i]] = n_slots; n_slots += B.ref_pair0[(size_t)refs[i] + 1] - B.ref_pair0[(size_t)refs[i]]; }
    // dense id of a pair = rank of its first piled-up read: per pair the smallest read index (atomic min, threads over
    // the reads), a prefix count of those first reads, then every read looks its pair's id up
    uvec<uint32_t> first, dense;
    par_assign(pool, first, (size_t)n_slots, 0xFFFFFFFFu);
    par_assign(pool, dense, (size_t)n_slots, 0xFFFFFFFFu);
    par_assign(pool, Q->pid, n_reads, 0u);
    const int n_tasks = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n_reads / 2048 + 1));
    auto lo_of = [&](int t) { return n_reads * (size_t)t / (size_t)n_tasks; };
    auto slot_of = [&](const Read &r) -> size_t { return (size_t)(slot0[(size_t)r.tid] + (r.pair_idx - B.ref_pair0[(size_t)r.tid])); };
    BamBatch *q = Q.get();
    // "is this read's pair in R2M" as ONE BIT per pair entry: the reads come in position order, their pair entries in hash order --
    // a look-up into the 72-byte entries is a DRAM miss per read (6 M reads: 20-40 ms of the hand-over), the bit plane stays in the L2
    // (only the words between the first and the last pair entry of the batch's references are made: a reference's entries are contiguous)
    const size_t n_pair_all = B.pairs.size();
    uvec<uint64_t> pass_bits((n_pair_all + 63) / 64);
    {
        size_t p_lo = n_pair_all, p_hi = 0;
        for (int32_t i = 0; i < n_refs; i++) { p_lo = std::min<size_t>(p_lo, (size_t)B.ref_pair0[(size_t)refs[i]]); p_hi = std::max<size_t>(p_hi, (size_t)B.ref_pair0[(size_t)refs[i] + 1]); }
        const size_t w_lo = p_lo >> 6, n_words = p_hi > p_lo ? ((p_hi + 63) >> 6) - w_lo : 0;
        const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)pool.size() * 4, n_words / 4096 + 1));
        const bool want_mm = !p->skip_mm;
        if (want_mm) { Q->pair_mm_lo = w_lo * 64; Q->pair_mm.resize(n_words * 64 + 64); }
        uint8_t *pmm = want_mm ? Q->pair_mm.data() : nullptr;
        pool.run(nt, [&](int t) {
            for (size_t wd = w_lo + n_words * (size_t)t / (size_t)nt; wd < w_lo + n_words * (size_t)(t + 1) / (size_t)nt; wd++) {
                uint64_t m = 0;
                const size_t i0 = wd * 64, i1 = std::min(n_pair_all, i0 + 64);
                for (size_t i = i0; i < i1; i++) {
                    m |= (uint64_t)(B.pairs[i].pass && B.pairs[i].reads != 0) << (i - i0);
                    if (pmm) pmm[i - w_lo * 64] = (uint8_t)std::min<int32_t>(255, std::max<int32_t>(0, B.level_of(B.pairs[i].mm)));
                }
                pass_bits[wd] = m;
            }
        });
    }
    pool.run(n_tasks, [&](int t) {
        for (size_t ri = lo_of(t); ri < lo_of(t + 1); ri++) {
            const Read &r = S.reads[ri];
            // R2M membership is by NAME on this scaffold (get_base_counts_mm looks up query_name)
            if (r.pair_idx == 0xFFFFFFFFu) continue;
            if (!((pass_bits[r.pair_idx >> 6] >> (r.pair_idx & 63u)) & 1u)) continue;
            q->emit[ri] = 1;
            uint32_t *f = &first[slot_of(r)];
            uint32_t cur = __atomic_load_n(f, __ATOMIC_RELAXED);
            while ((uint32_t)ri < cur && !__atomic_compare_exchange_n(f, &cur, (uint32_t)ri, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
    });
    std::vector<uint32_t> firsts((size_t)n_tasks + 1, 0);
    std::vector<uint64_t> outs((size_t)n_tasks + 1, 0), segs_of((size_t)n_tasks + 1, 0), cols_of((size_t)n_tasks + 1, 0);
    par_assign(pool, Q->out_at, n_reads + 1, (uint64_t)0);
    uvec<uint32_t> seg_cnt;                         // segments of every read
    if (as_segments) { Q->seg_at.resize(n_reads + 1); par_assign(pool, seg_cnt, n_reads, 0u); }      // (seg_at: every entry is written below)
    pool.run(n_tasks, [&](int t) {
        uint32_t nf = 0;
        uint64_t no = 0, ns = 0, nc = 0;
        for (size_t ri = lo_of(t); ri < lo_of(t + 1); ri++) {
            if (!q->emit[ri]) continue;
            nf += first[slot_of(S.reads[ri])] == (uint32_t)ri;
            if (as_segments) {                      // segments come from the CIGAR alone: no per-base pass
                uint64_t k = 0;
                q->for_segments(ri, [&](int64_t, int64_t, int64_t cols) { k++; nc += (uint64_t)cols; });
                seg_cnt[ri] = (uint32_t)k;
                ns += k;
                continue;
            }
            const uint64_t c = q->count_read(ri);
            q->out_at[ri + 1] = c;
            no += c;
        }
        firsts[(size_t)t + 1] = nf; outs[(size_t)t + 1] = no; segs_of[(size_t)t + 1] = ns; cols_of[(size_t)t + 1] = nc;
    });
    for (int t = 0; t < n_tasks; t++) { segs_of[(size_t)t + 1] += segs_of[(size_t)t]; Q->n_seg_bases += (int64_t)cols_of[(size_t)t + 1]; }
    if (as_segments) {
        Q->seg_gpos.resize((size_t)segs_of[(size_t)n_tasks]);
        pool.run(n_tasks, [&](int t) {
            uint64_t at = segs_of[(size_t)t];
            for (size_t ri = lo_of(t); ri < lo_of(t + 1); ri++) {
                q->seg_at[ri] = at;
                if (seg_cnt[ri]) q->for_segments(ri, [&](int64_t g, int64_t, int64_t) { q->seg_gpos[(size_t)at++] = (uint32_t)g; });
            }
        });
        q->seg_at[n_reads] = segs_of[(size_t)n_tasks];
    }
    for (int t = 0; t < n_tasks; t++) { firsts[(size_t)t + 1] += firsts[(size_t)t]; outs[(size_t)t + 1] += outs[(size_t)t]; }
    Q->next_pair = firsts[(size_t)n_tasks];
    const bool keep_names = !B.seg_names.empty();           // (isx_bam_drop_names not called: somebody wants the names of the pair ids)
    bam->last_dense_pair.assign(keep_names ? (size_t)Q->next_pair : 0, 0u);
    uint32_t *dpair = bam->last_dense_pair.data();
    pool.run(n_tasks, [&](int t) {
        uint32_t nf = firsts[(size_t)t];
        uint64_t at = outs[(size_t)t];
        for (size_t ri = lo_of(t); ri < lo_of(t + 1); ri++) {
            if (q->emit[ri]) {
                const size_t sl = slot_of(S.reads[ri]);
                if (first[sl] == (uint32_t)ri) { if (keep_names) dpair[nf] = S.reads[ri].pair_idx; dense[sl] = nf++; }
            }
            at += q->out_at[ri + 1];
            q->out_at[ri + 1] = at;
        }
    });
    pool.run(n_tasks, [&](int t) {
        for (size_t ri = lo_of(t); ri < lo_of(t + 1); ri++)
            if (q->emit[ri]) q->pid[ri] = dense[slot_of(S.reads[ri])];
    });
    if (Q->out_at.back() >= 0xFFFFFFFFull) { isx_set_error("isx_bam_expand_refs: more than 2^32 observations in one batch (expand fewer references at a time)"); return ISX_ERR_ARG; }
    stage("count");

    // ---- iterate_splits (fasta.py:56-73) on the batch's flat space ----
    const int64_t W = p->window_length > 0 ? p->window_length : 10000;
    for (int32_t i = 0; i < n_refs; i++) {
        const size_t t = (size_t)refs[i];
        const int64_t sLen = B.ref_len[t];
        if (sLen <= 0) continue;
        const int64_t n_chunks = sLen / W + 1;
        const int64_t chunk = (int64_t)((double)sLen / (double)n_chunks);
        int64_t start = 0;
        for (int64_t c = 0; c < n_chunks; c++) {
            Q->split_bounds.push_back(boff[t] + start);
            Q->split_ref.push_back((int32_t)t);
            if (c + 1 < n_chunks) start += chunk;
        }
    }
    Q->split_bounds.push_back(n_pos);
    {
        std::lock_guard<std::mutex> lk(bam->retire_mu);
        bam->batches_out++;
    }
    *out = Q.release();
    return ISX_OK;
}

static void batch_is_back(isx_bam *B)
{
    std::lock_guard<std::mutex> lk(B->retire_mu);
    if (--B->batches_out == 0) B->retire_cv.notify_all();
}

void bam_batch_free(BamBatch *q)
{
    if (!q) return;
    isx_bam *B = q->B;
    delete q;
    batch_is_back(B);
}


isx_bam::~isx_bam()
{
    for (auto &r : retired) delete r.first;
    if (map) munmap(const_cast<uint8_t *>(map), map_len);
    if (fd >= 0) close(fd);
}

void bam_batch_retire(BamBatch *q)
{
    if (!q) return;
    isx_bam *B = q->B;
    size_t bytes = q->S.reads.size() * sizeof(Read) + q->S.cigars.size() * 4 + q->emit.size() + q->pid.size() * 4 + q->out_at.size() * 8 +
                   q->seg_at.size() * 8 + q->seg_gpos.size() * 4;
    for (const auto &d : q->S.seg_data) bytes += d.size();
    std::vector<BamBatch *> dead;
    {
        std::lock_guard<std::mutex> lk(B->retire_mu);
        B->retired.emplace_back(q, bytes);
        B->retired_bytes += bytes;
        while (B->retired_bytes > isx_bam::RETIRE_LIMIT && !B->retired.empty()) {
            dead.push_back(B->retired.front().first);
            B->retired_bytes -= B->retired.front().second;
            B->retired.erase(B->retired.begin());
        }
    }
    for (BamBatch *d : dead) delete d;
    batch_is_back(B);
}
int64_t bam_batch_n_segs(const BamBatch *q) { return (int64_t)q->seg_gpos.size(); }
int64_t bam_batch_seg_bases(const BamBatch *q) { return q->n_seg_bases; }
const uint32_t *bam_batch_seg_gpos(const BamBatch *q) { return q->seg_gpos.data(); }
void bam_batch_emit_segs(const BamBatch *q, int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases)
{
    q->emit_segs(first, count, gpos, len, mm, pair, bases);
}
void bam_batch_emit_planes(const BamBatch *q, int64_t first, int64_t count, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint64_t *planes)
{
    q->emit_planes(first, count, gpos, len, mm, pair, planes);
}
int64_t bam_batch_n_obs(const BamBatch *q) { return q->n_obs(); }
int64_t bam_batch_n_pos(const BamBatch *q) { return q->n_pos; }
void bam_batch_emit(const BamBatch *q, int64_t first, uint32_t count, isx_obs *obs, uint32_t *pair) { q->emit_range(first, count, obs, pair); }
void bam_batch_info(const BamBatch *q, int32_t n_refs, isx_bam_info *info)
{
    *info = q->B->totals;
    info->n_refs = n_refs;
    info->n_splits = (int32_t)q->split_ref.size();
    info->n_pos = q->n_pos;
    info->n_obs = q->seg_at.empty() ? q->n_obs() : q->n_seg_bases;
    info->n_segs = (int64_t)q->seg_gpos.size();
    info->n_pairs = q->next_pair;
    info->max_mm = q->prm.skip_mm ? 0 : q->B->totals.max_mm;
}
const std::vector<int64_t> &bam_batch_bounds(const BamBatch *q) { return q->split_bounds; }

extern "C" {

static int expand_into_handle(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, isx_bam_info *info, int64_t reg_lo, int64_t reg_hi)
{
    if (!bam || !p || n_refs < 0 || (n_refs && !refs)) { isx_set_error("isx_bam_expand_refs: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    BamBatch *q = nullptr;
    const int rc = bam_batch_prepare(bam, p, refs, n_refs, &q, reg_lo, reg_hi, false);
    if (rc != ISX_OK) return rc;
    std::unique_ptr<BamBatch, void (*)(BamBatch *)> Q(q, bam_batch_free);
    // the whole stream into the handle (threads over contiguous pieces; file order is kept)
    const size_t n_out = (size_t)Q->n_obs();
    B.obs.reset(new isx_obs[std::max<size_t>(n_out, 1)]);
    B.pair.reset(new uint32_t[std::max<size_t>(n_out, 1)]);
    B.n_obs = n_out;
    isxenc::HostPool &pool = pool_of(B);
    const size_t piece = 1 << 16;
    const int n_tasks = (int)((n_out + piece - 1) / piece);
    pool.run(n_tasks, [&](int t) {
        const size_t a = (size_t)t * piece, e = std::min(n_out, a + piece);
        q->emit_range((int64_t)a, (uint32_t)(e - a), B.obs.get() + a, B.pair.get() + a);
    });
    B.split_bounds = Q->split_bounds;
    B.split_ref = Q->split_ref;
    B.expanded = true;
    if (info) bam_batch_info(q, n_refs, info);
    return ISX_OK;
}

int isx_bam_expand_refs(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, isx_bam_info *info)
{
    return expand_into_handle(bam, p, refs, n_refs, info, 0, -1);
}

// the same batch as READ SEGMENTS (isx_segs): what isx_pipe_submit_bam hands a read-level pipe; info->n_obs = the columns
// the segments cover (an upper bound of the observations), *n_seg = how many isx_bam_copy_segs will deliver
int isx_bam_segment_refs(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, isx_bam_info *info, int64_t *n_seg)
{
    if (!bam || !p || n_refs < 0 || (n_refs && !refs) || !n_seg) { isx_set_error("isx_bam_segment_refs: bad argument"); return ISX_ERR_ARG; }
    isx_bam &B = *bam;
    BamBatch *q = nullptr;
    const int rc = bam_batch_prepare(bam, p, refs, n_refs, &q, 0, -1, true);
    if (rc != ISX_OK) return rc;
    std::unique_ptr<BamBatch, void (*)(BamBatch *)> Q(q, bam_batch_free);
    const size_t n = Q->seg_gpos.size();
    B.seg_gpos.assign(Q->seg_gpos.begin(), Q->seg_gpos.end());
    B.seg_len.resize(n); B.seg_mm.resize(n); B.seg_pair.resize(n); B.seg_bases.resize(n * ISX_SEG_WORDS);
    B.seg_planes.resize(n * ISX_PLANE_WORDS);               // the same segments as bit planes (isx_bam_copy_read_planes)
    isxenc::HostPool &pool = pool_of(B);
    const size_t piece = 4096;
    std::vector<uint32_t> tmp_gpos(n), g2(n), p2(n);
    std::vector<uint8_t> l2(n), m2(n);
    pool.run((int)((n + piece - 1) / piece), [&](int t) {
        const size_t a = (size_t)t * piece, e = std::min(n, a + piece);
        q->emit_segs((int64_t)a, (int64_t)(e - a), tmp_gpos.data() + a, B.seg_len.data() + a, B.seg_mm.data() + a, B.seg_pair.data() + a,
                     B.seg_bases.data() + a * ISX_SEG_WORDS);
        q->emit_planes((int64_t)a, (int64_t)(e - a), g2.data() + a, l2.data() + a, m2.data() + a, p2.data() + a, B.seg_planes.data() + a * ISX_PLANE_WORDS);
    });
    bool mm_same = true;
    for (size_t i = 0; i < n && mm_same; i++) mm_same = m2[i] == (uint8_t)std::min<int>(B.seg_mm[i], 127);
    if (g2 != tmp_gpos || memcmp(l2.data(), B.seg_len.data(), n) != 0 || memcmp(p2.data(), B.seg_pair.data(), n * 4) != 0 || !mm_same) {
        isx_set_error("internal: the segment and the bit-plane emission differ");
        return ISX_ERR_STATE;
    }
    if (tmp_gpos != B.seg_gpos) { isx_set_error("internal: segment starts of the layout pass and of the emission differ"); return ISX_ERR_STATE; }
    B.split_bounds = Q->split_bounds;
    B.split_ref = Q->split_ref;
    B.expanded = false;
    if (info) bam_batch_info(q, n_refs, info);
    *n_seg = (int64_t)n;
    return ISX_OK;
}

int isx_bam_copy_segs(const isx_bam *bam, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases, int64_t *split_bounds,
                      int32_t *split_ref)
{
    if (!bam) { isx_set_error("isx_bam_copy_segs: NULL handle"); return ISX_ERR_ARG; }
    const size_t n = bam->seg_gpos.size();
    if (gpos && n) memcpy(gpos, bam->seg_gpos.data(), n * 4);
    if (len && n) memcpy(len, bam->seg_len.data(), n);
    if (mm && n) memcpy(mm, bam->seg_mm.data(), n);
    if (pair && n) memcpy(pair, bam->seg_pair.data(), n * 4);
    if (bases && n) memcpy(bases, bam->seg_bases.data(), n * ISX_SEG_WORDS * 4);
    if (split_bounds) memcpy(split_bounds, bam->split_bounds.data(), bam->split_bounds.size() * sizeof(int64_t));
    if (split_ref && !bam->split_ref.empty()) memcpy(split_ref, bam->split_ref.data(), bam->split_ref.size() * sizeof(int32_t));
    return ISX_OK;
}

int isx_bam_copy_read_planes(const isx_bam *bam, uint64_t *planes)
{
    if (!bam || !planes) { isx_set_error("isx_bam_copy_read_planes: bad argument"); return ISX_ERR_ARG; }
    if (!bam->seg_planes.empty()) memcpy(planes, bam->seg_planes.data(), bam->seg_planes.size() * sizeof(uint64_t));
    return ISX_OK;
}

// samfile.pileup(scaffold, start=..., stop=..., truncate=True) of the SNV-pooling re-pileup (polymorpher.py:287-293): only the
// columns [start, stop) of ONE reference, from the reads that overlap them; positions stay those of the whole reference
int isx_bam_expand_region(isx_bam *bam, const isx_bam_params *p, int32_t ref, int64_t start, int64_t stop, isx_bam_info *info)
{
    if (stop <= start || start < 0) { isx_set_error("isx_bam_expand_region: 0 <= start < stop"); return ISX_ERR_ARG; }
    return expand_into_handle(bam, p, &ref, 1, info, start, stop);
}

// scan + filter + expansion of every reference of the file
int isx_bam_expand(isx_bam *bam, const isx_bam_params *p, isx_bam_info *info)
{
    if (!bam || !p || !info) { isx_set_error("isx_bam_expand: bad argument"); return ISX_ERR_ARG; }
    int rc = isx_bam_scan(bam, nullptr);
    if (rc != ISX_OK) return rc;
    if ((rc = isx_bam_filter(bam, p, NAN, nullptr)) != ISX_OK) return rc;
    std::vector<int32_t> all(bam->ref_name.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int32_t)i;
    return isx_bam_expand_refs(bam, p, all.data(), (int32_t)all.size(), info);
}

int isx_bam_copy(const isx_bam *bam, isx_obs *obs, uint32_t *pair, int64_t *split_bounds, int32_t *split_ref)
{
    if (!bam || !bam->expanded) { isx_set_error("isx_bam_copy: expand first"); return ISX_ERR_STATE; }
    if (obs && bam->n_obs) memcpy(obs, bam->obs.get(), bam->n_obs * sizeof(isx_obs));
    if (pair && bam->n_obs) memcpy(pair, bam->pair.get(), bam->n_obs * sizeof(uint32_t));
    if (split_bounds) memcpy(split_bounds, bam->split_bounds.data(), bam->split_bounds.size() * sizeof(int64_t));
    if (split_ref && !bam->split_ref.empty()) memcpy(split_ref, bam->split_ref.data(), bam->split_ref.size() * sizeof(int32_t));
    return ISX_OK;
}

/* zero-copy access to what isx_bam_copy copies; valid until the next expand / isx_bam_close */
int isx_bam_view(const isx_bam *bam, const isx_obs **obs, const uint32_t **pair)
{
    if (!bam || !bam->expanded) { isx_set_error("isx_bam_view: expand first"); return ISX_ERR_STATE; }
    if (obs) *obs = bam->obs.get();
    if (pair) *pair = bam->pair.get();
    return ISX_OK;
}

}  // extern "C"


// the front end's own block decoder on the calling thread (tests, tools/inflate_rate.py): fast_inflate.h with NO fallback -- ISX_ERR_IO names
// the first block it does not decode
int isx_bgzf_inflate_fast(const uint8_t *file, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, int64_t out_bytes)
{
    if (!file || !blocks || n_blocks < 0 || (!out && out_bytes) || out_bytes < 0) { isx_set_error("isx_bgzf_inflate_fast: bad argument"); return ISX_ERR_ARG; }
    std::unique_ptr<isxinf::FastInflater> f(new isxinf::FastInflater());
    for (int64_t i = 0; i < n_blocks; i++) {
        const isx_bgzf_block &b = blocks[i];
        if (b.in_off < 0 || b.in_len < 0 || b.in_off + b.in_len > n_bytes || b.out_len < 0 || b.out_off < 0 || b.out_off + b.out_len > out_bytes) {
            isx_set_error("isx_bgzf_inflate_fast: a block reaches outside its buffer");
            return ISX_ERR_ARG;
        }
        if (b.out_len && !f->run(file + b.in_off, (size_t)b.in_len, out + b.out_off, (size_t)b.out_len)) {
            isx_set_error("BGZF block " + std::to_string(i) + ": not decoded by the fast decoder");
            return ISX_ERR_IO;
        }
    }
    return ISX_OK;
}
