// seg_encode.cpp -- read segments -> device record stream (see seg_encode.h), and the host helpers of the read-level
// hand-over declared in include/instrain_amd.h (isx_encode_segs, isx_count_read_segs, isx_pack_reads).
//
// Layout: the input is cut into tasks of TASK segments; a first pass over the segment STARTS alone (4 bytes each) counts
// the device groups every task needs -- 16 records per group, closed early where the starts of a group would span more
// than 65535 positions (the next contig / genome of a database, an uncovered stretch) --, a prefix sum places the tasks and
// the second pass writes headers, payload, group bases, the per-group position directory and the pair ids.
#include "seg_encode.h"

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void isx_set_error(const std::string &msg);

namespace isxenc {

namespace {

constexpr int64_t TASK = 4096;          // segments per task (a multiple of ISX_SEG_GROUP): 256 KiB of payload
constexpr uint32_t SPAN = 65535u;       // largest delta a header can carry

// groups the segments [a, e) need: greedy, arrival order
inline int64_t count_groups(const uint32_t *gpos, int64_t a, int64_t e, int64_t group = ISX_SEG_GROUP)
{
    int64_t n = 0;
    for (int64_t i = a; i < e;) {
        uint32_t lo = gpos[i], hi = gpos[i];
        int64_t j = i + 1;
        for (; j < e && j - i < group; j++) {
            const uint32_t p = gpos[j];
            const uint32_t nlo = p < lo ? p : lo, nhi = p > hi ? p : hi;
            if (nhi - nlo > SPAN) break;
            lo = nlo; hi = nhi;
        }
        i = j; n++;
    }
    return n;
}

struct Scratch {
    std::vector<uint32_t> gpos, pair, bases;
    std::vector<uint8_t> len, mm;
};

// one 64-byte device record = header word + 15 payload words.  With AVX-512 it is assembled in a register (a masked load of
// the payload shifted by one lane -- masked-off lanes never fault --, the header blended into lane 0) and leaves with one
// streaming store: the staging memory is written once and never read by this core (a 60-byte memcpy per record ran at a
// fifth of this).
inline bool cpu_has_avx512()
{
    static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    return v;
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
inline void put_record_avx512(uint32_t *o, uint32_t header, const uint32_t *payload)
{
    __m512i v = _mm512_maskz_loadu_epi32((__mmask16)0xFFFE, payload - 1);
    v = _mm512_mask_set1_epi32(v, (__mmask16)0x0001, (int)header);
    _mm512_stream_si512(reinterpret_cast<__m512i *>(o), v);
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
inline void put_padding_avx512(uint32_t *o)
{
    const __m512i v = _mm512_mask_set1_epi32(_mm512_set1_epi32((int)ISX_SEG_SKIP_WORD), (__mmask16)0x0001, 0);
    _mm512_stream_si512(reinterpret_cast<__m512i *>(o), v);
}

}  // namespace

// device groups the segment starts need, with encode_segs' own task cut (a task never shares a group with its neighbour):
// what a caller must size cap_rec by -- a sparse stream (starts thousands of positions apart) closes a group every few
// segments because of the 65 535 span, not only at jumps
int64_t seg_groups_needed(HostPool &pool, const uint32_t *gpos, int64_t n)
{
    const int n_tasks = (int)((n + TASK - 1) / TASK);
    std::vector<int64_t> g((size_t)std::max(n_tasks, 1), 0);
    pool.run(n_tasks, [&](int t) {
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        g[(size_t)t] = count_groups(gpos, a, e);
    });
    int64_t tot = 0;
    for (int64_t v : g) tot += v;
    return std::max<int64_t>(tot, 1);
}

int encode_segs(HostPool &pool, SegJob &J)
{
    const int64_t n = J.n_seg;
    const bool producer = (bool)J.produce;
    const uint32_t *gpos_all = producer ? J.gpos_all : J.in.gpos;
    const int n_tasks = (int)((n + TASK - 1) / TASK);
    std::vector<int64_t> g_at((size_t)n_tasks + 1, 0);
    pool.run(n_tasks, [&](int t) {
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        g_at[(size_t)t + 1] = count_groups(gpos_all, a, e);
    });
    for (int t = 0; t < n_tasks; t++) g_at[(size_t)t + 1] += g_at[(size_t)t];
    const int64_t n_groups = std::max<int64_t>(g_at[(size_t)n_tasks], 1);
    J.n_rec = n_groups * ISX_SEG_GROUP;
    if (J.n_rec > J.cap_rec) return SEG_CAPACITY;
    std::atomic<int> err{SEG_OK};
    std::vector<int64_t> bases_of((size_t)std::max(n_tasks, 1), 0);
    std::vector<uint32_t> maxp_of((size_t)std::max(n_tasks, 1), 0);
    const bool pairs = J.pair_out != nullptr && (producer ? J.want_pairs : J.in.pair != nullptr);
    const int64_t RG = J.ring_groups;
    const size_t group_words = (size_t)ISX_SEG_GROUP * ISX_SEG_REC_WORDS;
    if (n == 0) {                                   // one empty group: the kernels want a stream
        if (RG) J.wave_begin(0);
        for (int r = 0; r < ISX_SEG_GROUP; r++) {
            uint32_t *o = J.rec + (size_t)r * ISX_SEG_REC_WORDS;
            o[0] = 0;
            for (int k = 1; k < ISX_SEG_REC_WORDS; k++) o[k] = ISX_SEG_SKIP_WORD;
            if (J.pair_out) J.pair_out[r] = 0;
        }
        J.gbase[0] = 0; J.cmin[0] = 0xFFFFFFFFu; J.cmax[0] = 0; J.cany[0] = 0;
        J.n_bases = 0; J.max_pair = 0;
        if (RG) J.wave_flush(0, 0, 1);
        return SEG_OK;
    }
    const bool fast = cpu_has_avx512() && (reinterpret_cast<uintptr_t>(J.rec) & 63) == 0;
    auto run_task = [&](int t, int64_t wave_g0, int half) {
        if (err.load(std::memory_order_relaxed) != SEG_OK) return;
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        const uint32_t *gp, *pr, *bs;
        const uint8_t *ln, *mm;
        if (producer) {
            thread_local Scratch S;
            if (S.gpos.size() < (size_t)TASK) { S.gpos.resize((size_t)TASK); S.pair.resize((size_t)TASK); S.bases.resize((size_t)TASK * ISX_SEG_WORDS); S.len.resize((size_t)TASK); S.mm.resize((size_t)TASK); }
            J.produce(a, e - a, S.gpos.data(), S.len.data(), S.mm.data(), pairs ? S.pair.data() : nullptr, S.bases.data());
            gp = S.gpos.data() - a; ln = S.len.data() - a; mm = S.mm.data() - a; pr = pairs ? S.pair.data() - a : nullptr;
            bs = S.bases.data() - (size_t)a * ISX_SEG_WORDS;
        } else {
            gp = J.in.gpos; ln = J.in.len; mm = J.in.mm; pr = pairs ? J.in.pair : nullptr; bs = J.in.bases;
        }
        int64_t g = g_at[(size_t)t], nb = 0;
        uint32_t maxp = 0;
        for (int64_t i = a; i < e;) {
            uint32_t lo = gpos_all[i], hi = lo;
            int64_t j = i + 1;
            for (; j < e && j - i < ISX_SEG_GROUP; j++) {
                const uint32_t p = gpos_all[j];
                const uint32_t nlo = p < lo ? p : lo, nhi = p > hi ? p : hi;
                if (nhi - nlo > SPAN) break;
                lo = nlo; hi = nhi;
            }
            uint32_t *o = J.rec + (RG ? (size_t)(g - wave_g0 + (int64_t)half * RG) : (size_t)g) * group_words;
            uint32_t last = 0;
            for (int64_t s = i; s < j; s++, o += ISX_SEG_REC_WORDS) {
                const uint32_t L = ln[s], m = mm ? mm[s] : 0u, p = gp[s];
                if (L == 0 || L > ISX_SEG_BASES) { err.store(SEG_BAD_LEN); return; }
                if ((int64_t)p + (int64_t)L > J.n_pos || p != gpos_all[s]) { err.store(SEG_BAD_POS); return; }
                if ((int)m >= J.n_mm_bins) { err.store(SEG_MM_RANGE); return; }
                if (fast) put_record_avx512(o, (p - lo) | (L << 16) | (m << 24), bs + (size_t)s * ISX_SEG_WORDS);
                else {
                    o[0] = (p - lo) | (L << 16) | (m << 24);
                    memcpy(o + 1, bs + (size_t)s * ISX_SEG_WORDS, ISX_SEG_WORDS * sizeof(uint32_t));
                }
                last = std::max(last, p + L - 1);
                nb += L;
            }
            for (int64_t s = j - i; s < ISX_SEG_GROUP; s++, o += ISX_SEG_REC_WORDS) {
                if (fast) { put_padding_avx512(o); continue; }
                o[0] = 0;
                for (int k = 1; k < ISX_SEG_REC_WORDS; k++) o[k] = ISX_SEG_SKIP_WORD;
            }
            if (J.pair_out) {
                uint32_t *po = J.pair_out + (size_t)g * ISX_SEG_GROUP;
                for (int64_t s = i; s < j; s++) { const uint32_t v = pr ? pr[s] : 0u; po[s - i] = v; maxp = v > maxp ? v : maxp; }
                for (int64_t s = j - i; s < ISX_SEG_GROUP; s++) po[s] = 0;
            }
            J.gbase[g] = lo; J.cmin[g] = lo; J.cmax[g] = last; J.cany[g] = 1;
            i = j; g++;
        }
        if (fast) _mm_sfence();                     // the streaming stores are globally visible before the task counts as done
        bases_of[(size_t)t] = nb; maxp_of[(size_t)t] = maxp;
    };
    if (!RG) pool.run(n_tasks, [&](int t) { run_task(t, 0, 0); });
    else {
        // waves of consecutive tasks whose groups fit one half of the ring
        int half = 0;
        for (int t0 = 0; t0 < n_tasks && err.load() == SEG_OK;) {
            int t1 = t0 + 1;
            while (t1 < n_tasks && g_at[(size_t)t1 + 1] - g_at[(size_t)t0] <= RG) t1++;
            if (g_at[(size_t)t1] - g_at[(size_t)t0] > RG) return SEG_CAPACITY;         // one task alone outgrows a half (ring far too small)
            J.wave_begin(half);
            const int64_t wg0 = g_at[(size_t)t0];
            pool.run(t1 - t0, [&](int k) { run_task(t0 + k, wg0, half); });
            if (err.load() == SEG_OK) J.wave_flush(half, wg0, g_at[(size_t)t1]);
            t0 = t1; half ^= 1;
        }
    }
    if (err.load() != SEG_OK) return err.load();
    J.n_bases = 0; J.max_pair = 0;
    for (int t = 0; t < n_tasks; t++) { J.n_bases += bases_of[(size_t)t]; J.max_pair = std::max(J.max_pair, maxp_of[(size_t)t]); }
    return SEG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// reference-delta records (include/instrain_amd.h ISX_DREC_*): a segment is compared with the reference on the host and only
// what differs travels -- the columns it does not observe as a bit plane, the bases that are not the reference's as
// (offset, base) exceptions.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct Cols {                           // one segment against the reference: 160-bit column masks
    uint64_t skip[3];                   // no observation at the column (codes >= 4)
    uint64_t exc[3];                    // observed, and not the reference's base
    uint64_t nbase[3];                  // code 5: a base that is not A/C/T/G (mm profiling on: it makes its level present, profile_utilities.py:279-285)
    alignas(64) uint8_t code[192];      // base code per column
};

inline void cols_scalar(const uint32_t *bases, const uint8_t *ref, uint32_t L, Cols &C)
{
    C.skip[0] = C.skip[1] = C.skip[2] = 0;
    C.exc[0] = C.exc[1] = C.exc[2] = 0;
    C.nbase[0] = C.nbase[1] = C.nbase[2] = 0;
    for (uint32_t j = 0; j < L; j++) {
        const uint32_t c = (bases[j / 10] >> (3 * (j % 10))) & 7u;
        C.code[j] = (uint8_t)c;
        const uint64_t bit = 1ull << (j & 63);
        if (c >= 4) { C.skip[j >> 6] |= bit; if (c == 5) C.nbase[j >> 6] |= bit; }
        else if (c != ref[j]) C.exc[j >> 6] |= bit;
    }
}

inline bool cpu_has_vbmi()
{
    static const bool v = cpu_has_avx512() && __builtin_cpu_supports("avx512vbmi") && !getenv("ISX_NO_VBMI");     // (the switch: tests of the scalar path)
    return v;
}

// three vectors of 64 codes from the fifteen words: per 64-bit lane the two words its eight codes sit in (vpermd), the eight
// 3-bit fields picked by vpmultishiftqb
struct UnpackTables {
    alignas(64) uint32_t idx[3][16];
    alignas(64) uint8_t ctrl[3][64];
    UnpackTables()
    {
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 8; q++) {
                const int n0 = 64 * k + 8 * q, i = n0 / 10;
                idx[k][2 * q] = (uint32_t)std::min(i, 15);
                idx[k][2 * q + 1] = (uint32_t)std::min(i + 1, 15);
                for (int t = 0; t < 8; t++) {
                    const int n = n0 + t;
                    ctrl[k][8 * q + t] = (uint8_t)((n / 10 == i ? 0 : 32) + 3 * (n % 10));
                }
            }
    }
};

__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi")))
inline void cols_vbmi(const uint32_t *bases, const uint8_t *ref, uint32_t L, Cols &C)
{
    static const UnpackTables T;
    const __m512i in = _mm512_maskz_loadu_epi32((__mmask16)0x7FFF, bases);
    const __m512i seven = _mm512_set1_epi8(7), four = _mm512_set1_epi8(4), five = _mm512_set1_epi8(5);
#pragma GCC unroll 3
    for (int k = 0; k < 3; k++) {
        const __m512i src = _mm512_permutexvar_epi32(_mm512_load_si512(T.idx[k]), in);
        const __m512i c = _mm512_and_si512(_mm512_multishift_epi64_epi8(_mm512_load_si512(T.ctrl[k]), src), seven);
        _mm512_store_si512(C.code + 64 * k, c);
        const uint32_t have = L > (uint32_t)(64 * k) ? std::min<uint32_t>(64u, L - (uint32_t)(64 * k)) : 0u;
        const __mmask64 lm = have == 64 ? ~(__mmask64)0 : (((__mmask64)1 << have) - 1);
        const __m512i r = _mm512_maskz_loadu_epi8(lm, ref + 64 * k);               // (masked lanes never fault)
        const __mmask64 sk = _mm512_cmpge_epu8_mask(c, four) & lm;
        C.skip[k] = (uint64_t)sk;
        C.nbase[k] = (uint64_t)(_mm512_cmpeq_epi8_mask(c, five) & lm);
        C.exc[k] = (uint64_t)(_mm512_cmpneq_epi8_mask(c, r) & lm & ~sk);
    }
}

// bits [b, b + len) of a 160-bit mask, moved down to bit 0 (len <= 160 - b)
inline void window160(const uint64_t m[3], uint32_t b, uint32_t len, uint64_t out[3])
{
    const uint32_t ws = b >> 6, bs = b & 63;
    for (uint32_t i = 0; i < 3; i++) {
        const uint64_t lo = i + ws < 3 ? m[i + ws] : 0, hi = i + ws + 1 < 3 ? m[i + ws + 1] : 0;
        uint64_t v = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
        const uint32_t from = 64 * i;
        if (len <= from) v = 0;
        else if (len - from < 64) v &= ((uint64_t)1 << (len - from)) - 1;
        out[i] = v;
    }
}

struct GroupBuf {
    alignas(64) uint32_t rec[ISX_DREC_GROUP][ISX_DREC_WORDS];
    uint32_t start[2 * ISX_DREC_GROUP], last[2 * ISX_DREC_GROUP];      // per 16-byte half (a full record: its first)
    bool used[2 * ISX_DREC_GROUP];
    int n = 0;                  // records
    int open = -1;              // the dual record whose second half is still free (-1: none)
    uint32_t lo = 0, hi = 0;
};

__attribute__((target("avx512f,avx512bw,avx512vl")))
inline void store_group_avx512(uint32_t *o, const GroupBuf &G)
{
    for (int r = 0; r < ISX_DREC_GROUP; r += 2) _mm512_stream_si512(reinterpret_cast<__m512i *>(o + (size_t)r * ISX_DREC_WORDS), _mm512_load_si512(G.rec[r]));
    _mm_sfence();
}

}  // namespace

int64_t delta_groups_needed(HostPool &pool, const uint32_t *gpos, int64_t n, int64_t slack_groups)
{
    const int n_tasks = (int)((n + TASK - 1) / TASK);
    std::vector<int64_t> g((size_t)std::max(n_tasks, 1), 0);
    pool.run(n_tasks, [&](int t) {
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        g[(size_t)t] = count_groups(gpos, a, e, 2 * ISX_DREC_GROUP) + slack_groups;      // (a group holds up to 64 segments without skipped columns)
    });
    int64_t tot = 0;
    for (int64_t v : g) tot += v;
    return std::max<int64_t>(tot, 1);
}

int encode_delta(HostPool &pool, SegJob &J)
{
    const int64_t n = J.n_seg;
    const bool producer = (bool)J.produce;
    const uint32_t *gpos_all = producer ? J.gpos_all : J.in.gpos;
    const int n_tasks = (int)((n + TASK - 1) / TASK);
    const int64_t slack = std::max<int64_t>(J.slack_groups, 1);
    std::vector<int64_t> g_at((size_t)n_tasks + 1, 0);
    if (J.task_groups) for (int t = 0; t < n_tasks; t++) g_at[(size_t)t + 1] = std::max<int64_t>(J.task_groups[t], 1);
    else pool.run(n_tasks, [&](int t) {
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        g_at[(size_t)t + 1] = count_groups(gpos_all, a, e, 2 * ISX_DREC_GROUP) + slack;
    });
    for (int t = 0; t < n_tasks; t++) g_at[(size_t)t + 1] += g_at[(size_t)t];
    const int64_t n_groups = std::max<int64_t>(g_at[(size_t)n_tasks], 1);
    J.n_rec = n_groups * ISX_DREC_GROUP;
    J.need_slack = slack;
    if (J.n_rec > J.cap_rec) return SEG_CAPACITY;
    std::atomic<int> err{SEG_OK};
    std::vector<int64_t> bases_of((size_t)std::max(n_tasks, 1), 0), need_of((size_t)std::max(n_tasks, 1), 0), pieces_of((size_t)std::max(n_tasks, 1), 0);
    std::vector<uint32_t> maxp_of((size_t)std::max(n_tasks, 1), 0);
    const bool pairs = producer ? J.want_pairs : J.in.pair != nullptr;          // (the ids travel inside the records: pair_out is not written)
    const int64_t RG = J.ring_groups;
    constexpr size_t group_words = (size_t)ISX_DREC_GROUP * ISX_DREC_WORDS;
    const bool fast_store = cpu_has_avx512() && (reinterpret_cast<uintptr_t>(J.rec) & 63) == 0;
    const bool vbmi = cpu_has_vbmi();
    auto put_empty_group = [&](uint32_t *o) {     // full records of length 0
        for (int r = 0; r < ISX_DREC_GROUP; r++, o += ISX_DREC_WORDS) {
            o[0] = o[1] = o[2] = o[4] = o[5] = o[6] = o[7] = 0;
            o[3] = ISX_DREC_NO_EXC;
        }
    };
    if (n == 0) {                                   // one empty group: the kernels want a stream
        if (RG) J.wave_begin(0);
        put_empty_group(J.rec);
        J.gbase[0] = 0; J.cmin[0] = 0xFFFFFFFFu; J.cmax[0] = 0; J.cany[0] = 0;
        J.n_bases = 0; J.max_pair = 0; J.n_pieces = 0;
        if (RG) J.wave_flush(0, 0, 1);
        return SEG_OK;
    }
    auto run_task = [&](int t, int64_t wave_g0, int half) {
        if (err.load(std::memory_order_relaxed) != SEG_OK) return;
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        const uint32_t *gp, *pr, *bs;
        const uint8_t *ln, *mm;
        if (producer) {
            thread_local Scratch S;
            if (S.gpos.size() < (size_t)TASK) { S.gpos.resize((size_t)TASK); S.pair.resize((size_t)TASK); S.bases.resize((size_t)TASK * ISX_SEG_WORDS + 1); S.len.resize((size_t)TASK); S.mm.resize((size_t)TASK); }
            J.produce(a, e - a, S.gpos.data(), S.len.data(), S.mm.data(), pairs ? S.pair.data() : nullptr, S.bases.data());
            gp = S.gpos.data() - a; ln = S.len.data() - a; mm = S.mm.data() - a; pr = pairs ? S.pair.data() - a : nullptr;
            bs = S.bases.data() - (size_t)a * ISX_SEG_WORDS;
        } else {
            gp = J.in.gpos; ln = J.in.len; mm = J.in.mm; pr = pairs ? J.in.pair : nullptr; bs = J.in.bases;
        }
        const int64_t g_end = g_at[(size_t)t + 1];
        int64_t g = g_at[(size_t)t], nb = 0, used = 0, np = 0;     // used: groups this task needs (may exceed its region: counted, not written)
        uint32_t maxp = 0;
        GroupBuf G;
        auto rec_at = [&](int64_t gi) { return J.rec + (RG ? (size_t)(gi - wave_g0 + (int64_t)half * RG) : (size_t)gi) * group_words; };
        auto close_group = [&]() {
            if (!G.n) return;
            if (g < g_end) {
                uint32_t lo = 0xFFFFFFFFu, last = 0;
                for (int h = 0; h < 2 * G.n; h++) if (G.used[h]) { lo = std::min(lo, G.start[h]); last = std::max(last, G.last[h]); }
                for (int h = 0; h < 2 * G.n; h++) if (G.used[h]) G.rec[h >> 1][4 * (h & 1)] |= G.start[h] - lo;
                for (int r = G.n; r < ISX_DREC_GROUP; r++) {
                    uint32_t *o = G.rec[r];
                    o[0] = o[1] = o[2] = o[4] = o[5] = o[6] = o[7] = 0;
                    o[3] = ISX_DREC_NO_EXC;
                }
                uint32_t *o = rec_at(g);
                if (fast_store) store_group_avx512(o, G); else memcpy(o, G.rec, sizeof G.rec);
                J.gbase[g] = lo; J.cmin[g] = lo; J.cmax[g] = last; J.cany[g] = 1;
                g++;
            }
            used++;
            G.n = 0; G.open = -1;
        };
        // one piece of a segment -> a 16-byte half of a dual record when the segment has no skipped columns (`plain`: exc holds up to
        // ISX_DREC_EXC exceptions), else a full record (msk + exc[0]: up to ISX_DREC_EXC_FULL exceptions).  A plain piece fills the free
        // half of the record before it only while no full record has come in between: the stream keeps the segments' order.
        auto add_piece = [&](uint32_t start, uint32_t len, uint32_t pid, const uint64_t msk[3], const uint32_t exc[2], bool plain, uint32_t mmv) {
            len |= mmv << 8;                    // the pair's mm level rides in bits 24..30 of the header (0 with one mm bin)
            bool fill = plain && G.open >= 0;
            if (G.n) {
                const uint32_t nlo = std::min(G.lo, start), nhi = std::max(G.hi, start);
                if (nhi - nlo > SPAN || (!fill && G.n == ISX_DREC_GROUP)) { close_group(); fill = false; }
            }
            if (!G.n) { G.lo = G.hi = start; }
            else { G.lo = std::min(G.lo, start); G.hi = std::max(G.hi, start); }
            int h;
            if (fill) {
                h = 2 * G.open + 1;
                uint32_t *o = G.rec[G.open] + 4;
                o[0] = (len << 16) | ISX_DREC_DUAL; o[1] = pid; o[2] = exc[0]; o[3] = exc[1];
                G.open = -1;
            } else {
                h = 2 * G.n;
                uint32_t *o = G.rec[G.n];
                if (plain) {
                    o[0] = (len << 16) | ISX_DREC_DUAL; o[1] = pid; o[2] = exc[0]; o[3] = exc[1];
                    o[4] = ISX_DREC_DUAL; o[5] = 0; o[6] = o[7] = ISX_DREC_NO_EXC;         // second half: length 0 until a piece moves in
                    G.open = G.n;
                } else {
                    o[0] = len << 16;
                    o[1] = (uint32_t)msk[0]; o[2] = (uint32_t)(msk[0] >> 32); o[3] = exc[0];
                    o[4] = (uint32_t)msk[1]; o[5] = (uint32_t)(msk[1] >> 32); o[6] = (uint32_t)msk[2]; o[7] = pid;
                    G.open = -1;
                }
                G.used[h + 1] = false;
                G.n++;
            }
            G.used[h] = true; G.start[h] = start; G.last[h] = start + (len & 0xFFu) - 1;
            np++;
        };
        Cols C;
        for (int64_t s = a; s < e; s++) {
            const uint32_t L = ln[s], m = mm ? mm[s] : 0u, p = gp[s];
            if (L == 0 || L > ISX_SEG_BASES) { err.store(SEG_BAD_LEN); return; }
            if ((int64_t)p + (int64_t)L > J.n_pos || p != gpos_all[s]) { err.store(SEG_BAD_POS); return; }
            if ((int)m >= J.n_mm_bins) { err.store(SEG_MM_RANGE); return; }
            const uint32_t pid = pr ? pr[s] : 0u;
            maxp = std::max(maxp, pid);
            nb += L;
            if (vbmi) cols_vbmi(bs + (size_t)s * ISX_SEG_WORDS, J.ref + p, L, C);
            else cols_scalar(bs + (size_t)s * ISX_SEG_WORDS, J.ref + p, L, C);
            if (J.n_mm_bins > 1) {
                // mm profiling on: a base that is not A/C/T/G travels as an "exception" at a SKIPPED column (the kernel tells it from a
                // counted one by the column's skip bit): nothing is counted there, the pair's level becomes present
                C.exc[0] |= C.nbase[0]; C.exc[1] |= C.nbase[1]; C.exc[2] |= C.nbase[2];
            }
            const int n_exc = __builtin_popcountll(C.exc[0]) + __builtin_popcountll(C.exc[1]) + __builtin_popcountll(C.exc[2]);
            const bool plain = (C.skip[0] | C.skip[1] | C.skip[2]) == 0;           // no skipped column in the whole segment: its pieces are dual halves
            if (n_exc == 0) {
                const uint32_t none[2] = {ISX_DREC_NO_EXC, ISX_DREC_NO_EXC};
                add_piece(p, L, pid, C.skip, none, plain, m);
                continue;
            }
            // pieces of at most ISX_DREC_EXC (plain) / ISX_DREC_EXC_FULL exceptions: a piece ends right before the exception it has no room for
            const uint32_t max_exc = plain ? ISX_DREC_EXC : ISX_DREC_EXC_FULL;
            uint64_t ex[3] = {C.exc[0], C.exc[1], C.exc[2]};
            uint32_t b = 0;
            while (b < L) {
                uint32_t f[ISX_DREC_EXC], nf = 0, end = L;
                for (int k = 0; k < 3; k++) {
                    while (ex[k]) {
                        const uint32_t col = (uint32_t)(64 * k + __builtin_ctzll(ex[k]));
                        if (nf == max_exc) { end = col; goto cut; }
                        ex[k] &= ex[k] - 1;
                        f[nf++] = ((col - b) & 0xFFu) | ((uint32_t)(C.code[col] & 3u) << 8);
                    }
                }
            cut:
                uint32_t w[2] = {ISX_DREC_NO_EXC, ISX_DREC_NO_EXC};
                for (uint32_t i = 0; i < nf; i++) {
                    const int wi = i / 3, sh = 10 * (int)(i % 3);
                    w[wi] = (w[wi] & ~(0x3FFu << sh)) | (f[i] << sh);
                }
                uint64_t msk[3];
                window160(C.skip, b, end - b, msk);
                add_piece(p + b, end - b, pid, msk, w, plain, m);
                b = end;
            }
        }
        close_group();
        need_of[(size_t)t] = used;
        for (; g < g_end; g++) {                    // the spare groups of the region: empty
            put_empty_group(rec_at(g));
            J.gbase[g] = 0; J.cmin[g] = 0xFFFFFFFFu; J.cmax[g] = 0; J.cany[g] = 0;
        }
        bases_of[(size_t)t] = nb; maxp_of[(size_t)t] = maxp; pieces_of[(size_t)t] = np;
    };
    if (!RG) pool.run(n_tasks, [&](int t) { run_task(t, 0, 0); });
    else {
        int half = 0;
        for (int t0 = 0; t0 < n_tasks && err.load() == SEG_OK;) {
            int t1 = t0 + 1;
            while (t1 < n_tasks && g_at[(size_t)t1 + 1] - g_at[(size_t)t0] <= RG) t1++;
            if (g_at[(size_t)t1] - g_at[(size_t)t0] > RG) return SEG_CAPACITY;         // one task alone outgrows a half (ring far too small)
            J.wave_begin(half);
            const int64_t wg0 = g_at[(size_t)t0];
            pool.run(t1 - t0, [&](int k) { run_task(t0 + k, wg0, half); });
            if (err.load() == SEG_OK) J.wave_flush(half, wg0, g_at[(size_t)t1]);
            t0 = t1; half ^= 1;
        }
    }
    if (err.load() != SEG_OK) return err.load();
    J.n_bases = 0; J.max_pair = 0; J.n_pieces = 0;
    int64_t worst = 0;
    bool fits = true;
    J.task_need.assign(need_of.begin(), need_of.begin() + n_tasks);
    for (int t = 0; t < n_tasks; t++) {
        J.n_bases += bases_of[(size_t)t]; J.max_pair = std::max(J.max_pair, maxp_of[(size_t)t]); J.n_pieces += pieces_of[(size_t)t];
        const int64_t region = g_at[(size_t)t + 1] - g_at[(size_t)t];
        if (need_of[(size_t)t] > region) fits = false;
        worst = std::max(worst, need_of[(size_t)t] - (region - (J.task_groups ? 0 : slack)));
    }
    J.need_slack = fits ? slack : std::max<int64_t>(worst, slack + 1);
    if (!fits) return SEG_CAPACITY;                 // some task outgrew its region: encode again with task_groups = task_need
    return SEG_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// bit-plane input (include/instrain_amd.h isx_read_planes) -> the same reference-delta records.  Nothing is unpacked: the five
// words of 2-bit codes are XORed with the reference plane funnel-shifted to the segment's start (32 columns a step), the two
// bits of a column are folded and gathered with pext into the 160-bit "differs" mask, the skip plane is the input's own.
// ---------------------------------------------------------------------------------------------------------------------------
namespace {

struct LenMasks {                       // bits [0, L) of a 256-bit field (rows are loaded as one vector)
    alignas(32) uint64_t m[ISX_SEG_BASES + 1][4];
    LenMasks()
    {
        for (uint32_t L = 0; L <= ISX_SEG_BASES; L++)
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t from = 64 * k;
                m[L][k] = L <= from ? 0 : (L - from >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << (L - from)) - 1));
            }
    }
};

inline bool cpu_has_bmi2()
{
    static const bool v = __builtin_cpu_supports("bmi2") && !getenv("ISX_NO_BMI2");     // (the switch: tests of the portable path)
    return v;
}

inline uint64_t even_bits_portable(uint64_t x)       // bits 0, 2, 4 ... 62 -> bits 0 .. 31
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}

// bits [p, p + 160) of a bit plane; `q` = plane + p / 8 with at least 32 readable bytes
inline void bits160(const uint8_t *q, uint32_t sh, uint64_t d[3])
{
    uint64_t w[4];
    memcpy(w, q, 32);
    for (int i = 0; i < 3; i++) d[i] = (w[i] >> sh) | ((w[i + 1] << 1) << (63 - sh));
}

struct PlaneTask {                       // one task of encode_planes: segments [a, e) into device groups [ga, ge)
    int64_t a, e;
    const uint32_t *gp, *pr, *gpos_all;
    const uint8_t *ln;
    const uint8_t *mm = nullptr;        // the segments' mm levels (n_mm_bins > 1), NULL = 0
    const uint64_t *pl;
    uint32_t *rec0;                     // where device group g0 lies
    int64_t g0, ga, ge;
    bool fast_store;
    int64_t used = 0, np = 0, nb = 0;   // results: groups needed (may exceed the region: counted, not written), pieces, columns
    uint32_t maxp = 0;
};

inline void put_empty_drec(uint32_t *o)
{
    o[0] = o[1] = o[2] = o[4] = o[5] = o[6] = o[7] = 0;
    o[3] = ISX_DREC_NO_EXC;
}

#ifndef ISX_PLANES_PREFETCH
#define ISX_PLANES_PREFETCH 16
#endif
constexpr int64_t PLANES_PREFETCH = ISX_PLANES_PREFETCH;     // segments ahead of the one being encoded
struct PlaneGroup {                     // the group being filled (layout rules: encode_delta's add_piece / close_group)
    alignas(64) uint32_t rec[ISX_DREC_GROUP][ISX_DREC_WORDS];
    int n = 0, open = -1;               // records / the dual record whose second half is still free (-1: none)
    uint32_t lo = 0, hi = 0, last = 0;  // lowest / highest start, highest last position
    int64_t g = 0, used = 0;            // next device group to write / groups the task needs (may exceed its region: counted, not written)
};
// a start below the group's base (input that is not sorted by start): the deltas written so far move up
__attribute__((noinline)) void rebase_drecs(PlaneGroup &G, uint32_t new_lo)
{
    const uint32_t up = G.lo - new_lo;
    for (int r = 0; r < G.n; r++) {
        uint32_t *o = G.rec[r];
        if ((o[0] >> 16) & 0xFFu) o[0] += up;
        if ((o[0] & ISX_DREC_DUAL) && ((o[4] >> 16) & 0xFFu)) o[4] += up;
    }
    G.lo = new_lo;
}

struct PlaneScratch {
    std::vector<uint32_t> gpos, pair;
    std::vector<uint8_t> len, mm;
    std::vector<uint64_t> planes;       // (64-byte aligned inside)
    uint64_t *pl = nullptr;
};

// the per-segment pass, compiled three times: portable (shift-and-mask), BMI2 (pext gathers the "differs" bits), AVX-512 VBMI2 + GFNI
// (one funnel shift for the five words, one GF(2) affine transform instead of the gathers)
#define ISX_PLANES_FN planes_task_portable
#define ISX_CLOSE_FN close_group_portable
#define ISX_DIFFERS_FN differs160_portable
#define ISX_PLANES_VARIANT 0
#include "seg_planes.inc"
#undef ISX_PLANES_FN
#undef ISX_CLOSE_FN
#undef ISX_DIFFERS_FN
#undef ISX_PLANES_VARIANT
#pragma GCC push_options
#pragma GCC target("bmi,bmi2,popcnt,lzcnt")
#define ISX_PLANES_FN planes_task_bmi2
#define ISX_CLOSE_FN close_group_bmi2
#define ISX_DIFFERS_FN differs160_bmi2
#define ISX_PLANES_VARIANT 1
#include "seg_planes.inc"
#undef ISX_PLANES_FN
#undef ISX_CLOSE_FN
#undef ISX_DIFFERS_FN
#undef ISX_PLANES_VARIANT
#pragma GCC pop_options
#pragma GCC push_options
#pragma GCC target("bmi,bmi2,popcnt,lzcnt,avx512f,avx512bw,avx512vl,avx512vbmi2,gfni")
#define ISX_PLANES_FN planes_task_avx512
#define ISX_CLOSE_FN close_group_avx512
#define ISX_DIFFERS_FN differs160_avx512
#define ISX_PLANES_VARIANT 2
#include "seg_planes.inc"
#undef ISX_PLANES_FN
#undef ISX_CLOSE_FN
#undef ISX_DIFFERS_FN
#undef ISX_PLANES_VARIANT
#pragma GCC pop_options

inline int planes_variant()
{
    static const int v = [] {
        if (const char *e = getenv("ISX_PLANES_VARIANT")) return std::max(0, std::min(2, atoi(e)));     // (the switch: tests of every path)
        if (!cpu_has_bmi2()) return 0;
        return cpu_has_avx512() && __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("gfni") ? 2 : 1;
    }();
    if (v == 2 && !(cpu_has_avx512() && __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("gfni") && __builtin_cpu_supports("bmi2"))) return __builtin_cpu_supports("bmi2") ? 1 : 0;
    if (v == 1 && !__builtin_cpu_supports("bmi2")) return 0;
    return v;
}

}  // namespace

int encode_planes(HostPool &pool, SegJob &J)
{
    const int64_t n = J.n_seg;
    const bool producer = (bool)J.produce_planes;
    const uint32_t *gpos_all = producer ? J.gpos_all : J.in2.gpos;
    const int n_tasks = (int)((n + TASK - 1) / TASK);
    const int64_t slack = std::max<int64_t>(J.slack_groups, 1);
    std::vector<int64_t> g_at((size_t)n_tasks + 1, 0);
    if (J.task_groups) for (int t = 0; t < n_tasks; t++) g_at[(size_t)t + 1] = std::max<int64_t>(J.task_groups[t], 1);
    else pool.run(n_tasks, [&](int t) {
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        g_at[(size_t)t + 1] = count_groups(gpos_all, a, e, 2 * ISX_DREC_GROUP) + slack;
    });
    for (int t = 0; t < n_tasks; t++) g_at[(size_t)t + 1] += g_at[(size_t)t];
    const int64_t n_groups = std::max<int64_t>(g_at[(size_t)n_tasks], 1);
    J.n_rec = n_groups * ISX_DREC_GROUP;
    J.need_slack = slack;
    if (J.n_rec > J.cap_rec) return SEG_CAPACITY;
    std::atomic<int> err{SEG_OK};
    std::vector<int64_t> bases_of((size_t)std::max(n_tasks, 1), 0), need_of((size_t)std::max(n_tasks, 1), 0), pieces_of((size_t)std::max(n_tasks, 1), 0);
    std::vector<uint32_t> maxp_of((size_t)std::max(n_tasks, 1), 0);
    const bool pairs = producer ? J.want_pairs : J.in2.pair != nullptr;
    const int64_t RG = J.ring_groups;
    constexpr size_t group_words = (size_t)ISX_DREC_GROUP * ISX_DREC_WORDS;
    const bool fast_store = cpu_has_avx512() && (reinterpret_cast<uintptr_t>(J.rec) & 63) == 0;
    const int variant = planes_variant();
    if (n == 0) {                                   // one empty group: the kernels want a stream
        if (RG) J.wave_begin(0);
        for (int r = 0; r < ISX_DREC_GROUP; r++) put_empty_drec(J.rec + (size_t)r * ISX_DREC_WORDS);
        J.gbase[0] = 0; J.cmin[0] = 0xFFFFFFFFu; J.cmax[0] = 0; J.cany[0] = 0;
        J.n_bases = 0; J.max_pair = 0; J.n_pieces = 0;
        if (RG) J.wave_flush(0, 0, 1);
        return SEG_OK;
    }
    auto run_task = [&](int t, int64_t wave_g0, int half) {
        if (err.load(std::memory_order_relaxed) != SEG_OK) return;
        const int64_t a = (int64_t)t * TASK, e = std::min<int64_t>(n, a + TASK);
        const uint32_t *gp, *pr;
        const uint8_t *ln, *mmv = nullptr;
        const uint64_t *pl;
        if (producer) {
            thread_local PlaneScratch S;
            if (S.gpos.size() < (size_t)TASK) {
                S.gpos.resize((size_t)TASK); S.pair.resize((size_t)TASK); S.len.resize((size_t)TASK); S.mm.resize((size_t)TASK);
                S.planes.resize((size_t)TASK * ISX_PLANE_WORDS + 8);
                S.pl = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(S.planes.data()) + 63) & ~(uintptr_t)63);
            }
            J.produce_planes(a, e - a, S.gpos.data(), S.len.data(), J.n_mm_bins > 1 ? S.mm.data() : nullptr, pairs ? S.pair.data() : nullptr, S.pl);
            gp = S.gpos.data() - a; ln = S.len.data() - a; pr = pairs ? S.pair.data() - a : nullptr;
            if (J.n_mm_bins > 1) mmv = S.mm.data() - a;
            pl = S.pl - (size_t)a * ISX_PLANE_WORDS;
        } else {
            gp = J.in2.gpos; ln = J.in2.len; pr = pairs ? J.in2.pair : nullptr; pl = J.in2.planes;
            if (J.n_mm_bins > 1) mmv = J.in2.mm;
        }
        // (ring mode: device group gi of this wave lies at slot gi - wave_g0 of the wave's half)
        PlaneTask K;
        K.a = a; K.e = e; K.gp = gp; K.ln = ln; K.mm = mmv; K.pr = pr; K.pl = pl; K.gpos_all = gpos_all;
        K.rec0 = RG ? J.rec + (size_t)((int64_t)half * RG) * group_words : J.rec;
        K.g0 = RG ? wave_g0 : 0; K.ga = g_at[(size_t)t]; K.ge = g_at[(size_t)t + 1]; K.fast_store = fast_store;
        if (variant == 2) planes_task_avx512(J, err, K);
        else if (variant == 1) planes_task_bmi2(J, err, K);
        else planes_task_portable(J, err, K);
        if (err.load(std::memory_order_relaxed) != SEG_OK) return;
        need_of[(size_t)t] = K.used;
        bases_of[(size_t)t] = K.nb; maxp_of[(size_t)t] = K.maxp; pieces_of[(size_t)t] = K.np;
    };
    if (!RG) pool.run(n_tasks, [&](int t) { run_task(t, 0, 0); });
    else {
        int half = 0;
        for (int t0 = 0; t0 < n_tasks && err.load() == SEG_OK;) {
            int t1 = t0 + 1;
            while (t1 < n_tasks && g_at[(size_t)t1 + 1] - g_at[(size_t)t0] <= RG) t1++;
            if (g_at[(size_t)t1] - g_at[(size_t)t0] > RG) return SEG_CAPACITY;         // one task alone outgrows a half (ring far too small)
            J.wave_begin(half);
            const int64_t wg0 = g_at[(size_t)t0];
            pool.run(t1 - t0, [&](int k) { run_task(t0 + k, wg0, half); });
            if (err.load() == SEG_OK) J.wave_flush(half, wg0, g_at[(size_t)t1]);
            t0 = t1; half ^= 1;
        }
    }
    if (err.load() != SEG_OK) return err.load();
    J.n_bases = 0; J.max_pair = 0; J.n_pieces = 0;
    int64_t worst = 0;
    bool fits = true;
    J.task_need.assign(need_of.begin(), need_of.begin() + n_tasks);
    for (int t = 0; t < n_tasks; t++) {
        J.n_bases += bases_of[(size_t)t]; J.max_pair = std::max(J.max_pair, maxp_of[(size_t)t]); J.n_pieces += pieces_of[(size_t)t];
        const int64_t region = g_at[(size_t)t + 1] - g_at[(size_t)t];
        if (need_of[(size_t)t] > region) fits = false;
        worst = std::max(worst, need_of[(size_t)t] - (region - (J.task_groups ? 0 : slack)));
    }
    J.need_slack = fits ? slack : std::max<int64_t>(worst, slack + 1);
    if (!fits) return SEG_CAPACITY;                 // some task outgrew its region: encode again with task_groups = task_need
    return SEG_OK;
}


// The reference codes of a batch as they travel and lie in a slot: a 2-bit plane (A C T G; anything else as 0), four positions a
// byte, and a bit plane marking the positions that are not A/C/T/G.  plane2 holds (n_pos + 3) / 4 bytes, nplane (n_pos + 7) / 8;
// returns whether the N plane marks any position (it is always written).
namespace {
__attribute__((target("bmi2")))
inline void pack_ref_piece_bmi2(const uint8_t *r, int64_t n, uint8_t *o2, uint8_t *on, bool &seen)
{
    int64_t i = 0;
    uint64_t any = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, r + i, 8);
        const uint64_t bad = x & 0xFCFCFCFCFCFCFCFCull;                        // a code above 3 has one of these bits
        if (__builtin_expect(bad == 0, 1)) {
            const uint16_t v = (uint16_t)_pext_u64(x, 0x0303030303030303ull);
            memcpy(o2 + (i >> 2), &v, 2);
            on[i >> 3] = 0;
            continue;
        }
        uint64_t t = bad | (bad >> 4);
        t |= t >> 2; t |= t >> 1;                                               // bit 0 of every byte: the byte is not a base
        const uint64_t nb = t & 0x0101010101010101ull;
        const uint64_t keep = ~(nb * 0xFFull);                                  // bytes that are bases
        const uint16_t v = (uint16_t)_pext_u64(x & keep, 0x0303030303030303ull);
        memcpy(o2 + (i >> 2), &v, 2);
        on[i >> 3] = (uint8_t)_pext_u64(nb, 0x0101010101010101ull);
        any |= nb;
    }
    for (; i < n; i += 8) {                                                      // the last, partial byte group
        uint32_t lo = 0, hi = 0, nbits = 0;
        const int64_t m = std::min<int64_t>(8, n - i);
        for (int64_t k = 0; k < m; k++) {
            const uint32_t c = r[i + k];
            const uint32_t b = c > 3u;
            nbits |= b << k;
            const uint32_t v = b ? 0u : c;
            if (k < 4) lo |= v << (2 * k); else hi |= v << (2 * (k - 4));
        }
        o2[i >> 2] = (uint8_t)lo;
        if (m > 4) o2[(i >> 2) + 1] = (uint8_t)hi;
        on[i >> 3] = (uint8_t)nbits;
        any |= nbits;
    }
    seen = any != 0;
}

inline void pack_ref_piece_scalar(const uint8_t *r, int64_t n, uint8_t *o2, uint8_t *on, bool &seen)
{
    seen = false;
    for (int64_t i = 0; i < n; i += 8) {
        uint32_t lo = 0, hi = 0, nb = 0;
        const int64_t m = std::min<int64_t>(8, n - i);
        for (int64_t k = 0; k < m; k++) {
            const uint32_t c = r[i + k];
            const uint32_t bad = c > 3u;
            nb |= bad << k;
            const uint32_t v = bad ? 0u : c;
            if (k < 4) lo |= v << (2 * k); else hi |= v << (2 * (k - 4));
        }
        o2[i >> 2] = (uint8_t)lo;
        if (m > 4) o2[(i >> 2) + 1] = (uint8_t)hi;
        on[i >> 3] = (uint8_t)nb;
        seen |= nb != 0;
    }
}
}  // namespace

bool pack_ref_planes(HostPool &pool, const uint8_t *ref, int64_t n_pos, uint8_t *plane2, uint8_t *nplane)
{
    const int64_t piece = (int64_t)256 << 10;               // a multiple of 8
    const int n_tasks = (int)((n_pos + piece - 1) / piece);
    std::atomic<int> any{0};
    const bool bmi2 = cpu_has_bmi2();
    auto cp = [&](int t) {
        const int64_t a = (int64_t)t * piece, e = std::min<int64_t>(n_pos, a + piece);
        bool seen = false;
        if (bmi2) pack_ref_piece_bmi2(ref + a, e - a, plane2 + (a >> 2), nplane + (a >> 3), seen);
        else pack_ref_piece_scalar(ref + a, e - a, plane2 + (a >> 2), nplane + (a >> 3), seen);
        if (seen) any.store(1, std::memory_order_relaxed);
    };
    if (n_tasks > 1) pool.run(n_tasks, cp); else if (n_tasks == 1) cp(0);
    return any.load() != 0;
}

// fifteen words of ten 3-bit codes -> one line of planes (codes >= 4: skipped column, base bits 0 -- except code 5, a base that is not
// A/C/T/G but passed the filter: base bits 1 and the line's marker flag, bit 63 of word 7; isx_read_planes.mm in include/instrain_amd.h)
void planes_from_words(const uint32_t *w, uint32_t L, uint64_t *P)
{
    uint64_t b[5] = {0, 0, 0, 0, 0}, sk[3] = {0, 0, 0};
    for (uint32_t j = 0; j < L; j++) {
        const uint32_t c = (w[j / 10] >> (3 * (j % 10))) & 7u;
        if (c >= 4) {
            sk[j >> 6] |= (uint64_t)1 << (j & 63);
            if (c == 5) { b[j >> 5] |= (uint64_t)1 << (2 * (j & 31)); sk[2] |= (uint64_t)1 << 63; }
        } else b[j >> 5] |= (uint64_t)c << (2 * (j & 31));
    }
    for (int i = 0; i < 5; i++) P[i] = b[i];
    for (int i = 0; i < 3; i++) P[5 + i] = sk[i];
}

}  // namespace isxenc

namespace {

// ASCII base -> code of a base that passes the quality filter (P2C order A C T G, profile_utilities.py:34; anything
// else only makes its mm level present, :279-285)
inline uint32_t ascii_code(char c)
{
    switch (c) {
    case 'A': return 0; case 'C': return 1; case 'T': return 2; case 'G': return 3;
    default: return 5;
    }
}

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

// calls f(flat position of the run's first kept column, query offset of that column, columns) for every M / = / X run of a
// read, truncated to [clip_lo, clip_hi); returns false on a bad CIGAR operator
template <class F>
inline bool for_runs(const uint32_t *cig, int64_t n_cig, int64_t ref_start, int64_t clip_lo, int64_t clip_hi, F &&f)
{
    int64_t ref = ref_start, q = 0;
    for (int64_t k = 0; k < n_cig; k++) {
        const uint32_t op = cig[k] & 15u;
        const int64_t n = cig[k] >> 4;
        if (op == OP_M || op == OP_EQ || op == OP_X) {
            const int64_t j0 = std::max<int64_t>(0, clip_lo - ref), j1 = std::min<int64_t>(n, clip_hi - ref);
            if (j1 > j0) f(ref + j0, q + j0, j1 - j0);
            q += n; ref += n;
        } else if (op == OP_I || op == OP_S) q += n;
        else if (op == OP_D || op == OP_N) ref += n;
        else if (op != OP_H && op != OP_P) return false;
    }
    return true;
}

}  // namespace

extern "C" {

int isx_encode_segs(const isx_segs *segs, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int64_t cap_rec, uint32_t *rec,
                    uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec)
{
    return isx_encode_segs_ring(segs, n_pos, n_mm_bins, host_threads, cap_rec, 0, rec, gbase, pair_out, n_rec);
}

int isx_encode_segs_ring(const isx_segs *segs, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int64_t cap_rec, int64_t ring_records,
                         uint32_t *rec, uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec)
{
    if (!segs || !rec || !gbase || !n_rec || segs->n_seg < 0 || cap_rec < ISX_SEG_GROUP || (cap_rec % ISX_SEG_GROUP) ||
        (segs->n_seg && (!segs->gpos || !segs->len || !segs->bases)) || (segs->pair && !pair_out) || ring_records < 0 ||
        (ring_records % (2 * ISX_SEG_GROUP))) {
        isx_set_error("isx_encode_segs: bad argument");
        return ISX_ERR_ARG;
    }
    std::vector<uint32_t> ring;
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    std::vector<uint32_t> cmin((size_t)(cap_rec / ISX_SEG_GROUP)), cmax(cmin.size());
    std::vector<uint8_t> cany(cmin.size());
    isxenc::SegJob J;
    J.in = *segs; J.n_seg = segs->n_seg; J.n_pos = n_pos; J.n_mm_bins = std::max(1, n_mm_bins);
    J.rec = rec; J.gbase = gbase; J.pair_out = segs->pair ? pair_out : nullptr;
    J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data(); J.cap_rec = cap_rec;
    if (ring_records) {         // the pipe's ring mode with a memcpy standing in for the DMA engine
        const int64_t half = ring_records / 2;
        ring.assign((size_t)ring_records * ISX_SEG_REC_WORDS, 0xABABABABu);
        J.rec = ring.data();
        J.ring_groups = half / ISX_SEG_GROUP;
        J.wave_begin = [](int) {};
        J.wave_flush = [&](int h, int64_t g0, int64_t g1) {
            const size_t gw = (size_t)ISX_SEG_GROUP * ISX_SEG_REC_WORDS;
            memcpy(rec + (size_t)g0 * gw, ring.data() + (size_t)h * half * ISX_SEG_REC_WORDS, (size_t)(g1 - g0) * gw * 4);
            std::fill_n(ring.begin() + (ptrdiff_t)((size_t)h * half * ISX_SEG_REC_WORDS), (size_t)half * ISX_SEG_REC_WORDS, 0xABABABABu);   // stale data must never travel
        };
    }
    const int rc = isxenc::encode_segs(pool, J);
    if (rc == isxenc::SEG_CAPACITY) { isx_set_error("isx_encode_segs: the stream does not fit cap_rec records"); return ISX_ERR_CAPACITY; }
    if (rc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
    if (rc == isxenc::SEG_BAD_POS) { isx_set_error("a segment reaches beyond n_pos"); return ISX_ERR_ARG; }
    if (rc == isxenc::SEG_BAD_LEN) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
    *n_rec = J.n_rec;
    return ISX_OK;
}

int isx_encode_delta(const isx_segs *segs, const uint8_t *ref, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int32_t slack_groups,
                     int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec, int64_t *need_slack)
{
    if (!segs || !ref || !rec || !gbase || !n_rec || segs->n_seg < 0 || cap_rec < ISX_DREC_GROUP || (cap_rec % ISX_DREC_GROUP) || n_pos <= 0 ||
        (segs->n_seg && (!segs->gpos || !segs->len || !segs->bases)) || ring_records < 0 ||
        (ring_records % (2 * ISX_DREC_GROUP)) || slack_groups < 0) {
        isx_set_error("isx_encode_delta: bad argument");
        return ISX_ERR_ARG;
    }
    std::vector<uint32_t> ring;
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    std::vector<uint32_t> cmin((size_t)(cap_rec / ISX_DREC_GROUP)), cmax(cmin.size());
    std::vector<uint8_t> cany(cmin.size());
    isxenc::SegJob J;
    J.in = *segs; J.n_seg = segs->n_seg; J.n_pos = n_pos; J.n_mm_bins = std::max(1, n_mm_bins);
    J.ref = ref; J.slack_groups = std::max(1, slack_groups);
    (void)pair_out;             // (the read-pair ids travel inside the records)
    J.rec = rec; J.gbase = gbase; J.pair_out = nullptr;
    J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data(); J.cap_rec = cap_rec;
    if (ring_records) {         // the pipe's ring mode with a memcpy standing in for the DMA engine
        const int64_t half = ring_records / 2;
        ring.assign((size_t)ring_records * ISX_DREC_WORDS, 0xABABABABu);
        J.rec = ring.data();
        J.ring_groups = half / ISX_DREC_GROUP;
        J.wave_begin = [](int) {};
        J.wave_flush = [&](int h, int64_t g0, int64_t g1) {
            const size_t gw = (size_t)ISX_DREC_GROUP * ISX_DREC_WORDS;
            memcpy(rec + (size_t)g0 * gw, ring.data() + (size_t)h * half * ISX_DREC_WORDS, (size_t)(g1 - g0) * gw * 4);
            std::fill_n(ring.begin() + (ptrdiff_t)((size_t)h * half * ISX_DREC_WORDS), (size_t)half * ISX_DREC_WORDS, 0xABABABABu);
        };
    }
    const int rc = isxenc::encode_delta(pool, J);
    if (need_slack) *need_slack = J.need_slack;
    *n_rec = J.n_rec;
    if (rc == isxenc::SEG_CAPACITY) {
        isx_set_error(J.need_slack > J.slack_groups ? "isx_encode_delta: a task needs more spare groups than slack_groups (see *need_slack)"
                                                    : "isx_encode_delta: the stream does not fit cap_rec records");
        return ISX_ERR_CAPACITY;
    }
    if (rc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
    if (rc == isxenc::SEG_BAD_POS) { isx_set_error("a segment reaches beyond n_pos"); return ISX_ERR_ARG; }
    if (rc == isxenc::SEG_BAD_LEN) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
    return ISX_OK;
}

int64_t isx_delta_records_needed(const uint32_t *gpos, int64_t n_seg, int32_t host_threads, int32_t slack_groups)
{
    if (n_seg < 0 || (n_seg && !gpos) || slack_groups < 0) return -1;
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    return isxenc::delta_groups_needed(pool, gpos, n_seg, std::max(1, slack_groups)) * ISX_DREC_GROUP;
}

int64_t isx_seg_records_needed(const uint32_t *gpos, int64_t n_seg, int32_t host_threads)
{
    if (n_seg < 0 || (n_seg && !gpos)) return -1;
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    return isxenc::seg_groups_needed(pool, gpos, n_seg) * ISX_SEG_GROUP;
}

int isx_count_read_segs(int64_t n_reads, const uint32_t *cigar, const int64_t *cigar_off, const int64_t *ref_start,
                        const int64_t *clip_lo, const int64_t *clip_hi, int64_t *n_seg)
{
    if (n_reads < 0 || !n_seg || (n_reads && (!cigar || !cigar_off || !ref_start || !clip_lo || !clip_hi))) {
        isx_set_error("isx_count_read_segs: bad argument");
        return ISX_ERR_ARG;
    }
    int64_t n = 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const bool ok = for_runs(cigar + cigar_off[r], cigar_off[r + 1] - cigar_off[r], ref_start[r], clip_lo[r], clip_hi[r],
                                 [&](int64_t, int64_t, int64_t cols) { n += (cols + ISX_SEG_BASES - 1) / ISX_SEG_BASES; });
        if (!ok) { isx_set_error("isx_count_read_segs: unknown CIGAR operator"); return ISX_ERR_ARG; }
    }
    *n_seg = n;
    return ISX_OK;
}

int isx_pack_reads(int64_t n_reads, const int64_t *ref_start, const int64_t *clip_lo, const int64_t *clip_hi,
                   const uint32_t *cigar, const int64_t *cigar_off, const char *seq, const uint8_t *qual, const int64_t *seq_off,
                   const uint8_t *mm, const uint32_t *pair, int32_t min_base_quality, int64_t cap_seg, uint32_t *seg_gpos,
                   uint8_t *seg_len, uint8_t *seg_mm, uint32_t *seg_pair, uint32_t *seg_bases, int64_t *n_seg)
{
    if (n_reads < 0 || !n_seg || cap_seg < 0 || (n_reads && (!ref_start || !clip_lo || !clip_hi || !cigar || !cigar_off || !seq || !qual || !seq_off)) ||
        (cap_seg && (!seg_gpos || !seg_len || !seg_bases)) || (pair && cap_seg && !seg_pair)) {
        isx_set_error("isx_pack_reads: bad argument");
        return ISX_ERR_ARG;
    }
    int64_t n = 0;
    bool full = false, bad = false;
    for (int64_t r = 0; r < n_reads && !full && !bad; r++) {
        const char *sq = seq + seq_off[r];
        const uint8_t *ql = qual + seq_off[r];
        const int64_t q_len = seq_off[r + 1] - seq_off[r];
        const bool ok = for_runs(cigar + cigar_off[r], cigar_off[r + 1] - cigar_off[r], ref_start[r], clip_lo[r], clip_hi[r],
                                 [&](int64_t pos, int64_t q0, int64_t cols) {
            if (q0 + cols > q_len || pos < 0 || pos + cols > (int64_t)0xFFFFFFFFll) { bad = true; return; }
            for (int64_t c0 = 0; c0 < cols && !full; c0 += ISX_SEG_BASES) {
                const int L = (int)std::min<int64_t>(ISX_SEG_BASES, cols - c0);
                if (n >= cap_seg) { full = true; return; }
                uint32_t *w = seg_bases + (size_t)n * ISX_SEG_WORDS;
                for (int k = 0; k < ISX_SEG_WORDS; k++) w[k] = ISX_SEG_SKIP_WORD;
                for (int j = 0; j < L; j++) {
                    const int64_t qi = q0 + c0 + j;
                    const uint32_t code = (int)ql[qi] >= min_base_quality ? ascii_code(sq[qi]) : 4u;
                    w[j / 10] = (w[j / 10] & ~(7u << (3 * (j % 10)))) | (code << (3 * (j % 10)));
                }
                seg_gpos[n] = (uint32_t)(pos + c0); seg_len[n] = (uint8_t)L;
                if (seg_mm) seg_mm[n] = mm ? mm[r] : (uint8_t)0;
                if (seg_pair) seg_pair[n] = pair ? pair[r] : 0u;
                n++;
            }
        });
        if (!ok) { isx_set_error("isx_pack_reads: unknown CIGAR operator"); return ISX_ERR_ARG; }
    }
    if (bad) { isx_set_error("isx_pack_reads: a CIGAR reaches beyond its read's bases or the flat space"); return ISX_ERR_ARG; }
    if (full) { isx_set_error("isx_pack_reads: more segments than cap_seg (isx_count_read_segs gives the number)"); return ISX_ERR_CAPACITY; }
    *n_seg = n;
    return ISX_OK;
}

int isx_pack_ref_planes(const uint8_t *ref, int64_t n_pos, int32_t host_threads, uint8_t *plane2, uint8_t *nplane, int32_t *has_n)
{
    if (!ref || n_pos <= 0 || !plane2 || !nplane) { isx_set_error("isx_pack_ref_planes: bad argument"); return ISX_ERR_ARG; }
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    const bool any = isxenc::pack_ref_planes(pool, ref, n_pos, plane2, nplane);
    if (has_n) *has_n = any ? 1 : 0;
    return ISX_OK;
}

int isx_planes_from_segs(const isx_segs *segs, int32_t host_threads, uint64_t *planes)
{
    if (!segs || segs->n_seg < 0 || (segs->n_seg && (!segs->len || !segs->bases || !planes))) { isx_set_error("isx_planes_from_segs: bad argument"); return ISX_ERR_ARG; }
    const int64_t n = segs->n_seg, piece = 8192;
    std::atomic<bool> bad{false};
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    pool.run((int)((n + piece - 1) / piece), [&](int t) {
        const int64_t a = (int64_t)t * piece, e = std::min<int64_t>(n, a + piece);
        for (int64_t i = a; i < e; i++) {
            const uint32_t L = segs->len[i];
            if (L == 0 || L > ISX_SEG_BASES) { bad.store(true); return; }
            isxenc::planes_from_words(segs->bases + (size_t)i * ISX_SEG_WORDS, L, planes + (size_t)i * ISX_PLANE_WORDS);
        }
    });
    if (bad.load()) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
    return ISX_OK;
}

int isx_encode_planes(const isx_read_planes *reads, const isx_ref_planes *ref, int64_t n_pos, int32_t host_threads, int32_t slack_groups,
                      int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, int64_t *n_rec, int64_t *need_slack)
{
    return isx_encode_planes_mm(reads, ref, n_pos, 1, host_threads, slack_groups, cap_rec, ring_records, rec, gbase, n_rec, need_slack);
}

int isx_encode_planes_mm(const isx_read_planes *reads, const isx_ref_planes *ref, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int32_t slack_groups,
                         int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, int64_t *n_rec, int64_t *need_slack)
{
    if (n_mm_bins < 1 || n_mm_bins > 128) { isx_set_error("isx_encode_planes: n_mm_bins must be in [1, 128]"); return ISX_ERR_ARG; }
    if (!reads || !ref || !ref->plane2 || !rec || !gbase || !n_rec || reads->n_seg < 0 || cap_rec < ISX_DREC_GROUP || (cap_rec % ISX_DREC_GROUP) || n_pos <= 0 ||
        (reads->n_seg && (!reads->gpos || !reads->len || !reads->planes)) || ring_records < 0 ||
        (ring_records % (2 * ISX_DREC_GROUP)) || slack_groups < 0) {
        isx_set_error("isx_encode_planes: bad argument");
        return ISX_ERR_ARG;
    }
    std::vector<uint32_t> ring;
    isxenc::HostPool pool(std::max(1, host_threads), -1, false);
    std::vector<uint32_t> cmin((size_t)(cap_rec / ISX_DREC_GROUP)), cmax(cmin.size());
    std::vector<uint8_t> cany(cmin.size());
    isxenc::SegJob J;
    J.in2 = *reads; J.n_seg = reads->n_seg; J.n_pos = n_pos; J.n_mm_bins = n_mm_bins;
    J.ref2 = ref->plane2; J.refn = ref->nplane; J.slack_groups = std::max(1, slack_groups);
    J.rec = rec; J.gbase = gbase; J.pair_out = nullptr;
    J.cmin = cmin.data(); J.cmax = cmax.data(); J.cany = cany.data(); J.cap_rec = cap_rec;
    if (ring_records) {         // the pipe's ring mode with a memcpy standing in for the DMA engine
        const int64_t half = ring_records / 2;
        ring.assign((size_t)ring_records * ISX_DREC_WORDS + 16, 0xABABABABu);
        uint32_t *r0 = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(ring.data()) + 63) & ~(uintptr_t)63);
        J.rec = r0;
        J.ring_groups = half / ISX_DREC_GROUP;
        J.wave_begin = [](int) {};
        J.wave_flush = [&, r0](int h, int64_t g0, int64_t g1) {
            const size_t gw = (size_t)ISX_DREC_GROUP * ISX_DREC_WORDS;
            memcpy(rec + (size_t)g0 * gw, r0 + (size_t)h * half * ISX_DREC_WORDS, (size_t)(g1 - g0) * gw * 4);
            std::fill_n(r0 + (size_t)h * half * ISX_DREC_WORDS, (size_t)half * ISX_DREC_WORDS, 0xABABABABu);
        };
    }
    const int rc = isxenc::encode_planes(pool, J);
    if (need_slack) *need_slack = J.need_slack;
    *n_rec = J.n_rec;
    if (rc == isxenc::SEG_CAPACITY) {
        isx_set_error(J.need_slack > J.slack_groups ? "isx_encode_planes: a task needs more spare groups than slack_groups (see *need_slack)"
                                                    : "isx_encode_planes: the stream does not fit cap_rec records");
        return ISX_ERR_CAPACITY;
    }
    if (rc == isxenc::SEG_BAD_POS) { isx_set_error("a segment reaches beyond n_pos"); return ISX_ERR_ARG; }
    if (rc == isxenc::SEG_BAD_LEN) { isx_set_error("a segment's length is not in [1, 150]"); return ISX_ERR_ARG; }
    if (rc == isxenc::SEG_MM_RANGE) { isx_set_error("a segment has mm >= n_mm_bins"); return ISX_ERR_MM_RANGE; }
    return ISX_OK;
}

int isx_pack_read_planes(int64_t n_reads, const int64_t *ref_start, const int64_t *clip_lo, const int64_t *clip_hi,
                         const uint32_t *cigar, const int64_t *cigar_off, const char *seq, const uint8_t *qual, const int64_t *seq_off,
                         const uint32_t *pair, int32_t min_base_quality, int64_t cap_seg, uint32_t *seg_gpos,
                         uint8_t *seg_len, uint32_t *seg_pair, uint64_t *seg_planes, int64_t *n_seg)
{
    if (n_reads < 0 || !n_seg || cap_seg < 0 || (n_reads && (!ref_start || !clip_lo || !clip_hi || !cigar || !cigar_off || !seq || !qual || !seq_off)) ||
        (cap_seg && (!seg_gpos || !seg_len || !seg_planes)) || (pair && cap_seg && !seg_pair)) {
        isx_set_error("isx_pack_read_planes: bad argument");
        return ISX_ERR_ARG;
    }
    int64_t n = 0;
    bool full = false, bad = false;
    for (int64_t r = 0; r < n_reads && !full && !bad; r++) {
        const char *sq = seq + seq_off[r];
        const uint8_t *ql = qual + seq_off[r];
        const int64_t q_len = seq_off[r + 1] - seq_off[r];
        const bool ok = for_runs(cigar + cigar_off[r], cigar_off[r + 1] - cigar_off[r], ref_start[r], clip_lo[r], clip_hi[r],
                                 [&](int64_t pos, int64_t q0, int64_t cols) {
            if (q0 + cols > q_len || pos < 0 || pos + cols > (int64_t)0xFFFFFFFFll) { bad = true; return; }
            for (int64_t c0 = 0; c0 < cols && !full; c0 += ISX_SEG_BASES) {
                const int L = (int)std::min<int64_t>(ISX_SEG_BASES, cols - c0);
                if (n >= cap_seg) { full = true; return; }
                uint64_t *P = seg_planes + (size_t)n * ISX_PLANE_WORDS;
                for (int k = 0; k < ISX_PLANE_WORDS; k++) P[k] = 0;
                for (int j = 0; j < L; j++) {
                    const int64_t qi = q0 + c0 + j;
                    const uint32_t code = (int)ql[qi] >= min_base_quality ? ascii_code(sq[qi]) : 4u;
                    if (code >= 4) {
                        P[5 + (j >> 6)] |= (uint64_t)1 << (j & 63);
                        if (code == 5) { P[j >> 5] |= (uint64_t)1 << (2 * (j & 31)); P[7] |= (uint64_t)1 << 63; }     // a non-ACGT base that passed the filter: marked (isx_read_planes.mm)
                    } else P[j >> 5] |= (uint64_t)code << (2 * (j & 31));
                }
                seg_gpos[n] = (uint32_t)(pos + c0); seg_len[n] = (uint8_t)L;
                if (seg_pair) seg_pair[n] = pair ? pair[r] : 0u;
                n++;
            }
        });
        if (!ok) { isx_set_error("isx_pack_read_planes: unknown CIGAR operator"); return ISX_ERR_ARG; }
    }
    if (bad) { isx_set_error("isx_pack_read_planes: a CIGAR reaches beyond its read's bases or the flat space"); return ISX_ERR_ARG; }
    if (full) { isx_set_error("isx_pack_read_planes: more segments than cap_seg (isx_count_read_segs gives the number)"); return ISX_ERR_CAPACITY; }
    *n_seg = n;
    return ISX_OK;
}

}  // extern "C"
