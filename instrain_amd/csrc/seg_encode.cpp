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
#include <cstring>
#include <string>
#include <vector>

void isx_set_error(const std::string &msg);

namespace isxenc {

namespace {

constexpr int64_t TASK = 4096;          // segments per task (a multiple of ISX_SEG_GROUP): 256 KiB of payload
constexpr uint32_t SPAN = 65535u;       // largest delta a header can carry

// groups the segments [a, e) need: greedy, arrival order
inline int64_t count_groups(const uint32_t *gpos, int64_t a, int64_t e)
{
    int64_t n = 0;
    for (int64_t i = a; i < e;) {
        uint32_t lo = gpos[i], hi = gpos[i];
        int64_t j = i + 1;
        for (; j < e && j - i < ISX_SEG_GROUP; j++) {
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

}  // extern "C"
