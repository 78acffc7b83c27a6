// isx_summary.hip -- per-scaffold merge summaries on the device.
//
// Replaces the position-sized part of
//   make_coverage_table          /root/reference/inStrain/profile/profile_utilities.py:425-506
//   mm_counts_to_counts_shrunk   profile_utilities.py:508-532   (cumulative coverage over levels <= mm)
//   get_basewise_clons           profile_utilities.py:534-546   (per position the clonality of the highest
//                                                                 level <= mm that has one)
// i.e. for every (scaffold, mm): number of covered positions, sum / sum of squares / median of the
// cumulative coverage, count / sum / median of the clonalities (and of the rarefied ones).  The
// table-sized rest (SNP counts from the SNV table, ANI, expected breadth, column naming) stays on
// the host (instrain_amd/profile/profile_utilities.py make_coverage_table).
//
// Levels are applied in ascending order onto three flat per-position arrays that stay on the
// device; per level one pass of segmented reductions (register partials, one set of atomics per
// lane) and three rocPRIM segmented radix sorts for the medians.
#include <chrono>
#include <algorithm>
#include <vector>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "isx_internal.h"
#include "isx_summary.h"

namespace {

struct Acc {
    unsigned long long nonzero, sum, sumsq, counted, counted_r;
    double sum_clon, sum_clon_r;
    unsigned int present, pad;
};

__device__ __forceinline__ int find_seg(const int64_t *bounds, int n_seg, uint32_t g)
{
    int lo = 0, hi = n_seg;             // bounds[lo] <= g < bounds[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid] <= (int64_t)g) lo = mid; else hi = mid;
    }
    return lo;
}

// dense path (one level): counts / clonalities are already per position.  A pipe slot that keeps no count table has the
// coverage itself (16 bits, saturating) and the exact values of the few saturated positions in a list.
__global__ void k_level_dense(const uint4 *counts, const uint16_t *cov16, const float *clon, const float *clon_r, uint32_t n_pos,
                              uint32_t *cov, float *cv, float *cr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    if (counts) { const uint4 c = counts[i]; cov[i] = c.x + c.y + c.z + c.w; }
    else cov[i] = cov16[i];
    cv[i] = clon[i];
    cr[i] = clon_r[i];
}

__global__ void k_patch_saturated(const uint2 *sat, uint32_t n_sat, uint32_t *cov)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sat && sat[i].y >= 65535u) cov[sat[i].x] = sat[i].y;
}

static void launch_level_dense(const SummaryIn &in, dim3 grid, dim3 blk, hipStream_t s, uint32_t n_pos, uint32_t *cov, float *cv, float *cr)
{
    hipLaunchKernelGGL(k_level_dense, grid, blk, 0, s, in.counts, in.cov16, in.clon, in.clon_r, n_pos, cov, cv, cr);
    if (!in.counts && in.n_sat) hipLaunchKernelGGL(k_patch_saturated, dim3((in.n_sat + 255) / 256), blk, 0, s, in.sat, in.n_sat, cov);
}

// mm path: add level `mm` of the entry table (window slabs + overflow) onto the running arrays.
// Every (position, level) occurs once, so plain read-modify-writes are race free.
__global__ void k_level_apply(const isx_entry *entries, const uint32_t *win_nent, uint32_t slab, uint32_t n_win,
                              uint64_t ovf0, uint32_t n_ovf, uint32_t mm, uint32_t *cov, float *cv, float *cr,
                              const int64_t *bounds, int n_seg, Acc *acc)
{
    const uint64_t total = ovf0 + n_ovf;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (i < ovf0) {
            const uint32_t w = (uint32_t)(i / slab);
            if ((uint32_t)(i - (uint64_t)w * slab) >= win_nent[w]) continue;
        }
        const isx_entry e = entries[i];
        if (e.mm != mm) continue;
        const uint32_t s = e.cnt[0] + e.cnt[1] + e.cnt[2] + e.cnt[3];
        if (s) cov[e.gpos] += s;
        // covT has this level on this scaffold as soon as one column created it -- also when its only read showed a
        // non-ACGT base there (update_covT writes sum(count) = 0; shrink_basewise keeps the key with an empty Series)
        Acc *a = &acc[find_seg(bounds, n_seg, e.gpos)];
        if (!a->present) a->present = 1;
        if (e.clon == e.clon) cv[e.gpos] = e.clon;
        if (e.clon_rarefied == e.clon_rarefied) cr[e.gpos] = e.clon_rarefied;
    }
}

// segmented reductions: each lane walks a contiguous tile, keeps partials while the scaffold stays
// the same and flushes them with atomics when it changes / at the end
__global__ void __launch_bounds__(256) k_seg_reduce(const uint32_t *cov, const float *cv, const float *cr, uint32_t n_pos,
                                                    const int64_t *bounds, int n_seg, Acc *acc, int dense)
{
    const uint32_t TILE = 64;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = t * TILE;
    if (p0 >= n_pos) return;
    const uint32_t p1 = (uint32_t)min((uint64_t)n_pos, p0 + TILE);
    int seg = find_seg(bounds, n_seg, (uint32_t)p0);
    unsigned long long nz = 0, sum = 0, sq = 0, cn = 0, cnr = 0;
    double sc = 0.0, scr = 0.0;
    auto flush = [&]() {
        Acc *a = &acc[seg];
        if (nz) atomicAdd(&a->nonzero, nz);
        if (sum) atomicAdd(&a->sum, sum);
        if (sq) atomicAdd(&a->sumsq, sq);
        if (cn) { atomicAdd(&a->counted, cn); atomicAdd(&a->sum_clon, sc); }
        if (cnr) { atomicAdd(&a->counted_r, cnr); atomicAdd(&a->sum_clon_r, scr); }
        if (dense && nz && !a->present) a->present = 1;
        nz = sum = sq = cn = cnr = 0; sc = scr = 0.0;
    };
    for (uint32_t p = (uint32_t)p0; p < p1; p++) {
        if ((int64_t)p >= bounds[seg + 1]) { flush(); seg = find_seg(bounds, n_seg, p); }
        const unsigned long long c = cov[p];
        nz += c ? 1 : 0; sum += c; sq += c * c;
        const float v = cv[p], r = cr[p];
        if (v == v) { cn++; sc += (double)v; }
        if (r == r) { cnr++; scr += (double)r; }
    }
    flush();
}

// np.median over each segment of the sorted keys: the first `n` values of the segment count
// (n = segment length for coverage, = number of non-NaN values for the clonalities; NaN sorts last)
template <class T>
__global__ void k_pick_median(const T *sorted, const uint32_t *seg_off, int n_seg, const Acc *acc, int which, double *out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t b = seg_off[s];
    uint64_t n = seg_off[s + 1] - b;
    if (which == 1) n = acc[s].counted;
    if (which == 2) n = acc[s].counted_r;
    double m = __builtin_nan("");
    if (n) {
        const uint64_t h = n >> 1;
        m = (n & 1) ? (double)sorted[b + h] : ((double)sorted[b + h - 1] + (double)sorted[b + h]) / 2.0;
    }
    out[s] = m;
}

__global__ void k_pack_rows(const Acc *acc, const double *med_cov, const double *med_c, const double *med_r, int n_seg,
                            int mm, int M, isx_scaffold_level *out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const Acc a = acc[s];
    isx_scaffold_level r;
    r.nonzero = (int64_t)a.nonzero; r.sum_cov = a.sum; r.sumsq_cov = a.sumsq; r.median_cov = med_cov[s];
    r.counted = (int64_t)a.counted; r.sum_clon = a.sum_clon; r.median_clon = med_c[s];
    r.counted_rarefied = (int64_t)a.counted_r; r.sum_clon_rarefied = a.sum_clon_r; r.median_clon_rarefied = med_r[s];
    r.mm = mm; r.present = (int32_t)a.present;
    out[(size_t)s * M + mm] = r;
}

__global__ void k_reset_acc(Acc *acc, int n_seg)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    Acc z;
    z.nonzero = z.sum = z.sumsq = z.counted = z.counted_r = 0; z.sum_clon = z.sum_clon_r = 0.0; z.present = 0; z.pad = 0;
    acc[s] = z;
}

// compare (readComparer.py:145-191 calc_mm2overlap): positions where BOTH / EITHER sample has
// cumulative coverage >= min_cov, per scaffold
__global__ void __launch_bounds__(256) k_overlap_reduce(const uint32_t *cov_a, const uint32_t *cov_b, uint32_t n_pos,
                                                        uint32_t min_cov, const int64_t *bounds, int n_seg,
                                                        unsigned long long *both, unsigned long long *either)
{
    const uint32_t TILE = 64;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = t * TILE;
    if (p0 >= n_pos) return;
    const uint32_t p1 = (uint32_t)min((uint64_t)n_pos, p0 + TILE);
    int seg = find_seg(bounds, n_seg, (uint32_t)p0);
    unsigned long long nb = 0, ne = 0;
    for (uint32_t p = (uint32_t)p0; p < p1; p++) {
        if ((int64_t)p >= bounds[seg + 1]) {
            if (nb) atomicAdd(&both[seg], nb);
            if (ne) atomicAdd(&either[seg], ne);
            nb = ne = 0;
            seg = find_seg(bounds, n_seg, p);
        }
        const bool a = cov_a[p] >= min_cov, b = cov_b[p] >= min_cov;
        nb += (a && b) ? 1 : 0;
        ne += (a || b) ? 1 : 0;
    }
    if (nb) atomicAdd(&both[seg], nb);
    if (ne) atomicAdd(&either[seg], ne);
}

__global__ void k_pack_compare(const Acc *acc_a, const Acc *acc_b, const unsigned long long *both,
                               const unsigned long long *either, const unsigned long long *n_con, const unsigned long long *n_pop,
                               const uint32_t *failed, int n_seg, int mm, int M, isx_compare_level *out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    isx_compare_level r;
    r.both = (int64_t)both[s]; r.either = (int64_t)either[s];
    r.mm = mm; r.present_a = (int32_t)acc_a[s].present; r.present_b = (int32_t)acc_b[s].present; r.pad = 0;
    r.consensus_snps = failed ? (failed[s] ? -2 : (int64_t)n_con[s]) : -1;
    r.population_snps = failed ? (failed[s] ? -2 : (int64_t)n_pop[s]) : -1;
    out[(size_t)s * M + mm] = r;
}

// dense path: a level is "present" on a scaffold when it has any coverage there
__global__ void __launch_bounds__(256) k_present_dense(const uint32_t *cov, uint32_t n_pos, const int64_t *bounds, int n_seg, Acc *acc)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pos || cov[p] == 0) return;
    Acc *a = &acc[find_seg(bounds, n_seg, p)];
    if (!a->present) a->present = 1;
}

// ---- compare, SNP-table half (readComparer.py:205-290 _calc_SNP_count_alternate) ----
// compare sorts each sample's cumulative_snv_table by mm (compare_utils.py:124-138) and keeps, at EVERY
// compared mm, the LAST row of a position (drop_duplicates keep='last', no mm filter), i.e. the row of
// its highest level.  The consensus / population verdict of a position is therefore independent of the
// compared mm; only the covered-in-both mask changes.  So: one candidate pass, then per mm a mask pass.
struct SnpCand {
    uint32_t gpos;
    uint32_t row_a, row_b;          // index into the samples' SNV tables, 0xFFFFFFFF = no row
    uint32_t flags;                 // 1 consensus_SNP, 2 population_SNP
};

__global__ void k_snv_keys(const isx_snv *snv, uint32_t n, uint64_t *keys, uint32_t *idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((uint64_t)snv[i].gpos << 16) | snv[i].mm;
    idx[i] = i;
}

// index of the highest-mm row of gpos in a (gpos, mm)-sorted key list, or -1
__device__ __forceinline__ int64_t last_row_of(const uint64_t *keys, uint32_t n, uint32_t gpos)
{
    const uint64_t hi = ((uint64_t)gpos << 16) | 0xFFFFull;
    uint32_t lo = 0, up = n;                         // first index with key > hi
    while (lo < up) {
        const uint32_t mid = (lo + up) >> 1;
        if (keys[mid] <= hi) lo = mid + 1; else up = mid;
    }
    if (lo == 0 || (uint32_t)(keys[lo - 1] >> 16) != gpos) return -1;
    return (int64_t)lo - 1;
}

// readComparer.py:306-315 is_present
__device__ __forceinline__ bool is_present(uint32_t count, uint32_t total, const uint8_t *lut, int32_t lut_n, int32_t fallback,
                                           double min_freq)
{
    int32_t min_bases = fallback;
    if (total < (uint32_t)lut_n && lut[total] != 255) min_bases = lut[total];
    return (int64_t)count >= (int64_t)min_bases && ((double)count / (double)total) >= min_freq;
}

__global__ void __launch_bounds__(256) k_snp_candidates(const uint64_t *keys_a, const uint32_t *idx_a, uint32_t n_a, const isx_snv *snv_a,
                                                        const uint64_t *keys_b, const uint32_t *idx_b, uint32_t n_b, const isx_snv *snv_b,
                                                        const uint8_t *lut, int32_t lut_n, int32_t fallback, double min_freq,
                                                        const int64_t *bounds, int n_seg, SnpCand *cand, uint32_t *cursors)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_a + n_b) return;
    const bool from_a = t < n_a;
    const uint64_t *kx = from_a ? keys_a : keys_b;
    const uint32_t nx = from_a ? n_a : n_b, i = from_a ? t : t - n_a;
    const uint32_t gpos = (uint32_t)(kx[i] >> 16);
    if (i + 1 < nx && (uint32_t)(kx[i + 1] >> 16) == gpos) return;          // not the position's last row
    const int64_t j = from_a ? last_row_of(keys_b, n_b, gpos) : last_row_of(keys_a, n_a, gpos);
    if (!from_a && j >= 0) return;                                          // shared rows are judged from A's side
    SnpCand c;
    c.gpos = gpos;
    bool con, pop;
    if (j < 0) {                            // the row exists in one sample only (the other's columns are NaN)
        const uint32_t r = (from_a ? idx_a : idx_b)[i];
        const isx_snv x = (from_a ? snv_a : snv_b)[r];
        c.row_a = from_a ? r : 0xFFFFFFFFu;
        c.row_b = from_a ? 0xFFFFFFFFu : r;
        con = x.con_base != x.ref_base;                                     // call_con_snps :296-301
        if (x.ref_base > 3) {               // '{ref_base}_2' with ref_base N: the reference raises KeyError
            atomicOr(&cursors[2 + find_seg(bounds, n_seg, gpos)], 1u);
            return;
        }
        const uint32_t total = x.cnt[0] + x.cnt[1] + x.cnt[2] + x.cnt[3];
        pop = !is_present(x.cnt[x.ref_base], total, lut, lut_n, fallback, min_freq);   // call_pop_snps :329-344
    } else {
        const uint32_t ra = idx_a[i], rb = idx_b[j];
        const isx_snv a = snv_a[ra], b = snv_b[rb];
        c.row_a = ra; c.row_b = rb;
        con = a.con_base != b.con_base;                                     // :304
        const uint32_t ta = a.cnt[0] + a.cnt[1] + a.cnt[2] + a.cnt[3], tb = b.cnt[0] + b.cnt[1] + b.cnt[2] + b.cnt[3];
        if (!con) pop = false;                                                              // :325
        else if (is_present(b.cnt[a.con_base & 3], tb, lut, lut_n, fallback, min_freq)) pop = false;   // :349-355
        else if (is_present(a.cnt[b.con_base & 3], ta, lut, lut_n, fallback, min_freq)) pop = false;   // :358-361
        else if (a.allele_count > 1 && b.allele_count > 1 && a.var_base == b.var_base) pop = false;   // :364-367
        else pop = true;
    }
    if (!con && !pop) return;                                               // "Only keep SNPs" :281
    c.flags = (con ? 1u : 0u) | (pop ? 2u : 0u);
    cand[atomicAdd(&cursors[0], 1u)] = c;
}

__global__ void __launch_bounds__(256) k_snp_apply(const SnpCand *cand, uint32_t *cursors, const uint32_t *cov_a, const uint32_t *cov_b,
                                                   uint32_t min_cov, const int64_t *bounds, int n_seg, unsigned long long *n_con,
                                                   unsigned long long *n_pop, const isx_snv *snv_a, const isx_snv *snv_b, uint32_t mm,
                                                   isx_compare_snp *rows)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cursors[0]) return;
    const SnpCand c = cand[t];
    if (cov_a[c.gpos] < min_cov || cov_b[c.gpos] < min_cov) return;         // position not in mm2overlap[mm]
    const int seg = find_seg(bounds, n_seg, c.gpos);
    if (c.flags & 1u) atomicAdd(&n_con[seg], 1ull);
    if (c.flags & 2u) atomicAdd(&n_pop[seg], 1ull);
    isx_compare_snp r;
    memset(&r, 0, sizeof(r));
    r.gpos = c.gpos; r.mm = (uint16_t)mm;
    r.consensus_snp = c.flags & 1u; r.population_snp = (c.flags >> 1) & 1u;
    if (c.row_a != 0xFFFFFFFFu) {
        const isx_snv a = snv_a[c.row_a];
        r.has_a = 1; r.con_a = a.con_base; r.ref_a = a.ref_base; r.var_a = a.var_base;
        for (int k = 0; k < 4; k++) r.cnt_a[k] = a.cnt[k];
    }
    if (c.row_b != 0xFFFFFFFFu) {
        const isx_snv b = snv_b[c.row_b];
        r.has_b = 1; r.con_b = b.con_base; r.ref_b = b.ref_base; r.var_b = b.var_base;
        for (int k = 0; k < 4; k++) r.cnt_b[k] = b.cnt[k];
    }
    rows[atomicAdd(&cursors[1], 1u)] = r;
}

// ---- genome level (genomeUtilities.py:297-365 on generate_genome_coverage_array :932-981) ----
// covm[p] = cumulative coverage, or 0xFFFFFFFF when p lies in the `mask` positions at either end of its scaffold (or the
// scaffold is shorter than twice that): masked positions sort behind every real one and are left out of the sums.
struct GAcc { unsigned long long n, sum, sumsq; };

__global__ void __launch_bounds__(256) k_genome_mask_reduce(const uint32_t *cov, uint32_t n_pos, const int64_t *sbounds, int n_scaf,
                                                            const int64_t *gbounds, int n_gen, uint32_t mask, uint32_t *covm, GAcc *acc)
{
    const uint32_t TILE = 64;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = t * TILE;
    if (p0 >= n_pos) return;
    const uint32_t p1 = (uint32_t)min((uint64_t)n_pos, p0 + TILE);
    int sc = find_seg(sbounds, n_scaf, (uint32_t)p0), g = find_seg(gbounds, n_gen, (uint32_t)p0);
    unsigned long long n = 0, sum = 0, sq = 0;
    auto flush = [&]() {
        if (n) { atomicAdd(&acc[g].n, n); atomicAdd(&acc[g].sum, sum); atomicAdd(&acc[g].sumsq, sq); }
        n = sum = sq = 0;
    };
    for (uint32_t p = (uint32_t)p0; p < p1; p++) {
        if ((int64_t)p >= sbounds[sc + 1]) sc = find_seg(sbounds, n_scaf, p);
        if ((int64_t)p >= gbounds[g + 1]) { flush(); g = find_seg(gbounds, n_gen, p); }
        const int64_t s0 = sbounds[sc], s1 = sbounds[sc + 1];
        const bool valid = mask == 0 || (s1 - s0 >= 2 * (int64_t)mask && (int64_t)p - s0 >= (int64_t)mask && s1 - (int64_t)p > (int64_t)mask);
        const unsigned long long c = cov[p];
        covm[p] = valid ? (uint32_t)c : 0xFFFFFFFFu;
        if (valid) { n++; sum += c; sq += c * c; }
    }
    flush();
}

__global__ void k_genome_rows(const uint32_t *sorted, const uint32_t *seg_off, const GAcc *acc, int n_gen, int mm, int M,
                              isx_genome_level *out)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_gen) return;
    const GAcc a = acc[g];
    const uint32_t b = seg_off[g];
    isx_genome_level r;
    r.n = (int64_t)a.n; r.sum_cov = a.sum; r.sumsq_cov = a.sumsq;
    r.median_cov = 0.0;                         // no position left: the reference takes the median of [0]
    if (a.n) {
        const uint64_t h = a.n >> 1;
        r.median_cov = (a.n & 1) ? (double)sorted[b + h] : ((double)sorted[b + h - 1] + (double)sorted[b + h]) / 2.0;
    }
    r.mm = mm; r.pad = 0;
    out[(size_t)g * M + mm] = r;
}

template <class T>
int dev_alloc(T **p, size_t n)
{
    if (*p) return ISX_OK;
    HIP_TRY(isx_raw_dev_malloc(p, std::max<size_t>(n, 1) * sizeof(T)));
    return ISX_OK;
}

}  // namespace

// ---- isx_batch_fetch_entries: the used prefixes of the window slabs + the overflow region, in
// (gpos, mm) order, gathered on the device so that only the entries themselves cross PCIe ----
namespace {
__global__ void __launch_bounds__(256) k_entry_keys(const isx_entry *entries, const uint32_t *win_nent, uint32_t slab, uint64_t ovf0,
                                                    uint32_t n_ovf, uint64_t *keys, uint32_t *idx, uint32_t *cursor, uint32_t cap)
{
    const uint64_t total = ovf0 + n_ovf;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    const uint64_t rounds = (total + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; r++) {                         // whole waves stay in the loop (ballot below)
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool ok = i < total;
        if (ok && i < ovf0) {
            const uint32_t w = (uint32_t)(i / slab);
            ok = (uint32_t)(i - (uint64_t)w * slab) < win_nent[w];
        }
        const unsigned long long bal = __ballot(ok);
        if (!bal) continue;
        uint32_t base = 0;
        const int first = __ffsll((long long)bal) - 1;
        if (lane == first) base = atomicAdd(cursor, (uint32_t)__popcll(bal));   // one atomic per wave
        base = __shfl(base, first);
        if (!ok) continue;
        const uint32_t k = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (k >= cap) continue;
        keys[k] = ((uint64_t)entries[i].gpos << 16) | entries[i].mm;
        idx[k] = (uint32_t)i;
    }
}

__global__ void k_gather_entries(const isx_entry *entries, const uint32_t *idx, uint32_t n, isx_entry *out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(&entries[idx[k]]);
    uint4 *dst = reinterpret_cast<uint4 *>(&out[k]);
    dst[0] = src[0]; dst[1] = src[1];
}

// the same gather, shrunk to what shrink_basewise keeps of a (position, mm) level: its coverage, not its four counts --
// four 4-byte columns (position | mm:8 cov:24 | clonality | rarefied clonality), 16 bytes an entry instead of 32
__global__ void k_gather_entries_soa(const isx_entry *entries, const uint32_t *idx, uint32_t n, uint32_t *gpos, uint32_t *mm_cov,
                                     float *clon, float *clon_r, uint32_t *too_deep)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(&entries[idx[k]]);
    const uint4 a = src[0], b = src[1];             // gpos | mm, flags | cnt[0] | cnt[1]  ;  cnt[2] | cnt[3] | clon | clon_rarefied
    const uint32_t cov = a.z + a.w + b.x + b.y, mm = a.y & 0xFFFFu;
    if (cov >= (1u << 24) || mm >= 256u) atomicOr(too_deep, 1u);
    gpos[k] = a.x;
    mm_cov[k] = (mm << 24) | (cov & 0xFFFFFFu);
    clon[k] = __uint_as_float(b.z);
    clon_r[k] = __uint_as_float(b.w);
}
}  // namespace

int fetch_entries_sorted(hipStream_t s, const isx_entry *entries, const uint32_t *win_nent, uint32_t slab, uint32_t n_win,
                         uint32_t n_ovf, uint64_t n_entries, isx_entry *host_out, const EntryCopier *copier, const EntrySoa *soa)
{
    const uint64_t ovf0 = (uint64_t)n_win * slab;
    if (ovf0 + n_ovf >= 0xFFFFFFFFull || n_entries >= 0xFFFFFFFFull) { isx_set_error("entry table too large to fetch in one piece"); return ISX_ERR_CAPACITY; }
    const uint32_t n = (uint32_t)n_entries;
    const bool timing = getenv("ISX_PIPE_TIMING") != nullptr;      // tuning aid (stderr only)
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now_ms();
    uint64_t *keys = nullptr;
    uint32_t *idx = nullptr, *cursor = nullptr;
    uint8_t *out = nullptr;
    void *temp = nullptr;
    int rc = ISX_OK;
    auto done = [&](int code) {
        void *ps[] = {keys, idx, cursor, out, temp};
        for (void *p : ps) if (p) isx_dev_free(p);
        return code;
    };
    const size_t out_bytes = (size_t)n * (soa ? 16 : sizeof(isx_entry));
#define FE_TRY(expr) do { if ((expr) != hipSuccess) { isx_set_error(std::string("HIP error in fetch_entries: ") + #expr); return done(ISX_ERR_HIP); } } while (0)
    // (cached blocks: a profile fetches a table of this size per batch)
    FE_TRY(isx_dev_malloc(reinterpret_cast<void **>(&keys), (size_t)n * 2 * sizeof(uint64_t)));
    FE_TRY(isx_dev_malloc(reinterpret_cast<void **>(&idx), (size_t)n * 2 * sizeof(uint32_t)));
    FE_TRY(isx_dev_malloc(reinterpret_cast<void **>(&cursor), 8));
    FE_TRY(isx_dev_malloc(reinterpret_cast<void **>(&out), out_bytes));
    FE_TRY(hipMemsetAsync(cursor, 0, 8, s));
    hipLaunchKernelGGL(k_entry_keys, dim3(2048), dim3(256), 0, s, entries, win_nent, slab, ovf0, n_ovf, keys, idx, cursor, n);
    size_t tb = 0;
    FE_TRY(rocprim::radix_sort_pairs(nullptr, tb, keys, keys + n, idx, idx + n, (size_t)n, 0, 48, s));
    FE_TRY(isx_dev_malloc(&temp, tb + 256));
    FE_TRY(rocprim::radix_sort_pairs(temp, tb, keys, keys + n, idx, idx + n, (size_t)n, 0, 48, s));
    uint32_t *col = reinterpret_cast<uint32_t *>(out);
    if (soa)
        hipLaunchKernelGGL(k_gather_entries_soa, dim3((n + 255) / 256), dim3(256), 0, s, entries, idx + n, n, col, col + n,
                           reinterpret_cast<float *>(col + 2 * (size_t)n), reinterpret_cast<float *>(col + 3 * (size_t)n), cursor + 1);
    else
        hipLaunchKernelGGL(k_gather_entries, dim3((n + 255) / 256), dim3(256), 0, s, entries, idx + n, n, reinterpret_cast<isx_entry *>(out));
    uint32_t got[2] = {0, 0};
    FE_TRY(hipMemcpyAsync(got, cursor, 8, hipMemcpyDeviceToHost, s));
    FE_TRY(isx_wait_stream(s));
    if (got[0] != n) { isx_set_error("entry table inconsistent: " + std::to_string(got[0]) + " gathered vs " + std::to_string(n)); return done(ISX_ERR_STATE); }
    if (soa && got[1]) { isx_set_error("a (position, mm) level with coverage >= 2^24 or mm >= 256: fetch the full entries (isx_pipe_fetch_entries)"); return done(ISX_ERR_CAPACITY); }
    const double t_dev = now_ms();
    void *dsts[4] = {host_out, nullptr, nullptr, nullptr};
    const int n_parts = soa ? 4 : 1;
    if (soa) { dsts[0] = soa->gpos; dsts[1] = soa->mm_cov; dsts[2] = soa->clon; dsts[3] = soa->clon_rarefied; }
    const size_t part_bytes = out_bytes / (size_t)n_parts;
    for (int k = 0; k < n_parts; k++) {
        if (copier) {
            const int crc = (*copier)(out + (size_t)k * part_bytes, dsts[k], part_bytes, s);
            if (crc != ISX_OK) return done(crc);
        } else {
            FE_TRY(hipMemcpyAsync(dsts[k], out + (size_t)k * part_bytes, part_bytes, hipMemcpyDeviceToHost, s));
        }
    }
    FE_TRY(isx_wait_stream(s));
#undef FE_TRY
    if (timing) fprintf(stderr, "[fetch_entries] %u entries: keys + sort + gather %.1f ms, copy of %.1f MB %.1f ms (from %.1f)\n", n, t_dev - t_0, out_bytes / 1e6, now_ms() - t_dev, t_0);
    return done(rc);
}

int sort_pairs_by_position(hipStream_t s, const uint2 *in, uint2 *out, size_t n, void **temp, size_t *temp_bytes)
{
    if (!n) return ISX_OK;
    static_assert(sizeof(uint2) == sizeof(uint64_t), "an entry is one 64-bit key, position in its low word");
    const uint64_t *ki = reinterpret_cast<const uint64_t *>(in);
    uint64_t *ko = reinterpret_cast<uint64_t *>(out);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, ki, ko, n, 0, 32, s));
    if (tb > *temp_bytes || !*temp) {
        if (*temp) isx_dev_free(*temp);
        *temp = nullptr; *temp_bytes = 0;
        HIP_TRY(isx_dev_malloc(temp, tb + 256));
        *temp_bytes = tb + 256;
    }
    size_t t = *temp_bytes;
    HIP_TRY(rocprim::radix_sort_keys(*temp, t, ki, ko, n, 0, 32, s));
    return ISX_OK;
}

// the position-sized arrays are kept between calls; a batch with more positions than any before it gets new ones
void SummaryBuffers::fit_positions(size_t n_pos)
{
    if (n_pos <= cap_pos) return;
    void *ps[] = {cov, cv, cr, k_u32, k_f32};
    for (void *p : ps) if (p) isx_dev_free(p);
    cov = nullptr; cv = nullptr; cr = nullptr; k_u32 = nullptr; k_f32 = nullptr;
    cap_pos = n_pos;
}

void SummaryBuffers::release()
{
    void *ps[] = {cov, cv, cr, k_u32, k_f32, seg_off, seg_be, bounds, acc, med, rows, temp};
    for (void *p : ps) if (p) isx_dev_free(p);
    *this = SummaryBuffers();
}

int run_summary(const SummaryIn &in, SummaryBuffers &B, isx_scaffold_level *host_out, float *ms)
{
    hipStream_t s = in.stream;
    const uint32_t n_pos = in.n_pos;
    const int n_seg = in.n_scaffolds, M = in.M;
    int rc;
    B.fit_positions(n_pos);
    if ((rc = dev_alloc(&B.cov, B.cap_pos)) || (rc = dev_alloc(&B.cv, B.cap_pos)) || (rc = dev_alloc(&B.cr, B.cap_pos)) ||
        (rc = dev_alloc(&B.k_u32, B.cap_pos)) || (rc = dev_alloc(&B.k_f32, B.cap_pos))) return rc;
    if (B.n_seg != n_seg) {
        void *ps[] = {B.seg_off, B.seg_be, B.bounds, B.acc, B.med, B.rows};
        for (void *p : ps) if (p) isx_dev_free(p);
        B.seg_off = nullptr; B.seg_be = nullptr; B.bounds = nullptr; B.acc = nullptr; B.med = nullptr; B.rows = nullptr;
        B.n_seg = n_seg;
    }
    if ((rc = dev_alloc(&B.seg_off, (size_t)n_seg + 1)) || (rc = dev_alloc(&B.bounds, (size_t)n_seg + 1)) ||
        (rc = dev_alloc(reinterpret_cast<Acc **>(&B.acc), (size_t)n_seg)) || (rc = dev_alloc(&B.med, (size_t)n_seg * 3)) ||
        (rc = dev_alloc(&B.rows, (size_t)n_seg * M))) return rc;
    std::vector<uint32_t> off((size_t)n_seg + 1);
    for (int i = 0; i <= n_seg; i++) off[(size_t)i] = (uint32_t)in.scaffold_bounds[i];
    HIP_TRY(hipMemcpyAsync(B.seg_off, off.data(), off.size() * 4, hipMemcpyHostToDevice, s));
    // Medians come from sorted copies.  rocPRIM's segmented sort gives one workgroup to a segment, which
    // is right for thousands of contigs and hopeless for a 5 Mbp genome, so segments above BIG_SEG are
    // sorted one by one with the device-wide radix sort (into the same output array) and handed to the
    // segmented sort as empty ranges.
    constexpr uint32_t BIG_SEG = 1u << 17;
    std::vector<uint32_t> seg_b((size_t)n_seg), seg_e((size_t)n_seg);
    std::vector<int> big;
    uint32_t longest = 1;
    for (int i = 0; i < n_seg; i++) {
        seg_b[(size_t)i] = off[(size_t)i]; seg_e[(size_t)i] = off[(size_t)i + 1];
        if (seg_e[(size_t)i] - seg_b[(size_t)i] > BIG_SEG) {
            big.push_back(i);
            longest = std::max(longest, seg_e[(size_t)i] - seg_b[(size_t)i]);
            seg_e[(size_t)i] = seg_b[(size_t)i];
        }
    }
    if ((rc = dev_alloc(&B.seg_be, (size_t)n_seg * 2))) return rc;
    HIP_TRY(hipMemcpyAsync(B.seg_be, seg_b.data(), (size_t)n_seg * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(B.seg_be + n_seg, seg_e.data(), (size_t)n_seg * 4, hipMemcpyHostToDevice, s));
    const bool any_small = (int)big.size() < n_seg;
    HIP_TRY(hipMemcpyAsync(B.bounds, in.scaffold_bounds, ((size_t)n_seg + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(in.ev[0], s));
    Acc *acc = reinterpret_cast<Acc *>(B.acc);
    const dim3 blk(256), gpos((n_pos + 255) / 256), gseg((n_seg + 255) / 256);
    if (M > 1) {
        HIP_TRY(hipMemsetAsync(B.cov, 0, (size_t)n_pos * 4, s));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(B.cv), 0x7FC00000, n_pos, s));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(B.cr), 0x7FC00000, n_pos, s));
    }
    size_t tb = 0, tb2 = 0;
    HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, tb, B.cov, B.k_u32, n_pos, (unsigned)n_seg, B.seg_be, B.seg_be + n_seg, 0, 32, s));
    HIP_TRY(rocprim::segmented_radix_sort_keys(nullptr, tb2, B.cv, B.k_f32, n_pos, (unsigned)n_seg, B.seg_be, B.seg_be + n_seg, 0, 32, s));
    tb = std::max(tb, tb2);
    if (!big.empty()) {
        HIP_TRY(rocprim::radix_sort_keys(nullptr, tb2, B.cov, B.k_u32, (size_t)longest, 0, 32, s));
        tb = std::max(tb, tb2);
        HIP_TRY(rocprim::radix_sort_keys(nullptr, tb2, B.cv, B.k_f32, (size_t)longest, 0, 32, s));
        tb = std::max(tb, tb2);
    }
    if (B.temp_bytes < tb) {
        if (B.temp) isx_dev_free(B.temp);
        B.temp = nullptr;
        HIP_TRY(isx_raw_dev_malloc(&B.temp, tb + 256));
        B.temp_bytes = tb + 256;
    }
    auto sort_u32 = [&](const uint32_t *src, uint32_t *dst) -> int {
        size_t t = B.temp_bytes;
        if (any_small) HIP_TRY(rocprim::segmented_radix_sort_keys(B.temp, t, src, dst, n_pos, (unsigned)n_seg, B.seg_be, B.seg_be + n_seg, 0, 32, s));
        for (int i : big) {
            t = B.temp_bytes;
            HIP_TRY(rocprim::radix_sort_keys(B.temp, t, src + off[(size_t)i], dst + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i]), 0, 32, s));
        }
        return ISX_OK;
    };
    auto sort_f32 = [&](const float *src, float *dst) -> int {
        size_t t = B.temp_bytes;
        if (any_small) HIP_TRY(rocprim::segmented_radix_sort_keys(B.temp, t, src, dst, n_pos, (unsigned)n_seg, B.seg_be, B.seg_be + n_seg, 0, 32, s));
        for (int i : big) {
            t = B.temp_bytes;
            HIP_TRY(rocprim::radix_sort_keys(B.temp, t, src + off[(size_t)i], dst + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i]), 0, 32, s));
        }
        return ISX_OK;
    };
    for (int mm = 0; mm < M; mm++) {
        hipLaunchKernelGGL(k_reset_acc, gseg, blk, 0, s, acc, n_seg);
        if (M == 1) {
            launch_level_dense(in, gpos, blk, s, n_pos, B.cov, B.cv, B.cr);
        } else {
            hipLaunchKernelGGL(k_level_apply, dim3(2048), blk, 0, s, in.entries, in.win_nent, in.slab, in.n_win, in.ovf0,
                               in.n_ovf, (uint32_t)mm, B.cov, B.cv, B.cr, B.bounds, n_seg, acc);
        }
        const uint32_t tiles = (n_pos + 63) / 64;
        hipLaunchKernelGGL(k_seg_reduce, dim3((tiles + 255) / 256), blk, 0, s, B.cov, B.cv, B.cr, n_pos, B.bounds, n_seg, acc,
                           M == 1 ? 1 : 0);
        if ((rc = sort_u32(B.cov, B.k_u32))) return rc;
        hipLaunchKernelGGL(k_pick_median<uint32_t>, gseg, blk, 0, s, B.k_u32, B.seg_off, n_seg, acc, 0, B.med);
        if ((rc = sort_f32(B.cv, B.k_f32))) return rc;
        hipLaunchKernelGGL(k_pick_median<float>, gseg, blk, 0, s, B.k_f32, B.seg_off, n_seg, acc, 1, B.med + n_seg);
        if ((rc = sort_f32(B.cr, B.k_f32))) return rc;
        hipLaunchKernelGGL(k_pick_median<float>, gseg, blk, 0, s, B.k_f32, B.seg_off, n_seg, acc, 2, B.med + 2 * n_seg);
        hipLaunchKernelGGL(k_pack_rows, gseg, blk, 0, s, acc, B.med, B.med + n_seg, B.med + 2 * n_seg, n_seg, mm, M, B.rows);
    }
    HIP_TRY(hipEventRecord(in.ev[1], s));
    HIP_TRY(hipMemcpyAsync(host_out, B.rows, (size_t)n_seg * M * sizeof(isx_scaffold_level), hipMemcpyDeviceToHost, s));
    HIP_TRY(isx_wait_stream(s));
    if (ms) { float v = 0.f; (void)hipEventElapsedTime(&v, in.ev[0], in.ev[1]); *ms = v; }
    return ISX_OK;
}

// Genome-level coverage roll-up: a genome = consecutive scaffolds of the batch; per (genome, mm) the number of positions left
// after cutting `mask_edges` from both ends of every scaffold, the exact sums of the cumulative coverage over them and its median.
int run_genome_summary(const SummaryIn &in, SummaryBuffers &B, int n_genomes, const int32_t *genome_first, int mask_edges,
                       isx_genome_level *host_out, float *ms)
{
    hipStream_t s = in.stream;
    const uint32_t n_pos = in.n_pos;
    const int n_scaf = in.n_scaffolds, M = in.M;
    int rc;
    B.fit_positions(n_pos);
    if ((rc = dev_alloc(&B.cov, B.cap_pos)) || (rc = dev_alloc(&B.cv, B.cap_pos)) || (rc = dev_alloc(&B.cr, B.cap_pos)) ||
        (rc = dev_alloc(&B.k_u32, B.cap_pos)) || (rc = dev_alloc(&B.k_f32, B.cap_pos))) return rc;
    // small per-call tables (this is a once-per-batch pass)
    std::vector<int64_t> gb((size_t)n_genomes + 1);
    for (int g = 0; g <= n_genomes; g++) gb[(size_t)g] = in.scaffold_bounds[genome_first[g]];
    std::vector<uint32_t> off((size_t)n_genomes + 1);
    for (int g = 0; g <= n_genomes; g++) off[(size_t)g] = (uint32_t)gb[(size_t)g];
    int64_t *d_sb = nullptr, *d_gb = nullptr;
    uint32_t *d_off = nullptr, *d_be = nullptr, *covm = reinterpret_cast<uint32_t *>(B.k_f32);     // k_f32 is free in this pass
    GAcc *d_acc = nullptr;
    Acc *d_sacc = nullptr;
    isx_genome_level *d_rows = nullptr;
    void *temp = nullptr;
    auto done = [&](int code) {
        void *ps[] = {d_sb, d_gb, d_off, d_be, d_acc, d_sacc, d_rows, temp};
        for (void *p : ps) if (p) isx_dev_free(p);
        return code;
    };
#define GS_TRY(expr) do { if ((expr) != hipSuccess) { isx_set_error(std::string("HIP error in the genome summary: ") + #expr); return done(ISX_ERR_HIP); } } while (0)
    GS_TRY(isx_raw_dev_malloc(&d_sb, ((size_t)n_scaf + 1) * sizeof(int64_t)));
    GS_TRY(isx_raw_dev_malloc(&d_gb, ((size_t)n_genomes + 1) * sizeof(int64_t)));
    GS_TRY(isx_raw_dev_malloc(&d_off, ((size_t)n_genomes + 1) * sizeof(uint32_t)));
    GS_TRY(isx_raw_dev_malloc(&d_be, (size_t)n_genomes * 2 * sizeof(uint32_t)));
    GS_TRY(isx_raw_dev_malloc(&d_acc, (size_t)n_genomes * sizeof(GAcc)));
    GS_TRY(isx_raw_dev_malloc(&d_sacc, (size_t)n_scaf * sizeof(Acc)));
    GS_TRY(isx_raw_dev_malloc(&d_rows, (size_t)n_genomes * M * sizeof(isx_genome_level)));
    GS_TRY(hipMemsetAsync(d_sacc, 0, (size_t)n_scaf * sizeof(Acc), s));
    GS_TRY(hipMemcpyAsync(d_sb, in.scaffold_bounds, ((size_t)n_scaf + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    GS_TRY(hipMemcpyAsync(d_gb, gb.data(), gb.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    GS_TRY(hipMemcpyAsync(d_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    // medians from sorted copies, big genomes one by one with the device-wide sort (see run_summary)
    constexpr uint32_t BIG_SEG = 1u << 17;
    std::vector<uint32_t> seg_b((size_t)n_genomes), seg_e((size_t)n_genomes);
    std::vector<int> big;
    uint32_t longest = 1;
    for (int g = 0; g < n_genomes; g++) {
        seg_b[(size_t)g] = off[(size_t)g]; seg_e[(size_t)g] = off[(size_t)g + 1];
        if (seg_e[(size_t)g] - seg_b[(size_t)g] > BIG_SEG) {
            big.push_back(g);
            longest = std::max(longest, seg_e[(size_t)g] - seg_b[(size_t)g]);
            seg_e[(size_t)g] = seg_b[(size_t)g];
        }
    }
    GS_TRY(hipMemcpyAsync(d_be, seg_b.data(), (size_t)n_genomes * 4, hipMemcpyHostToDevice, s));
    GS_TRY(hipMemcpyAsync(d_be + n_genomes, seg_e.data(), (size_t)n_genomes * 4, hipMemcpyHostToDevice, s));
    const bool any_small = (int)big.size() < n_genomes;
    size_t tb = 0, tb2 = 0;
    GS_TRY(rocprim::segmented_radix_sort_keys(nullptr, tb, covm, B.k_u32, n_pos, (unsigned)n_genomes, d_be, d_be + n_genomes, 0, 32, s));
    if (!big.empty()) { GS_TRY(rocprim::radix_sort_keys(nullptr, tb2, covm, B.k_u32, (size_t)longest, 0, 32, s)); tb = std::max(tb, tb2); }
    GS_TRY(isx_raw_dev_malloc(&temp, tb + 256));
    GS_TRY(hipEventRecord(in.ev[0], s));
    const dim3 blk(256), gpos((n_pos + 255) / 256), ggen((n_genomes + 255) / 256);
    if (M > 1) {
        GS_TRY(hipMemsetAsync(B.cov, 0, (size_t)n_pos * 4, s));
        GS_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(B.cv), 0x7FC00000, n_pos, s));
        GS_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(B.cr), 0x7FC00000, n_pos, s));
    }
    for (int mm = 0; mm < M; mm++) {
        if (M == 1) launch_level_dense(in, gpos, blk, s, n_pos, B.cov, B.cv, B.cr);
        else hipLaunchKernelGGL(k_level_apply, dim3(2048), blk, 0, s, in.entries, in.win_nent, in.slab, in.n_win, in.ovf0, in.n_ovf,
                                (uint32_t)mm, B.cov, B.cv, B.cr, d_sb, n_scaf, d_sacc);
        GS_TRY(hipMemsetAsync(d_acc, 0, (size_t)n_genomes * sizeof(GAcc), s));
        const uint32_t tiles = (n_pos + 63) / 64;
        hipLaunchKernelGGL(k_genome_mask_reduce, dim3((tiles + 255) / 256), blk, 0, s, B.cov, n_pos, d_sb, n_scaf, d_gb, n_genomes,
                           (uint32_t)std::max(mask_edges, 0), covm, d_acc);
        size_t t = tb + 256;
        if (any_small) GS_TRY(rocprim::segmented_radix_sort_keys(temp, t, covm, B.k_u32, n_pos, (unsigned)n_genomes, d_be, d_be + n_genomes, 0, 32, s));
        for (int g : big) {
            t = tb + 256;
            GS_TRY(rocprim::radix_sort_keys(temp, t, covm + off[(size_t)g], B.k_u32 + off[(size_t)g], (size_t)(off[(size_t)g + 1] - off[(size_t)g]), 0, 32, s));
        }
        hipLaunchKernelGGL(k_genome_rows, ggen, blk, 0, s, B.k_u32, d_off, d_acc, n_genomes, mm, M, d_rows);
    }
    GS_TRY(hipEventRecord(in.ev[1], s));
    GS_TRY(hipMemcpyAsync(host_out, d_rows, (size_t)n_genomes * M * sizeof(isx_genome_level), hipMemcpyDeviceToHost, s));
    GS_TRY(isx_wait_stream(s));
#undef GS_TRY
    if (ms) { float v = 0.f; (void)hipEventElapsedTime(&v, in.ev[0], in.ev[1]); *ms = v; }
    return done(ISX_OK);
}

void CompareBuffers::release()
{
    void *ps[] = {cov_a, cov_b, scratch_f, bounds, acc_a, acc_b, both, rows, keys, idx, cand, snp_rows, cursors, temp};
    for (void *p : ps) if (p) isx_dev_free(p);
    *this = CompareBuffers();
}

template <class T>
static int ensure(T **p, size_t have, size_t want)
{
    if (*p && have >= want) return ISX_OK;
    if (*p) isx_dev_free(*p);
    *p = nullptr;
    HIP_TRY(isx_raw_dev_malloc(p, std::max<size_t>(want, 1) * sizeof(T)));
    return ISX_OK;
}

int run_compare(const SummaryIn &a, const SummaryIn &b, uint32_t min_cov, const CompareSnpIn &snp, CompareBuffers &B,
                isx_compare_level *host_out, float *ms)
{
    hipStream_t s = a.stream;
    const uint32_t n_pos = a.n_pos;
    const int n_seg = a.n_scaffolds, M = std::max(a.M, b.M);
    const bool do_snp = snp.lut != nullptr;
    const size_t n_snv = do_snp ? std::max<size_t>(snp.n_a, snp.n_b) : 0, n_ab = do_snp ? (size_t)snp.n_a + snp.n_b : 0;
    int rc;
    if ((rc = ensure(&B.cov_a, B.cap_pos, n_pos)) || (rc = ensure(&B.cov_b, B.cap_pos, n_pos)) ||
        (rc = ensure(&B.scratch_f, B.cap_pos * 2, (size_t)n_pos * 2))) return rc;
    B.cap_pos = std::max<size_t>(B.cap_pos, n_pos);
    if ((rc = ensure(&B.bounds, B.cap_seg + 1, (size_t)n_seg + 1)) ||
        (rc = ensure(reinterpret_cast<Acc **>(&B.acc_a), B.cap_seg, (size_t)n_seg)) ||
        (rc = ensure(reinterpret_cast<Acc **>(&B.acc_b), B.cap_seg, (size_t)n_seg)) ||
        (rc = ensure(&B.both, B.cap_seg * 4, (size_t)n_seg * 4)) ||
        (rc = ensure(&B.cursors, B.cap_seg + 2, (size_t)n_seg + 2))) return rc;
    B.cap_seg = std::max<size_t>(B.cap_seg, (size_t)n_seg);
    if ((rc = ensure(&B.rows, B.cap_rows, (size_t)n_seg * M))) return rc;
    B.cap_rows = std::max(B.cap_rows, (size_t)n_seg * M);
    if (do_snp) {
        if ((rc = ensure(&B.keys, B.cap_snv * 4, n_snv * 4)) || (rc = ensure(&B.idx, B.cap_snv * 4, n_snv * 4)) ||
            (rc = ensure(reinterpret_cast<SnpCand **>(&B.cand), B.cap_snv * 2, n_snv * 2))) return rc;
        B.cap_snv = std::max(B.cap_snv, n_snv);
        if ((rc = ensure(&B.snp_rows, B.cap_snp_rows, n_ab * M))) return rc;
        B.cap_snp_rows = std::max(B.cap_snp_rows, n_ab * M);
        size_t tb = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, B.keys, B.keys, B.idx, B.idx, std::max<size_t>(n_snv, 1), 0, 48, s));
        if (B.temp_bytes < tb) {
            if (B.temp) isx_dev_free(B.temp);
            B.temp = nullptr;
            HIP_TRY(isx_raw_dev_malloc(&B.temp, tb + 256));
            B.temp_bytes = tb + 256;
        }
    }
    B.n_snp_rows = 0;
    HIP_TRY(hipMemcpyAsync(B.bounds, a.scaffold_bounds, ((size_t)n_seg + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(a.ev[0], s));
    HIP_TRY(hipMemsetAsync(B.cov_a, 0, (size_t)n_pos * 4, s));
    HIP_TRY(hipMemsetAsync(B.cov_b, 0, (size_t)n_pos * 4, s));
    HIP_TRY(hipMemsetAsync(B.cursors, 0, ((size_t)n_seg + 2) * 4, s));
    Acc *acc_a = reinterpret_cast<Acc *>(B.acc_a), *acc_b = reinterpret_cast<Acc *>(B.acc_b);
    const dim3 blk(256), gpos((n_pos + 255) / 256), gseg((n_seg + 255) / 256);
    float *f0 = B.scratch_f, *f1 = B.scratch_f + n_pos;
    uint64_t *keys_a = B.keys, *keys_b = B.keys + n_snv, *keys_in = B.keys + 2 * n_snv;
    uint32_t *idx_a = B.idx, *idx_b = B.idx + n_snv, *idx_in = B.idx + 2 * n_snv;
    SnpCand *cand = reinterpret_cast<SnpCand *>(B.cand);
    const dim3 gab((unsigned)((n_ab + 255) / 256));
    if (do_snp && n_ab) {
        auto sorted = [&](const isx_snv *snv, uint32_t n, uint64_t *keys, uint32_t *idx) -> int {
            if (!n) return ISX_OK;
            hipLaunchKernelGGL(k_snv_keys, dim3((n + 255) / 256), blk, 0, s, snv, n, keys_in, idx_in);
            size_t t = B.temp_bytes;
            HIP_TRY(rocprim::radix_sort_pairs(B.temp, t, keys_in, keys, idx_in, idx, n, 0, 48, s));
            return ISX_OK;
        };
        if ((rc = sorted(snp.snv_a, snp.n_a, keys_a, idx_a)) || (rc = sorted(snp.snv_b, snp.n_b, keys_b, idx_b))) return rc;
        hipLaunchKernelGGL(k_snp_candidates, gab, blk, 0, s, keys_a, idx_a, snp.n_a, snp.snv_a, keys_b, idx_b, snp.n_b, snp.snv_b,
                           snp.lut, snp.lut_n, snp.fallback, snp.min_freq, B.bounds, n_seg, cand, B.cursors);
    }
    auto apply = [&](const SummaryIn &in, int mm, uint32_t *cov, Acc *acc) {
        hipLaunchKernelGGL(k_reset_acc, gseg, blk, 0, s, acc, n_seg);
        if (mm >= in.M) return;                 // no such level in this sample: coverage carries over
        if (in.M == 1) {
            launch_level_dense(in, gpos, blk, s, n_pos, cov, f0, f1);
            hipLaunchKernelGGL(k_present_dense, gpos, blk, 0, s, cov, n_pos, B.bounds, n_seg, acc);
        } else {
            hipLaunchKernelGGL(k_level_apply, dim3(2048), blk, 0, s, in.entries, in.win_nent, in.slab, in.n_win, in.ovf0, in.n_ovf,
                               (uint32_t)mm, cov, f0, f1, B.bounds, n_seg, acc);
        }
    };
    unsigned long long *both = B.both, *either = B.both + n_seg, *n_con = B.both + 2 * (size_t)n_seg, *n_pop = B.both + 3 * (size_t)n_seg;
    for (int mm = 0; mm < M; mm++) {
        apply(a, mm, B.cov_a, acc_a);
        apply(b, mm, B.cov_b, acc_b);
        HIP_TRY(hipMemsetAsync(B.both, 0, (size_t)n_seg * 4 * sizeof(unsigned long long), s));
        const uint32_t tiles = (n_pos + 63) / 64;
        hipLaunchKernelGGL(k_overlap_reduce, dim3((tiles + 255) / 256), blk, 0, s, B.cov_a, B.cov_b, n_pos, min_cov, B.bounds, n_seg,
                           both, either);
        if (do_snp && n_ab)
            hipLaunchKernelGGL(k_snp_apply, gab, blk, 0, s, cand, B.cursors, B.cov_a, B.cov_b, min_cov, B.bounds, n_seg, n_con, n_pop,
                               snp.snv_a, snp.snv_b, (uint32_t)mm, B.snp_rows);
        hipLaunchKernelGGL(k_pack_compare, gseg, blk, 0, s, acc_a, acc_b, both, either, n_con, n_pop,
                           do_snp ? B.cursors + 2 : (const uint32_t *)nullptr, n_seg, mm, M, B.rows);
    }
    HIP_TRY(hipEventRecord(a.ev[1], s));
    HIP_TRY(hipMemcpyAsync(host_out, B.rows, (size_t)n_seg * M * sizeof(isx_compare_level), hipMemcpyDeviceToHost, s));
    uint32_t cur[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(cur, B.cursors, sizeof(cur), hipMemcpyDeviceToHost, s));
    HIP_TRY(isx_wait_stream(s));
    B.n_snp_rows = cur[1];
    if (ms) { float v = 0.f; (void)hipEventElapsedTime(&v, a.ev[0], a.ev[1]); *ms = v; }
    return ISX_OK;
}
