// isx_linkage.hip -- pairwise SNV linkage on the device (sparse pair-increment path).
//
// Replaces
//   update_linked_reads           /root/reference/inStrain/profile/linkage.py:254-283 (consumer side;
//                                 the producer is allele_pass in isx_pileup.hip)
//   calc_mm_SNV_linkage_network   linkage.py:14-44
//   calculate_ld                  linkage.py:46-75
//   _iterator_ld_sites            linkage.py:78-131
//   major_minor_allele            linkage.py:133-136
//   _calc_ld_single               linkage.py:138-196 (the unseeded-random *_normalized part,
//                                 :200-228, is excluded from parity like in the reference's tests)
//
// Pipeline (all on the ctx stream; sorts/scans are rocPRIM device primitives):
//   1 sort the SNP-site records emitted by k_pileup_call by position  -> site rank
//   2 (isx_pileup.hip allele_pass, fused into the pileup kernels) observations at SNP sites whose
//     base is in the site's `bases` set, in exactly sized per-site slabs; k_ao_rank maps their
//     position to the site rank
//   3 radix sort of the allele observations by read-pair id           -> read_to_snvs[mm][name]
//   4 k_pair_incr    every i<j combination inside a pair (same split) -> 64-bit key
//                    (site1, site2, mm, b1, b2), oriented by (position, arrival order)
//   5 radix sort + run-length encode of the keys                      -> mm2combo2counts
//   6 k_ld_rows      one lane per edge: ascending mm on the edge, cumulative combo counts,
//                    site counts <= mm, gates, r2 / D' in fp64 in the reference's order.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "isx_internal.h"
#include "isx_linkage.h"

#pragma clang fp contract(off)

namespace {

// key layout: site1:sb | site2:sb | mm:8 | b1:2 | b2:2 with sb = bits needed for the site ranks of
// this batch (<= 26), so the radix sorts only walk 2*sb + 12 bits
__device__ __forceinline__ uint64_t make_key(int sb, uint32_t s1, uint32_t s2, uint32_t mm, uint32_t b1, uint32_t b2)
{
    return ((uint64_t)s1 << (12 + sb)) | ((uint64_t)s2 << 12) | ((uint64_t)mm << 4) | (b1 << 2) | b2;
}

__global__ void k_site_keys(const isx_site *sites, uint32_t n, uint32_t *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = sites[i].gpos;
}

// split id of every (sorted) site: number of split bounds <= gpos, minus one
__global__ void k_site_split(const isx_site *sites, uint32_t n, const int64_t *bounds, int n_splits,
                             uint32_t *site_gpos, uint32_t *site_split)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = sites[i].gpos;
    int lo = 0, hi = n_splits;          // bounds[lo] <= g < bounds[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid] <= (int64_t)g) lo = mid; else hi = mid;
    }
    site_gpos[i] = g;
    site_split[i] = (uint32_t)lo;
}

// The pileup kernel leaves the flat position in isx_ao::site; replace it by the rank of the site
// in the position-sorted site table and emit the grouping key (read-pair id).
__global__ void __launch_bounds__(256) k_ao_rank(isx_ao *ao, uint32_t n, const uint32_t *site_gpos, uint32_t n_sites,
                                                 uint32_t *ao_key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = ao[i].site;
    uint32_t lo = 0, hi = n_sites;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (site_gpos[mid] < g) lo = mid + 1; else hi = mid;
    }
    ao[i].site = lo;
    ao_key[i] = ao[i].pair;
}

// calc_mm_SNV_linkage_network (linkage.py:26-42): itertools.combinations(snvs, 2) per (mm, read).
// EMIT = false counts the combinations of element i with the later elements of its pair group,
// EMIT = true writes their keys at off[i].
// SELF_ONLY keeps only the same-site combinations (both mates visible at one column): the dense
// MFMA path takes every cross-site count from X^T X and needs just these from the pair lists.
template <bool EMIT, bool SELF_ONLY>
__global__ void __launch_bounds__(256) k_pair_incr(const isx_ao *ao, uint32_t n, const uint32_t *site_split,
                                                   uint32_t *cnt, const uint32_t *off, uint64_t *keys, int sb)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const isx_ao a = ao[i];
    const uint32_t sa = site_split[a.site];
    uint32_t c = 0;
    uint32_t o = EMIT ? off[i] : 0;
    for (uint32_t j = i + 1; j < n; j++) {
        const isx_ao b = ao[j];
        if (b.pair != a.pair) break;
        if (site_split[b.site] != sa) continue;      // read_to_snvs is per profile_split call
        if (SELF_ONLY && b.site != a.site) continue;
        if (EMIT) {
            // list order = (column, arrival order inside the column)
            const bool a_first = (a.site < b.site) || (a.site == b.site && a.obs_idx < b.obs_idx);
            keys[o + c] = a_first ? make_key(sb, a.site, b.site, a.mm, a.base, b.base)
                                  : make_key(sb, b.site, a.site, a.mm, b.base, a.base);
        }
        c++;
    }
    if (!EMIT) cnt[i] = c;
}

// cumulative counts over levels <= mm at a SNP site (mm_counts_to_counts on snv2mm2counts[p])
struct SiteView {
    const isx_site *sites;
    const isx_slev *slev;       // mm path: per-level counts of the sites
    const isx_snv *snv;         // dense path: a site's entry_off is the index of its SNV row, whose counts are the site's
    int dense;
};

__device__ __forceinline__ void site_cum(const SiteView &v, uint32_t s, uint32_t mm, uint32_t *out, bool &has_mm)
{
    const isx_site st = v.sites[s];
    if (v.dense) {
        const isx_snv r = v.snv[st.entry_off];
        out[0] = r.cnt[0]; out[1] = r.cnt[1]; out[2] = r.cnt[2]; out[3] = r.cnt[3];
        has_mm = (mm == 0);
        return;
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    has_mm = false;
    for (uint32_t k = 0; k < st.n_levels; k++) {
        const isx_slev e = v.slev[st.entry_off + k];
        if (e.mm == mm) has_mm = true;
        if (e.mm <= mm) { out[0] += e.cnt[0]; out[1] += e.cnt[1]; out[2] += e.cnt[2]; out[3] += e.cnt[3]; }
    }
}

// linkage.py:133-136: sorted(d, key=d.get, reverse=True) -- stable, ties resolve A<C<T<G
__device__ __forceinline__ void major_minor(const uint32_t *c, int &maj, int &mnr)
{
    int order[4] = {0, 1, 2, 3};
#pragma unroll
    for (int i = 1; i < 4; i++) {
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && c[order[j]] < c[v]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    maj = order[0]; mnr = order[1];
}

// The keys of one edge (site1, site2) from its first key `u` on: ascending mm, then combo; key >> 12 identifies the edge, bits 4..11
// are the mm level, bits 0..3 the combo (b1, b2).  _iterator_ld_sites / _calc_ld_single (linkage.py:93-196) per mm level that both
// sites hold.  EMIT = false counts the rows, EMIT = true writes them from out[0] on.  Returns the number of rows.
template <bool EMIT>
__device__ __forceinline__ uint32_t ld_edge_rows(const uint64_t *ukeys, const uint32_t *ucnt, uint32_t u, uint32_t n_u, uint32_t s1, uint32_t s2,
                                                 const SiteView &v, int min_snp, isx_ld *out, const Philox &ph)
{
    const uint64_t edge = ukeys[u] >> 12;
    uint32_t combo[16];
#pragma unroll
    for (int i = 0; i < 16; i++) combo[i] = 0;
    uint32_t rows = 0;
    uint32_t i = u;
    while (i < n_u && (ukeys[i] >> 12) == edge) {
        const uint32_t mm = (uint32_t)((ukeys[i] >> 4) & 0xFFu);
        while (i < n_u && (ukeys[i] >> 12) == edge && (uint32_t)((ukeys[i] >> 4) & 0xFFu) == mm) {
            const uint32_t cb = (uint32_t)(ukeys[i] & 0xFu);
            const uint32_t add = ucnt[i];
            // runtime-indexed register arrays spill; select with a static loop instead
#pragma unroll
            for (int q = 0; q < 16; q++) combo[q] += (q == (int)cb) ? add : 0u;
            i++;
        }
        // _iterator_ld_sites (linkage.py:93-131) for this mm
        uint32_t cA[4], cB[4];
        bool h1, h2;
        site_cum(v, s1, mm, cA, h1);
        site_cum(v, s2, mm, cB, h2);
        if (!h1 || !h2) continue;                               // mm not in updateMMs
        const uint64_t ssum = (uint64_t)cA[0] + cA[1] + cA[2] + cA[3] + cB[0] + cB[1] + cB[2] + cB[3];
        if (ssum < (uint64_t)min_snp) continue;
        int A, a, B, b;
        major_minor(cA, A, a);
        major_minor(cB, B, b);
        uint32_t nA = 0, na = 0, nB = 0, nb = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            nA += (q == A) ? cA[q] : 0u; na += (q == a) ? cA[q] : 0u;
            nB += (q == B) ? cB[q] : 0u; nb += (q == b) ? cB[q] : 0u;
        }
        if (nA == 0 || na == 0 || nB == 0 || nb == 0) continue;
        uint32_t AB = 0, Ab = 0, aB = 0, ab = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int q1 = q >> 2, q2 = q & 3;
            AB += (q1 == A && q2 == B) ? combo[q] : 0u;
            Ab += (q1 == A && q2 == b) ? combo[q] : 0u;
            aB += (q1 == a && q2 == B) ? combo[q] : 0u;
            ab += (q1 == a && q2 == b) ? combo[q] : 0u;
        }
        const uint32_t total = AB + Ab + aB + ab;
        if (!((int64_t)total > (int64_t)min_snp)) continue;     // _calc_ld_single: strict
        if (EMIT) {
            const double t = (double)total;
            const double fAB = (double)AB / t, fAb = (double)Ab / t, faB = (double)aB / t, fab = (double)ab / t;
            const double fA = fAB + fAb, fa = fab + faB, fB = fAB + faB, fb = fab + fAb;
            const double linkD = fAB - fA * fB;
            double r2 = __builtin_nan("");
            if (!(fa == 0 || fA == 0 || fB == 0 || fb == 0)) {
                double den = fA * fa;
                den = den * fB;
                den = den * fb;
                r2 = linkD * linkD / den;
            }
            const double linkd = fab - fa * fb;
            double dp = __builtin_nan("");
            if (linkd < 0) {
                const double d1 = (-fA) * fB, d2 = (-fa) * fb;
                dp = linkd / (d1 > d2 ? d1 : d2);
            } else if (linkD > 0) {
                const double d1 = fA * fb, d2 = fa * fB;
                dp = linkd / (d2 < d1 ? d2 : d1);
            }
            isx_ld r;
            r.gpos_a = v.sites[s1].gpos; r.gpos_b = v.sites[s2].gpos;
            r.mm = (uint16_t)mm;
            r.allele_A = (uint8_t)A; r.allele_a = (uint8_t)a; r.allele_B = (uint8_t)B; r.allele_b = (uint8_t)b;
            r.pad = 0;
            r.total = total; r.countAB = AB; r.countAb = Ab; r.countaB = aB; r.countab = ab;
            r.pad2 = 0;
            r.r2 = r2; r.d_prime = dp;
            // rarefied variants (linkage.py:200-228): min_snp draws from [fAB, fAb, faB, fab]
            r.r2_normalized = __builtin_nan(""); r.d_prime_normalized = __builtin_nan("");
            if (min_snp > 0) {
                const double p4[4] = {fAB, fAb, faB, fab};
                uint32_t rc[4];
                rarefy4(ph, r.gpos_a, r.gpos_b, 0x4C440000u /* 'LD' */ | mm, p4, min_snp, rc);
                const double n = (double)min_snp;
                const double gAB = (double)rc[0] / n, gAb = (double)rc[1] / n, gaB = (double)rc[2] / n, gab = (double)rc[3] / n;
                const double gA = gAB + gAb, ga = gab + gaB, gB = gAB + gaB, gb = gab + gAb;
                const double ln = gab - ga * gb;
                if (!(ga == 0 || gA == 0 || gB == 0 || gb == 0)) {
                    double den = gA * ga;
                    den = den * gB;
                    den = den * gb;
                    r.r2_normalized = ln * ln / den;
                }
                if (ln < 0) {
                    const double d1 = (-gA) * gB, d2 = (-ga) * gb;
                    r.d_prime_normalized = ln / (d1 > d2 ? d1 : d2);
                } else if (ln > 0) {
                    const double d1 = gA * gb, d2 = ga * gB;
                    r.d_prime_normalized = ln / (d2 < d1 ? d2 : d1);
                }
            }
            out[rows] = r;
        }
        rows++;
    }
    return rows;
}

// One lane per unique key; the lane holding the first key of an edge (site1, site2) walks the
// edge's keys (ascending mm, then combo).  EMIT=false: count rows per edge.  EMIT=true: write.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_ld_rows(const uint64_t *ukeys, const uint32_t *ucnt, uint32_t n_u,
                                                 SiteView v, int min_snp, uint32_t *rows_per, const uint32_t *row_off,
                                                 isx_ld *out, uint32_t *n_edges, Philox ph, int sb)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_u) return;
    const uint64_t k0 = ukeys[u];
    const uint64_t edge = k0 >> 12;
    const bool head = (u == 0) || ((ukeys[u - 1] >> 12) != edge);
    if (!head) { if (!EMIT) rows_per[u] = 0; return; }
    if (!EMIT) atomicAdd(n_edges, 1u);
    const uint32_t s1 = (uint32_t)(k0 >> (12 + sb)), s2 = (uint32_t)((k0 >> 12) & ((1u << sb) - 1u));
    const uint32_t rows = ld_edge_rows<EMIT>(ukeys, ucnt, u, n_u, s1, s2, v, min_snp, EMIT ? out + row_off[u] : nullptr, ph);
    if (!EMIT) rows_per[u] = rows;
}


// ------------------------------------------------------------------------------------------
// Bucket chain (round 6): steps 3-6 without a sort library, nine launches and one host sync a batch whatever its size.
//   k_link_prep     split id + position of every site, the per-site counters and the chain's state words cleared
//   k_ao_chain      k_ao_rank + every allele observation pushed onto the chain of its read pair (an atomic exchange on a table of
//                   (epoch, index) words: entries of an earlier batch read as "empty", so the table is never cleared)
//   k_pair_walk<0>  every observation walks the OLDER entries of its chain: each unordered combination inside a pair is met exactly
//                   once (itertools.combinations, linkage.py:30); counted at the site that comes first in the pair's list
//   k_scan_u32      exclusive scan of the per-site counts -> a bucket per first site, the list of sites that have one
//   k_pair_walk<1>  the same walk writes (site2, mm, b1, b2) into the first site's bucket
//   k_site_edges    one wave per first site: its bucket aggregated in an LDS hash table (mm2combo2counts of every edge of the
//                   site), the unique keys sorted by (site2, mm, combo) and written back over the bucket; its edges on the edge list
//   k_edge_rows<0>  one lane per edge: its LD rows counted
//   k_scan_u32      rows per site -> first row of every site
//   k_edge_rows<1>  one lane per edge: the rows written behind those of the site's earlier edges -- (site1, site2, mm) order, as the sorted chain
// What the host learns (increments, unique keys, edges, rows, three overflow flags) comes with the first 2048 rows in one read-back at the
// end; a table that was too small is grown and the chain continues from the kernel that needed it; a site with more unique keys than the
// LDS table holds sends the batch through the sorted chain (sparse_path) instead.
// ------------------------------------------------------------------------------------------
constexpr uint32_t LK_NIL = 0xFFFFFFFFu;
constexpr int LK_SLOTS = 1024;          // LDS hash slots of a site's wave
constexpr int LK_MAXU = 768;            // unique keys of one first site beyond which the batch takes the sorted chain
constexpr uint64_t LK_EMPTY = ~0ull;
constexpr uint32_t LK_STAGE = 256;
constexpr uint32_t LK_ROUND = 16;          // sites a wave of k_site_edges takes at a time
enum { LS_NINC = 0, LS_NU = 1, LS_NEDGES = 2, LS_NLD = 3, LS_FLAGS = 4, LS_NLIST1 = 5 /* sites with increments */, LS_NLIST2 = 6 /* sites with rows */, LS_WORDS = 16 };
constexpr uint32_t LKF_KEYS = 1u, LKF_BUCKET = 2u, LKF_LD = 4u;

__global__ void __launch_bounds__(256) k_link_prep(const isx_site *sites, uint32_t n, const int64_t *bounds, int n_splits,
                                                   uint32_t *site_gpos, uint32_t *site_split, uint32_t *site_cnt, uint32_t *site_rows, uint32_t *state)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LS_WORDS) state[i] = 0;
    if (i >= n) return;
    const uint32_t g = sites[i].gpos;
    int lo = 0, hi = n_splits;          // bounds[lo] <= g < bounds[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds[mid] <= (int64_t)g) lo = mid; else hi = mid;
    }
    site_gpos[i] = g;
    site_split[i] = (uint32_t)lo;
    site_cnt[i] = 0;
    site_rows[i] = 0;
}

__global__ void __launch_bounds__(256) k_ao_chain(isx_ao *ao, uint32_t n, const uint32_t *site_gpos, uint32_t n_sites,
                                                  unsigned long long *head, int hshift, uint32_t direct, uint32_t epoch, uint32_t *next)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = ao[i].site;
    uint32_t lo = 0, hi = n_sites;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (site_gpos[mid] < g) lo = mid + 1; else hi = mid;
    }
    ao[i].site = lo;
    // the slot of the pair: its id when the table has one per pair (`direct` = its size: the observations of a site come from reads
    // with neighbouring ids, so do their slots; no two pairs share a chain), a multiplicative hash otherwise
    const uint32_t pr = ao[i].pair;
    const uint32_t h = pr < direct ? pr : (pr * 0x9E3779B1u) >> hshift;
    const unsigned long long old = atomicExch(&head[h], ((unsigned long long)epoch << 32) | i);
    next[i] = (uint32_t)(old >> 32) == epoch ? (uint32_t)old : LK_NIL;
}

template <bool EMIT>
__global__ void __launch_bounds__(256) k_pair_walk(const isx_ao *ao, uint32_t n, const uint32_t *next, const uint32_t *site_split,
                                                   uint32_t *site_cnt, const uint32_t *site_off, uint64_t *keys, uint64_t cap_keys,
                                                   uint32_t *state)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, l = threadIdx.x & 63;
    if (EMIT && (uint64_t)state[LS_NINC] > cap_keys) {       // the buckets do not fit: the host grows them and comes back
        if (i == 0) atomicOr(&state[LS_FLAGS], LKF_KEYS);
        return;
    }
    isx_ao a{};
    uint32_t sa = 0, j = LK_NIL;
    if (i < n) { a = ao[i]; sa = site_split[a.site]; j = next[i]; }
    while (__ballot(j != LK_NIL)) {                          // (wave-uniform: the lanes' slots are taken together below)
        bool hit = false;
        uint32_t s1 = 0;
        uint64_t key = 0;
        if (j != LK_NIL) {
            const isx_ao b = ao[j];
            const uint32_t nj = next[j];
            if (b.pair == a.pair && site_split[b.site] == sa) {      // (another pair on the same hash slot; read_to_snvs is per profile_split call)
                hit = true;
                // list order = (column, arrival order inside the column)
                const bool a_first = (a.site < b.site) || (a.site == b.site && a.obs_idx < b.obs_idx);
                s1 = a_first ? a.site : b.site;
                if (EMIT) key = a_first ? make_key(0, 0, b.site, a.mm, a.base, b.base) : make_key(0, 0, a.site, a.mm, b.base, a.base);
            }
            j = nj;
        }
        // one atomic per first site and wave step: the observations of a site sit side by side, so most lanes of a wave ask for the same counter
        unsigned long long m = __ballot(hit);
        uint32_t slot = 0;
        while (m) {
            const int leader = __builtin_ctzll(m);
            const uint32_t ls1 = __shfl(s1, leader);
            const bool mine = hit && s1 == ls1;
            const unsigned long long same = __ballot(mine);
            uint32_t base = 0;
            if ((int)l == leader) base = atomicAdd(&site_cnt[ls1], (uint32_t)__popcll(same));
            if (EMIT) {
                base = __shfl(base, leader);
                if (mine) slot = base + (uint32_t)__popcll(same & ((1ull << l) - 1ull));
            }
            m &= ~same;
        }
        if (EMIT && hit) keys[site_off[s1] + slot] = key;
    }
}

// exclusive scan of in[0, n) into out[0, n] (out[n] = the total, also stored to *total) by ONE workgroup; `zero` (optional) is cleared
// along the way -- the cursors the next kernel fills the buckets with; `list` receives the indices with in[i] != 0, ascending, *n_list
// their number (the next kernel's waves walk that list: a metagenome batch has increments at a tenth of its sites)
__global__ void __launch_bounds__(1024) k_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, uint32_t *zero,
                                                   uint32_t *list, uint32_t *n_list)
{
    // a thread owns four consecutive elements of a 4096-element tile (one 16-byte load, 16-byte stores: whole lines per wave instruction --
    // sixteen elements a thread made every store touch 64 lines); the next tile's load is in flight while this one is scanned
    __shared__ uint32_t wsum[16], wnz[16];
    __shared__ uint32_t carry_s, carry_nz;
    __shared__ unsigned long long total_s;      // (the offsets are 32-bit: a total that is not is reported as 0xFFFFFFFF)
    const uint32_t t = threadIdx.x, l = t & 63, w = t >> 6;
    if (t == 0) { carry_s = 0; carry_nz = 0; total_s = 0; }
    __syncthreads();
    auto load4 = [&](uint32_t i0) -> uint4 {
        if (i0 + 4 <= n) return *reinterpret_cast<const uint4 *>(in + i0);       // (hipMalloc'ed arrays: 16-byte aligned)
        uint4 x;
        x.x = i0 < n ? in[i0] : 0u; x.y = i0 + 1 < n ? in[i0 + 1] : 0u; x.z = i0 + 2 < n ? in[i0 + 2] : 0u; x.w = 0u;
        return x;
    };
    uint4 nxt = load4(4 * t);
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t i0 = base + 4 * t;
        const uint4 x = nxt;
        if (base + 4096 < n) nxt = load4(i0 + 4096);
        const uint32_t v[4] = {x.x, x.y, x.z, x.w};
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        const uint32_t mine_nz = (v[0] != 0) + (v[1] != 0) + (v[2] != 0) + (v[3] != 0);
        uint32_t inc = mine, inz = mine_nz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d), oz = __shfl_up(inz, d);
            if ((int)l >= d) { inc += o; inz += oz; }
        }
        if (l == 63) { wsum[w] = inc; wnz[w] = inz; }
        __syncthreads();
        uint32_t before = carry_s, before_nz = carry_nz;
        for (uint32_t q = 0; q < w; q++) { before += wsum[q]; before_nz += wnz[q]; }
        uint32_t run = before + inc - mine, rnz = before_nz + inz - mine_nz;
        uint4 o;
        o.x = run; o.y = run + v[0]; o.z = o.y + v[1]; o.w = o.z + v[2];
        run = o.w + v[3];
        if (i0 + 4 <= n) {
            *reinterpret_cast<uint4 *>(out + i0) = o;
            if (zero) *reinterpret_cast<uint4 *>(zero + i0) = make_uint4(0, 0, 0, 0);
        } else {
            if (i0 < n) { out[i0] = o.x; if (zero) zero[i0] = 0; }
            if (i0 + 1 < n) { out[i0 + 1] = o.y; if (zero) zero[i0 + 1] = 0; }
            if (i0 + 2 < n) { out[i0 + 2] = o.z; if (zero) zero[i0 + 2] = 0; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) if (v[q] != 0) list[rnz++] = i0 + q;
        __syncthreads();
        if (t == 1023) { total_s += (unsigned long long)(run - carry_s); carry_s = run; carry_nz = rnz; }
        __syncthreads();
    }
    if (t == 0) { const uint32_t tot = total_s > 0xFFFFFFFEull ? 0xFFFFFFFFu : (uint32_t)total_s; out[n] = tot; *total = tot; *n_list = carry_nz; }
}

__device__ __forceinline__ uint32_t lk_hash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 40); }

// bitonic sort of one 64-bit key a lane inside groups of G lanes (G a power of two <= 64), ascending, by shuffles: no LDS, no barrier
template <int G>
__device__ __forceinline__ uint64_t lk_sort_in_groups(uint64_t k, uint32_t l)
{
#pragma unroll
    for (int kk = 2; kk <= G; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const uint64_t o = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), j) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)k, j);
            const bool up = ((l & (uint32_t)(G - 1)) & (uint32_t)kk) == 0, lower = (l & (uint32_t)j) == 0;
            k = (lower == up) ? (k < o ? k : o) : (k > o ? k : o);
        }
    }
    return k;
}

// edges of a wave's sites on their way to the list: one atomic per LK_STAGE edges (a counter every site's wave adds to takes ~9 ns a
// time: 1 ms for the 10^5 sites of a C3 batch)
__device__ __forceinline__ void lk_flush(uint2 *stage, uint32_t &n_stage, uint2 *edge_list, uint32_t *state, uint32_t l)
{
    uint32_t eb = 0;
    if (l == 0) eb = atomicAdd(&state[LS_NEDGES], n_stage);
    eb = __shfl(eb, 0);
    __syncthreads();
    for (uint32_t q = l; q < n_stage; q += 64) edge_list[eb + q] = stage[q];
    __syncthreads();
    n_stage = 0;
}

// Buckets of at most G increments, 64 / G of them side by side (`todo`: the lanes of the round that hold such a site): a lane holds one
// increment, a group's keys are sorted by shuffles, equal neighbours are one unique key with their number as its count -- registers only.
template <int G>
__device__ __forceinline__ void lk_small_buckets(unsigned long long todo, uint32_t my_s1, uint32_t my_off, uint32_t my_nb, uint32_t l,
                                                 uint64_t *keys, uint32_t *ucnt, uint32_t *rows_per, uint32_t *site_nu, uint2 *stage,
                                                 uint32_t &n_stage, uint32_t &nu_sum, uint2 *edge_list, uint32_t *state)
{
    constexpr int NG = 64 / G;
    const uint32_t g = l / G, t = l % G;
    while (todo) {
        int src = -1;
#pragma unroll
        for (int q = 0; q < NG; q++) {                          // the q-th group takes the q-th site that is left
            const int sq = todo ? __builtin_ctzll(todo) : -1;
            if (todo) todo &= todo - 1;
            if ((int)g == q) src = sq;
        }
        const bool has = src >= 0;
        const int sl = has ? src : 0;
        const uint32_t s1 = __shfl(my_s1, sl), off = __shfl(my_off, sl), n_b = has ? __shfl(my_nb, sl) : 0u;
        uint64_t k = t < n_b ? keys[off + t] : LK_EMPTY;
        k = lk_sort_in_groups<G>(k, l);
        const bool valid = k != LK_EMPTY;
        const uint64_t prev = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(k >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)k, 1);
        const bool uhead = valid && (t == 0 || k != prev);                              // first of its key
        const bool ehead = uhead && (t == 0 || (k >> 12) != (prev >> 12));              // first key of its edge (site1, site2)
        const unsigned long long ub = __ballot(uhead), eb = __ballot(ehead);
        const unsigned long long gb = G == 64 ? ub : (ub >> (g * G)) & ((1ull << (G & 63)) - 1ull);
        const uint32_t rank = (uint32_t)__popcll(gb & ((1ull << t) - 1ull));
        if (uhead) {
            const unsigned long long above = t == G - 1 ? 0ull : gb >> (t + 1);
            const uint32_t next = above ? t + 1 + (uint32_t)__builtin_ctzll(above) : n_b;
            keys[off + rank] = k;
            ucnt[off + rank] = next - t;
            rows_per[off + rank] = 0;
        }
        if (has && t == 0) site_nu[s1] = (uint32_t)__popcll(gb);
        if (n_stage + (uint32_t)__popcll(eb) > LK_STAGE) lk_flush(stage, n_stage, edge_list, state, l);
        if (ehead) stage[n_stage + (uint32_t)__popcll(eb & ((1ull << l) - 1ull))] = make_uint2(s1, rank);
        n_stage += (uint32_t)__popcll(eb);
        nu_sum += (uint32_t)__popcll(ub);
    }
}

// one wave per LK_ROUND first sites with increments (the list of k_scan_u32), taken in turns by a fixed grid
__global__ void __launch_bounds__(64) k_site_edges(uint64_t *keys, uint32_t *ucnt, uint32_t *rows_per, const uint32_t *site_off,
                                                   uint32_t *site_nu, uint2 *edge_list, uint32_t *state, uint32_t max_u, const uint32_t *site_list)
{
    __shared__ unsigned long long tk[LK_SLOTS];
    __shared__ uint32_t tc[LK_SLOTS];
    __shared__ uint2 stage[LK_STAGE];
    __shared__ unsigned long long cmp[64];
    const uint32_t l = threadIdx.x;
    if (state[LS_FLAGS] & LKF_KEYS) return;
    const uint32_t n_list = state[LS_NLIST1];
    uint32_t n_stage = 0, nu_sum = 0;
    // a wave takes LK_ROUND sites of the list at a time: their buckets' bounds are loaded side by side (a site at a time the wave waited
    // for three dependent loads per site).  A site with a single increment -- most sites of a shallow metagenome batch -- is done by its lane
    // on the spot; buckets of up to 16 increments go four at a time, up to 64 one at a time, through the register path above; the deep ones
    // through an LDS hash table, one after the other
    for (uint32_t lb = blockIdx.x * LK_ROUND; lb < n_list; lb += gridDim.x * LK_ROUND) {
    uint32_t my_s1 = 0, my_off = 0, my_nb = 0;
    if (l < LK_ROUND && lb + l < n_list) { my_s1 = site_list[lb + l]; my_off = site_off[my_s1]; my_nb = site_off[my_s1 + 1] - my_off; }
    {
        const bool single = my_nb == 1;                         // its own edge, counted once
        const unsigned long long sb = __ballot(single);
        if (sb) {
            if (n_stage + (uint32_t)__popcll(sb) > LK_STAGE) lk_flush(stage, n_stage, edge_list, state, l);
            if (single) {
                ucnt[my_off] = 1; rows_per[my_off] = 0; site_nu[my_s1] = 1;
                stage[n_stage + (uint32_t)__popcll(sb & ((1ull << l) - 1ull))] = make_uint2(my_s1, 0);
            }
            n_stage += (uint32_t)__popcll(sb); nu_sum += (uint32_t)__popcll(sb);
        }
    }
    const uint32_t reg_max = max_u >= 64 ? 64u : 1u;            // (ISX_LINK_MAXU below 64, a test's setting: everything through the table, which knows the limit)
    lk_small_buckets<16>(__ballot(my_nb >= 2 && my_nb <= min(16u, reg_max)), my_s1, my_off, my_nb, l, keys, ucnt, rows_per, site_nu, stage, n_stage, nu_sum, edge_list, state);
    lk_small_buckets<64>(__ballot(my_nb > 16 && my_nb <= reg_max), my_s1, my_off, my_nb, l, keys, ucnt, rows_per, site_nu, stage, n_stage, nu_sum, edge_list, state);
    for (unsigned long long rest = __ballot(my_nb > reg_max); rest; rest &= rest - 1) {
        const int src = __builtin_ctzll(rest);
        const uint32_t s1 = __shfl(my_s1, src), off = __shfl(my_off, src), n_b = __shfl(my_nb, src);
        // a table of at least twice the bucket's increments (more unique keys than increments there are not), 64 .. LK_SLOTS slots
        uint32_t slots = 64;
        while (slots < LK_SLOTS && slots < 2 * n_b) slots <<= 1;
        const uint32_t lim = min(max_u, slots - slots / 4);
        __syncthreads();                                        // (the table of the wave's previous site has been read)
        for (uint32_t q = l; q < slots; q += 64) { tk[q] = LK_EMPTY; tc[q] = 0; }
        __syncthreads();
        // mm2combo2counts of every edge of this site: key -> count
        uint32_t nu = 0;
        bool full = false;
        for (uint32_t c = 0; c < n_b && !full; c += 64) {
            const bool have = c + l < n_b;
            const uint64_t k = have ? keys[off + c + l] : 0;
            bool fresh = false;
            if (have) {
                uint32_t slot = lk_hash(k) & (slots - 1);
                for (;;) {
                    const unsigned long long old = atomicCAS(&tk[slot], LK_EMPTY, (unsigned long long)k);
                    if (old == LK_EMPTY || old == k) { fresh = old == LK_EMPTY; atomicAdd(&tc[slot], 1u); break; }
                    slot = (slot + 1) & (slots - 1);
                }
            }
            nu += (uint32_t)__popcll(__ballot(fresh));
            full = nu > lim;                                    // (wave-uniform; a 1024-slot table takes 64 more than `lim` before it is full,
        }                                                       //  a smaller one never gets there: it has two slots per increment)
        if (full) {                                             // too many different keys for the table: the sorted chain takes the batch
            if (l == 0) atomicOr(&state[LS_FLAGS], LKF_BUCKET);
            return;
        }
        __syncthreads();
        if (nu <= 64 && n_b < (1u << 26)) {
            // the few unique keys of a deep bucket (a C3 site: ~240 increments, ~40 keys): gathered from the table one to a lane as
            // key << 26 | count, sorted in registers -- no barrier per sorting step, no in-place compaction
            uint32_t cb = 0;
            for (uint32_t r = 0; r < slots; r += 64) {
                const unsigned long long k = tk[r + l];
                const bool valid = k != LK_EMPTY;
                const unsigned long long bal = __ballot(valid);
                if (valid) cmp[cb + (uint32_t)__popcll(bal & ((1ull << l) - 1ull))] = (k << 26) | tc[r + l];
                cb += (uint32_t)__popcll(bal);
            }
            __syncthreads();
            uint64_t v = l < nu ? cmp[l] : LK_EMPTY;
            v = lk_sort_in_groups<64>(v, l);
            const bool valid = l < nu;                              // (the padding sorts behind every key)
            const uint64_t k = v >> 26;
            const uint64_t pk = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(k >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)k, 1);
            const bool head = valid && (l == 0 || (pk >> 12) != (k >> 12));
            if (valid) { keys[off + l] = k; ucnt[off + l] = (uint32_t)(v & ((1u << 26) - 1u)); rows_per[off + l] = 0; }
            const unsigned long long hb = __ballot(head);
            if (n_stage + (uint32_t)__popcll(hb) > LK_STAGE) lk_flush(stage, n_stage, edge_list, state, l);
            if (head) stage[n_stage + (uint32_t)__popcll(hb & ((1ull << l) - 1ull))] = make_uint2(s1, l);
            n_stage += (uint32_t)__popcll(hb);
            if (l == 0) site_nu[s1] = nu;
            nu_sum += nu;
            continue;
        }
        // compact in place (the write cursor never passes the slots being read), pad to a power of two, bitonic sort by key
        uint32_t base = 0;
        for (uint32_t r = 0; r < slots; r += 64) {
            const unsigned long long k = tk[r + l];
            const uint32_t c = tc[r + l];
            const bool valid = k != LK_EMPTY;
            const unsigned long long bal = __ballot(valid);
            const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << l) - 1ull));
            __syncthreads();
            if (valid) { tk[base + rank] = k; tc[base + rank] = c; }
            base += (uint32_t)__popcll(bal);
            __syncthreads();
        }
        uint32_t n_pad = 2;
        while (n_pad < nu) n_pad <<= 1;
        for (uint32_t q = nu + l; q < n_pad; q += 64) { tk[q] = LK_EMPTY; tc[q] = 0; }
        __syncthreads();
        for (uint32_t kk = 2; kk <= n_pad; kk <<= 1) {
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t t = l; t < n_pad / 2; t += 64) {
                    const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                    const bool up = (i & kk) == 0;
                    const unsigned long long x = tk[i], y = tk[p];
                    if ((x > y) == up) {
                        const uint32_t cx = tc[i], cy = tc[p];
                        tk[i] = y; tk[p] = x; tc[i] = cy; tc[p] = cx;
                    }
                }
                __syncthreads();
            }
        }
        // the unique keys back over the bucket; the first key of every edge (site1, site2) goes on the edge list: the LD kernels take a lane per edge
        for (uint32_t c = 0; c < nu; c += 64) {
            const uint32_t u = c + l;
            bool head = false;
            if (u < nu) {
                const uint64_t k = tk[u];
                keys[off + u] = k;
                ucnt[off + u] = tc[u];
                rows_per[off + u] = 0;
                head = u == 0 || (tk[u - 1] >> 12) != (k >> 12);
            }
            const unsigned long long hb = __ballot(head);
            if (n_stage + (uint32_t)__popcll(hb) > LK_STAGE) lk_flush(stage, n_stage, edge_list, state, l);
            if (head) stage[n_stage + (uint32_t)__popcll(hb & ((1ull << l) - 1ull))] = make_uint2(s1, u);
            n_stage += (uint32_t)__popcll(hb);
        }
        if (l == 0) site_nu[s1] = nu;
        nu_sum += nu;
    }
    }
    if (n_stage) lk_flush(stage, n_stage, edge_list, state, l);
    if (l == 0 && nu_sum) atomicAdd(&state[LS_NU], nu_sum);
}

// one lane per edge (site1, site2): EMIT = false counts its LD rows (rows_per at the edge's first key, summed per site), EMIT = true
// writes them behind the rows of the site's earlier edges -- (site1, site2, mm) order, the order the sorted chain gave
template <bool EMIT>
__global__ void __launch_bounds__(256) k_edge_rows(const uint64_t *keys, const uint32_t *ucnt, uint32_t *rows_per, const uint32_t *site_off,
                                                   const uint32_t *site_nu, uint32_t *site_rows, const uint32_t *site_row_off, const uint2 *edge_list,
                                                   SiteView v, int min_snp, isx_ld *out, uint64_t cap_ld, uint32_t *state, Philox ph)
{
    if (state[LS_FLAGS] & (LKF_KEYS | LKF_BUCKET)) return;
    if (EMIT && (uint64_t)state[LS_NLD] > cap_ld) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&state[LS_FLAGS], LKF_LD); return; }
    const uint32_t n_edges = state[LS_NEDGES];
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += gridDim.x * blockDim.x) {
        const uint2 ed = edge_list[e];
        const uint32_t s1 = ed.x, u = ed.y, off = site_off[s1], nu = site_nu[s1];
        const uint32_t s2 = (uint32_t)(keys[off + u] >> 12);
        if (!EMIT) {
            const uint32_t rows = ld_edge_rows<false>(keys + off, ucnt + off, u, nu, s1, s2, v, min_snp, nullptr, ph);
            if (rows) { rows_per[off + u] = rows; atomicAdd(&site_rows[s1], rows); }
        } else {
            if (rows_per[off + u] == 0) continue;
            uint32_t before = site_row_off[s1];
            for (uint32_t q = 0; q < u; q++) before += rows_per[off + q];
            (void)ld_edge_rows<true>(keys + off, ucnt + off, u, nu, s1, s2, v, min_snp, out + before, ph);
        }
    }
}

// mm path: the site table in position order without a device-wide sort.  k_pileup_mm writes a window's sites side by side (one allocation a
// window) and records (first, count) per window; k_scan_u32 over the counts gives every window's place in position order and the list of
// windows that have sites; here one wave per listed window ranks its few sites by position (a count of the smaller ones) and copies them there.
__global__ void __launch_bounds__(64) k_site_order(const isx_site *sites, const uint32_t *win_base, const uint32_t *win_off, const uint32_t *win_list,
                                                   const uint32_t *n_list_p, isx_site *sorted)
{
    const uint32_t l = threadIdx.x, n_list = *n_list_p;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t w = win_list[li], off = win_off[w], n = win_off[w + 1] - off;
        const isx_site *src = sites + win_base[w];
        for (uint32_t e = l; e < n; e += 64) {
            const isx_site me = src[e];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) rank += src[j].gpos < me.gpos ? 1u : 0u;     // (a position is a site once)
            sorted[off + rank] = me;
        }
    }
}

__global__ void k_ao_pair_keys(const isx_ao *ao, uint32_t n, uint32_t *ao_key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ao_key[i] = ao[i].pair;
}


// ------------------------------------------------------------------------------------------
// Dense co-occurrence path (linkage_mode 2, n_mm_bins == 1): per split the read-pair x allele
// incidence matrix X (int8; rows = read pairs with an allele observation in the split, columns =
// 4 per SNP site) is materialised column-major, and every cross-site co-occurrence count
//     G[p1][p2]['mm2combo2counts'][0]["b1:b2"]  (linkage.py:30-42)  ==  (X^T X)[(p1,b1), (p2,b2)]
// comes out of v_mfma_i32_32x32x32_i8 tiles (one wave per 32x32 tile of the upper triangle;
// lane l feeds 16 consecutive pair-rows of column l%32 for A and for B -- the k order only has to
// agree between A and B, tools/mfma_probe.hip).  Same-site combinations (self pairs) are not a
// product of X with itself (identical bases contribute C(x,2), differing bases are ordered by
// arrival) and come from k_pair_incr<.., SELF_ONLY>.
// ------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_ao_key64(const isx_ao *ao, uint32_t n, const uint32_t *site_split, uint64_t *key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = ((uint64_t)site_split[ao[i].site] << 32) | ao[i].pair;
}

__global__ void k_row_heads(const uint64_t *key, uint32_t n, uint32_t *head)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// first row / first site of every split that has any (0xFFFFFFFF elsewhere; filled on the host)
__global__ void k_split_first_row(const uint64_t *key, const uint32_t *row_id, uint32_t n, uint32_t *first_row)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t sp = (uint32_t)(key[i] >> 32);
    if (i == 0 || (uint32_t)(key[i - 1] >> 32) != sp) first_row[sp] = row_id[i];
}

__global__ void k_split_first_site(const uint32_t *site_split, uint32_t n_sites, uint32_t *first_site)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sites) return;
    if (i == 0 || site_split[i - 1] != site_split[i]) first_site[site_split[i]] = i;
}

__global__ void k_dense_scatter(const isx_ao *ao, const uint64_t *key, const uint32_t *row_id, const uint32_t *head,
                                uint32_t n, const uint32_t *split_slot, const DenseSplit *splits, uint8_t *xt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = split_slot[(uint32_t)(key[i] >> 32)];
    if (slot == 0xFFFFFFFFu) return;
    const DenseSplit ds = splits[slot];
    const isx_ao a = ao[i];
    const uint64_t row = row_id[i] + head[i] - 1 - ds.first_row;     // row_id = exclusive scan of the heads
    const uint64_t col = (uint64_t)(a.site - ds.first_site) * 4 + a.base;
    const uint64_t idx = ds.xt_off + col * ds.rpad + row;
    atomicAdd(reinterpret_cast<uint32_t *>(xt + (idx & ~3ull)), 1u << (8 * (idx & 3)));
}

// X^T X on the matrix cores.  A workgroup (4 waves) owns a 128 x 128 block of allele columns (I <= J) and
// walks the rows (read pairs) in chunks of 128: both 128 x 128-byte operand panels go global -> registers
// -> LDS (double buffered, 16-byte pieces XOR-swizzled by the row so that the b128 reads of 32 different
// rows spread over all banks), every wave computes a 64 x 64 sub-block = 2 x 2 tiles of
// v_mfma_i32_32x32x32_i8, so each operand fragment read from LDS feeds two MFMAs and each byte of X^T is
// fetched from L2 once per 128 columns instead of once per 32.
// Operand layout: lane l holds row (l & 31) and 16 k-values of half (l >> 5); which k-values a (half,
// step) slot carries is free as long as A and B agree (a dot product has no order).
// EMIT=false: per 32 x 32 tile the number of (site1 < site2) non-zero counts; EMIT=true: keys + counts.
constexpr int DG_LDS_BYTES = 2 * 2 * 128 * 128;

template <bool EMIT>
__global__ void __launch_bounds__(256) k_dense_gemm(const DenseTile *blocks, const DenseSplit *splits, const uint8_t *xt,
                                                    uint32_t *tile_cnt, const uint32_t *tile_off, uint64_t *keys,
                                                    uint32_t *cnts, int sb)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t dg_lds[];
    const DenseTile bl = blocks[blockIdx.x];
    const DenseSplit ds = splits[bl.slot];
    const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wi = wave >> 1, wj = wave & 1;
    const uint32_t ncols = ds.ctiles * 32;
    const bool diag = bl.I == bl.J;
    // which of this wave's 2 x 2 tiles exist (inside the matrix, upper triangle incl. the diagonal tile)
    bool use[2][2];
    uint32_t tI[2], tJ[2];
#pragma unroll
    for (int a = 0; a < 2; a++) { tI[a] = bl.I * 4 + wi * 2 + a; tJ[a] = bl.J * 4 + wj * 2 + a; }
    bool any = false;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) { use[a][b] = tI[a] < ds.ctiles && tJ[b] < ds.ctiles && tJ[b] >= tI[a]; any |= use[a][b]; }

    // global -> register staging: thread handles pieces tid + 256 u (u < 4) of a 128-row x 8-piece panel
    const uint8_t *src_a[4], *src_b[4];
    uint32_t dst[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t idx = (uint32_t)tid + 256u * u, row = idx >> 3, piece = idx & 7;
        const uint32_t ra = min(bl.I * 128 + row, ncols - 1), rb = min(bl.J * 128 + row, ncols - 1);   // clamp: padding rows are masked at the output
        src_a[u] = xt + ds.xt_off + (uint64_t)ra * ds.rpad + piece * 16;
        src_b[u] = xt + ds.xt_off + (uint64_t)rb * ds.rpad + piece * 16;
        dst[u] = row * 128 + ((piece ^ (row & 7)) << 4);
    }
    v4i ga[4], gb[4];
    auto gload = [&](uint32_t k0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ga[u] = *reinterpret_cast<const v4i *>(src_a[u] + k0);
            if (!diag) gb[u] = *reinterpret_cast<const v4i *>(src_b[u] + k0);
        }
    };
    auto lstore = [&](int buf) {
        uint8_t *pa = dg_lds + buf * (2 * 128 * 128), *pb = pa + 128 * 128;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            *reinterpret_cast<v4i *>(pa + dst[u]) = ga[u];
            if (!diag) *reinterpret_cast<v4i *>(pb + dst[u]) = gb[u];
        }
    };
    v16i acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = v16i{0};
    const uint32_t n_chunks = ds.rpad / 128;
    if (n_chunks) { gload(0); lstore(0); }
    __syncthreads();
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (c + 1 < n_chunks) gload((c + 1) * 128);             // in flight behind this chunk's MFMAs
        const uint8_t *pa = dg_lds + (c & 1) * (2 * 128 * 128), *pb = diag ? pa : pa + 128 * 128;
        if (any) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t piece = 2 * q + (l >> 5);
                v4i fa[2], fb[2];
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    const uint32_t ra = wi * 64 + a * 32 + (l & 31), rb = wj * 64 + a * 32 + (l & 31);
                    fa[a] = *reinterpret_cast<const v4i *>(pa + ra * 128 + ((piece ^ (ra & 7)) << 4));
                    fb[a] = *reinterpret_cast<const v4i *>(pb + rb * 128 + ((piece ^ (rb & 7)) << 4));
                }
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++)
                        if (use[a][b]) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (c + 1 < n_chunks) lstore((c + 1) & 1);              // that buffer was last read in round c - 1
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if (!use[a][b]) continue;                           // wave-uniform
            const uint32_t I = tI[a], J = tJ[b];
            const uint32_t t = ds.tile0 + I * ds.ctiles - (I * (I - 1)) / 2 + (J - I);
            uint32_t run = EMIT ? tile_off[t] : 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t ci = I * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);      // C/D layout of the 32x32 forms
                const uint32_t cj = J * 32 + (l & 31);
                const uint32_t si = ci >> 2, sj = cj >> 2;
                const int v = acc[a][b][r];
                const bool hit = v > 0 && si < sj && sj < ds.n_sites;
                const unsigned long long bal = __ballot(hit);
                if (EMIT && hit) {
                    const uint32_t o = run + (uint32_t)__popcll(bal & ((1ull << l) - 1ull));
                    keys[o] = make_key(sb, ds.first_site + si, ds.first_site + sj, 0, ci & 3, cj & 3);
                    cnts[o] = (uint32_t)v;
                }
                run += (uint32_t)__popcll(bal);
            }
            if (!EMIT && l == 0) tile_cnt[t] = run;
        }
    }
}

__global__ void k_fill_ones(uint32_t *p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 1u;
}

template <class T>
int ensure(DevBuf<T> &b, size_t n)
{
    if (b.cap >= n && b.p) return ISX_OK;
    if (b.p) isx_dev_free(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = n + n / 4 + 256;
    HIP_TRY(isx_dev_malloc(reinterpret_cast<void **>(&b.p), want * sizeof(T)));
    b.cap = want;
    return ISX_OK;
}

int ensure_temp(DevBuf<uint8_t> &b, size_t bytes) { return ensure(b, bytes); }

// room for n LD rows behind the head of ld_block
int ensure_ld(LinkageBuffers &B, size_t n)
{
    const int rc = ensure(B.ld_block, n + LD_HEAD);
    if (rc) return rc;
    B.ld.p = B.ld_block.p + LD_HEAD;
    B.ld.cap = B.ld_block.cap - LD_HEAD;
    return ISX_OK;
}

inline int bits_for(uint64_t n)
{
    int b = 1;
    while (b < 64 && (n >> b)) b++;
    return b;
}

}  // namespace

void LinkageBuffers::release()
{
    void *ps[] = {site_keys.p, site_keys2.p, sites_sorted.p, site_gpos.p, site_split.p, ao_key.p, ao2.p,
                  ao_key2.p, incr_cnt.p, incr_off.p, keys.p, keys2.p, ukeys.p, ucnt.p, n_runs.p, rows_per.p,
                  row_off.p, ld_block.p, temp.p, chain_head.p, next.p, site_cnt.p, site_off.p, site_cur.p, site_nu.p, site_rows.p, site_row_off.p, site_list1.p, site_list2.p, edge_list.p, win_off.p, win_list.p, key64.p, key64b.p, head.p, row_id.p, first_row.p, first_site.p,
                  split_slot.p, tile_cnt.p, tile_off.p, vals.p, vals2.p, dsplits.p, dtiles.p, xt.p};
    for (void *p : ps) if (p) isx_dev_free(p);
    *this = LinkageBuffers();
}

// rocPRIM sorts fewer than 2^20 items by block sort + ~log2(n / block) merge passes (two dozen launches for the 10^5 .. 10^6 allele
// observations / pair increments of a batch); in a stream of batches every launch of a finisher's chain queues behind other batches'
// kernels, so the chain's length in LAUNCHES is what it costs.  Onesweep: one histogram + one pass per 8 bits (ISX_SORT_MERGE=1: rocPRIM's choice)
using isx_sort_onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
static const bool g_sort_merge = getenv("ISX_SORT_MERGE") != nullptr;
#define EV(i) HIP_TRY(hipEventRecord(in.ev[i], s))
#define RP(call_with_temp)                                                                       \
    do {                                                                                         \
        void *tp = nullptr; size_t tb = 0;                                                       \
        HIP_TRY(call_with_temp);                                                                 \
        if ((rc = ensure_temp(B.temp, tb))) return rc;                                           \
        tp = B.temp.p; tb = B.temp.cap;                                                          \
        HIP_TRY(call_with_temp);                                                                 \
    } while (0)

namespace {

// exclusive scan of cnt[0..n) into off, returns the total (one small D2H + sync)
int scan_total(LinkageBuffers &B, hipStream_t s, uint32_t *cnt, uint32_t *off, uint32_t n, uint64_t &total)
{
    int rc;
    RP(rocprim::exclusive_scan(tp, tb, cnt, off, 0u, n, rocprim::plus<uint32_t>(), s));
    uint32_t last_off = 0, last_cnt = 0;
    HIP_TRY(isx_read_back(&last_off, off + (n - 1), 4, s));
    HIP_TRY(isx_read_back(&last_cnt, cnt + (n - 1), 4, s));
    HIP_TRY(isx_read_sync(s));
    total = (uint64_t)last_off + last_cnt;
    return ISX_OK;
}

// steps 3-5 of the sparse path: group by pair, all i<j combinations as keys, sort, run-length encode
int sparse_path(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out, uint32_t n_ao, uint32_t &n_u)
{
    hipStream_t s = in.stream;
    int rc;
    const int sb = bits_for(in.n_sites);
    n_u = 0;
    const int pair_bits = bits_for(in.n_pairs ? in.n_pairs : 0xFFFFFFFFull);
    if (g_sort_merge) RP(rocprim::radix_sort_pairs(tp, tb, B.ao_key.p, B.ao_key2.p, in.ao, B.ao2.p, n_ao, 0, pair_bits, s));
    else RP(rocprim::radix_sort_pairs<isx_sort_onesweep>(tp, tb, B.ao_key.p, B.ao_key2.p, in.ao, B.ao2.p, n_ao, 0, pair_bits, s));
    EV(3);
    if ((rc = ensure(B.incr_cnt, n_ao)) || (rc = ensure(B.incr_off, (size_t)n_ao + 1))) return rc;
    hipLaunchKernelGGL((k_pair_incr<false, false>), dim3((n_ao + 255) / 256), dim3(256), 0, s, B.ao2.p, n_ao,
                       B.site_split.p, B.incr_cnt.p, nullptr, nullptr, sb);
    uint64_t n_inc = 0;
    if ((rc = scan_total(B, s, B.incr_cnt.p, B.incr_off.p, n_ao, n_inc))) return rc;
    out.n_increments = n_inc;
    if (n_inc == 0) { EV(4); return ISX_OK; }
    if (n_inc >= 0xFFFFFFFFull) { isx_set_error("more than 2^32 pair increments in one batch"); return ISX_ERR_CAPACITY; }
    if ((rc = ensure(B.keys, n_inc)) || (rc = ensure(B.keys2, n_inc)) || (rc = ensure(B.ukeys, n_inc)) ||
        (rc = ensure(B.ucnt, n_inc)) || (rc = ensure(B.n_runs, 2))) return rc;
    hipLaunchKernelGGL((k_pair_incr<true, false>), dim3((n_ao + 255) / 256), dim3(256), 0, s, B.ao2.p, n_ao,
                       B.site_split.p, nullptr, B.incr_off.p, B.keys.p, sb);
    if (g_sort_merge) RP(rocprim::radix_sort_keys(tp, tb, B.keys.p, B.keys2.p, (size_t)n_inc, 0, 2 * sb + 12, s));
    else RP(rocprim::radix_sort_keys<isx_sort_onesweep>(tp, tb, B.keys.p, B.keys2.p, (size_t)n_inc, 0, 2 * sb + 12, s));
    RP(rocprim::run_length_encode(tp, tb, B.keys2.p, (size_t)n_inc, B.ukeys.p, B.ucnt.p, B.n_runs.p, s));
    HIP_TRY(isx_read_back(&n_u, B.n_runs.p, 4, s));
    EV(4);
    HIP_TRY(isx_read_sync(s));
    return ISX_OK;
}

// steps 3-6 as the bucket chain (kernels above); *fell_back: a site's keys did not fit the LDS table, nothing of the chain's output is valid
int bucket_chain(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out, const isx_site *sites_sorted, bool *fell_back)
{
    hipStream_t s = in.stream;
    int rc;
    *fell_back = false;
    const uint32_t n_ao = in.n_ao, n_sites = in.n_sites;
    // the chains' heads: a slot per read pair when the ids are dense enough (no shared chains, neighbouring slots for the observations of a
    // site), twice as many slots as observations behind a hash otherwise
    const bool direct = in.n_pairs && in.n_pairs <= 8ull * n_ao + (1ull << 20) && in.n_pairs <= (1ull << 31);
    int logH = 12;
    while (logH < 31 && (1ull << logH) < (direct ? in.n_pairs : 2ull * n_ao)) logH++;
    const size_t H = (size_t)1 << logH;
    if (B.chain_head.cap < H || !B.chain_head.p) {
        if ((rc = ensure(B.chain_head, H))) return rc;
        HIP_TRY(hipMemsetAsync(B.chain_head.p, 0, B.chain_head.cap * sizeof(uint64_t), s));
        B.chain_epoch = 0;
    }
    if (++B.chain_epoch == 0) {                                 // 2^32 batches later: the tags start over
        HIP_TRY(hipMemsetAsync(B.chain_head.p, 0, B.chain_head.cap * sizeof(uint64_t), s));
        B.chain_epoch = 1;
    }
    size_t cap_keys = std::max<size_t>(B.keys.cap, std::max<size_t>((size_t)n_ao * 2, 65536));
    size_t cap_ld = std::max<size_t>(B.ld_block.cap > LD_HEAD ? B.ld_block.cap - LD_HEAD : 0, 4096);
    static_assert(sizeof(isx_ld) * LD_HEAD >= LS_WORDS * sizeof(uint32_t), "the state words fit the head of ld_block");
    // test aids: ISX_LINK_MAXU = unique keys a site may have before the batch falls back (<= LK_MAXU); ISX_LINK_TEST_CAPS = "keys,rows":
    // what the first attempt tells the kernels the tables hold (the growth steps without a batch that needs them)
    uint32_t max_u = LK_MAXU;
    if (const char *e = getenv("ISX_LINK_MAXU")) max_u = (uint32_t)std::min<long>(std::max<long>(atol(e), 1), LK_MAXU);
    size_t test_keys = 0, test_ld = 0;
    if (const char *e = getenv("ISX_LINK_TEST_CAPS")) { unsigned long a = 0, b = 0; if (sscanf(e, "%lu,%lu", &a, &b) == 2) { test_keys = a; test_ld = b; } }
    auto size_tables = [&]() -> int {
        int r;
        if ((r = ensure(B.keys, cap_keys)) || (r = ensure(B.ucnt, cap_keys)) || (r = ensure(B.rows_per, cap_keys)) || (r = ensure(B.edge_list, cap_keys))) return r;
        cap_keys = std::min(std::min(B.keys.cap, B.ucnt.cap), std::min(B.rows_per.cap, B.edge_list.cap));
        return ISX_OK;
    };
    auto size_ld = [&]() -> int {
        const int r = ensure_ld(B, cap_ld);
        if (r) return r;
        cap_ld = B.ld.cap;
        return ISX_OK;
    };
    if ((rc = ensure(B.next, n_ao)) || (rc = ensure(B.site_cnt, (size_t)n_sites + 1)) || (rc = ensure(B.site_off, (size_t)n_sites + 1)) ||
        (rc = ensure(B.site_cur, (size_t)n_sites + 1)) || (rc = ensure(B.site_nu, n_sites)) || (rc = ensure(B.site_rows, (size_t)n_sites + 1)) ||
        (rc = ensure(B.site_row_off, (size_t)n_sites + 1)) || (rc = ensure(B.site_list1, n_sites)) || (rc = ensure(B.site_list2, n_sites)) || (rc = size_tables()) || (rc = size_ld())) return rc;
    uint32_t *state = reinterpret_cast<uint32_t *>(B.ld_block.p);           // the chain's state words live in front of the rows: one read-back
    const dim3 blk(256), g_sites((n_sites + 255) / 256), g_ao((n_ao + 255) / 256);
    SiteView v{sites_sorted, in.slev, in.snv, in.M == 1 ? 1 : 0};
    hipLaunchKernelGGL(k_link_prep, g_sites, blk, 0, s, sites_sorted, n_sites, in.split_bounds, in.n_splits, B.site_gpos.p, B.site_split.p,
                       B.site_cnt.p, B.site_rows.p, state);
    EV(1);
    hipLaunchKernelGGL(k_ao_chain, g_ao, blk, 0, s, in.ao, n_ao, B.site_gpos.p, n_sites, reinterpret_cast<unsigned long long *>(B.chain_head.p),
                       32 - logH, direct ? (uint32_t)H : 0u, B.chain_epoch, B.next.p);
    EV(2);
    hipLaunchKernelGGL((k_pair_walk<false>), g_ao, blk, 0, s, in.ao, n_ao, B.next.p, B.site_split.p, B.site_cnt.p, nullptr, nullptr, (uint64_t)0, state);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, B.site_cnt.p, B.site_off.p, n_sites, state + LS_NINC, B.site_cur.p, B.site_list1.p, state + LS_NLIST1);
    const dim3 g_waves(std::min<uint32_t>((n_sites + LK_ROUND - 1) / LK_ROUND, 4096)), g_edges(std::min<uint32_t>((uint32_t)std::min<size_t>((cap_keys + 255) / 256, 0x7FFFFFFF), 2048));
    uint32_t h[LS_WORDS] = {0};
    for (int stage = 0, attempt = 0;; attempt++) {               // stage 0: from the buckets on; 1: the rows only (after a table grew)
        if (attempt == 4) { isx_set_error("linkage tables still too small after three growth steps"); return ISX_ERR_CAPACITY; }
        if (stage == 0) {
            hipLaunchKernelGGL((k_pair_walk<true>), g_ao, blk, 0, s, in.ao, n_ao, B.next.p, B.site_split.p, B.site_cur.p, B.site_off.p, B.keys.p,
                               (uint64_t)(attempt == 0 && test_keys ? std::min(test_keys, cap_keys) : cap_keys), state);
            EV(3);
            hipLaunchKernelGGL(k_site_edges, g_waves, dim3(64), 0, s, B.keys.p, B.ucnt.p, B.rows_per.p, B.site_off.p, B.site_nu.p,
                               reinterpret_cast<uint2 *>(B.edge_list.p), state, max_u, B.site_list1.p);
            hipLaunchKernelGGL((k_edge_rows<false>), g_edges, blk, 0, s, B.keys.p, B.ucnt.p, B.rows_per.p, B.site_off.p, B.site_nu.p, B.site_rows.p,
                               (const uint32_t *)nullptr, reinterpret_cast<const uint2 *>(B.edge_list.p), v, in.min_snp, (isx_ld *)nullptr, (uint64_t)0, state, in.philox);
            hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, B.site_rows.p, B.site_row_off.p, n_sites, state + LS_NLD, (uint32_t *)nullptr, B.site_list2.p, state + LS_NLIST2);
            EV(4);
        }
        hipLaunchKernelGGL((k_edge_rows<true>), g_edges, blk, 0, s, B.keys.p, B.ucnt.p, B.rows_per.p, B.site_off.p, B.site_nu.p, B.site_rows.p,
                           B.site_row_off.p, reinterpret_cast<const uint2 *>(B.edge_list.p), v, in.min_snp, B.ld.p,
                           (uint64_t)(attempt == 0 && test_ld ? std::min(test_ld, cap_ld) : cap_ld), state, in.philox);
        HIP_TRY(hipGetLastError());
        // the state words and, with them, the first rows (most batches of a metagenome have a few hundred): one copy, one wait
        const size_t n_pre = std::min<size_t>(cap_ld, LD_PREFIX_ROWS);
        B.h_ld.resize(LD_HEAD + n_pre);
        HIP_TRY(isx_read_back(B.h_ld.data(), B.ld_block.p, (LD_HEAD + n_pre) * sizeof(isx_ld), s));
        EV(5);
        HIP_TRY(isx_read_sync(s));
        memcpy(h, B.h_ld.data(), sizeof(h));
        if (h[LS_FLAGS] & LKF_KEYS) {
            if (h[LS_NINC] == 0xFFFFFFFFu) { isx_set_error("more than 2^32 pair increments in one batch"); return ISX_ERR_CAPACITY; }
            cap_keys = (size_t)h[LS_NINC] + h[LS_NINC] / 4 + 4096;
            B.keys.cap = B.ucnt.cap = B.rows_per.cap = B.edge_list.cap = 0;        // (ensure() frees and reallocates: nothing in them is needed)
            if ((rc = size_tables())) return rc;
            HIP_TRY(hipMemsetAsync(state + LS_NU, 0, (LS_FLAGS + 1 - LS_NU) * sizeof(uint32_t), s));      // (the total and the site list stay)
            HIP_TRY(hipMemsetAsync(B.site_cur.p, 0, ((size_t)n_sites + 1) * sizeof(uint32_t), s));
            stage = 0;
            continue;
        }
        if (h[LS_FLAGS] & LKF_BUCKET) { *fell_back = true; return ISX_OK; }
        if (h[LS_FLAGS] & LKF_LD) {
            if (h[LS_NLD] == 0xFFFFFFFFu) { isx_set_error("more than 2^32 LD rows in one batch"); return ISX_ERR_CAPACITY; }
            // (the state words sit in the block that is about to move: they are put back in front of the new rows)
            cap_ld = (size_t)h[LS_NLD] + h[LS_NLD] / 4 + 4096;
            B.ld_block.cap = 0;
            if ((rc = size_ld())) return rc;
            state = reinterpret_cast<uint32_t *>(B.ld_block.p);
            h[LS_FLAGS] &= ~LKF_LD;
            HIP_TRY(hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, s));
            HIP_TRY(isx_wait_stream(s));
            stage = 1;
            continue;
        }
        break;
    }
    out.n_increments = h[LS_NINC];
    out.n_edges = h[LS_NEDGES];
    out.n_ld = h[LS_NLD];
    out.ld_host = B.h_ld.data() + LD_HEAD;
    out.n_ld_host = std::min<uint64_t>(out.n_ld, B.h_ld.size() - LD_HEAD);
    return ISX_OK;
}

// steps 3-5 of the dense path: X^T per split, int8 MFMA tiles, self pairs, sort + reduce by key
int dense_path(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out, uint32_t n_ao, uint32_t n_sites, uint32_t &n_u)
{
    hipStream_t s = in.stream;
    int rc;
    n_u = 0;
    const int sb = bits_for(n_sites);
    const uint32_t nsp = (uint32_t)in.n_splits;
    // rows: allele observations sorted by (split, pair); one row per distinct (split, pair)
    if ((rc = ensure(B.key64, n_ao)) || (rc = ensure(B.key64b, n_ao)) || (rc = ensure(B.head, n_ao)) ||
        (rc = ensure(B.row_id, (size_t)n_ao + 1)) || (rc = ensure(B.first_row, (size_t)nsp + 1)) ||
        (rc = ensure(B.first_site, (size_t)nsp + 1)) || (rc = ensure(B.split_slot, nsp))) return rc;
    const dim3 ga((n_ao + 255) / 256), blk(256);
    hipLaunchKernelGGL(k_ao_key64, ga, blk, 0, s, in.ao, n_ao, B.site_split.p, B.key64.p);
    const int key_bits = 32 + bits_for(nsp);
    RP(rocprim::radix_sort_pairs(tp, tb, B.key64.p, B.key64b.p, in.ao, B.ao2.p, n_ao, 0, key_bits, s));
    EV(3);
    hipLaunchKernelGGL(k_row_heads, ga, blk, 0, s, B.key64b.p, n_ao, B.head.p);
    uint64_t n_rows = 0;
    if ((rc = scan_total(B, s, B.head.p, B.row_id.p, n_ao, n_rows))) return rc;
    HIP_TRY(hipMemsetAsync(B.first_row.p, 0xFF, ((size_t)nsp + 1) * 4, s));
    HIP_TRY(hipMemsetAsync(B.first_site.p, 0xFF, ((size_t)nsp + 1) * 4, s));
    hipLaunchKernelGGL(k_split_first_row, ga, blk, 0, s, B.key64b.p, B.row_id.p, n_ao, B.first_row.p);
    hipLaunchKernelGGL(k_split_first_site, dim3((n_sites + 255) / 256), blk, 0, s, B.site_split.p, n_sites, B.first_site.p);
    std::vector<uint32_t> frow((size_t)nsp + 1), fsite((size_t)nsp + 1);
    HIP_TRY(isx_read_back(frow.data(), B.first_row.p, frow.size() * 4, s));
    HIP_TRY(isx_read_back(fsite.data(), B.first_site.p, fsite.size() * 4, s));
    HIP_TRY(isx_read_sync(s));
    frow[nsp] = (uint32_t)n_rows; fsite[nsp] = n_sites;
    for (uint32_t i = nsp; i-- > 0;) {
        if (frow[i] == 0xFFFFFFFFu) frow[i] = frow[i + 1];
        if (fsite[i] == 0xFFFFFFFFu) fsite[i] = fsite[i + 1];
    }
    // split table + tile list (host; a few entries per split)
    std::vector<DenseSplit> ds;
    std::vector<DenseTile> blocks;
    std::vector<uint32_t> slot(nsp, 0xFFFFFFFFu);
    uint64_t bytes = 0, macs = 0, n_tiles64 = 0;
    for (uint32_t sp = 0; sp < nsp; sp++) {
        const uint32_t rows = frow[sp + 1] - frow[sp], ns = fsite[sp + 1] - fsite[sp];
        if (rows == 0 || ns < 2) continue;           // no cross-site pair possible
        DenseSplit d{};
        d.xt_off = bytes;
        d.rpad = (rows + 127) / 128 * 128;
        d.ctiles = (ns * 4 + 31) / 32;
        d.first_site = fsite[sp]; d.n_sites = ns; d.first_row = frow[sp];
        d.tile0 = (uint32_t)n_tiles64;
        bytes += (uint64_t)d.ctiles * 32 * d.rpad;
        slot[sp] = (uint32_t)ds.size();
        const uint32_t nb = (d.ctiles + 3) / 4;
        for (uint32_t I = 0; I < nb; I++)
            for (uint32_t J = I; J < nb; J++) blocks.push_back(DenseTile{(uint32_t)ds.size(), I, J, 0});
        n_tiles64 += (uint64_t)d.ctiles * (d.ctiles + 1) / 2;
        macs += (uint64_t)d.ctiles * (d.ctiles + 1) / 2 * 32 * 32 * d.rpad;       // useful tiles only (no block padding)
        ds.push_back(d);
    }
    out.dense_tiles = n_tiles64; out.dense_bytes = bytes; out.dense_macs = macs;
    if (bytes > (16ull << 30)) { isx_set_error("dense linkage path needs > 16 GiB for X^T: use the sparse path"); return ISX_ERR_CAPACITY; }
    if (n_tiles64 >= 0xFFFFFFFFull || blocks.size() >= 0x7FFFFFFFull) { isx_set_error("too many dense tiles"); return ISX_ERR_CAPACITY; }
    const uint32_t n_tiles = (uint32_t)n_tiles64, n_blocks = (uint32_t)blocks.size();
    uint64_t n_gemm = 0;
    if (n_tiles) {
        if ((rc = ensure(B.dsplits, ds.size())) || (rc = ensure(B.dtiles, blocks.size())) || (rc = ensure(B.xt, bytes + 16)) ||
            (rc = ensure(B.tile_cnt, n_tiles)) || (rc = ensure(B.tile_off, (size_t)n_tiles + 1))) return rc;
        HIP_TRY(hipMemcpyAsync(B.dsplits.p, ds.data(), ds.size() * sizeof(DenseSplit), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.dtiles.p, blocks.data(), blocks.size() * sizeof(DenseTile), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(B.split_slot.p, slot.data(), slot.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(B.xt.p, 0, bytes + 16, s));
        hipLaunchKernelGGL(k_dense_scatter, ga, blk, 0, s, B.ao2.p, B.key64b.p, B.row_id.p, B.head.p, n_ao, B.split_slot.p,
                           B.dsplits.p, B.xt.p);
        HIP_TRY(hipEventRecord(in.ev_mfma[0], s));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense_gemm<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dense_gemm<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS_BYTES);
        hipLaunchKernelGGL((k_dense_gemm<false>), dim3(n_blocks), blk, DG_LDS_BYTES, s, B.dtiles.p, B.dsplits.p,
                           B.xt.p, B.tile_cnt.p, nullptr, nullptr, nullptr, sb);
        HIP_TRY(hipEventRecord(in.ev_mfma[1], s));
        if ((rc = scan_total(B, s, B.tile_cnt.p, B.tile_off.p, n_tiles, n_gemm))) return rc;
    } else {
        HIP_TRY(hipEventRecord(in.ev_mfma[0], s));
        HIP_TRY(hipEventRecord(in.ev_mfma[1], s));
    }
    // same-site combinations from the pair lists
    if ((rc = ensure(B.incr_cnt, n_ao)) || (rc = ensure(B.incr_off, (size_t)n_ao + 1))) return rc;
    hipLaunchKernelGGL((k_pair_incr<false, true>), ga, blk, 0, s, B.ao2.p, n_ao, B.site_split.p, B.incr_cnt.p, nullptr, nullptr, sb);
    uint64_t n_self = 0;
    if ((rc = scan_total(B, s, B.incr_cnt.p, B.incr_off.p, n_ao, n_self))) return rc;
    const uint64_t n_k = n_gemm + n_self;
    if (n_k == 0) { EV(4); return ISX_OK; }
    if (n_k >= 0xFFFFFFFFull) { isx_set_error("more than 2^32 co-occurrence keys in one batch"); return ISX_ERR_CAPACITY; }
    if ((rc = ensure(B.keys, n_k)) || (rc = ensure(B.keys2, n_k)) || (rc = ensure(B.vals, n_k)) || (rc = ensure(B.vals2, n_k)) ||
        (rc = ensure(B.ukeys, n_k)) || (rc = ensure(B.ucnt, n_k)) || (rc = ensure(B.n_runs, 2))) return rc;
    if (n_gemm)
        hipLaunchKernelGGL((k_dense_gemm<true>), dim3(n_blocks), blk, DG_LDS_BYTES, s, B.dtiles.p, B.dsplits.p, B.xt.p,
                           nullptr, B.tile_off.p, B.keys.p, B.vals.p, sb);
    if (n_self) {
        hipLaunchKernelGGL((k_pair_incr<true, true>), ga, blk, 0, s, B.ao2.p, n_ao, B.site_split.p, nullptr, B.incr_off.p,
                           B.keys.p + n_gemm, sb);
        hipLaunchKernelGGL(k_fill_ones, dim3(((uint32_t)n_self + 255) / 256), blk, 0, s, B.vals.p + n_gemm, (uint32_t)n_self);
    }
    RP(rocprim::radix_sort_pairs(tp, tb, B.keys.p, B.keys2.p, B.vals.p, B.vals2.p, (size_t)n_k, 0, 2 * sb + 12, s));
    RP(rocprim::reduce_by_key(tp, tb, B.keys2.p, B.vals2.p, (size_t)n_k, B.ukeys.p, B.ucnt.p, B.n_runs.p,
                              rocprim::plus<uint32_t>(), rocprim::equal_to<uint64_t>(), s));
    RP(rocprim::reduce(tp, tb, B.vals2.p, B.n_runs.p + 1, 0u, (size_t)n_k, rocprim::plus<uint32_t>(), s));
    uint32_t h2[2] = {0, 0};
    HIP_TRY(isx_read_back(h2, B.n_runs.p, 8, s));
    EV(4);
    HIP_TRY(isx_read_sync(s));
    n_u = h2[0];
    out.n_increments = h2[1];
    return ISX_OK;
}

}  // namespace

int run_linkage(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out)
{
    hipStream_t s = in.stream;
    out = LinkageOut();
    const uint32_t n_sites = in.n_sites;
    int rc;
    const bool lt = getenv("ISX_LINK_TIMING") != nullptr;          // tuning aid: host time stamps on stderr
    const auto lt0 = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        if (lt) fprintf(stderr, "[run_linkage] %-22s +%.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lt0).count());
    };
    EV(0);
    HIP_TRY(hipEventRecord(in.ev_mfma[0], s));
    HIP_TRY(hipEventRecord(in.ev_mfma[1], s));
    if (n_sites == 0) { EV(1); EV(2); EV(3); EV(4); EV(5); return ISX_OK; }
    if (n_sites >= (1u << 26)) { isx_set_error("more than 2^26 SNP sites in one batch: split the batch"); return ISX_ERR_ARG; }
    if (in.M > 256) { isx_set_error("linkage supports at most 256 mm bins"); return ISX_ERR_ARG; }
    if (in.mode == 2 && in.M != 1) { isx_set_error("the dense MFMA linkage path needs n_mm_bins == 1"); return ISX_ERR_ARG; }

    // ---- 1. sites by position ----
    if ((rc = ensure(B.site_keys, n_sites)) || (rc = ensure(B.site_keys2, n_sites)) ||
        (rc = ensure(B.sites_sorted, n_sites)) || (rc = ensure(B.site_gpos, n_sites)) ||
        (rc = ensure(B.site_split, n_sites))) return rc;
    tick("site buffers");
    const isx_site *sites_sorted = in.sites;
    const char *site_env = getenv("ISX_LINK_SITE_SORT");            // "rocprim": the device-wide sort also where the window table would do (check / A/B)
    if (!in.sites_ordered && in.win_site_cnt && in.n_win > 0 && !(site_env && !strcmp(site_env, "rocprim"))) {
        if ((rc = ensure(B.win_off, (size_t)in.n_win + 1)) || (rc = ensure(B.win_list, (size_t)in.n_win)) || (rc = ensure(B.n_runs, 2))) return rc;
        hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, s, in.win_site_cnt, B.win_off.p, (uint32_t)in.n_win, B.n_runs.p, (uint32_t *)nullptr, B.win_list.p, B.n_runs.p + 1);
        hipLaunchKernelGGL(k_site_order, dim3(std::min<uint32_t>((uint32_t)in.n_win, 4096)), dim3(64), 0, s, in.sites, in.win_site_base, B.win_off.p, B.win_list.p,
                           B.n_runs.p + 1, B.sites_sorted.p);
        sites_sorted = B.sites_sorted.p;
    } else if (!in.sites_ordered) {
        hipLaunchKernelGGL(k_site_keys, dim3((n_sites + 255) / 256), dim3(256), 0, s, in.sites, n_sites, B.site_keys.p);
        RP(rocprim::radix_sort_pairs(tp, tb, B.site_keys.p, B.site_keys2.p, const_cast<isx_site *>(in.sites),
                                     B.sites_sorted.p, n_sites, 0, 32, s));
        sites_sorted = B.sites_sorted.p;
    }
    // ---- 2. allele observations (produced by the pileup kernel): position -> site rank ----
    const uint32_t n_ao = in.n_ao;
    out.n_ao = n_ao;
    // ISX_LINK_CHAIN=sorted: steps 3-6 through the device-wide sorts (the round-2 chain, kept as the bucket chain's fallback and its check)
    const char *chain_env = getenv("ISX_LINK_CHAIN");
    const bool bucket = in.mode != 2 && n_ao != 0 && !(chain_env && !strcmp(chain_env, "sorted"));
    bool ranked = false;
    if (bucket) {
        bool fell_back = false;
        if ((rc = bucket_chain(in, B, out, sites_sorted, &fell_back)) != ISX_OK) return rc;
        tick("bucket chain");
        if (!fell_back) { out.chain = 3; return ISX_OK; }
        ranked = true;                                      // (k_link_prep / k_ao_chain have run: sites split, observations ranked)
    } else {
        hipLaunchKernelGGL(k_site_split, dim3((n_sites + 255) / 256), dim3(256), 0, s, sites_sorted, n_sites,
                           in.split_bounds, in.n_splits, B.site_gpos.p, B.site_split.p);
        tick("site sort enqueued");
        EV(1);
    }
    if (n_ao == 0) { EV(2); EV(3); EV(4); EV(5); return ISX_OK; }
    if ((rc = ensure(B.ao_key, n_ao)) || (rc = ensure(B.ao2, n_ao)) || (rc = ensure(B.ao_key2, n_ao))) return rc;
    tick("ao buffers");
    if (ranked) hipLaunchKernelGGL(k_ao_pair_keys, dim3((n_ao + 255) / 256), dim3(256), 0, s, in.ao, n_ao, B.ao_key.p);
    else hipLaunchKernelGGL(k_ao_rank, dim3((n_ao + 255) / 256), dim3(256), 0, s, in.ao, n_ao, B.site_gpos.p, n_sites, B.ao_key.p);
    EV(2);
    out.chain = in.mode == 2 ? 2 : 1;

    // ---- 3-5. co-occurrence counts as sorted unique keys ----
    uint32_t n_u = 0;
    if (in.mode == 2) rc = dense_path(in, B, out, n_ao, n_sites, n_u);
    else rc = sparse_path(in, B, out, n_ao, n_u);
    if (rc != ISX_OK) return rc;
    tick("pair counts");
    if (n_u == 0) { if (out.n_increments == 0) { /* EV(3)/EV(4) recorded by the path */ } EV(5); return ISX_OK; }

    // ---- 6. LD rows ----
    const int sb = bits_for(n_sites);
    if ((rc = ensure(B.rows_per, n_u)) || (rc = ensure(B.row_off, (size_t)n_u + 1))) return rc;
    HIP_TRY(hipMemsetAsync(B.n_runs.p + 1, 0, 4, s));
    SiteView v{sites_sorted, in.slev, in.snv, in.M == 1 ? 1 : 0};
    hipLaunchKernelGGL(k_ld_rows<false>, dim3((n_u + 255) / 256), dim3(256), 0, s, B.ukeys.p, B.ucnt.p, n_u, v,
                       in.min_snp, B.rows_per.p, nullptr, nullptr, B.n_runs.p + 1, in.philox, sb);
    uint64_t n_ld = 0;
    if ((rc = scan_total(B, s, B.rows_per.p, B.row_off.p, n_u, n_ld))) return rc;
    uint32_t n_edges = 0;
    HIP_TRY(isx_read_back(&n_edges, B.n_runs.p + 1, 4, s));
    HIP_TRY(isx_read_sync(s));
    out.n_edges = n_edges;
    out.n_ld = n_ld;
    if (n_ld) {
        if ((rc = ensure_ld(B, n_ld))) return rc;
        hipLaunchKernelGGL(k_ld_rows<true>, dim3((n_u + 255) / 256), dim3(256), 0, s, B.ukeys.p, B.ucnt.p, n_u, v,
                           in.min_snp, nullptr, B.row_off.p, B.ld.p, nullptr, in.philox, sb);
    }
    EV(5);
    tick("ld rows enqueued");
    return ISX_OK;
}
