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
#include <algorithm>
#include <cstring>
#include <string.h>
#include <rocprim/rocprim.hpp>

#include "isx_internal.h"
#include "isx_linkage.h"

#pragma clang fp contract(off)

namespace {

// key layout: site1:26 | site2:26 | mm:8 | b1:2 | b2:2
__device__ __forceinline__ uint64_t make_key(uint32_t s1, uint32_t s2, uint32_t mm, uint32_t b1, uint32_t b2)
{
    return ((uint64_t)s1 << 38) | ((uint64_t)s2 << 12) | ((uint64_t)mm << 4) | (b1 << 2) | b2;
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
template <bool EMIT>
__global__ void __launch_bounds__(256) k_pair_incr(const isx_ao *ao, uint32_t n, const uint32_t *site_split,
                                                   uint32_t *cnt, const uint32_t *off, uint64_t *keys)
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
        if (EMIT) {
            // list order = (column, arrival order inside the column)
            const bool a_first = (a.site < b.site) || (a.site == b.site && a.obs_idx < b.obs_idx);
            keys[o + c] = a_first ? make_key(a.site, b.site, a.mm, a.base, b.base)
                                  : make_key(b.site, a.site, a.mm, b.base, a.base);
        }
        c++;
    }
    if (!EMIT) cnt[i] = c;
}

// cumulative counts over levels <= mm at a SNP site (mm_counts_to_counts on snv2mm2counts[p])
struct SiteView {
    const isx_site *sites;
    const isx_entry *entries;   // mm path
    const uint4 *counts;        // dense path
    int dense;
};

__device__ __forceinline__ void site_cum(const SiteView &v, uint32_t s, uint32_t mm, uint32_t *out, bool &has_mm)
{
    const isx_site st = v.sites[s];
    if (v.dense) {
        const uint4 c = v.counts[st.gpos];
        out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
        has_mm = (mm == 0);
        return;
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    has_mm = false;
    for (uint32_t k = 0; k < st.n_levels; k++) {
        const isx_entry e = v.entries[st.entry_off + k];
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

// One lane per unique key; the lane holding the first key of an edge (site1, site2) walks the
// edge's keys (ascending mm, then combo).  EMIT=false: count rows per edge.  EMIT=true: write.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_ld_rows(const uint64_t *ukeys, const uint32_t *ucnt, uint32_t n_u,
                                                 SiteView v, int min_snp, uint32_t *rows_per, const uint32_t *row_off,
                                                 isx_ld *out, uint32_t *n_edges)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_u) return;
    const uint64_t k0 = ukeys[u];
    const uint64_t edge = k0 >> 12;
    const bool head = (u == 0) || ((ukeys[u - 1] >> 12) != edge);
    if (!head) { if (!EMIT) rows_per[u] = 0; return; }
    if (!EMIT) atomicAdd(n_edges, 1u);
    const uint32_t s1 = (uint32_t)(k0 >> 38), s2 = (uint32_t)((k0 >> 12) & 0x3FFFFFFu);
    uint32_t combo[16];
#pragma unroll
    for (int i = 0; i < 16; i++) combo[i] = 0;
    uint32_t rows = 0;
    uint32_t o = EMIT ? row_off[u] : 0;
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
            out[o + rows] = r;
        }
        rows++;
    }
    if (!EMIT) rows_per[u] = rows;
}

template <class T>
int ensure(DevBuf<T> &b, size_t n)
{
    if (b.cap >= n && b.p) return ISX_OK;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = n + n / 4 + 256;
    HIP_TRY(hipMalloc(&b.p, want * sizeof(T)));
    b.cap = want;
    return ISX_OK;
}

int ensure_temp(DevBuf<uint8_t> &b, size_t bytes) { return ensure(b, bytes); }

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
                  row_off.p, ld.p, temp.p};
    for (void *p : ps) if (p) (void)hipFree(p);
    *this = LinkageBuffers();
}

#define EV(i) HIP_TRY(hipEventRecord(in.ev[i], s))

int run_linkage(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out)
{
    hipStream_t s = in.stream;
    out = LinkageOut();
    const uint32_t n_sites = in.n_sites;
    EV(0);
    if (n_sites == 0) { EV(1); EV(2); EV(3); EV(4); EV(5); return ISX_OK; }
    if (n_sites >= (1u << 26)) { isx_set_error("more than 2^26 SNP sites in one batch: split the batch"); return ISX_ERR_ARG; }
    if (in.M > 256) { isx_set_error("linkage supports at most 256 mm bins"); return ISX_ERR_ARG; }

    // ---- 1. sites by position ----
    int rc;
    if ((rc = ensure(B.site_keys, n_sites)) || (rc = ensure(B.site_keys2, n_sites)) ||
        (rc = ensure(B.sites_sorted, n_sites)) || (rc = ensure(B.site_gpos, n_sites)) ||
        (rc = ensure(B.site_split, n_sites))) return rc;
    hipLaunchKernelGGL(k_site_keys, dim3((n_sites + 255) / 256), dim3(256), 0, s, in.sites, n_sites, B.site_keys.p);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, B.site_keys.p, B.site_keys2.p, const_cast<isx_site *>(in.sites),
                                      B.sites_sorted.p, n_sites, 0, 32, s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::radix_sort_pairs(B.temp.p, tb, B.site_keys.p, B.site_keys2.p, const_cast<isx_site *>(in.sites),
                                      B.sites_sorted.p, n_sites, 0, 32, s));
    hipLaunchKernelGGL(k_site_split, dim3((n_sites + 255) / 256), dim3(256), 0, s, B.sites_sorted.p, n_sites,
                       in.split_bounds, in.n_splits, B.site_gpos.p, B.site_split.p);
    EV(1);

    // ---- 2. allele observations (produced by the pileup kernel): position -> site rank ----
    const uint32_t n_ao = in.n_ao;
    out.n_ao = n_ao;
    if (n_ao == 0) { EV(2); EV(3); EV(4); EV(5); return ISX_OK; }
    if ((rc = ensure(B.ao_key, n_ao)) || (rc = ensure(B.ao2, n_ao)) || (rc = ensure(B.ao_key2, n_ao))) return rc;
    hipLaunchKernelGGL(k_ao_rank, dim3((n_ao + 255) / 256), dim3(256), 0, s, in.ao, n_ao, B.site_gpos.p, n_sites,
                       B.ao_key.p);
    EV(2);

    // ---- 3. group by pair ----
    tb = 0;
    const int pair_bits = bits_for(in.n_pairs ? in.n_pairs : 0xFFFFFFFFull);
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, B.ao_key.p, B.ao_key2.p, in.ao, B.ao2.p, n_ao, 0, pair_bits, s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::radix_sort_pairs(B.temp.p, tb, B.ao_key.p, B.ao_key2.p, in.ao, B.ao2.p, n_ao, 0, pair_bits, s));
    EV(3);

    // ---- 4. pair increments ----
    if ((rc = ensure(B.incr_cnt, n_ao)) || (rc = ensure(B.incr_off, (size_t)n_ao + 1))) return rc;
    hipLaunchKernelGGL(k_pair_incr<false>, dim3((n_ao + 255) / 256), dim3(256), 0, s, B.ao2.p, n_ao, B.site_split.p,
                       B.incr_cnt.p, nullptr, nullptr);
    tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, B.incr_cnt.p, B.incr_off.p, 0u, n_ao, rocprim::plus<uint32_t>(), s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::exclusive_scan(B.temp.p, tb, B.incr_cnt.p, B.incr_off.p, 0u, n_ao, rocprim::plus<uint32_t>(), s));
    uint32_t last_off = 0, last_cnt = 0;
    HIP_TRY(hipMemcpyAsync(&last_off, B.incr_off.p + (n_ao - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last_cnt, B.incr_cnt.p + (n_ao - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint64_t n_inc = (uint64_t)last_off + last_cnt;
    out.n_increments = n_inc;
    if (n_inc == 0) { EV(4); EV(5); return ISX_OK; }
    if (n_inc >= 0xFFFFFFFFull) { isx_set_error("more than 2^32 pair increments in one batch"); return ISX_ERR_CAPACITY; }
    if ((rc = ensure(B.keys, n_inc)) || (rc = ensure(B.keys2, n_inc)) || (rc = ensure(B.ukeys, n_inc)) ||
        (rc = ensure(B.ucnt, n_inc)) || (rc = ensure(B.n_runs, 2))) return rc;
    hipLaunchKernelGGL(k_pair_incr<true>, dim3((n_ao + 255) / 256), dim3(256), 0, s, B.ao2.p, n_ao, B.site_split.p,
                       nullptr, B.incr_off.p, B.keys.p);
    // ---- 5. sort + run-length encode ----
    const int site_bits = bits_for(n_sites);
    (void)site_bits;
    tb = 0;
    HIP_TRY(rocprim::radix_sort_keys(nullptr, tb, B.keys.p, B.keys2.p, (size_t)n_inc, 0, 64, s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::radix_sort_keys(B.temp.p, tb, B.keys.p, B.keys2.p, (size_t)n_inc, 0, 64, s));
    tb = 0;
    HIP_TRY(rocprim::run_length_encode(nullptr, tb, B.keys2.p, (size_t)n_inc, B.ukeys.p, B.ucnt.p, B.n_runs.p, s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::run_length_encode(B.temp.p, tb, B.keys2.p, (size_t)n_inc, B.ukeys.p, B.ucnt.p, B.n_runs.p, s));
    uint32_t n_u = 0;
    HIP_TRY(hipMemcpyAsync(&n_u, B.n_runs.p, 4, hipMemcpyDeviceToHost, s));
    EV(4);
    HIP_TRY(hipStreamSynchronize(s));

    // ---- 6. LD rows ----
    if ((rc = ensure(B.rows_per, n_u)) || (rc = ensure(B.row_off, (size_t)n_u + 1))) return rc;
    HIP_TRY(hipMemsetAsync(B.n_runs.p + 1, 0, 4, s));
    SiteView v{B.sites_sorted.p, in.entries, in.counts, in.M == 1 ? 1 : 0};
    hipLaunchKernelGGL(k_ld_rows<false>, dim3((n_u + 255) / 256), dim3(256), 0, s, B.ukeys.p, B.ucnt.p, n_u, v,
                       in.min_snp, B.rows_per.p, nullptr, nullptr, B.n_runs.p + 1);
    tb = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tb, B.rows_per.p, B.row_off.p, 0u, n_u, rocprim::plus<uint32_t>(), s));
    if ((rc = ensure_temp(B.temp, tb))) return rc;
    tb = B.temp.cap;
    HIP_TRY(rocprim::exclusive_scan(B.temp.p, tb, B.rows_per.p, B.row_off.p, 0u, n_u, rocprim::plus<uint32_t>(), s));
    uint32_t lo2 = 0, lc2 = 0, n_edges = 0;
    HIP_TRY(hipMemcpyAsync(&lo2, B.row_off.p + (n_u - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&lc2, B.rows_per.p + (n_u - 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&n_edges, B.n_runs.p + 1, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t n_ld = lo2 + lc2;
    out.n_edges = n_edges;
    out.n_ld = n_ld;
    if (n_ld) {
        if ((rc = ensure(B.ld, n_ld))) return rc;
        hipLaunchKernelGGL(k_ld_rows<true>, dim3((n_u + 255) / 256), dim3(256), 0, s, B.ukeys.p, B.ucnt.p, n_u, v,
                           in.min_snp, nullptr, B.row_off.p, B.ld.p, nullptr);
    }
    EV(5);
    return ISX_OK;
}
