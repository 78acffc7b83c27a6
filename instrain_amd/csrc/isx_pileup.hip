// isx_pileup.hip -- k_pileup_call: per-window LDS pileup histogram + fused SNV-call epilogue.
//
// Replaces, for every position of a batch of splits at once, the reference's per-column loop
//   process_bam_sites        /root/reference/inStrain/profile/profile_utilities.py:218-266
//   get_base_counts_mm       profile_utilities.py:268-286
//   update_covT              profile_utilities.py:288-295
//   mm_counts_to_counts      profile_utilities.py:297-312
//   update_snp_table         /root/reference/inStrain/profile/snv_utilities.py:40-145
//   call_snv_site            snv_utilities.py:147-196
//   calc_snp_class           snv_utilities.py:198-223 (+ readComparer.py:307-316 is_present)
//   calculate_clonality      snv_utilities.py:225-231
//
// Design (gfx950): the flat position space is cut into windows of W positions; ONE workgroup
// owns a window exclusively, so its counters live in LDS (no global atomics, no inter-workgroup
// traffic) and the SNV-call epilogue runs straight out of LDS.  Observations arrive in BAM
// order, i.e. position-clustered, so the records that can touch a window form one contiguous
// range [lo, hi) of the stream (computed at upload from a per-1024-record min/max directory);
// the workgroup streams that range with 16-byte coalesced loads (2 records per lane per load,
// 4 loads in flight per lane) and drops records outside its window.  HBM-bound: 8 B per
// observation in, 20 B (dense, M==1) or 28 B per present (pos, mm) entry out.
//
// LDS layout: cnt[(mm*4 + base) * W + p]  (u32)  -> a wave touching consecutive positions of
// one read hits consecutive banks, and the epilogue (lane = position) reads conflict-free;
// then pres[k * W + p] bitmasks (mm path only) for levels made present by a non-ACGT base
// (profile_utilities.py:279-285 creates table[mm] before the KeyError).
#include "isx_internal.h"

#pragma clang fp contract(off)

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lut_min_bases(const uint8_t *lut, int lut_n, int fallback, uint32_t total)
{
    // snv_utilities.py:174-177 / readComparer.py:311-314
    if (total < (uint32_t)lut_n) {
        uint8_t v = lut[total];
        if (v != 255) return v;
    }
    return fallback;
}

__device__ __forceinline__ int argmax4(const uint32_t *c)
{
    int b = 0;
#pragma unroll
    for (int k = 1; k < 4; k++) if (c[k] > c[b]) b = k;
    return b;
}

// snv_utilities.py:147-196. returns -2 = None (uncounted), -1 = not a SNP, 0..3 = consensus base
__device__ __forceinline__ int call_snv_site(const uint32_t *c, uint32_t total, int ref_base, int min_bases,
                                             int min_cov, double min_freq, int &morphia)
{
    morphia = 0;
    if ((int64_t)total < (int64_t)min_cov) return -2;
    int i = 0;
    const double dt = (double)total;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if ((int)c[k] >= min_bases && (double)c[k] / dt >= min_freq) i++;
    morphia = i;
    const int am = argmax4(c);
    if (i > 1) return am;
    if (i == 1 && am != ref_base) return am;
    if (i == 0) return am;
    return -1;
}

// snv_utilities.py:225-231, fp64 in source order, no contraction
__device__ __forceinline__ double clonality(const uint32_t *c, uint32_t total)
{
    const double ds = (double)total;
    const double f0 = (double)c[0] / ds, f1 = (double)c[1] / ds, f2 = (double)c[2] / ds, f3 = (double)c[3] / ds;
    double prob = f0 * f0;
    prob = prob + f1 * f1;
    prob = prob + f2 * f2;
    prob = prob + f3 * f3;
    return prob;
}

// snv_utilities.py:198-223
__device__ __forceinline__ int snp_class(int con, int ref, int var, const uint32_t *c, uint32_t total, int morphia,
                                         int min_bases, double min_freq)
{
    if (ref > 3) return 0;
    if (morphia == 0) return 1;
    if (morphia == 1) return 2;
    if (ref == con) return 3;
    if (ref == var) return 4;
    if ((int)c[ref] >= min_bases && ((double)c[ref] / (double)total) >= min_freq) return 4;
    return 5;
}

// blockIdx -> window so that consecutive windows (which share boundary chunks of the stream)
// run on the same XCD (block b is dispatched to XCD b % 8; each XCD has a private L2).
__device__ __forceinline__ int xcd_window(int b, int nb)
{
    const int per = nb >> 3;            // grid is a multiple of 8
    return (b & 7) * per + (b >> 3);
}

template <bool MM>
__global__ void __launch_bounds__(512) k_pileup_call(const PileupArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int w = xcd_window(blockIdx.x, gridDim.x);
    if (w >= a.n_win) return;
    const int W = a.W, M = MM ? a.M : 1;
    const uint32_t w0 = (uint32_t)w << a.logW;
    const int n_cnt = M * 4 * W;
    const int pres_words = MM ? ((M + 31) >> 5) : 0;
    uint32_t *cnt = lds;
    uint32_t *pres = lds + n_cnt;
    uint32_t *scratch = pres + pres_words * W;      // [0] entry total, [1] entry base

    {   // zero the window
        uint4 *z = reinterpret_cast<uint4 *>(lds);
        const int n4 = (n_cnt + pres_words * W) >> 2;       // W is a multiple of 64
        for (int i = tid; i < n4; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
        if (tid < 4) scratch[tid] = 0;
    }
    __syncthreads();

    // ---- get_base_counts_mm over the window's slice of the stream ----
    const uint2 rng = a.win_range[w];
    const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(a.rec);
    const uint32_t lo = rng.x >> 1, hi = rng.y >> 1;
    uint32_t bad_mm = 0;
    auto visit = [&](uint32_t gpos, uint32_t attr) {
        const uint32_t rel = gpos - w0;
        if (rel < (uint32_t)W) {
            const uint32_t base = (attr >> 16) & 0xFFu;
            if (MM) {
                const uint32_t mm = attr & 0xFFFFu;
                if (mm >= (uint32_t)M) { bad_mm = 1; return; }
                if (base < 4) atomicAdd(&cnt[(mm * 4 + base) * W + rel], 1u);
                else atomicOr(&pres[(mm >> 5) * W + rel], 1u << (mm & 31));
            } else {
                if (base < 4) atomicAdd(&cnt[base * W + rel], 1u);
            }
        }
    };
    for (uint32_t i = lo + tid; i < hi; i += 4 * nthr) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = i + u * nthr;
            if (j < hi) v[u] = __builtin_nontemporal_load(&rec4[j]);
            else { v[u].x = ISX_SENTINEL; v[u].y = 0; v[u].z = ISX_SENTINEL; v[u].w = 0; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { visit(v[u].x, v[u].y); visit(v[u].z, v[u].w); }
    }
    if (bad_mm) atomicOr(a.flags, ISX_FLAG_MM_RANGE);
    __syncthreads();

    // ---- epilogue: one lane per position, straight out of LDS ----
    uint32_t e_off = 0;
    if (MM) {
        uint32_t my_e = 0;
        for (int p = tid; p < W; p += nthr) {
            if (w0 + p >= a.n_pos) break;
            for (int m = 0; m < M; m++) {
                const uint32_t any = cnt[(m * 4 + 0) * W + p] | cnt[(m * 4 + 1) * W + p] | cnt[(m * 4 + 2) * W + p] |
                                     cnt[(m * 4 + 3) * W + p] | ((pres[(m >> 5) * W + p] >> (m & 31)) & 1u);
                my_e += any ? 1u : 0u;
            }
        }
        const uint32_t my_off = atomicAdd(&scratch[0], my_e);
        __syncthreads();
        if (tid == 0) scratch[1] = atomicAdd(&a.cursors[CUR_ENTRIES], scratch[0]);
        __syncthreads();
        e_off = scratch[1] + my_off;
        if (scratch[1] + scratch[0] > a.cap_entries) {
            if (tid == 0) atomicOr(a.flags, ISX_FLAG_CAP_ENTRIES);
            return;
        }
    }

    for (int p = tid; p < W; p += nthr) {
        const uint32_t gpos = w0 + p;
        if (gpos >= a.n_pos) break;
        const int ref_base = a.ref[gpos];
        uint32_t cum[4] = {0, 0, 0, 0};
        int anySNP = 0, cryptic = 0, nrows = 0, nlev = 0;
        uint32_t mask = 0;
        const uint32_t first_entry = e_off;
        float clon_last = __builtin_nanf("");

        // one pass of update_snp_table's `for mm in sorted(MMcounts)`; EMIT writes SNV rows
        auto levels = [&](bool emit, uint32_t row_base) {
            cum[0] = cum[1] = cum[2] = cum[3] = 0;
            int any = 0, cry = 0, rows = 0;
            for (int m = 0; m < M; m++) {
                uint32_t l[4];
#pragma unroll
                for (int k = 0; k < 4; k++) l[k] = cnt[(m * 4 + k) * W + p];
                uint32_t present = l[0] | l[1] | l[2] | l[3];
                if (MM) present |= (pres[(m >> 5) * W + p] >> (m & 31)) & 1u;
                if (!present) continue;
#pragma unroll
                for (int k = 0; k < 4; k++) cum[k] += l[k];
                const uint32_t total = cum[0] + cum[1] + cum[2] + cum[3];
                const int min_bases = ((int64_t)total >= (int64_t)a.min_cov) ? lut_min_bases(a.lut, a.lut_n, a.fallback, total) : 0;
                int morphia;
                const int snp = call_snv_site(cum, total, ref_base, min_bases, a.min_cov, a.min_freq, morphia);
                if (!emit) {
                    float cl = __builtin_nanf("");
                    if ((int64_t)total >= (int64_t)a.min_cov) cl = (float)clonality(cum, total);
                    clon_last = cl;
                    if (MM) {
                        isx_entry e;
                        e.gpos = gpos; e.mm = (uint16_t)m; e.flags = 0;
                        e.cnt[0] = l[0]; e.cnt[1] = l[1]; e.cnt[2] = l[2]; e.cnt[3] = l[3];
                        e.clon = cl;
                        a.entries[e_off] = e;
                        e_off++;
                        nlev++;
                    }
                }
                if (snp == -2) continue;
                if (snp != -1) {
                    uint32_t tmp[4] = {cum[0], cum[1], cum[2], cum[3]};
                    tmp[snp] = 0;
                    const int var = argmax4(tmp);
                    if (emit) {
                        isx_snv r;
                        r.gpos = gpos; r.mm = (uint16_t)m;
                        r.con_base = (uint8_t)snp; r.var_base = (uint8_t)var;
                        r.allele_count = (uint8_t)morphia;
                        r.cls = (uint8_t)snp_class(snp, ref_base, var, cum, total, morphia, min_bases, a.min_freq);
                        r.cryptic = (uint8_t)cryptic;       // position-level flag from the first pass (p2c map)
                        r.ref_base = (uint8_t)ref_base;
                        r.cnt[0] = cum[0]; r.cnt[1] = cum[1]; r.cnt[2] = cum[2]; r.cnt[3] = cum[3];
                        a.snv[row_base + rows] = r;
                    }
                    rows++;
                    if (morphia >= 2) { any = 1; mask |= (1u << snp) | (1u << var); }
                    else if (morphia == 1 && any) cry = 1;
                } else if (any) {
                    cry = 1;
                }
            }
            anySNP = any; cryptic = cry; nrows = rows;
        };

        levels(false, 0);
        if (!MM) {
            a.counts[gpos] = make_uint4(cum[0], cum[1], cum[2], cum[3]);
            a.clon[gpos] = clon_last;
        }
        a.site_mask[gpos] = anySNP ? (uint8_t)mask : (uint8_t)0;
        if (nrows) {
            const uint32_t row_base = atomicAdd(&a.cursors[CUR_SNV], (uint32_t)nrows);
            if (row_base + nrows > a.cap_snv) atomicOr(a.flags, ISX_FLAG_CAP_SNV);
            else levels(true, row_base);
        }
        if (anySNP) {
            const uint32_t s = atomicAdd(&a.cursors[CUR_SITES], 1u);
            if (s >= a.cap_sites) atomicOr(a.flags, ISX_FLAG_CAP_SITES);
            else {
                isx_site st;
                st.gpos = gpos; st.entry_off = first_entry; st.n_levels = (uint16_t)nlev;
                st.mask = (uint8_t)mask; st.pad = 0;
                a.sites[s] = st;
            }
        }
    }
}

}  // namespace

size_t pileup_lds_bytes(int W, int M)
{
    const size_t pres_words = M > 1 ? (size_t)((M + 31) / 32) : 0;
    return ((size_t)M * 4 * W + pres_words * W + 4) * sizeof(uint32_t);
}

void launch_pileup(const PileupArgs &a, int block, size_t lds, hipStream_t s)
{
    const int grid = ((a.n_win + 7) / 8) * 8;
    if (a.M > 1) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pileup_call<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pileup_call<true>, dim3(grid), dim3(block), lds, s, a);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pileup_call<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pileup_call<false>, dim3(grid), dim3(block), lds, s, a);
    }
}
