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

#define THR_LDS 1024        // coverages below this read their folded threshold from LDS

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
__global__ void __launch_bounds__(1024) k_pileup_call(const PileupArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int w = xcd_window(blockIdx.x, gridDim.x);
    if (w >= a.n_win) return;
    const int W = a.W, M = MM ? a.M : 1;
    const uint32_t w0 = (uint32_t)w * (uint32_t)a.W;
    const int n_cnt = M * 4 * W;
    const int pres_words = MM ? ((M + 31) >> 5) : 0;
    uint32_t *cnt = lds;
    uint32_t *pres = lds + n_cnt;
    uint32_t *scratch = pres + pres_words * W;      // [0] entry total, [1] entry base, [2] queue length, [4..] queue
    uint16_t *thr_lds = reinterpret_cast<uint16_t *>(scratch + 4 + 2 * a.qcap);

    {   // zero the window
        uint4 *z = reinterpret_cast<uint4 *>(lds);
        const int n4 = (n_cnt + pres_words * W) >> 2;       // W is a multiple of 64
        for (int i = tid; i < n4; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
        if (tid < 4) scratch[tid] = 0;
    }
    // Issued now, consumed only in the epilogue (the waits land after the streaming loop): this
    // lane's slice of the folded-threshold table and the reference bases of its positions, so the
    // epilogue has no dependent global load on its common path.
    uint32_t thr_stage[4];
    const int n_thr32 = min(THR_LDS, a.lut_n & ~1) >> 1;
    {
        const uint32_t *t32 = reinterpret_cast<const uint32_t *>(a.thr);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int i = tid + it * nthr;
            thr_stage[it] = (i < n_thr32) ? t32[i] : 0u;
        }
    }
    uint8_t ref_raw[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const uint32_t gp = w0 + tid + it * nthr;
        ref_raw[it] = (tid + it * nthr < W && gp < a.n_pos) ? a.ref[gp] : (uint8_t)4;
    }
    __syncthreads();

    // ---- get_base_counts_mm over the window's slice of the stream ----
    const uint2 rng = a.win_range[w];
    const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(a.rec);
    const uint32_t lo = rng.x >> 1, hi = rng.y >> 1;
    uint32_t bad_mm = 0;
    const int dbg = a.debug_mode;                   // ablation switches (tools/tune_pileup.py), 0 in production
    uint32_t sink = 0;
    auto visit = [&](uint32_t gpos, uint32_t attr) {
        const uint32_t rel = gpos - w0;
        if (dbg & 1) { sink ^= gpos + attr; return; }
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
    for (uint32_t i = lo + tid; i < ((dbg & 4) ? lo : hi); i += 4 * nthr) {
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
    if ((dbg & 1) && sink == 0x12345678u) atomicOr(a.flags + 1, 1u);     // keep the loads alive
    {
        uint32_t *t32 = reinterpret_cast<uint32_t *>(thr_lds);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int i = tid + it * nthr;
            if (i < n_thr32) t32[i] = thr_stage[it];
        }
    }
    __syncthreads();
    if (dbg & 2) return;

    // ---- epilogue: one lane per position, straight out of LDS ----
    // Integer-only on the common path: for coverage < lut_n the two per-base tests of
    // call_snv_site (c >= null_model[total] and float(c)/total >= min_freq, snv_utilities.py:179)
    // are folded on the host into ONE exact threshold thr[total] (isx_api.hip build_thresholds);
    // clonality is exactly 1.0 when a single base is present, otherwise the (pos, level) is queued
    // in LDS and the fp64 divisions run densely packed afterwards (no divergent lanes idling).
    uint32_t *qn = scratch + 2;                          // queue length
    uint32_t *queue = scratch + 4;                       // [QCAP][2]: target index, (mm << 16) | p
    const uint32_t QCAP = (uint32_t)a.qcap;
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

    int ep_it = 0;
    for (int p = tid; p < W; p += nthr, ep_it++) {
        const uint32_t gpos = w0 + p;
        if (gpos >= a.n_pos) break;
        int ref_base;
        if (ep_it == 0) ref_base = ref_raw[0];
        else if (ep_it == 1) ref_base = ref_raw[1];
        else if (ep_it == 2) ref_base = ref_raw[2];
        else if (ep_it == 3) ref_base = ref_raw[3];
        else ref_base = a.ref[gpos];
        uint32_t cum[4] = {0, 0, 0, 0};
        int anySNP = 0, cryptic = 0, nrows = 0, nlev = 0;
        uint32_t mask = 0;
        const uint32_t first_entry = e_off;
        float clon_last = __builtin_nanf("");
        bool clon_deferred = false;

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
                const bool counted = (int64_t)total >= (int64_t)a.min_cov;
                const bool fast = total < (uint32_t)a.lut_n;           // folded threshold available
                int morphia = 0, snp = -2, min_bases = 0;
                uint32_t thr = 0;
                if (counted) {
                    if (fast) {
                        thr = (total < THR_LDS && nthr >= 128) ? (uint32_t)thr_lds[total] : (uint32_t)a.thr[total];
#pragma unroll
                        for (int k = 0; k < 4; k++) morphia += (cum[k] >= thr) ? 1 : 0;
                        const int am = argmax4(cum);
                        snp = (morphia > 1) ? am : (morphia == 1 ? (am != ref_base ? am : -1) : am);
                    } else {                                           // coverage >= lut_n: the reference's own arithmetic
                        min_bases = a.fallback;
                        snp = call_snv_site(cum, total, ref_base, min_bases, a.min_cov, a.min_freq, morphia);
                    }
                }
                if (!emit) {
                    float cl = __builtin_nanf("");
                    bool defer = false;
                    if (counted) {
                        const uint32_t mx = max(max(cum[0], cum[1]), max(cum[2], cum[3]));
                        if (mx == total) cl = 1.0f;                    // (s/s)^2 + 0 + 0 + 0
                        else {
                            const uint32_t slot = atomicAdd(qn, 1u);
                            if (slot < QCAP) {
                                defer = true;
                                queue[slot * 2 + 0] = MM ? e_off : gpos;
                                queue[slot * 2 + 1] = ((uint32_t)m << 16) | (uint32_t)p;
                            } else {
                                cl = (float)clonality(cum, total);     // queue full: inline
                            }
                        }
                    }
                    clon_last = cl;
                    clon_deferred = defer;
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
#pragma unroll
                    for (int k = 0; k < 4; k++) tmp[k] = (k == snp) ? 0u : tmp[k];
                    const int var = argmax4(tmp);
                    if (emit) {
                        int cls;
                        if (fast) {                                    // calc_snp_class with is_present folded into thr
                            uint32_t cref = 0;
#pragma unroll
                            for (int k = 0; k < 4; k++) cref = (k == ref_base) ? cum[k] : cref;
                            if (ref_base > 3) cls = 0;
                            else if (morphia == 0) cls = 1;
                            else if (morphia == 1) cls = 2;
                            else if (ref_base == snp) cls = 3;
                            else if (ref_base == var) cls = 4;
                            else cls = (cref >= thr) ? 4 : 5;
                        } else {
                            cls = snp_class(snp, ref_base, var, cum, total, morphia, min_bases, a.min_freq);
                        }
                        isx_snv r;
                        r.gpos = gpos; r.mm = (uint16_t)m;
                        r.con_base = (uint8_t)snp; r.var_base = (uint8_t)var;
                        r.allele_count = (uint8_t)morphia;
                        r.cls = (uint8_t)cls;
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

        if (!(dbg & 16)) levels(false, 0);
        if (!MM && !(dbg & 8)) {
            a.counts[gpos] = make_uint4(cum[0], cum[1], cum[2], cum[3]);
            if (!clon_deferred) a.clon[gpos] = clon_last;
        }
        if (!(dbg & 8) && !(dbg & 32)) a.site_mask[gpos] = anySNP ? (uint8_t)mask : (uint8_t)0;
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
    __syncthreads();
    // ---- deferred clonalities: calculate_clonality (snv_utilities.py:225-231) in fp64, densely packed ----
    const uint32_t nq = min(*qn, QCAP);
    for (uint32_t q = tid; q < nq; q += nthr) {
        const uint32_t pm = queue[q * 2 + 1];
        const int p = (int)(pm & 0xFFFFu), mq = (int)(pm >> 16);
        uint32_t c[4] = {0, 0, 0, 0};
        for (int m = 0; m <= mq; m++) {                 // mm_counts_to_counts(MMcounts, mm)
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] += cnt[(m * 4 + k) * W + p];
        }
        const float cl = (float)clonality(c, c[0] + c[1] + c[2] + c[3]);
        if (MM) a.entries[queue[q * 2]].clon = cl;
        else a.clon[queue[q * 2]] = cl;
    }
}


// ---------------------------------------------------------------------------------------------
// k_pileup_dense: the n_mm_bins == 1 (--skip_mm_profiling / --database_mode) specialisation.
// Persistent workgroups: each walks windows slot, slot + grid, ... so the per-window fixed costs
// (launch, threshold staging, dependent global latencies of the epilogue) are paid behind the
// NEXT window's first loads, which are issued before the epilogue starts.  SNV rows / SNP sites
// are allocated with ONE global atomic per window (LDS-aggregated), clonality divisions and row
// emission run densely packed from an LDS queue.
// LDS: cnt[4][W] | queue[W] | thr_lds[THR_LDS] | scratch[8]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024, 8) k_pileup_dense(const PileupArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int W = a.W;
    uint32_t *cnt = lds;
    uint32_t *queue = lds + 4 * W;
    uint32_t *scratch = queue + W;              // [0] queue length [1] rows [2] sites [3] row base [4] site base
    uint16_t *thr_lds = reinterpret_cast<uint16_t *>(scratch + 8);
    const int grid = gridDim.x, per = grid >> 3;
    const int slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // consecutive windows share an XCD's L2
    const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(a.rec);
    const int dbg = a.debug_mode;

    {   // once per workgroup: folded thresholds of the low coverages
        const int n = min(THR_LDS, a.lut_n);
        for (int i = tid; i < n; i += nthr) thr_lds[i] = a.thr[i];
    }

    u32x4 v[4];
    uint32_t lo = 0, hi = 0;
    auto issue = [&](uint32_t i0) {             // 4 coalesced 16-byte loads per lane (2 records each)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = i0 + tid + u * nthr;
            if (j < hi) v[u] = __builtin_nontemporal_load(&rec4[j]);
            else { v[u].x = ISX_SENTINEL; v[u].y = 0; v[u].z = ISX_SENTINEL; v[u].w = 0; }
        }
    };
    if (slot < a.n_win) {
        const uint2 rng = a.win_range[slot];
        lo = rng.x >> 1; hi = rng.y >> 1;
        if (lo < hi) issue(lo);
    }

    for (int w = slot; w < a.n_win; w += grid) {
        const uint32_t w0 = (uint32_t)w * (uint32_t)W;
        {   // zero the window's counters
            uint4 *z = reinterpret_cast<uint4 *>(cnt);
            for (int i = tid; i < W; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
            if (tid < 8) scratch[tid] = 0;
        }
        uint8_t ref_raw[2];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const uint32_t gp = w0 + tid + it * nthr;
            ref_raw[it] = (tid + it * nthr < W && gp < a.n_pos) ? a.ref[gp] : (uint8_t)4;
        }
        __syncthreads();

        // ---- get_base_counts_mm (profile_utilities.py:268-286) over the window's slice ----
        for (uint32_t i0 = lo; i0 < hi; i0 += 4 * nthr) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t g0 = v[u].x, a0 = v[u].y, g1 = v[u].z, a1 = v[u].w;
                const uint32_t r0 = g0 - w0, r1 = g1 - w0;
                const uint32_t b0 = (a0 >> 16) & 0xFFu, b1 = (a1 >> 16) & 0xFFu;
                if (r0 < (uint32_t)W && b0 < 4) atomicAdd(&cnt[b0 * W + r0], 1u);
                if (r1 < (uint32_t)W && b1 < 4) atomicAdd(&cnt[b1 * W + r1], 1u);
            }
            const uint32_t nxt = i0 + 4 * nthr;
            if (nxt < hi) issue(nxt);
        }
        __syncthreads();

        // ---- first loads of the NEXT window go out before the epilogue ----
        {
            const int wn = w + grid;
            lo = hi = 0;
            if (wn < a.n_win) {
                const uint2 rng = a.win_range[wn];
                lo = rng.x >> 1; hi = rng.y >> 1;
                if (lo < hi) issue(lo);
            }
        }

        // ---- epilogue pass 1: integer only (update_snp_table / call_snv_site, single mm level) ----
        int ep_it = 0;
        for (int p = tid; p < ((dbg & 2) ? 0 : W); p += nthr, ep_it++) {
            const uint32_t gpos = w0 + p;
            if (gpos >= a.n_pos) break;
            const uint32_t c[4] = {cnt[p], cnt[W + p], cnt[2 * W + p], cnt[3 * W + p]};
            const uint32_t total = c[0] + c[1] + c[2] + c[3];
            a.counts[gpos] = make_uint4(c[0], c[1], c[2], c[3]);
            uint32_t mask = 0;
            float cl = __builtin_nanf("");
            bool defer = false;
            if ((int64_t)total >= (int64_t)a.min_cov) {
                const int ref_base = ep_it == 0 ? ref_raw[0] : (ep_it == 1 ? ref_raw[1] : a.ref[gpos]);
                int morphia = 0, snp;
                if (total < (uint32_t)a.lut_n) {
                    const uint32_t thr = total < THR_LDS ? (uint32_t)thr_lds[total] : (uint32_t)a.thr[total];
#pragma unroll
                    for (int k = 0; k < 4; k++) morphia += (c[k] >= thr) ? 1 : 0;
                    const int am = argmax4(c);
                    snp = (morphia > 1) ? am : (morphia == 1 ? (am != ref_base ? am : -1) : am);
                } else {
                    snp = call_snv_site(c, total, ref_base, a.fallback, a.min_cov, a.min_freq, morphia);
                }
                const uint32_t mx = max(max(c[0], c[1]), max(c[2], c[3]));
                if (mx == total) cl = 1.0f; else defer = true;
                uint32_t entry = (uint32_t)p;
                if (defer) entry |= 1u << 13;
                if (snp != -1) {
                    entry |= 1u << 14;
                    atomicAdd(&scratch[1], 1u);
                    if (morphia >= 2) {
                        uint32_t tmp[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
                        for (int k = 0; k < 4; k++) tmp[k] = (k == snp) ? 0u : tmp[k];
                        mask = (1u << snp) | (1u << argmax4(tmp));
                        entry |= (atomicAdd(&scratch[2], 1u) + 1u) << 16;
                    }
                }
                if (entry != (uint32_t)p) queue[atomicAdd(&scratch[0], 1u)] = entry;
            }
            if (!defer) a.clon[gpos] = cl;
            a.site_mask[gpos] = (uint8_t)mask;
        }
        __syncthreads();
        const uint32_t nq = scratch[0], nrows = scratch[1], nsites = scratch[2];
        if (tid == 0 && nrows) scratch[3] = atomicAdd(&a.cursors[CUR_SNV], nrows);
        if (tid == 32 && nsites) scratch[4] = atomicAdd(&a.cursors[CUR_SITES], nsites);
        // ---- deferred clonalities (snv_utilities.py:225-231), densely packed ----
        for (uint32_t q = tid; q < nq; q += nthr) {
            const uint32_t e = queue[q];
            if (!(e & (1u << 13))) continue;
            const int p = (int)(e & 0x1FFFu);
            const uint32_t c[4] = {cnt[p], cnt[W + p], cnt[2 * W + p], cnt[3 * W + p]};
            a.clon[w0 + p] = (float)clonality(c, c[0] + c[1] + c[2] + c[3]);
        }
        if (nrows) __syncthreads();             // uniform: scratch[3..4] from the two atomics above
        // ---- SNV rows / SNP sites (snv_utilities.py:107-133) ----
        const uint32_t row_base = scratch[3], site_base = scratch[4];
        bool emit_rows = nrows != 0;
        if (emit_rows && (row_base + nrows > a.cap_snv || site_base + nsites > a.cap_sites)) {
            if (tid == 0) atomicOr(a.flags, row_base + nrows > a.cap_snv ? ISX_FLAG_CAP_SNV : ISX_FLAG_CAP_SITES);
            emit_rows = false;
        }
        uint32_t my_row = 0;                    // rank among the row entries (LDS counter)
        for (uint32_t q0 = 0; q0 < (emit_rows ? nq : 0u); q0 += nthr) {
            const uint32_t q = q0 + tid;
            const uint32_t e = q < nq ? queue[q] : 0u;
            const bool is_row = (e >> 14) & 1u;
            if (is_row) my_row = atomicAdd(&scratch[5], 1u);
            if (!is_row) continue;
            const int p = (int)(e & 0x1FFFu);
            const uint32_t gpos = w0 + p;
            const uint32_t c[4] = {cnt[p], cnt[W + p], cnt[2 * W + p], cnt[3 * W + p]};
            const uint32_t total = c[0] + c[1] + c[2] + c[3];
            const int ref_base = a.ref[gpos];
            int morphia = 0, snp, cls;
            const bool fast = total < (uint32_t)a.lut_n;
            uint32_t thr = 0;
            if (fast) {
                thr = a.thr[total];
#pragma unroll
                for (int k = 0; k < 4; k++) morphia += (c[k] >= thr) ? 1 : 0;
                const int am = argmax4(c);
                snp = (morphia > 1) ? am : (morphia == 1 ? (am != ref_base ? am : -1) : am);
            } else {
                snp = call_snv_site(c, total, ref_base, a.fallback, a.min_cov, a.min_freq, morphia);
            }
            uint32_t tmp[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int k = 0; k < 4; k++) tmp[k] = (k == snp) ? 0u : tmp[k];
            const int var = argmax4(tmp);
            if (fast) {
                uint32_t cref = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) cref = (k == ref_base) ? c[k] : cref;
                if (ref_base > 3) cls = 0;
                else if (morphia == 0) cls = 1;
                else if (morphia == 1) cls = 2;
                else if (ref_base == snp) cls = 3;
                else if (ref_base == var) cls = 4;
                else cls = (cref >= thr) ? 4 : 5;
            } else {
                cls = snp_class(snp, ref_base, var, c, total, morphia, a.fallback, a.min_freq);
            }
            isx_snv r;
            r.gpos = gpos; r.mm = 0;
            r.con_base = (uint8_t)snp; r.var_base = (uint8_t)var;
            r.allele_count = (uint8_t)morphia; r.cls = (uint8_t)cls;
            r.cryptic = 0;                      // a single mm level cannot turn cryptic (snv_utilities.py:135-140)
            r.ref_base = (uint8_t)ref_base;
            r.cnt[0] = c[0]; r.cnt[1] = c[1]; r.cnt[2] = c[2]; r.cnt[3] = c[3];
            a.snv[row_base + my_row] = r;
            const uint32_t ss = e >> 16;
            if (ss) {
                isx_site st;
                st.gpos = gpos; st.entry_off = 0; st.n_levels = 1;
                st.mask = (uint8_t)((1u << snp) | (1u << var)); st.pad = 0;
                a.sites[site_base + ss - 1] = st;
            }
        }
        // the zeroing + barrier at the top of the next window protect cnt / queue / scratch
        __syncthreads();
    }
}

}  // namespace

size_t pileup_lds_bytes(int W, int M, int qcap)
{
    const size_t pres_words = M > 1 ? (size_t)((M + 31) / 32) : 0;
    return ((size_t)M * 4 * W + pres_words * W + 4 + (size_t)qcap * 2) * sizeof(uint32_t) + THR_LDS * sizeof(uint16_t);
}

size_t pileup_dense_lds_bytes(int W)
{
    return ((size_t)5 * W + 8) * sizeof(uint32_t) + THR_LDS * sizeof(uint16_t);
}

void launch_pileup(const PileupArgs &a, int block, size_t lds, int grid_dense, hipStream_t s)
{
    if (a.M > 1) {
        const int grid = ((a.n_win + 7) / 8) * 8;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pileup_call<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pileup_call<true>, dim3(grid), dim3(block), lds, s, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pileup_dense),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_pileup_dense, dim3(grid_dense), dim3(block), lds, s, a);
    }
}
