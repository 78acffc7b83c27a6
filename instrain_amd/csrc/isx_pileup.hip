// isx_pileup.hip -- per-window LDS pileup histogram + fused SNV-call epilogue + fused
// allele-observation pass (the producer side of linkage).
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
//   update_linked_reads      /root/reference/inStrain/profile/linkage.py:254-283
//
// Design (gfx950): the flat position space is cut into windows of W positions; ONE workgroup
// owns a window exclusively, so its counters live in LDS (no global atomics on the data path,
// no inter-workgroup traffic) and the SNV-call epilogue runs straight out of LDS.  Observations
// arrive in BAM order, i.e. position-clustered, so the records that can touch a window form one
// contiguous range [lo, hi) of the stream (computed at upload from a per-1024-record min/max
// directory); the workgroup streams that range with 16-byte coalesced loads (2 records per
// lane per load, 4 loads in flight per lane) and drops records outside its window.
// HBM-bound: 8 B per observation in, 20 B (dense, M==1) or 28 B per present (pos, mm) entry out.
//
// LDS layout: cnt[(mm*4 + base) * W + p] (u32) -> a wave touching consecutive positions of one
// read hits consecutive banks, and the epilogue (lane = position) reads conflict-free.
//
// Integer-only epilogue on the common path: for coverage < lut_n the two per-base tests of
// call_snv_site (c >= null_model[total] and float(c)/total >= min_freq, snv_utilities.py:179) are
// folded on the host into ONE exact threshold thr[total] (isx_api.hip build_thresholds);
// clonality is exactly 1.0 when a single base is present, otherwise the (pos, level) is queued in
// LDS and the fp64 divisions run densely packed afterwards (no divergent lanes idling).
//
// Linkage producer: a SNP site's `bases` set and its exact number of qualifying observations
// (sum of its counts over the set) are known in the epilogue, so every site gets an exactly
// sized slab of the allele-observation table (one global atomic per WINDOW); the workgroup then
// re-streams its record range and drops each qualifying observation into its site's slab
// through a per-position LDS cursor.
#include <hip/hip_ext.h>
#include "isx_internal.h"

#pragma clang fp contract(off)

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define THR_LDS 1024        // coverages below this read their folded threshold from LDS

// scratch words (LDS)
enum { S_NQ = 0, S_ROWS, S_SITES, S_ROW_BASE, S_SITE_BASE, S_ROW_RANK, S_NAO, S_AO_BASE, S_ENT_TOT, S_ENT_BASE, S_SLEV, S_SLEV_BASE, S_NRARE, S_RARE_BASE, S_RARE_RANK,
       S_NCLON, S_CLON_BASE, S_CLON_RANK, S_COVX, S_COVX_BASE, S_N = 20 };
#define S_RNG 128           // k_pileup_dense: behind the scratch words, the record ranges of the workgroup's next 64 windows (64 x uint2)

// table cursors run on across launches; a run's slots are relative to the values it started from
__device__ __forceinline__ uint32_t cur_add(const PileupArgs &a, int which, uint32_t n)
{
    return atomicAdd(&a.cursors[which], n) - a.base[which];
}

// error flags: the returned value is consumed so the atomic is complete before the workgroup's ticket
__device__ __forceinline__ void flag_or(const PileupArgs &a, uint32_t bit)
{
    if (atomicOr(a.flags, bit) == 0xFFFFFFFFu) a.flags[3] = 1;
}

// After the pileup kernel a one-wave kernel copies cursors | flags to mapped pinned host memory: the
// host then needs no copy, only the stream synchronisation it does anyway.  (A last-workgroup ticket
// inside the pileup kernel was tried: 512 returning atomics on one word at the kernel's tail cost
// more than this extra kernel boundary.)
__global__ void k_publish_state(const uint32_t *cursors, uint32_t *host_state, uint32_t epoch)
{
    const int i = threadIdx.x;
    if (i < CUR_N + 4) host_state[i] = __hip_atomic_load(&cursors[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    // the host spins on this word (coherent pinned memory) instead of sleeping in a stream wait
    if (i == 0) __hip_atomic_store(&host_state[CUR_N + 4], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the same publication from inside a pileup kernel (first wave of workgroup 0): the state of the pass that
// ran before this kernel on the stream, complete at this point by stream order
__device__ __forceinline__ void publish_previous(const PileupArgs &a, int tid)
{
    if (a.pub_cursors == nullptr || blockIdx.x != 0 || tid >= 64) return;
    if (tid < CUR_N + 4) a.pub_host_state[tid] = __hip_atomic_load(&a.pub_cursors[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (tid == 0) __hip_atomic_store(&a.pub_host_state[CUR_N + 4], a.pub_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// reference base code of a flat position (see PileupArgs::ref_packed)
// Inclusive prefix sum over the 64 lanes of a wave, in the VALU: four row-shift DPP adds inside every row of 16 lanes, then the last
// lane of rows 0 / 2 is broadcast into rows 1 / 3 and lane 31 into the upper half (row_bcast:15 / :31, gfx9).  A lane whose DPP source
// does not exist, or whose row the row mask excludes, adds the `old` operand: 0.  (The same scan through __shfl_up is six dependent
// ds_bpermute round trips through the LDS crossbar.)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);      // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ uint8_t ref_at(const PileupArgs &a, uint32_t gpos)
{
    if (a.ref_packed == 2) {
        if (a.ref_n && ((a.ref_n[gpos >> 3] >> (gpos & 7u)) & 1u)) return 4;
        return (uint8_t)((a.ref[gpos >> 2] >> ((gpos & 3u) << 1)) & 3u);
    }
    if (a.ref_packed) return (uint8_t)((a.ref[gpos >> 1] >> ((gpos & 1u) << 2)) & 0xFu);
    return a.ref[gpos];
}

__device__ __forceinline__ int argmax4(const uint32_t *c)
{
    int b = 0;
#pragma unroll
    for (int k = 1; k < 4; k++) if (c[k] > c[b]) b = k;
    return b;
}

// snv_utilities.py:147-196 with the reference's own fp64 arithmetic (coverage >= lut_n only).
// returns -2 = None (uncounted), -1 = not a SNP, 0..3 = consensus base
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

// snv_utilities.py:198-223 (reference arithmetic; coverage >= lut_n only)
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

// One mm level of update_snp_table on cumulative counts `c`.
struct SiteCall { int snp, morphia, var, cls; };

__device__ __forceinline__ SiteCall call_level(const PileupArgs &a, const uint16_t *thr_lds, const uint32_t *c,
                                               uint32_t total, int ref_base, bool want_class)
{
    SiteCall r{-2, 0, 0, 0};
    if ((int64_t)total < (int64_t)a.min_cov) return r;
    const bool fast = total < (uint32_t)a.lut_n;
    uint32_t thr = 0;
    if (fast) {
        if (thr_lds) {          // two typed loads (a select of the LDS and the global pointer would be ONE flat load, which also counts as an LDS access)
            thr = thr_lds[min(total, (uint32_t)THR_LDS - 1u)];
            asm volatile("" : "+v"(thr));
            if (total >= THR_LDS) thr = a.thr[total];
        } else thr = a.thr[total];
#pragma unroll
        for (int k = 0; k < 4; k++) r.morphia += (c[k] >= thr) ? 1 : 0;
        const int am = argmax4(c);
        r.snp = (r.morphia > 1) ? am : (r.morphia == 1 ? (am != ref_base ? am : -1) : am);
    } else {
        r.snp = call_snv_site(c, total, ref_base, a.fallback, a.min_cov, a.min_freq, r.morphia);
    }
    if (r.snp >= 0) {
        uint32_t tmp[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
        for (int k = 0; k < 4; k++) tmp[k] = (k == r.snp) ? 0u : tmp[k];
        r.var = argmax4(tmp);               // list.index(max): first maximum (snv_utilities.py:110-112)
        if (want_class) {
            if (fast) {                     // calc_snp_class with is_present folded into thr
                uint32_t cref = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) cref = (k == ref_base) ? c[k] : cref;
                if (ref_base > 3) r.cls = 0;
                else if (r.morphia == 0) r.cls = 1;
                else if (r.morphia == 1) r.cls = 2;
                else if (ref_base == r.snp) r.cls = 3;
                else if (ref_base == r.var) r.cls = 4;
                else r.cls = (cref >= thr) ? 4 : 5;
            } else {
                r.cls = snp_class(r.snp, ref_base, r.var, c, total, r.morphia, a.fallback, a.min_freq);
            }
        }
    }
    return r;
}

// calculate_rarefied_clonality (snv_utilities.py:233-247) with a counter-based generator
__device__ __forceinline__ float rarefied_clonality(const PileupArgs &a, const uint32_t *c, uint32_t gpos, uint32_t mm)
{
    const double s = (double)(c[0] + c[1] + c[2] + c[3]);
    const double p[4] = {(double)c[0] / s, (double)c[1] / s, (double)c[2] / s, (double)c[3] / s};
    uint32_t rc[4];
    const Philox ph{a.seed_lo, a.seed_hi};
    rarefy4(ph, gpos, mm, 0x434C4F4Eu /* 'CLON' */, p, a.min_cov_r, rc);
    return (float)clonality(rc, rc[0] + rc[1] + rc[2] + rc[3]);
}

__device__ __forceinline__ uint32_t masked_sum(const uint32_t *c, uint32_t mask)
{
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += ((mask >> k) & 1u) ? c[k] : 0u;
    return s;
}

// update_linked_reads (linkage.py:254-283): an observation at a SNP site whose base is in the site's
// `bases` set goes to the next free slot of the site's slab.  The window's records are streamed a
// second time, but only their positions (a.gpos, 4 of the 8 bytes; the full record and the pair id are
// fetched for the ~1 % of records that sit on a site).  Those candidates are rare but nearly every
// wave-wide step has one, and a candidate costs dependent global loads + a scattered 16-byte store:
// handled in place, every step of every wave would wait on that latency with one lane alive.  So they
// are compacted (ballot + mbcnt) into a per-wave LDS stage of 64 entries and drained with all lanes
// busy.  `stage` = 128 words per wave of LDS that is dead during this pass (the window's counters; the
// caller has a barrier before).
__device__ __forceinline__ void allele_drain(const PileupArgs &a, const uint32_t *st, uint32_t n, uint32_t w0,
                                             const uint8_t *maskl, uint32_t *slabc, uint32_t ao_base, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((uint32_t)lane < n) {
        const uint32_t i = st[2 * lane], rel = st[2 * lane + 1];
        uint32_t base, mm;
        if (a.rec16) { base = (uint32_t)a.rec16[i] >> 13; mm = 0; }
        else if (a.rec32) { const uint32_t x = a.rec32[i]; base = (x >> 24) & 7u; mm = (x >> 16) & 0xFFu; }
        else { const uint32_t at = a.rec[i].y; base = (at >> 16) & 0xFFu; mm = at & 0xFFFFu; }
        if (base < 4 && ((maskl[rel] >> base) & 1u)) {
            const uint32_t slot = atomicAdd(&slabc[rel], 1u);
            isx_ao o;
            if (a.pair) o.pair = a.pair[i];
            else {                              // run table: a read's records are consecutive, a chunk holds a handful of runs
                uint32_t r = a.run_index[i >> 10];
                while (r + 1 < a.n_runs && a.pair_runs[r + 1].x <= i) r++;
                o.pair = a.pair_runs[r].y;
            }
            o.site = w0 + rel; o.obs_idx = i;
            o.mm = (uint16_t)mm; o.base = (uint8_t)base; o.pad = 0;
            a.ao[ao_base + slot] = o;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// lo, hi: the window's record range in units of TWO records (as the counting pass uses them); multiples of 512
__device__ __forceinline__ void allele_pass(const PileupArgs &a, uint32_t lo, uint32_t hi,
                                            uint32_t w0, int W, const uint8_t *maskl, uint32_t *slabc,
                                            uint32_t ao_base, uint32_t *stage, int tid, int nthr)
{
    const int lane = tid & 63;
    uint32_t *st = stage + (tid >> 6) * 128;
    uint32_t nst = 0;                           // wave-uniform fill of the stage
    auto consider = [&](uint32_t rel, uint32_t idx) {               // called by all lanes of the wave together
        bool cand = false;
        if (rel < (uint32_t)W) cand = maskl[rel] != 0;
        const uint64_t bal = __ballot(cand);
        if (bal == 0) return;                                       // wave-uniform
        const uint32_t n = (uint32_t)__popcll(bal);
        if (nst + n > 64u) { allele_drain(a, st, nst, w0, maskl, slabc, ao_base, lane); nst = 0; }
        if (cand) {
            const uint32_t r = nst + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            st[2 * r] = idx;
            st[2 * r + 1] = rel;
        }
        nst += n;
    };
    if (a.rec16) {                              // short stream: the records themselves are 2 bytes, base included
        const u32x4 *g8 = reinterpret_cast<const u32x4 *>(a.rec16);
        const uint32_t q_lo = lo >> 2, q_hi = hi >> 2;              // units of EIGHT records
        for (uint32_t i0 = q_lo; i0 < q_hi; i0 += 2 * nthr) {
            u32x4 v[2];
            uint32_t bw[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t j = i0 + tid + u * nthr;
                if (j < q_hi) { v[u] = __builtin_nontemporal_load(&g8[j]); bw[u] = a.gbase[__builtin_amdgcn_readfirstlane(j >> 6)] - w0; }
                else { v[u].x = v[u].y = v[u].z = v[u].w = 0xFFFFFFFFu; bw[u] = 0; }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const uint32_t word = (h >> 1) == 0 ? v[u].x : ((h >> 1) == 1 ? v[u].y : ((h >> 1) == 2 ? v[u].z : v[u].w));
                    const uint32_t d = (h & 1) ? (word >> 16) : (word & 0xFFFFu);
                    const uint32_t bb = d >> 13;
                    uint32_t rel = (d & 0x1FFFu) + bw[u];
                    if (bb >= 4 || rel >= (uint32_t)W || !((maskl[rel] >> bb) & 1u)) rel = 0xFFFFFFFFu;   // not an allele of a site
                    consider(rel, 8u * (i0 + tid + u * nthr) + (uint32_t)h);
                }
            }
        }
    } else if (a.gpos16) {                      // 2-byte deltas: 8 records per 16-byte load
        const u32x4 *g8 = reinterpret_cast<const u32x4 *>(a.gpos16);
        const uint32_t q_lo = lo >> 2, q_hi = hi >> 2;              // units of EIGHT records
        for (uint32_t i0 = q_lo; i0 < q_hi; i0 += 2 * nthr) {
            u32x4 v[2];
            uint32_t cb[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint32_t j = i0 + tid + u * nthr;
                if (j < q_hi) { v[u] = __builtin_nontemporal_load(&g8[j]); cb[u] = a.chunk_base[j >> a.gpos16_shift]; }
                else { v[u].x = v[u].y = v[u].z = v[u].w = 0xFFFFFFFFu; cb[u] = 0; }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const uint32_t word = (h >> 1) == 0 ? v[u].x : ((h >> 1) == 1 ? v[u].y : ((h >> 1) == 2 ? v[u].z : v[u].w));
                    const uint32_t d = (h & 1) ? (word >> 16) : (word & 0xFFFFu);
                    const uint32_t rel = d == 0xFFFFu ? 0xFFFFFFFFu : cb[u] + d - w0;
                    consider(rel, 8u * (i0 + tid + u * nthr) + (uint32_t)h);
                }
            }
        }
    } else {
        const u32x4 *g4 = reinterpret_cast<const u32x4 *>(a.gpos);
        const uint32_t q_lo = lo >> 1, q_hi = hi >> 1;              // units of FOUR records
        for (uint32_t i0 = q_lo; i0 < q_hi; i0 += 4 * nthr) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = i0 + tid + u * nthr;
                if (j < q_hi) v[u] = __builtin_nontemporal_load(&g4[j]);
                else { v[u].x = ISX_SENTINEL; v[u].y = ISX_SENTINEL; v[u].z = ISX_SENTINEL; v[u].w = ISX_SENTINEL; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t g = h == 0 ? v[u].x : (h == 1 ? v[u].y : (h == 2 ? v[u].z : v[u].w));
                    consider(g - w0, 4u * (i0 + tid + u * nthr) + (uint32_t)h);
                }
            }
        }
    }
    if (nst) allele_drain(a, st, nst, w0, maskl, slabc, ao_base, lane);
}

// quad broadcast: every lane of a quad of four gets lane 0's value (the record header sits in the first 16-byte quarter)
__device__ __forceinline__ uint32_t quad_first(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x00 /* quad_perm [0,0,0,0] */, 0xF, 0xF, true);
}

#define SEG_SKIPW 0x24924924u       // ten codes 4: nothing to count in this word

// update_linked_reads on the read-segment stream: the window's records are walked a second time, but a 10-base word is
// only opened when the window's SNP-site bitmap has a bit under it (sites are ~1 % of the positions); a qualifying base
// is staged and drained exactly like allele_pass does.  An allele observation's arrival order (obs_idx) is its RECORD:
// two observations of one pair at one site come from its two mates, whose records keep the BAM order.
// lo16, hi16: the window's range in 16-byte quarters (multiples of 64).
__device__ __forceinline__ void allele_pass_segs(const PileupArgs &a, uint32_t lo16, uint32_t hi16, uint32_t w0, int W,
                                                 const uint8_t *maskl, uint32_t *slabc, uint32_t ao_base, uint32_t *stage,
                                                 int tid, int nthr)
{
    const int lane = tid & 63;
    uint32_t *st = stage + (tid >> 6) * 128;
    uint32_t *sitebits = stage + (nthr >> 6) * 128;         // [W / 32 + 2]
    for (int p = tid; p < W; p += nthr) {                   // W and nthr are multiples of 64: whole waves
        const uint64_t bal = __ballot(maskl[p] != 0);
        if (lane == 0) { sitebits[p >> 5] = (uint32_t)bal; sitebits[(p >> 5) + 1] = (uint32_t)(bal >> 32); }
    }
    if (tid < 2) sitebits[(W >> 5) + tid] = 0;
    __syncthreads();
    uint32_t nst = 0;
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < nst) {
            const uint32_t rec = st[2 * lane], info = st[2 * lane + 1];
            const uint32_t rel = info & 0xFFFFu;
            const uint32_t slot = atomicAdd(&slabc[rel], 1u);
            isx_ao o;
            o.pair = a.pair[rec]; o.site = w0 + rel; o.obs_idx = rec;
            o.mm = (uint16_t)(info >> 24); o.base = (uint8_t)((info >> 16) & 7u); o.pad = 0;
            a.ao[ao_base + slot] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const uint32_t q = (uint32_t)tid & 3u;
    for (uint32_t i0 = lo16; i0 < hi16; i0 += (uint32_t)nthr) {
        const uint32_t i = i0 + (uint32_t)tid;
        if ((uint32_t)__builtin_amdgcn_readfirstlane(i) >= hi16) break;         // wave-uniform
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.seg) + i);
        const uint32_t gb = a.gbase[__builtin_amdgcn_readfirstlane(i >> 6)];
        const uint32_t hdr = quad_first(v.x);
        const uint32_t mm = a.M > 1 ? hdr >> 24 : 0u;
        const int32_t r0 = (int32_t)(gb + (hdr & 0xFFFFu) - w0) + (int32_t)(q * 40u) - 10;
        const uint32_t wd[4] = {q ? v.x : SEG_SKIPW, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int32_t r = r0 + 10 * k;
            const uint32_t w = wd[k];
            uint32_t bits = 0;
            if (w != SEG_SKIPW && (uint32_t)(r + 9) < (uint32_t)(W + 9)) {
                const int32_t base = r < 0 ? 0 : r;
                const uint32_t wi = (uint32_t)base >> 5;
                const uint64_t b64 = (uint64_t)sitebits[wi] | ((uint64_t)sitebits[wi + 1] << 32);
                bits = ((uint32_t)(b64 >> (base & 31)) << (base - r)) & 0x3FFu;
            }
            while (__ballot(bits != 0)) {                                       // wave-uniform
                const bool has = bits != 0;
                const int j = has ? __ffs((int)bits) - 1 : 0;
                bits &= bits - 1u;                                              // 0 stays 0
                const uint32_t code = (w >> (3 * j)) & 7u;
                const uint32_t rel = (uint32_t)(r + j);
                const bool cand = has && code < 4u && ((maskl[has ? rel : 0u] >> code) & 1u);
                const uint64_t bal = __ballot(cand);
                if (bal == 0) continue;
                const uint32_t n = (uint32_t)__popcll(bal);
                if (nst + n > 64u) { drain(); nst = 0; }
                if (cand) {
                    const uint32_t at = nst + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    st[2 * at] = i >> 2;
                    st[2 * at + 1] = rel | (code << 16) | (mm << 24);
                }
                nst += n;
            }
        }
    }
    if (nst) drain();
}

// the skip bits of a 32-column word whose first column is window position r, cut to the columns that lie inside [0, W): the walk below
// then needs no per-bit window test (a word wholly inside the window -- nearly every one -- costs one compare)
__device__ __forceinline__ uint32_t skip_bits_in_window(uint32_t bits, int32_t r, uint32_t uW)
{
    if ((uint32_t)r > uW - 32u) {               // crosses an edge of the window or lies outside (a negative r wraps around)
        const int32_t lo = r < 0 ? -r : 0, hi = (int32_t)uW - r < 32 ? (int32_t)uW - r : 32;
        if (hi <= lo) bits = 0;
        else bits &= (hi >= 32 ? 0xFFFFFFFFu : (1u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    return bits;
}

// ---- reference-delta records (include/instrain_amd.h ISX_DREC_*): a PAIR of lanes holds one 32-byte record ----
// lane 0 of the pair: header, skip bits of columns 0..63, exceptions 0-2; lane 1: skip bits of columns 64..159, exceptions 3-5
__device__ __forceinline__ uint32_t pair_first(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t pair_other(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// four reference codes (one per byte) of the positions gpos .. gpos + 3 (gpos a multiple of 4); 4 beyond n_pos
__device__ __forceinline__ uint32_t ref4_at(const PileupArgs &a, uint32_t gpos)
{
    if (gpos + 3u < a.n_pos) {
        if (a.ref_packed == 2) {
            const uint32_t h = a.ref[gpos >> 2];
            uint32_t r = (h & 3u) | ((h & 0xCu) << 6) | ((h & 0x30u) << 12) | ((h & 0xC0u) << 18);
            if (a.ref_n) {
                const uint32_t nb = ((uint32_t)a.ref_n[gpos >> 3] >> (gpos & 4u)) & 0xFu;
                if (nb) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if ((nb >> k) & 1u) r = (r & ~(0xFFu << (8 * k))) | (4u << (8 * k));
                }
            }
            return r;
        }
        if (a.ref_packed) {
            const uint32_t h = *reinterpret_cast<const uint16_t *>(a.ref + (gpos >> 1));
            return (h & 0xFu) | ((h & 0xF0u) << 4) | ((h & 0xF00u) << 8) | ((h & 0xF000u) << 12);
        }
        return *reinterpret_cast<const uint32_t *>(a.ref + gpos);
    }
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) r |= (gpos + k < a.n_pos ? (uint32_t)ref_at(a, gpos + k) : 4u) << (8 * k);
    return r;
}

// The same in two steps, so that the global loads can be issued at a window's top and their values first touched behind the stream loop:
// ref4_at uses what it loads at once (a branch on the non-ACGT bits, the unpacking arithmetic), and the compiler put the wait -- a
// vmcnt(0), which also waits for the records prefetched during the window before -- right behind each load: up to four serialized
// memory round trips at the top of every window of a pipe slot.  ref4_raw only loads (the batch's last positions: the codes themselves)
__device__ __forceinline__ void ref4_raw(const PileupArgs &a, uint32_t gpos, uint32_t &raw, uint32_t &rawn)
{
    rawn = 0;
    if (gpos + 3u < a.n_pos) {
        if (a.ref_packed == 2) {
            raw = a.ref[gpos >> 2];
            if (a.ref_n) rawn = a.ref_n[gpos >> 3];
        } else if (a.ref_packed) raw = *reinterpret_cast<const uint16_t *>(a.ref + (gpos >> 1));
        else raw = *reinterpret_cast<const uint32_t *>(a.ref + gpos);
    } else raw = ref4_at(a, gpos);
}
__device__ __forceinline__ uint32_t ref4_expand(const PileupArgs &a, uint32_t gpos, uint32_t raw, uint32_t rawn)
{
    if (gpos + 3u >= a.n_pos) return raw;
    if (a.ref_packed == 2) {
        uint32_t r = (raw & 3u) | ((raw & 0xCu) << 6) | ((raw & 0x30u) << 12) | ((raw & 0xC0u) << 18);
        const uint32_t nb = (rawn >> (gpos & 4u)) & 0xFu;
#pragma unroll
        for (int k = 0; k < 4; k++) r = ((nb >> k) & 1u) ? ((r & ~(0xFFu << (8 * k))) | (4u << (8 * k))) : r;
        return r;
    }
    if (a.ref_packed) return (raw & 0xFu) | ((raw & 0xF0u) << 4) | ((raw & 0xF00u) << 8) | ((raw & 0xF000u) << 12);
    return raw;
}

// update_linked_reads on the reference-delta stream: like allele_pass_segs, the window's records are walked a second time against
// the bitmap of the window's SNP sites; a record's base at a site is its exception there, else the reference's (unless skipped).
// lo16, hi16: the window's range in 16-byte halves of records (multiples of 64).
__device__ __forceinline__ void allele_pass_delta(const PileupArgs &a, uint32_t lo16, uint32_t hi16, uint32_t w0, int W,
                                                  const uint8_t *maskl, uint32_t *slabc, uint32_t ao_base, uint32_t *stage,
                                                  int tid, int nthr)
{
    const int lane = tid & 63;
    uint32_t *st = stage + (tid >> 6) * 128;
    uint32_t *sitebits = stage + (nthr >> 6) * 128;         // [W / 32 + 2]
    for (int p = tid; p < W; p += nthr) {                   // W and nthr are multiples of 64: whole waves
        const uint64_t bal = __ballot(maskl[p] != 0);
        if (lane == 0) { sitebits[p >> 5] = (uint32_t)bal; sitebits[(p >> 5) + 1] = (uint32_t)(bal >> 32); }
    }
    if (tid < 2) sitebits[(W >> 5) + tid] = 0;
    __syncthreads();
    uint32_t nst = 0;
    auto drain = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < nst) {
            const uint32_t half = st[2 * lane], info = st[2 * lane + 1];     // half: index of the segment's 16-byte half (a full record: its first)
            const uint32_t rel = info & 0xFFFFu;
            const uint32_t slot = atomicAdd(&slabc[rel], 1u);
            isx_ao o;
            // the read-pair id travels inside the record: word 1 of a dual half, word 7 of a full record
            o.pair = reinterpret_cast<const uint32_t *>(a.drec)[(info >> 20) & 1u ? ((half | 1u) << 2) + 3u : (half << 2) + 1u];
            o.site = w0 + rel; o.obs_idx = half;
            o.mm = (uint16_t)(info >> 24); o.base = (uint8_t)((info >> 16) & 7u); o.pad = 0;
            a.ao[ao_base + slot] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const uint32_t odd = (uint32_t)tid & 1u;
    for (uint32_t i0 = lo16; i0 < hi16; i0 += (uint32_t)nthr) {
        const uint32_t i = i0 + (uint32_t)tid;
        if ((uint32_t)__builtin_amdgcn_readfirstlane(i) >= hi16) break;         // wave-uniform
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.drec) + i);
        const uint32_t gb = a.gbase[__builtin_amdgcn_readfirstlane(i >> 6)];
        const uint32_t hdr0 = pair_first(v.x);
        const bool dual = (hdr0 >> 31) != 0u;                   // two segments without skipped columns, a lane each; else one with its skip plane
        const uint32_t hdr = dual ? v.x : hdr0;
        const uint32_t len = (hdr >> 16) & 0xFFu;
        const uint32_t w3 = pair_first(v.w);                    // a full record's exceptions (its word 3)
        const uint32_t e_a = dual ? v.z : w3, e_b = dual ? v.w : ISX_DREC_NO_EXC;
        const int32_t s = (int32_t)(gb + (hdr & 0xFFFFu) - w0);
        // the five 32-column chunks of a segment: a dual half walks all of its own; of a full record the first lane walks chunks 0, 1 (skip
        // words 1, 2), the second 2, 3, 4 (words 4, 5, 6)
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const uint32_t c0 = 32u * (uint32_t)k;                 // first column of this 32-column chunk
            const int32_t r = s + (int32_t)c0;
            const uint32_t skw = dual ? 0u : (k == 0 ? v.y : (k == 1 ? v.z : (k == 2 ? v.x : (k == 3 ? v.y : v.z))));
            const bool mine = dual || (odd != 0u) == (k >= 2);
            uint32_t bits = 0;
            if (len > c0 && mine && (uint32_t)(r + 31) < (uint32_t)(W + 31)) {
                const uint32_t ncol = min(len - c0, 32u);
                const int32_t base = r < 0 ? 0 : r;
                const uint32_t wi = (uint32_t)base >> 5;
                const uint64_t b64 = (uint64_t)sitebits[wi] | ((uint64_t)sitebits[wi + 1] << 32);
                bits = (uint32_t)(b64 >> (base & 31)) << (base - r);
                bits &= ~skw & (ncol == 32u ? 0xFFFFFFFFu : (1u << ncol) - 1u);
            }
            while (__ballot(bits != 0)) {                                       // wave-uniform
                const bool has = bits != 0;
                const int j = has ? __ffs((int)bits) - 1 : 0;
                bits &= bits - 1u;                                              // 0 stays 0
                const uint32_t col = c0 + (uint32_t)j;
                const uint32_t rel = has ? (uint32_t)(r + j) : 0u;
                uint32_t code = 8u;
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const uint32_t x = (e_a >> (10 * f)) & 0x3FFu, y = (e_b >> (10 * f)) & 0x3FFu;
                    if ((x & 0xFFu) == col) code = x >> 8;
                    if ((y & 0xFFu) == col) code = y >> 8;
                }
                if (has && code == 8u) code = ref_at(a, w0 + rel);
                const bool cand = has && code < 4u && ((maskl[rel] >> code) & 1u);
                const uint64_t bal = __ballot(cand);
                if (bal == 0) continue;
                const uint32_t n = (uint32_t)__popcll(bal);
                if (nst + n > 64u) { drain(); nst = 0; }
                if (cand) {
                    const uint32_t at = nst + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    st[2 * at] = dual ? i : (i & ~1u);
                    st[2 * at + 1] = rel | (code << 16) | (dual ? 0u : 1u << 20) | (((hdr >> 24) & 0x7Fu) << 24);     // (the pair's mm level: bits 24..30 of the header)
                }
                nst += n;
            }
        }
    }
    if (nst) drain();
}

// ---------------------------------------------------------------------------------------------
// k_pileup_dense: the n_mm_bins == 1 (--skip_mm_profiling / --database_mode) specialisation.
// Persistent workgroups: each walks windows slot, slot + grid, ... so the per-window fixed costs
// (launch, threshold staging, dependent global latencies of the epilogue) are paid behind the
// NEXT window's first loads, which are issued before the epilogue starts.  SNV rows / SNP sites /
// allele-observation slabs are allocated with ONE global atomic each per window (LDS-aggregated);
// clonality divisions and row emission run densely packed from an LDS queue.
// LDS: cnt[4][S] | queue[S] | scratch[16] | thr_lds[THR_LDS] | (linkage) slabc[W] | maskl[W bytes], S = W + ISX_DENSE_PAD:
// the 8 extra words of a row are the junk columns of the packed decode (records that belong to another window), and the
// queue region -- idle during the stream -- is the junk row of records without an A/C/T/G base.
// ---------------------------------------------------------------------------------------------
#ifndef ISX_LB_WAVES
#define ISX_LB_WAVES 8
#endif
#ifdef ISX_TUNING
// timeline of workgroup 0 (debug bit 4096): 16 stamps a window, first 32 windows; read back with isx_debug_read_ts (tools/timeline.py)
__device__ unsigned long long g_isx_ts[32 * 16];
extern "C" int isx_debug_read_ts(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_isx_ts), sizeof(g_isx_ts)); }
#define ISX_TS(k) do { if ((dbg & 4096) && blockIdx.x == 0 && tid == 0 && ts_w < 32) g_isx_ts[ts_w * 16 + (k)] = wall_clock64(); } while (0)
#else
#define ISX_TS(k) do { } while (0)
#endif
template <bool LINKAGE, int FMT, bool PK16 = false>   // FMT = bytes per resident record: 8 (isx_obs), 4 (compact), 2 (short), 64 (read
__global__ void __launch_bounds__(1024, ISX_LB_WAVES) k_pileup_dense(const PileupArgs a_kernarg)     // segments); PK16: short records decoded two at a time
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // The ~150 dwords of arguments stay in the kernarg segment and every phase of a window reads the few it needs from there again
    // (scalar loads, cache hits): held in SGPRs across the whole window loop they were spilled to VGPR lanes, and a third of the VALU
    // instructions issued were v_readlane / v_writelane.  ISX_ARGS_FRESH() makes the compiler forget what it has loaded so far.
    typedef const __attribute__((address_space(4))) PileupArgs KernArgs;
    KernArgs *kargs = (KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#define a (*(const PileupArgs *)kargs)
    // The same for what derives from the thread index: the compiler hoists every tid-based address and predicate out of the window loop
    // and then has no registers to keep them in -- they went to scratch memory, and a scratch reload at a window's top is a vector memory
    // operation: its s_waitcnt vmcnt(0) also waited for the record loads prefetched during the window before (the whole point of the
    // prefetch).  At every phase boundary the thread index is an unknown again; what a phase needs of it is recomputed in a few VALU.
    int tid = threadIdx.x;
    const int nthr = blockDim.x;
#define ISX_ARGS_FRESH() asm volatile("" : "+s"(kargs), "+v"(tid))
    publish_previous(a, tid);
    const int W = a.W;
    constexpr bool SEGS = FMT == 64;
    constexpr bool DREC = FMT == 32;                            // reference-delta records: see count_slot and the materialise phase
    // PKL (reference-delta records of a batch none of whose windows streams 32768 records): 16-bit counters, two per LDS word --
    // rows (A | C) and (T | G), and ONE row for skipped columns (low half) and coverage differences (high half, wrapping) that
    // becomes the queue after the materialise phase: 12 instead of 24 bytes a position, so a window is twice as wide and the
    // per-window fixed costs (barriers, dependent global latencies, cursor round trips) are paid half as often
    constexpr bool PKL = DREC && PK16;
    constexpr int NR = PKL ? 2 : 4;                             // counter rows
    const int S = W + (SEGS ? ISX_SEG_PAD : ISX_DENSE_PAD);     // row stride of the counters
    uint32_t *cnt = lds + (SEGS ? ISX_SEG_LM : 0);              // segments: ISX_SEG_LM margin columns on either side of the window
    uint32_t *queue = lds + NR * S;
    uint32_t *scratch = queue + S;
    uint2 *rngl = reinterpret_cast<uint2 *>(scratch + S_N);
    uint16_t *thr_lds = reinterpret_cast<uint16_t *>(scratch + S_N + S_RNG);
    uint32_t *slabc = scratch + S_N + S_RNG + THR_LDS / 2;
    uint8_t *maskl = reinterpret_cast<uint8_t *>(slabc + W);
    // DREC: two more rows -- the coverage differences (+1 where a record starts, -1 behind its end) and, in the queue's row (idle
    // until the epilogue), the skipped columns; the counter rows hold the EXCEPTIONS until the materialise phase
    uint32_t *dlt = PKL ? queue : lds + a.dlt_off;
    uint32_t *wtot = lds + a.dlt_off + (PKL ? 0 : S);           // [16] per-wave totals of the prefix sum
    uint8_t *refl = reinterpret_cast<uint8_t *>(wtot + 16);     // PKL: the window's reference codes [W], stashed by the materialise phase
    // the four counts of window position p
    auto ld4 = [&](int p, uint32_t *c) {
        if (PKL) {
            const uint32_t x = cnt[p], y = cnt[S + p];
            c[0] = x & 0xFFFFu; c[1] = x >> 16; c[2] = y & 0xFFFFu; c[3] = y >> 16;
        } else { c[0] = cnt[p]; c[1] = cnt[S + p]; c[2] = cnt[2 * S + p]; c[3] = cnt[3 * S + p]; }
    };
    constexpr bool linkage = LINKAGE;            // compile-time: the linkage-off kernels carry none of the allele pass
    const int grid = gridDim.x, per = grid >> 3;
    const int slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // consecutive windows share an XCD's L2
    // COMPACT: 4-byte records (4 per 16-byte load); a wave-wide load covers exactly one ISX_GROUP of 256
    // records, so the group's position base is a scalar load.  Otherwise the 8-byte isx_obs (2 per load).
    constexpr bool COMPACT = FMT != 8;
    const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(SEGS ? (const void *)a.seg : (DREC ? (const void *)a.drec : (FMT == 2 ? (const void *)a.rec16 : (FMT == 4 ? (const void *)a.rec32 : (const void *)a.rec))));
    constexpr int RSH = FMT == 2 ? 3 : (FMT == 4 ? 2 : 1);         // record index -> 16-byte load index (segments: a record is FOUR loads)
    const int dbg = a.debug_mode;               // ablation switches (tools/), 0 in production
    const bool stripe_k = PKL && a.stripe != 0; // packed rows without a count table: materialise + first epilogue pass by stripes (below), window by window

    {   // once per workgroup: folded thresholds of the low coverages
        const int n = min(THR_LDS, a.lut_n);
        for (int i = tid; i < n; i += nthr) thr_lds[i] = a.thr[i];
    }

    u32x4 v[4];
    uint32_t gb[4] = {0, 0, 0, 0};              // COMPACT: position base of each load's group (wave-uniform)
    bool live[4] = {false, false, false, false};   // COMPACT: the wave really loaded slot u (wave-uniform); a skipped slot holds nothing to count
    uint32_t lo = 0, hi = 0;
    const uint32_t lane16 = (uint32_t)tid;      // lane offset in 16-byte units: the one address register of the stream loads
    auto issue_one = [&](int u, uint32_t i0) {  // one coalesced 16-byte load per lane into slot u
        const uint32_t j = i0 + tid + u * nthr;
        // compact streams: unconditional (ISX_TAIL_BYTES of padding follow the stream; records past `hi` lie beyond the
        // window and are dropped like any other) -- no per-lane branch, no select against a padding value
        // (a wave whose 64 loads all lie past `hi` skips the load -- a uniform branch -- and counts padding instead)
        const uint32_t jw = __builtin_amdgcn_readfirstlane(j);
        if (COMPACT) live[u] = jw < hi;
        if (COMPACT ? jw < hi : j < hi) {
            // wave-uniform base in scalar registers + one 32-bit lane offset (saddr form): no 64-bit address pair per
            // load in flight -- with four loads rotating the pairs were spilled and every reload drained vmcnt
            const uint64_t ub = reinterpret_cast<uint64_t>(rec4 + (i0 + (uint32_t)(u * nthr)));
            typedef __attribute__((address_space(1))) const u32x4 gvec;        // global, not flat: vmcnt only
            const gvec *sb = reinterpret_cast<const gvec *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32)) << 32) |
                                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ub));
            // (with linkage the allele pass walks the window's records a second time: the first walk leaves them in the L2, the second --
            // non-temporal -- is their last use)
            v[u] = linkage ? *(sb + lane16) : __builtin_nontemporal_load(sb + lane16);
            if (COMPACT) gb[u] = a.gbase[__builtin_amdgcn_readfirstlane(j >> 6)];
        } else if (FMT == 2) { v[u].x = v[u].y = v[u].z = v[u].w = 0xFFFFFFFFu; }
        else if (SEGS || DREC) { }             // (a slot the wave did not load is never counted: live[u])
        else if (FMT == 4) { v[u].x = v[u].y = v[u].z = v[u].w = ISX_PAD32; }
        else { v[u].x = ISX_SENTINEL; v[u].y = 0; v[u].z = ISX_SENTINEL; v[u].w = 0; }
    };
    // a window's record range is loaded one window AHEAD of its first record loads (at the top of the window before): the loads
    // that depend on it are then issued without waiting for a global round trip
    uint2 rng_next = make_uint2(0, 0);
    auto prefetch_window = [&](int wn) {
        lo = hi = 0;
        if (wn < a.n_win) {
            const uint2 rng = rng_next;
            if (SEGS) { lo = rng.x << 2; hi = rng.y << 2; } else if (DREC) { lo = rng.x << 1; hi = rng.y << 1; } else { lo = rng.x >> RSH; hi = rng.y >> RSH; }
            if (lo < hi) { issue_one(0, lo); issue_one(1, lo); }        // the first half-round; the stream loop issues the rest
        }
    };
    if (slot < a.n_win) rng_next = a.win_range[slot];
    prefetch_window(slot);
    int win_it = 0;                             // this workgroup's windows so far

#ifdef ISX_TUNING
    int ts_w = -1;
#endif
    for (int w = slot; w < a.n_win; w += grid) {
        const uint32_t w0 = (uint32_t)w * (uint32_t)W;
        const uint32_t cur_lo = lo, cur_hi = hi;
        // the stripe path pays where many positions stay below min_cov; in a deep window every position goes on, the compaction is wasted and
        // the loops behind it walk a full queue for its few flagged entries (resident C2, depth 18: 68 us against 59).  A window that streams
        // more than W / 16 records (mean depth beyond ~8 at 135 kept bases a record) takes the per-position epilogue
        const bool stripe = stripe_k && (cur_hi - cur_lo) * 8u <= (uint32_t)W;
#ifdef ISX_TUNING
        ++ts_w;
#endif
        ISX_TS(0);
        ISX_ARGS_FRESH();
        // the record ranges of the workgroup's next 64 windows wait in LDS (filled here every 64th window; visible behind the barrier
        // below, read at the window's end): loaded window by window, the value -- wanted in scalar registers, which the compiler has none
        // to spare for -- was waited for on the spot, a memory round trip at the top of every window
        if ((win_it & 63) == 0 && tid < 64) {
            const int wn = w + (tid + 1) * grid;
            rngl[tid] = wn < a.n_win ? a.win_range[wn] : make_uint2(0, 0);
        }
        const uint32_t dummy = 4u * (uint32_t)S + (uint32_t)(tid & 63);     // see the stream loop
#ifdef ISX_TUNING
        uint32_t ablate_acc = 0;
#endif
        {   // zero the window's counters
            uint4 *z = reinterpret_cast<uint4 *>(cnt);
            const int nz = ((NR + (DREC ? 1 : 0)) * S) >> 2;        // the counter rows (+ the skipped-columns row behind them)
            for (int i = tid; i < nz; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
            if (DREC && !PKL) {
                uint4 *zd = reinterpret_cast<uint4 *>(dlt);
                for (int i = tid; i < (S >> 2); i += nthr) zd[i] = make_uint4(0, 0, 0, 0);
            }
            if (linkage) {
                uint4 *zm = reinterpret_cast<uint4 *>(maskl);
                for (int i = tid; i < (W >> 4); i += nthr) zm[i] = make_uint4(0, 0, 0, 0);
            }
            if (tid < S_N) scratch[tid] = 0;
        }
        // DREC: the reference codes of this thread's four positions of the materialise phase (ref4); with packed rows the window's
        // codes go to LDS here (coalesced word loads; read back after the stream's barrier by whichever thread owns a position)
        uint32_t ref4 = 0x04040404u, ref4b = 0x04040404u;      // (loaded here, stored behind the stream loop: the latency hides there)
        uint32_t ref4n = 0, ref4bn = 0;
        if (PKL) {                              // (raw: unpacked where they are stored)
            if (4 * tid < W) ref4_raw(a, w0 + 4u * (uint32_t)tid, ref4, ref4n);
            if (4 * (tid + nthr) < W) ref4_raw(a, w0 + 4u * (uint32_t)(tid + nthr), ref4b, ref4bn);
        } else if (DREC && 4 * tid < W) ref4 = ref4_at(a, w0 + 4u * (uint32_t)tid);
        uint8_t ref_raw[2];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const uint32_t gp = w0 + tid + it * nthr;
            ref_raw[it] = (!PKL && tid + it * nthr < W && gp < a.n_pos) ? ref_at(a, gp) : (uint8_t)4;
        }
        __syncthreads();
        ISX_TS(1);
        ISX_ARGS_FRESH();

        // ---- get_base_counts_mm (profile_utilities.py:268-286) over the window's slice ----
        // Branch-free: a record outside the window / without an A,C,T,G base adds to a per-lane dummy word (the queue
        // region, idle during the stream) -- no exec-mask juggling per record, and the counter index is a 24-bit mad
        // (v_mul_lo_u32 is quarter rate).
        auto count_slot = [&](int u) {
            if (COMPACT && !live[u]) return;        // uniform: the last round of a window is half empty on average
            if (DREC) {
                // Reference-delta records: a pair of lanes holds one 32-byte record -- ONE segment with its plane of skipped columns (a full
                // record), or TWO segments without skipped columns, a lane each (a dual record: include/instrain_amd.h ISX_DREC_*).  Coverage
                // is counted by DIFFERENCE -- +1 at a segment's first column, -1 behind its last (2 LDS atomics per segment) -- and only what
                // is not "the reference's base, observed" touches a counter of its own: one atomic per skipped column (the set bits of this
                // lane's share of the skip plane) and one per exception.  ~17 atomics per 150-base read instead of 135.
                const uint32_t odd = (uint32_t)tid & 1u;
                const uint32_t hdr0 = pair_first(v[u].x);
                const bool dual = (hdr0 >> 31) != 0u;
                const uint32_t hdr = dual ? v[u].x : hdr0;
                const uint32_t len = (hdr >> 16) & 0xFFu;
                const int32_t s = (int32_t)(gb[u] + (hdr & 0xFFFFu) - w0);
                const uint32_t uW = (uint32_t)W;
#ifdef ISX_TUNING
                if (dbg & 64) { ablate_acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w ^ hdr; return; }        // loads only
#endif
                if ((dual || !odd) && len) {
                    const int32_t hi_c = s + (int32_t)len;
                    if (hi_c > 0 && s < W) {
                        atomicAdd(&dlt[s < 0 ? 0 : s], PKL ? 0x00010000u : 1u);
                        if (hi_c < W) atomicAdd(&dlt[hi_c], PKL ? 0xFFFF0000u : 0xFFFFFFFFu);
                    }
                }
                uint32_t sk[3] = {odd ? v[u].x : v[u].y, odd ? v[u].y : v[u].z, odd ? v[u].z : 0u};
                if (dual) sk[0] = sk[1] = sk[2] = 0;                    // (a dual record's segments have no skipped columns)
#ifdef ISX_TUNING
                if (dbg & 8) sk[0] = sk[1] = sk[2] = 0;                 // no skip-plane walk
#endif
                const int32_t c0 = s + (odd ? 64 : 0);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int32_t r = c0 + 32 * k;
                    uint32_t bits = skip_bits_in_window(sk[k], r, uW);
                    while (__ballot(bits != 0)) {                               // wave-uniform
                        if (bits != 0) atomicAdd(&queue[r + (int32_t)__builtin_ctz(bits)], 1u);
                        bits &= bits - 1u;                                      // 0 stays 0
                    }
                }
                // exceptions: a dual half carries six (its words 2, 3); a full record three (word 3 = the first lane's fourth word; its
                // word 7 is the read-pair id)
                const uint32_t exw[2] = {dual ? v[u].z : (odd ? ISX_DREC_NO_EXC : v[u].w), dual ? v[u].w : ISX_DREC_NO_EXC};
                // (fields fill up in order; the second word -- a dual half's exceptions four to six -- is empty in nearly every wave)
                const int nf = __ballot(exw[1] != ISX_DREC_NO_EXC) ? 6 : 3;
#pragma unroll
                for (int f = 0; f < 6; f++) {
                    if (f == 3 && nf == 3) break;               // wave-uniform
                    const uint32_t ex = exw[f / 3];
                    const uint32_t off = (ex >> (10 * (f % 3))) & 0xFFu, code = (ex >> (10 * (f % 3) + 8)) & 3u;
                    const uint32_t rel = (uint32_t)(s + (int32_t)off);
                    if (off < len && rel < uW) {                 // (an empty field has offset 255)
                        if (PKL) atomicAdd(&cnt[__umul24(code >> 1, (uint32_t)S) + rel], 1u << (16 * (code & 1u)));
                        else atomicAdd(&cnt[__umul24(code, (uint32_t)S) + rel], 1u);
                    }
                }
                return;
            }
            if (SEGS) {
                // Read segments: a quad of lanes holds one 64-byte record, lane q its quarter -- the header (lane 0's first
                // word, broadcast inside the quad) and three words of ten bases, or four words: bases 40 q - 10 + 10 k ... of
                // the segment.  A word lies at ten consecutive positions, so its counters are ten consecutive columns (rows
                // = base code; row 4 -- the idle queue region -- swallows code 4) at one LDS address + immediate offsets:
                // 2 VALU + one LDS atomic per base and no per-base window test (the margin columns absorb the edge words).
                const uint32_t q = (uint32_t)tid & 3u;
                const uint32_t hdr = quad_first(v[u].x);
                const int32_t r0 = (int32_t)(gb[u] + (hdr & 0xFFFFu) - w0) + (int32_t)(q * 40u) - 10;
                const uint32_t wd[4] = {q ? v[u].x : SEG_SKIPW, v[u].y, v[u].z, v[u].w};
                char *lds_b = reinterpret_cast<char *>(lds);
                const uint32_t S4 = (uint32_t)S * 4u;
#ifdef ISX_TUNING       // ablations of the stream loop (tools/tune_reads.py): what bounds it?
                if (dbg & 64) { ablate_acc += wd[0] ^ wd[1] ^ wd[2] ^ wd[3] ^ hdr; return; }            // loads only
#endif
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t r = r0 + 10 * k;
                    uint32_t w = wd[k];
                    if (w == SEG_SKIPW || (uint32_t)(r + 9) >= (uint32_t)(W + 9)) continue;
                    const uint32_t m = w & SEG_SKIPW;                   // codes 5..7 count like 4
                    w &= ~((m >> 1) | (m >> 2));
                    const uint32_t a0 = (uint32_t)(r + ISX_SEG_LM) << 2;
#pragma unroll
                    for (int j = 0; j < 10; j++) {
                        const uint32_t code = __builtin_amdgcn_ubfe(w, 3 * j, 3);
#ifdef ISX_TUNING
                        if (dbg & 8) { ablate_acc += __umul24(code, S4) + a0 + 4u * (uint32_t)j; continue; }                          // decode, no LDS
                        if (dbg & 16) { atomicAdd(reinterpret_cast<uint32_t *>(lds_b + ((dummy + (code & 0u)) << 2)), 1u); continue; }  // lane-private word: no conflicts
#endif
                        atomicAdd(reinterpret_cast<uint32_t *>(lds_b + (__umul24(code, S4) + a0 + 4u * (uint32_t)j)), 1u);
                    }
                }
                return;
            }
            if (FMT == 2) {
                const uint32_t bw = gb[u] - w0;
                const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#ifdef ISX_TUNING
                if (dbg & 64) { ablate_acc += x[0] ^ x[1] ^ x[2] ^ x[3] ^ bw; return; }         // loads only
#endif
                if (PK16
#ifdef ISX_TUNING
                    && !(dbg & (8 | 16 | 32))
#endif
                ) {
                    // Two records per 32-bit word, decoded together with packed 16-bit ALU ops: the stream phase is VALU-bound
                    // (SQ counters) and this is 5 VALU per record instead of 8 VALU + 1 SALU.
                    //   rel  = delta + (group base - window start) mod 2^16; anything >= W (a later window; an earlier one
                    //          wraps to >= 57345) is clamped into the lane's junk column W + (lane & 7)
                    //   row  = min(base, 4): row 4 is the junk row (the idle queue region) for non-ACGT / padding records
                    //   byte offset = (row * S + rel) * 4 < 65536 because W <= ISX_PK16_MAX_W (checked at launch)
                    // The group's offset is wave-uniform; one that cannot belong to a group overlapping the window (a chunk of
                    // the directory may drag in a group of another genome) would alias in 16 bits and is replaced by one that
                    // sends every record of the load to the junk columns.
                    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                    const uint32_t o16 = (bw + 8192u < 24576u) ? (bw & 0xFFFFu) : 0x7000u;
                    const u16x2 bw2 = {(unsigned short)o16, (unsigned short)o16};
                    const unsigned short wl = (unsigned short)(W + (tid & 7));
                    const u16x2 wl2 = {wl, wl};
                    const u16x2 sv = {(unsigned short)S, (unsigned short)S};
                    const u16x2 m13 = {0x1FFF, 0x1FFF};
                    const u16x2 four = {4, 4};
                    char *lds_b = reinterpret_cast<char *>(lds);
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const u16x2 xx = __builtin_bit_cast(u16x2, x[h]);
                        const u16x2 r = __builtin_elementwise_min((u16x2)((xx & m13) + bw2), wl2);
                        const u16x2 row = __builtin_elementwise_min((u16x2)(xx >> 13), four);
                        const u16x2 f = (u16x2)(row * sv + r) << 2;
                        const uint32_t fa = __builtin_bit_cast(uint32_t, f);
                        atomicAdd(reinterpret_cast<uint32_t *>(lds_b + (fa & 0xFFFFu)), 1u);
                        atomicAdd(reinterpret_cast<uint32_t *>(lds_b + (fa >> 16)), 1u);
                    }
                    return;
                }
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const uint32_t r = __builtin_amdgcn_ubfe(x[h >> 1], 16 * (h & 1), 13) + bw;
                    const uint32_t bb = __builtin_amdgcn_ubfe(x[h >> 1], 16 * (h & 1) + 13, 3);
#ifdef ISX_TUNING       // ablations of the stream loop (tools/ablate_dense.py): what bounds it?
                    if (dbg & 8) { ablate_acc += (r < (uint32_t)W && bb < 4) ? __umul24(bb, (uint32_t)S) + r : dummy; continue; }     // decode, no LDS
                    if (dbg & 16) { atomicAdd(&cnt[dummy], (r < (uint32_t)W && bb < 4) ? 1u : 0u); continue; }                         // lane-private word
                    if (dbg & 32) { if (h == 0) atomicAdd(&cnt[(r < (uint32_t)W && bb < 4) ? __umul24(bb, (uint32_t)S) + r : dummy], 1u); continue; }   // 1 of 8 records
#endif
                    atomicAdd(&cnt[(r < (uint32_t)W && bb < 4) ? __umul24(bb, (uint32_t)S) + r : dummy], 1u);
                }
            } else if (FMT == 4) {
                const uint32_t bw = gb[u] - w0;
                const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const uint32_t r = (x[h] & 0xFFFFu) + bw, bb = __builtin_amdgcn_ubfe(x[h], 24, 3);
                    atomicAdd(&cnt[(r < (uint32_t)W && bb < 4) ? __umul24(bb, (uint32_t)S) + r : dummy], 1u);
                }
            } else {
                const uint32_t g0 = v[u].x, a0 = v[u].y, g1 = v[u].z, a1 = v[u].w;
                const uint32_t r0 = g0 - w0, r1 = g1 - w0;
                const uint32_t b0 = (a0 >> 16) & 0xFFu, b1 = (a1 >> 16) & 0xFFu;
                if (r0 < (uint32_t)W && b0 < 4) atomicAdd(&cnt[b0 * S + r0], 1u);
                if (r1 < (uint32_t)W && b1 < 4) atomicAdd(&cnt[b1 * S + r1], 1u);
            }
        };
        // Two half-rounds in flight: slots 0,1 (loaded during the previous half / the previous window's epilogue) are
        // counted while slots 2,3 load, and slots 0,1 of the NEXT round load while 2,3 are counted.  Every slot is
        // loaded, then consumed, then reloaded in static program order: no register copies, the waits are vmcnt(2).
        if (DREC) {
            // reference-delta records: the stream is a fifth of the observation records' -- one half-round (two slots) in flight is
            // enough, and the eight registers of the other one keep the kernel out of scratch memory
            for (uint32_t i0 = lo; i0 < hi; i0 += 2 * nthr) {
                count_slot(0); count_slot(1);
                if (i0 + 2 * nthr < hi) { issue_one(0, i0 + 2 * nthr); issue_one(1, i0 + 2 * nthr); }
            }
        } else
        for (uint32_t i0 = lo; i0 < hi; i0 += 4 * nthr) {
            issue_one(2, i0); issue_one(3, i0);
            count_slot(0); count_slot(1);
            if (i0 + 4 * nthr < hi) { issue_one(0, i0 + 4 * nthr); issue_one(1, i0 + 4 * nthr); }
            count_slot(2); count_slot(3);
        }
#ifdef ISX_TUNING
        if (ablate_acc == 0xDEADBEEFu) cnt[dummy] = ablate_acc;             // keeps the ablated decode alive
#endif
        if (PKL) {                              // the window's reference codes, for whichever thread owns a position afterwards
            if (4 * tid < W) reinterpret_cast<uint32_t *>(refl)[tid] = ref4_expand(a, w0 + 4u * (uint32_t)tid, ref4, ref4n);
            if (4 * (tid + nthr) < W) reinterpret_cast<uint32_t *>(refl)[tid + nthr] = ref4_expand(a, w0 + 4u * (uint32_t)(tid + nthr), ref4b, ref4bn);
        }
        __syncthreads();
        ISX_TS(2);
        ISX_ARGS_FRESH();

        // first loads of the NEXT window go out before the epilogue (with linkage the registers
        // are needed by the allele pass first, so the prefetch follows it)
        if (!linkage) { rng_next = rngl[win_it & 63]; prefetch_window(w + grid); }

        if (PKL && !stripe && !(dbg & 16)) {
            // ---- materialise, packed rows: a thread owns PT consecutive positions (PT odd: a wave's 64 lanes then hit 64 different
            //      LDS banks at every step) ----
            const int lane = tid & 63;
            const int PT = ((W + nthr - 1) / nthr) | 1, p0 = tid * PT;
            int32_t sum = 0;
            for (int k = 0; k < PT; k++) if (p0 + k < W) sum += (int32_t)dlt[p0 + k] >> 16;
            const int32_t inc = (int32_t)wave_scan_incl((uint32_t)sum);
            if (lane == 63) wtot[tid >> 6] = (uint32_t)inc;
            __syncthreads();
            ISX_TS(3);
            int32_t run = inc - sum;
            for (int k = 0; k < (tid >> 6); k++) run += (int32_t)wtot[k];
            bool beyond15 = false;                  // (4-bit coverage plane: does the window need its 16-bit row?)
            for (int k = 0; k < PT; k++) {
                const int p = p0 + k;
                if (p >= W) break;
                const uint32_t x = dlt[p], a01 = cnt[p], a23 = cnt[S + p];
                run += (int32_t)x >> 16;
                beyond15 |= run - (int32_t)(x & 0xFFFFu) > 15;        // (what is observed: covered minus skipped)
                const uint32_t r = refl[p];
                if (r < 4u) {
                    const uint32_t v = (uint32_t)run - (x & 0xFFFFu) - ((a01 & 0xFFFFu) + (a01 >> 16) + (a23 & 0xFFFFu) + (a23 >> 16));
                    cnt[(r >> 1) * S + p] = ((r >> 1) ? a23 : a01) + (v << (16 * (r & 1u)));
                }
            }
            if (a.cov4 && __ballot(beyond15) && lane == 0) scratch[S_COVX] = 1u;       // (every writer writes the same 1)
            __syncthreads();
            ISX_TS(4);
            ISX_ARGS_FRESH();
        }
        if (DREC && !PKL && !(dbg & 16)) {
            // ---- materialise: the reference base's count of every position from the coverage differences ----
            //   covered[p] = prefix sum of the difference row; observed[p] = covered[p] - skipped[p];
            //   count of the reference's base = observed[p] - sum of the exceptions counted at p
            // (a thread owns four consecutive positions; prefix sum over the wave, per-wave totals through LDS) -- afterwards the
            // counter rows are what the other record formats leave behind and the epilogue below does not know the difference
            const int t4 = tid * 4, lane = tid & 63;
            const bool act = t4 < W;
            uint32_t run[4] = {0, 0, 0, 0};
            if (act) {
                const uint4 d = *reinterpret_cast<const uint4 *>(dlt + t4);
                run[0] = d.x; run[1] = run[0] + d.y; run[2] = run[1] + d.z; run[3] = run[2] + d.w;
            }
            const uint32_t inc = wave_scan_incl(run[3]);
            if (lane == 63) wtot[tid >> 6] = inc;
            __syncthreads();
            ISX_TS(3);
            uint32_t off = inc - run[3];
            for (int k = 0; k < (tid >> 6); k++) off += wtot[k];
            if (act) {
                const uint4 m4 = *reinterpret_cast<const uint4 *>(queue + t4);
                const uint4 c0 = *reinterpret_cast<const uint4 *>(cnt + t4), c1 = *reinterpret_cast<const uint4 *>(cnt + S + t4),
                            c2 = *reinterpret_cast<const uint4 *>(cnt + 2 * S + t4), c3 = *reinterpret_cast<const uint4 *>(cnt + 3 * S + t4);
                const uint32_t mk[4] = {m4.x, m4.y, m4.z, m4.w};
                const uint32_t ek[4] = {c0.x + c1.x + c2.x + c3.x, c0.y + c1.y + c2.y + c3.y, c0.z + c1.z + c2.z + c3.z, c0.w + c1.w + c2.w + c3.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t r = (ref4 >> (8 * k)) & 0xFFu;
                    if (r < 4u) cnt[__umul24(r, (uint32_t)S) + (uint32_t)(t4 + k)] = off + run[k] - mk[k] - ek[k];
                }
            }
            __syncthreads();
            ISX_TS(4);
            ISX_ARGS_FRESH();
        }

        // ---- epilogue pass 1: integer only (update_snp_table, single mm level) ----
        // what a position that reaches min_cov (or rarefied_coverage) needs: the SNV call, its clonality class, its table slots.  `qi` < 0: the
        // position joins the queue if anything is left to do for it; else it IS queue entry qi (the stripe path) and its entry is rewritten in place
        // stripe path: its queue holds EVERY position that reaches min_cov, and few of them have anything left to do after the first pass
        // (a clonality to divide, a row, a clonTR value).  Their queue indices are listed in the free top of the queue row (a shallow
        // window's queue is short: room for 1024 two-byte indices) so that the loops below run over them densely instead of having every
        // wave walk the whole queue for a lane or two; a deep window -- no room, or more than 1024 -- walks the queue as before
        uint16_t *const flist = reinterpret_cast<uint16_t *>(queue + S - 512);
        bool use_fl = false;
        auto site_pass1 = [&](int p, uint32_t gpos, const uint32_t *c, uint32_t total, int ref_base, int qi, bool st_ok, bool call_ok) {
            float cl = __builtin_nanf("");
            bool defer = false;
            uint32_t entry = (uint32_t)p;
            if ((int64_t)total >= (int64_t)a.min_cov && call_ok) {
#ifdef ISX_TUNING
                const SiteCall sc = (dbg & 1024) ? SiteCall{-1, 1, 0, 0} : call_level(a, thr_lds, c, total, ref_base, false);    // 1024: no SNV call
                const uint32_t mx = (dbg & 2048) ? total : max(max(c[0], c[1]), max(c[2], c[3]));                               // 2048: every clonality 1.0
#else
                const SiteCall sc = call_level(a, thr_lds, c, total, ref_base, false);
                const uint32_t mx = max(max(c[0], c[1]), max(c[2], c[3]));
#endif
                if (mx == total) cl = 1.0f; else defer = true;
                if (defer) entry |= 1u << 13;
                if (a.clon_list && defer) atomicAdd(&scratch[S_NCLON], 1u);     // the list of clonalities other than 1.0: written below, once the window has its slots
                if (sc.snp != -1) {
                    entry |= 1u << 14;
                    atomicAdd(&scratch[S_ROWS], 1u);
                    if (sc.morphia >= 2) {
                        const uint32_t mask = (1u << sc.snp) | (1u << sc.var);
                        entry |= (atomicAdd(&scratch[S_SITES], 1u) + 1u) << 17;
                        if (linkage) {
                            maskl[p] = (uint8_t)mask;
                            slabc[p] = atomicAdd(&scratch[S_NAO], masked_sum(c, mask));
                        }
                    }
                }
            }
            // clonTR is gated on rarefied_coverage alone (snv_utilities.py:100-102), also below min_cov
            if (a.min_cov_r > 0 && (int64_t)total >= (int64_t)a.min_cov_r) { entry |= 1u << 15; if (a.rare) atomicAdd(&scratch[S_NRARE], 1u); }
            if (qi >= 0) {
                queue[qi] = entry;
                if (use_fl && entry != (uint32_t)p) {
                    const uint32_t k = atomicAdd(&scratch[S_ENT_TOT], 1u);      // (S_ENT_TOT: a word of the mm kernel, free here)
                    if (k < 1024u) flist[k] = (uint16_t)qi;
                }
            }
            else if (entry != (uint32_t)p) queue[atomicAdd(&scratch[S_NQ], 1u)] = entry;
            if (!defer && st_ok && a.clon) a.clon[gpos] = cl;      // (a lean slot keeps no dense clonality array)
        };
        uint32_t tot_pk[4] = {0, 0, 0, 0};      // stripe path: this thread's eight coverages, 16 bits each (kept for the window's 16-bit row)
        if (PKL && stripe) {
            // ---- stripe path (round 6; packed rows, no count table wanted): materialise and first epilogue pass in one ----
            // A thread owns EIGHT consecutive positions.  Their coverage is the prefix sum of the difference row minus the skipped columns -- it
            // needs neither the exception counters nor the reference -- and coverage is all a position below min_cov (and rarefied_coverage)
            // hands back: the 4-bit plane leaves as one 32-bit word a thread (cov16: one 16-byte store, cov8: 8 bytes), and only the positions
            // that do reach min_cov are COMPACTED into the queue (wave prefix sum of their number, one LDS atomic a wave) and get the reference
            // base's count, the SNV call and the rest from a second pass with every lane busy.  At metagenome depth (70 % of a C5 batch's
            // positions lie below min_cov) that pass runs over a third of the window.  (profile_utilities.py:288-295, snv_utilities.py:85-104)
            const int lane = tid & 63;
            const int p0 = tid * 8;
            const bool act = p0 < W;
            uint32_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int32_t sum = 0;
            if (act) {
                const uint4 xa = *reinterpret_cast<const uint4 *>(dlt + p0), xb = *reinterpret_cast<const uint4 *>(dlt + p0 + 4);
                x[0] = xa.x; x[1] = xa.y; x[2] = xa.z; x[3] = xa.w; x[4] = xb.x; x[5] = xb.y; x[6] = xb.z; x[7] = xb.w;
#pragma unroll
                for (int k = 0; k < 8; k++) sum += (int32_t)x[k] >> 16;
            }
            const int32_t inc = (int32_t)wave_scan_incl((uint32_t)sum);
            if (lane == 63) wtot[tid >> 6] = (uint32_t)inc;
            __syncthreads();                    // (from here on nobody reads the difference row: it becomes the queue)
            ISX_TS(3);
            // (branch-free on purpose: written with a test per position the compiler made sixteen branches of it, each re-reading n_pos from
            //  the kernarg segment.  The arguments this phase needs are read once.)
            const uint32_t n_pos = a.n_pos, sat_thr = a.sat_thr;
            const int32_t need = a.min_cov_r > 0 ? min(a.min_cov, a.min_cov_r) : a.min_cov;
            // what the waves before this one add up to: every lane reads one of the sixteen totals, a prefix sum inside each row of 16 lanes
            uint32_t wv = wtot[lane & 15];
            wv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv, 0x111, 0xF, 0xF, false);      // row_shr:1
            wv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv, 0x112, 0xF, 0xF, false);      // row_shr:2
            wv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv, 0x114, 0xF, 0xF, false);      // row_shr:4
            wv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv, 0x118, 0xF, 0xF, false);      // row_shr:8
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            int32_t run = inc - sum + (wave ? __builtin_amdgcn_readlane((int)wv, wave - 1) : 0);
            const uint32_t g0 = w0 + (uint32_t)p0;
            const uint32_t nv = (act && g0 < n_pos) ? min(n_pos - g0, 8u) : 0u;        // positions of this stripe that exist (8 but for the batch's last stripe and the lanes beyond the window)
            uint32_t tot[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                run += (int32_t)x[k] >> 16;
                tot[k] = (uint32_t)run - (x[k] & 0xFFFFu);
            }
            if (nv < 8u) {                      // (a position that does not exist never goes on, whatever min_cov is)
#pragma unroll
                for (int k = 0; k < 8; k++) if ((uint32_t)k >= nv) tot[k] = 0x80000000u;
            }
            bool go[8];
#pragma unroll
            for (int k = 0; k < 8; k++) go[k] = (int32_t)tot[k] >= need;
            if (nv < 8u) {
#pragma unroll
                for (int k = 0; k < 8; k++) if ((uint32_t)k >= nv) tot[k] = 0u;
            }
            uint32_t nib = 0, tmax = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                tmax = max(tmax, tot[k]);
                nib |= min(tot[k], 15u) << (4 * k);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) tot_pk[k] = tot[2 * k] | (tot[2 * k + 1] << 16);
            if (nv) {
                const bool whole = nv == 8u;    // (the batch's last stripe is stored position by position: the tables end at n_pos)
                // (every loop over the eight is unrolled: an index that is not a constant would send the array to scratch memory)
                uint8_t *const cov4 = a.cov4, *const cov8 = a.cov8;
                uint16_t *const cov16 = a.cov16;
                float *const clon = a.clon;
                if (cov4) {
                    if (whole) *reinterpret_cast<uint32_t *>(cov4 + (g0 >> 1)) = nib;
                    else {
#pragma unroll
                        for (int k = 0; k < 8; k += 2) if ((uint32_t)k < nv) cov4[(g0 + (uint32_t)k) >> 1] = (uint8_t)(nib >> (4 * k));
                    }
                }
                if (cov16) {
                    if (whole) *reinterpret_cast<uint4 *>(cov16 + g0) = make_uint4(tot_pk[0], tot_pk[1], tot_pk[2], tot_pk[3]);
                    else {
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((uint32_t)k < nv) cov16[g0 + (uint32_t)k] = (uint16_t)tot[k];
                    }
                }
                if (cov8) {
                    uint32_t b0 = 0, b1 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) { b0 |= min(tot[k], 255u) << (8 * k); b1 |= min(tot[4 + k], 255u) << (8 * k); }
                    if (whole) *reinterpret_cast<uint2 *>(cov8 + g0) = make_uint2(b0, b1);
                    else {
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((uint32_t)k < nv) cov8[g0 + (uint32_t)k] = (uint8_t)min(tot[k], 255u);
                    }
                }
                if ((cov16 || cov8 || cov4) && tmax >= sat_thr) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (tot[k] >= sat_thr) {
                            const uint32_t at = cur_add(a, CUR_SAT, 1u);
                            if (at < a.cap_sat) a.sat[at] = make_uint2(g0 + (uint32_t)k, tot[k]);
                        }
                    }
                }
                if (clon) {                     // dense clonality array: NaN below min_cov; the positions of the queue are written again by the second pass
                    const float qn = __builtin_nanf("");
                    if (whole) {
                        reinterpret_cast<float4 *>(clon + g0)[0] = make_float4(qn, qn, qn, qn);
                        reinterpret_cast<float4 *>(clon + g0)[1] = make_float4(qn, qn, qn, qn);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) if ((uint32_t)k < nv) clon[g0 + (uint32_t)k] = qn;
                    }
                }
                if (cov4 && tmax > 15u) scratch[S_COVX] = 1u;       // (every writer writes the same 1)
            }
            // compaction of the positions that go on: the lanes whose position k does are one ballot; a lane's slot is the wave's base (one
            // LDS atomic a wave) + the ballots before k + the lanes below it in ballot k.  (The queue is then column-major inside a wave's
            // 512 positions; nothing downstream depends on the order of the queue.)
            uint64_t bal[8];
            uint32_t wn = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { bal[k] = __ballot(go[k]); wn += (uint32_t)__popcll(bal[k]); }
            if (wn) {                           // wave-uniform
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(&scratch[S_NQ], wn);
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    if (go[k]) queue[__builtin_amdgcn_mbcnt_hi((uint32_t)(bal[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[k], qb))] = (uint32_t)(p0 + k) | (tot[k] << 16);
                    qb += (uint32_t)__popcll(bal[k]);
                }
            }
            __syncthreads();
            ISX_TS(4);
            ISX_ARGS_FRESH();
            const uint32_t nq1 = scratch[S_NQ];
            use_fl = nq1 + 512u <= (uint32_t)S;
            for (uint32_t q = tid; q < nq1; q += nthr) {
                const uint32_t e = queue[q];
                const int p = (int)(e & 0x1FFFu);
                uint32_t a01 = cnt[p], a23 = cnt[S + p];
                const uint32_t r = refl[p];
                if (r < 4u) {                   // the reference base's count: what was observed and is no exception
                    const uint32_t v = (e >> 16) - ((a01 & 0xFFFFu) + (a01 >> 16) + (a23 & 0xFFFFu) + (a23 >> 16));
                    if (r >> 1) { a23 += v << (16 * (r & 1u)); cnt[S + p] = a23; }
                    else { a01 += v << (16 * (r & 1u)); cnt[p] = a01; }
                }
                const uint32_t c[4] = {a01 & 0xFFFFu, a01 >> 16, a23 & 0xFFFFu, a23 >> 16};
                site_pass1(p, w0 + (uint32_t)p, c, c[0] + c[1] + c[2] + c[3], (int)r, (int)q, true, true);
            }
        } else {
        int ep_it = 0;
        for (int p = tid; p < ((dbg & 2) ? 0 : W); p += nthr, ep_it++) {
            const uint32_t gpos = w0 + p;
            if (gpos >= a.n_pos) break;
            uint32_t c[4];
            ld4(p, c);
            const uint32_t total = c[0] + c[1] + c[2] + c[3];
#ifdef ISX_TUNING       // ablations of the epilogue (tools/tune_reads.py)
            const bool st_ok = !(dbg & 128);    // 128: no global stores of the position-sized tables
            const bool call_ok = !(dbg & 256);  // 256: no SNV call / clonality (as if below min_cov)
#else
            constexpr bool st_ok = true, call_ok = true;
#endif
            if (a.counts && st_ok) a.counts[gpos] = make_uint4(c[0], c[1], c[2], c[3]);
            if (PKL && a.cov4 && st_ok) {       // 4-bit plane: a pair of lanes holds the two positions of a byte
                const uint32_t nib = min(total, 15u), other = pair_other(nib);          // (a lane beyond n_pos has left the loop: it reads as 0)
                if (!(tid & 1)) a.cov4[gpos >> 1] = (uint8_t)(nib | (other << 4));
            }
            if ((a.cov16 || a.cov8 || (PKL && a.cov4)) && st_ok) { // shrunk hand-back of a pipe slot: coverage alone, 2 (or 1) bytes per position,
                if (a.cov16) a.cov16[gpos] = (uint16_t)min(total, 65535u);      // exact values of the few positions beyond that in a list
                if (a.cov8) a.cov8[gpos] = (uint8_t)min(total, 255u);
                if (total >= a.sat_thr) {
                    const uint32_t k = cur_add(a, CUR_SAT, 1u);
                    if (k < a.cap_sat) a.sat[k] = make_uint2(gpos, total);
                }
            }
            const int ref_base = PKL ? (int)refl[p] : (ep_it == 0 ? ref_raw[0] : (ep_it == 1 ? ref_raw[1] : ref_at(a, gpos)));
            site_pass1(p, gpos, c, total, ref_base, -1, st_ok, call_ok);
        }
        }
        __syncthreads();
        ISX_TS(5);
        ISX_ARGS_FRESH();
#ifdef ISX_TUNING
        if (dbg & 512) { win_it++; __syncthreads(); continue; }       // 512: nothing after the first epilogue pass (no table slots, no rows)
#endif
        const uint32_t nq = scratch[S_NQ], nrows = scratch[S_ROWS], nsites = scratch[S_SITES], nao = scratch[S_NAO];
        const uint32_t nfl = scratch[S_ENT_TOT];
        const bool fl = PKL && use_fl && nfl <= 1024u;       // (uniform) the loops below walk the listed queue entries only
        const uint32_t nit = fl ? nfl : nq;
        const uint32_t nrare = a.rare ? scratch[S_NRARE] : 0u;
        const uint32_t nclon = a.clon_list ? scratch[S_NCLON] : 0u;
        const uint32_t covx = PKL ? scratch[S_COVX] : 0u;                   // 4-bit coverage plane: this window also writes its 16-bit row
        {   // the window's table slots: six lanes of the LAST wave (the one least likely to hold queue entries), one atomic instruction --
            // that wave waits for the round trip, the others go on with what needs no slot yet
            const int k = tid - (nthr - 64);
            if (k >= 0 && k < 6) {
                const uint32_t n = k == 0 ? nrows : (k == 1 ? nsites : (k == 2 ? nao : (k == 3 ? nrare : (k == 4 ? nclon : (covx ? 1u : 0u)))));
                const int which = k == 0 ? CUR_SNV : (k == 1 ? CUR_SITES : (k == 2 ? CUR_AO : (k == 3 ? CUR_RARE : (k == 4 ? CUR_CLON : CUR_COVX))));
                const int slot_w = k == 0 ? S_ROW_BASE : (k == 1 ? S_SITE_BASE : (k == 2 ? S_AO_BASE : (k == 3 ? S_RARE_BASE : (k == 4 ? S_CLON_BASE : S_COVX_BASE))));
                if (n) scratch[slot_w] = atomicAdd(&a.cursors[which], n) - a.base[which];
            }
        }
        bool ok = nrows != 0;
        uint32_t ao_base = 0;
        auto slots_ok = [&](uint32_t row_base, uint32_t site_base) {    // (after the barrier that makes the slots visible)
            ao_base = scratch[S_AO_BASE];
            if (ok && (row_base + nrows > a.cap_snv || site_base + nsites > a.cap_sites || ao_base + nao > a.cap_ao)) {
                if (tid == 0) flag_or(a, row_base + nrows > a.cap_snv ? ISX_FLAG_CAP_SNV
                                                : (site_base + nsites > a.cap_sites ? ISX_FLAG_CAP_SITES : ISX_FLAG_CAP_AO));
                ok = false;
            }
            if (a.win_rec && tid == 0) {
                // where this window's rows / sites / list entries went and how many there are: the tables are filled in the order the
                // windows reach their cursors, k_win_gather puts them into position order afterwards (windows are position ranges, a
                // window's entries lie together) -- instead of sorting the tables
                uint32_t *wr = a.win_rec + 8 * (size_t)w;
                const uint32_t cb = scratch[S_CLON_BASE], rb = scratch[S_RARE_BASE];
                wr[0] = row_base; wr[1] = ok ? nrows : 0u;
                wr[2] = site_base; wr[3] = ok ? nsites : 0u;
                wr[4] = cb; wr[5] = (nclon && cb + nclon <= a.cap_clon) ? nclon : 0u;
                wr[6] = rb; wr[7] = (nrare && rb + nrare <= a.cap_rare) ? nrare : 0u;
            }
        };
        auto covx_row = [&]() {
            if (PKL && covx) {
                const uint32_t k = scratch[S_COVX_BASE];
                const uint32_t at = k * (uint32_t)W;
                if (at + (uint32_t)W <= a.cap_cov_rows) {
                    if (stripe) {                   // the stripe path still holds its eight coverages
                        if (8 * tid < W) *reinterpret_cast<uint4 *>(a.cov_rows + at + 8u * (uint32_t)tid) = make_uint4(tot_pk[0], tot_pk[1], tot_pk[2], tot_pk[3]);
                    } else
                    for (int p = tid; p < W; p += nthr) {
                        uint32_t c[4];
                        ld4(p, c);
                        a.cov_rows[at + (uint32_t)p] = (uint16_t)min(c[0] + c[1] + c[2] + c[3], 65535u);
                    }
                }
                if (tid == 0) a.cov_row_win[k] = (uint32_t)w;
            }
        };
        if (fl && nit <= (uint32_t)nthr) {
            // ---- a shallow window's tail in one step: a thread has at most ONE listed queue entry.  What needs no table slot -- the fp64
            //      clonality (snv_utilities.py:225-231), the rarefied one (:233-247), the SNV row's call and class (:107-133) -- is computed
            //      while the slot atomics are under way; ONE barrier; then the stores.  (The loops below pay the atomics' round trip and
            //      two barriers before they start.) ----
            const bool have = (uint32_t)tid < nit;
            const uint32_t e = have ? queue[(uint32_t)flist[tid]] : 0u;
            const int p = (int)(e & 0x1FFFu);
            const uint32_t gpos = w0 + (uint32_t)p;
            uint32_t c[4] = {0, 0, 0, 0};
            if (have) ld4(p, c);
            const uint32_t total = c[0] + c[1] + c[2] + c[3];
            const bool f_clon = (e >> 13) & 1u, f_row = (e >> 14) & 1u, f_rare = a.min_cov_r > 0 && ((e >> 15) & 1u);
            float v_clon = 0.f, v_rare = 0.f;
            if (f_clon) v_clon = (float)clonality(c, total);
            if (f_rare) v_rare = rarefied_clonality(a, c, gpos, 0);
            const int ref_base = f_row ? (int)refl[p] : 4;
            SiteCall sc{-2, 0, 0, 0};
            if (f_row) sc = call_level(a, nullptr, c, total, ref_base, true);
            __syncthreads();                    // the slots are there
            ISX_TS(6);
            ISX_ARGS_FRESH();
            const uint32_t row_base = scratch[S_ROW_BASE], site_base = scratch[S_SITE_BASE];
            slots_ok(row_base, site_base);
            covx_row();
            if (f_clon) {
                const uint32_t clon_base = scratch[S_CLON_BASE];
                if (a.clon) a.clon[gpos] = v_clon;
                if (nclon && clon_base + nclon <= a.cap_clon) a.clon_list[clon_base + atomicAdd(&scratch[S_CLON_RANK], 1u)] = make_uint2(gpos, __float_as_uint(v_clon));
            }
            if (f_rare) {
                const uint32_t rare_base = scratch[S_RARE_BASE];
                if (a.clon_r) a.clon_r[gpos] = v_rare;
                if (nrare && rare_base + nrare <= a.cap_rare) a.rare[rare_base + atomicAdd(&scratch[S_RARE_RANK], 1u)] = make_uint2(gpos, __float_as_uint(v_rare));
            }
            if (f_row && ok) {
                const uint32_t my_row = atomicAdd(&scratch[S_ROW_RANK], 1u);
                isx_snv r;
                r.gpos = gpos; r.mm = 0;
                r.con_base = (uint8_t)sc.snp; r.var_base = (uint8_t)sc.var;
                r.allele_count = (uint8_t)sc.morphia; r.cls = (uint8_t)sc.cls;
                r.cryptic = 0;
                r.ref_base = (uint8_t)ref_base;
                r.cnt[0] = c[0]; r.cnt[1] = c[1]; r.cnt[2] = c[2]; r.cnt[3] = c[3];
                a.snv[row_base + my_row] = r;
                const uint32_t ss = e >> 17;
                if (ss) {
                    isx_site st;
                    st.gpos = gpos; st.entry_off = row_base + my_row; st.n_levels = 1;
                    st.mask = (uint8_t)((1u << sc.snp) | (1u << sc.var)); st.pad = 0;
                    a.sites[site_base + ss - 1] = st;
                }
            }
            ISX_TS(7);
        } else {
        if (nclon) __syncthreads();             // uniform: the list entries below need the window's base
        ISX_TS(6);
        // ---- deferred clonalities (snv_utilities.py:225-231), densely packed; the sparse clonality list ----
        {
            const uint32_t clon_base = scratch[S_CLON_BASE];
            const bool list = nclon && clon_base + nclon <= a.cap_clon;      // else the host reads the dense array
            for (uint32_t q = tid; q < nit; q += nthr) {
                const uint32_t e = queue[fl ? (uint32_t)flist[q] : q];
                if (!(e & (1u << 13))) continue;
                const int p = (int)(e & 0x1FFFu);
                uint32_t c[4];
            ld4(p, c);
                const float v = (float)clonality(c, c[0] + c[1] + c[2] + c[3]);
                if (a.clon) a.clon[w0 + p] = v;
                if (list) a.clon_list[clon_base + atomicAdd(&scratch[S_CLON_RANK], 1u)] = make_uint2(w0 + p, __float_as_uint(v));
            }
        }
        if (nrows | nrare | covx) __syncthreads();     // uniform: scratch bases from the atomics above
        covx_row();
        ISX_TS(7);
        ISX_ARGS_FRESH();
        if (a.min_cov_r > 0) {                  // rarefied clonality (snv_utilities.py:233-247), own loop: fewer live registers
            const uint32_t rare_base = scratch[S_RARE_BASE];
            const bool list = nrare && rare_base + nrare <= a.cap_rare;      // else the host reads the dense array
            for (uint32_t q = tid; q < nit; q += nthr) {
                const uint32_t e = queue[fl ? (uint32_t)flist[q] : q];
                if (!(e & (1u << 15))) continue;
                const int p = (int)(e & 0x1FFFu);
                uint32_t c[4];
            ld4(p, c);
                const float v = rarefied_clonality(a, c, w0 + p, 0);
                if (a.clon_r) a.clon_r[w0 + p] = v;
                if (list) a.rare[rare_base + atomicAdd(&scratch[S_RARE_RANK], 1u)] = make_uint2(w0 + p, __float_as_uint(v));
            }
        }
        // ---- SNV rows / SNP sites (snv_utilities.py:107-133) ----
        const uint32_t row_base = scratch[S_ROW_BASE], site_base = scratch[S_SITE_BASE];
        slots_ok(row_base, site_base);
        for (uint32_t q0 = 0; q0 < (ok ? nit : 0u); q0 += nthr) {
            const uint32_t q = q0 + tid;
            const uint32_t e = q < nit ? queue[fl ? (uint32_t)flist[q] : q] : 0u;
            if (!((e >> 14) & 1u)) continue;
            const uint32_t my_row = atomicAdd(&scratch[S_ROW_RANK], 1u);
            const int p = (int)(e & 0x1FFFu);
            const uint32_t gpos = w0 + p;
            uint32_t c[4];
            ld4(p, c);
            const uint32_t total = c[0] + c[1] + c[2] + c[3];
            const int ref_base = PKL ? (int)refl[p] : (int)ref_at(a, gpos);
            const SiteCall sc = call_level(a, nullptr, c, total, ref_base, true);
            isx_snv r;
            r.gpos = gpos; r.mm = 0;
            r.con_base = (uint8_t)sc.snp; r.var_base = (uint8_t)sc.var;
            r.allele_count = (uint8_t)sc.morphia; r.cls = (uint8_t)sc.cls;
            r.cryptic = 0;                      // a single mm level cannot turn cryptic (snv_utilities.py:135-140)
            r.ref_base = (uint8_t)ref_base;
            r.cnt[0] = c[0]; r.cnt[1] = c[1]; r.cnt[2] = c[2]; r.cnt[3] = c[3];
            a.snv[row_base + my_row] = r;
            const uint32_t ss = e >> 17;
            if (ss) {
                isx_site st;
                st.gpos = gpos; st.entry_off = row_base + my_row; st.n_levels = 1;      // linkage reads the site's counts from its SNV row
                st.mask = (uint8_t)((1u << sc.snp) | (1u << sc.var)); st.pad = 0;
                a.sites[site_base + ss - 1] = st;
            }
        }
        }
        ISX_TS(8);
        ISX_ARGS_FRESH();
        if (linkage) {
            if (ok && nao) {
                __syncthreads();                // every wave is done with cnt: it becomes the allele pass's stage
                if (SEGS) allele_pass_segs(a, cur_lo, cur_hi, w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
                else if (DREC) allele_pass_delta(a, cur_lo, cur_hi, w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
                else allele_pass(a, cur_lo << (RSH - 1), cur_hi << (RSH - 1), w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
            }
            ISX_TS(9);
            rng_next = rngl[win_it & 63];
            prefetch_window(w + grid);
        }
        win_it++;
        // the zeroing + barrier at the top of the next window protect cnt / queue / scratch
        __syncthreads();
        ISX_TS(10);
    }
}
#undef a
#undef ISX_ARGS_FRESH

// ---------------------------------------------------------------------------------------------
// k_pileup_mm: n_mm_bins > 1 (mm profiling on, the reference's default).  Persistent workgroups
// like k_pileup_dense.  PACKED: two u16 counters per LDS word ((A,C) and (T,G) of a level) -- legal
// when no window streams >= 65536 records (no counter can overflow), halves the LDS per position
// and so doubles the window (less over-scan).  All table slots (entries, SNV rows, SNP sites,
// allele slabs) come from ONE global atomic each per window: pass A runs update_snp_table's level
// loop to size everything, pass B runs it again to write.
// LDS: cnt[M*(PACKED?2:4)][W] | pres[ceil(M/32)][W] | scratch[16] | queue[QCAP][2] | rowq[RQCAP][4] | thr_lds |
//      (linkage) slabc[W] | maskl[W bytes]
// pres = levels made present by a non-ACGT base only (profile_utilities.py:279-285 creates
// table[mm] before the KeyError), a presence the SNV loop must see.
// ---------------------------------------------------------------------------------------------
// SPARSE (pipe slots, M <= 32): the levels also go home as a per-position level mask + one coverage byte (or two) per present level
// + the list of clonalities other than 1.0 (PileupArgs::lev_*); level indices are position-major inside a window (a block-wide
// prefix sum of the positions' level counts), the window's first index comes from one global atomic.  The 32-byte entries are then a
// flat table indexed by level (a.entries; NULL in a lean slot: nothing but what travels home is written).
// DREC (round 6): the stream is reference-delta records with the pair's mm level in bits 24..30 of a segment's header.  As in
// k_pileup_dense a level's coverage is counted by DIFFERENCE (+1 at a segment's first column, -1 behind its last, into the level's
// difference row), one atomic per skipped column and one per exception; a materialise phase -- a thread owns PT consecutive
// positions, per level a wave prefix sum of the differences -- then rebuilds the reference base's count of every (position, level)
// and the position's level mask, after which the counters are what the other record formats leave behind.  ~17 LDS atomics per
// 150-base read instead of 135.  LDS: cnt | aux[M][W] (PACKED: skipped columns in the low half, coverage differences in the
// wrapping high half; else two rows a level) | pres | ...
template <bool PACKED, bool LINKAGE, bool COMPACT, bool SEGS = false, bool SPARSE = false, bool DREC = false>    // SEGS: the read-segment stream (COMPACT is true then)
__global__ void __launch_bounds__(1024) k_pileup_mm(const PileupArgs a_kernarg)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    // (arguments read from the kernarg segment phase by phase, as in k_pileup_dense)
    typedef const __attribute__((address_space(4))) PileupArgs KernArgs;
    KernArgs *kargs = (KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#define a (*(const PileupArgs *)kargs)
    int tid = threadIdx.x;                      // (an unknown again at every phase boundary, like the arguments: see k_pileup_dense)
    const int nthr = blockDim.x;
#define ISX_ARGS_FRESH() asm volatile("" : "+s"(kargs), "+v"(tid))
    publish_previous(a, tid);
    const int W = a.W, M = a.M;
#ifdef ISX_TUNING
    int ts_w = -1;                              // timeline of workgroup 0 (debug bit 4096, tools/timeline_mm.py)
#endif
    const int n_cnt = M * (PACKED ? 2 : 4) * W + (DREC ? M * (PACKED ? 1 : 2) * W : 0);
    const int pres_words = (M + 31) >> 5;
    uint32_t *cnt = lds;
    uint32_t *aux = lds + M * (PACKED ? 2 : 4) * W;      // DREC: per level the skipped-columns / coverage-difference row(s)
    uint32_t *pres = lds + n_cnt;
    uint32_t *scratch = pres + pres_words * W;
    uint32_t *queue = scratch + S_N;                    // [QCAP][2]: entry index, flags | (mm << 16) | p
    uint16_t *thr_lds = reinterpret_cast<uint16_t *>(queue + 2 * a.qcap + 4 * a.rqcap);
    uint32_t *slabc = queue + 2 * a.qcap + 4 * a.rqcap + THR_LDS / 2;
    uint8_t *maskl = reinterpret_cast<uint8_t *>(slabc + W);
    const uint32_t QCAP = (uint32_t)a.qcap;
    constexpr bool linkage = LINKAGE;            // compile-time: the linkage-off kernels carry none of the allele pass
    const int grid = gridDim.x, per = grid >> 3;
    const int slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);      // consecutive windows share an XCD's L2
    const u32x4 *rec4 = reinterpret_cast<const u32x4 *>(DREC ? (const void *)a.drec : (SEGS ? (const void *)a.seg : (COMPACT ? (const void *)a.rec32 : (const void *)a.rec)));
    constexpr int RSH = COMPACT ? 2 : 1;        // record index -> 16-byte load index (see k_pileup_dense)
    {   // once per workgroup: folded thresholds of the low coverages
        const int n = min(THR_LDS, a.lut_n);
        for (int i = tid; i < n; i += nthr) thr_lds[i] = a.thr[i];
    }

    auto rd = [&](int m, int k, int p) -> uint32_t {    // count of base k at level m, position p
        if (PACKED) {
            const uint32_t w = cnt[(m * 2 + (k >> 1)) * W + p];
            return (k & 1) ? (w >> 16) : (w & 0xFFFFu);
        }
        return cnt[(m * 4 + k) * W + p];
    };

    u32x4 v[4];
    uint32_t gb[4] = {0, 0, 0, 0};
    uint32_t lo = 0, hi = 0;
    uint32_t my_entries = 0;                    // thread 0: entries of all windows of this workgroup
    // same load scheme as k_pileup_dense: wave-uniform bound, scalar base + lane offset, two half-rounds in flight
    auto issue_one = [&](int u, uint32_t i0) {
        const uint32_t j = i0 + tid + u * nthr;
        const uint32_t jw = __builtin_amdgcn_readfirstlane(j);
        if (COMPACT ? jw < hi : j < hi) {
            const uint64_t ub = reinterpret_cast<uint64_t>(rec4 + (i0 + (uint32_t)(u * nthr)));
            typedef __attribute__((address_space(1))) const u32x4 gvec;
            const gvec *sb = reinterpret_cast<const gvec *>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32)) << 32) |
                                                            (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ub));
            v[u] = __builtin_nontemporal_load(sb + (uint32_t)tid);
            if (COMPACT) gb[u] = a.gbase[__builtin_amdgcn_readfirstlane(j >> 6)];
        } else if (DREC) { v[u].x = v[u].y = v[u].z = v[u].w = 0u; }       // (length 0: nothing to count)
        else if (COMPACT) { v[u].x = v[u].y = v[u].z = v[u].w = ISX_PAD32; }
        else { v[u].x = ISX_SENTINEL; v[u].y = 0; v[u].z = ISX_SENTINEL; v[u].w = 0; }
    };
    auto prefetch_window = [&](int wn) {
        lo = hi = 0;
        if (wn < a.n_win) {
            const uint2 rng = a.win_range[wn];
            if (DREC) { lo = rng.x << 1; hi = rng.y << 1; } else if (SEGS) { lo = rng.x << 2; hi = rng.y << 2; } else { lo = rng.x >> RSH; hi = rng.y >> RSH; }
            if (lo < hi) { issue_one(0, lo); issue_one(1, lo); }
        }
    };
    prefetch_window(slot);

    for (int w = slot; w < a.n_win; w += grid) {
        const uint32_t w0 = (uint32_t)w * (uint32_t)W;
        const uint32_t cur_lo = lo, cur_hi = hi;
        const int dbg = a.debug_mode;           // ablation switches (tools/ablate_mm.py), 0 in production
#ifdef ISX_TUNING
        ++ts_w;
#endif
        ISX_TS(0);
        {   // zero the window
            uint4 *z = reinterpret_cast<uint4 *>(lds);
            const int n4 = (n_cnt + pres_words * W) >> 2;   // W is a multiple of 64
            for (int i = tid; i < n4; i += nthr) z[i] = make_uint4(0, 0, 0, 0);
            if (linkage) {
                uint4 *zm = reinterpret_cast<uint4 *>(maskl);
                for (int i = tid; i < (W >> 4); i += nthr) zm[i] = make_uint4(0, 0, 0, 0);
            }
            if (tid < S_N) scratch[tid] = 0;
        }
        __syncthreads();
        ISX_TS(1);
        ISX_ARGS_FRESH();

        // DREC: the reference codes of the materialise phase's positions (a lane's position of chunk wave / wave + waves), loaded
        // here so that their latency hides behind the stream
        uint32_t rcode[2] = {4u, 4u};
        if (DREC) {
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const int p = 64 * ((tid >> 6) + sl * (nthr >> 6)) + (tid & 63);
                const uint32_t gp = w0 + (uint32_t)p;
                if (p < W && gp < a.n_pos) rcode[sl] = (uint32_t)ref_at(a, gp);
            }
        }
        // ---- get_base_counts_mm over the window's slice of the stream ----
        uint32_t bad_mm = 0;
        auto count_slot = [&](int u) {
            const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (DREC) {
                // reference-delta records (k_pileup_dense's count_slot, with the level's rows): a pair of lanes holds one 32-byte record --
                // ONE segment with its plane of skipped columns (full), or TWO segments without skipped columns, a lane each (dual)
                const uint32_t odd = (uint32_t)tid & 1u;
                const uint32_t hdr0 = pair_first(x[0]);
                const bool dual = (hdr0 >> 31) != 0u;
                const uint32_t hdr = dual ? x[0] : hdr0;
                const uint32_t len = (hdr >> 16) & 0xFFu, mm = (hdr >> 24) & 0x7Fu;
                if (len == 0) return;
                if (mm >= (uint32_t)M) { bad_mm = 1; return; }
                const int32_t s = (int32_t)(gb[u] + (hdr & 0xFFFFu) - w0);
                const uint32_t uW = (uint32_t)W;
                uint32_t *dlt = aux + __umul24(PACKED ? mm : 2u * mm + 1u, uW);          // coverage differences (PACKED: high half of the level's aux word)
                uint32_t *skp = aux + __umul24(PACKED ? mm : 2u * mm, uW);               // skipped columns (PACKED: low half)
                if (dual || !odd) {
                    const int32_t hi_c = s + (int32_t)len;
                    if (hi_c > 0 && s < W) {
                        atomicAdd(&dlt[s < 0 ? 0 : s], PACKED ? 0x00010000u : 1u);
                        if (hi_c < W) atomicAdd(&dlt[hi_c], PACKED ? 0xFFFF0000u : 0xFFFFFFFFu);
                    }
                }
                // a full record's skip plane: the first lane holds columns 0..63 (words 1, 2), the second 64..159 (words 4..6)
                uint32_t sk[3] = {odd ? x[0] : x[1], odd ? x[1] : x[2], odd ? x[2] : 0u};
                if (dual) sk[0] = sk[1] = sk[2] = 0;
                const int32_t c0 = s + (odd ? 64 : 0);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int32_t r = c0 + 32 * k;
                    uint32_t bits = skip_bits_in_window(sk[k], r, uW);
                    while (__ballot(bits != 0)) {                               // wave-uniform
                        if (bits != 0) atomicAdd(&skp[r + (int32_t)__builtin_ctz(bits)], 1u);
                        bits &= bits - 1u;                                      // 0 stays 0
                    }
                }
                // exceptions (the first lane of a full record: word 3; a dual half: words 2, 3).  An exception at a SKIPPED column of a full
                // record is a base that is not A/C/T/G: nothing to count, the level is present there (profile_utilities.py:279-285)
                const uint32_t o4 = pair_other(x[0]), o5 = pair_other(x[1]), o6 = pair_other(x[2]);      // (the second lane's skip words)
                const uint32_t exw[2] = {dual ? x[2] : (odd ? ISX_DREC_NO_EXC : x[3]), dual ? x[3] : ISX_DREC_NO_EXC};
                const int nf = __ballot(exw[1] != ISX_DREC_NO_EXC) ? 6 : 3;       // (the second word is empty in nearly every wave)
#pragma unroll
                for (int f = 0; f < 6; f++) {
                    if (f == 3 && nf == 3) break;               // wave-uniform
                    const uint32_t ex = exw[f / 3];
                    const uint32_t off = (ex >> (10 * (f % 3))) & 0xFFu, code = (ex >> (10 * (f % 3) + 8)) & 3u;
                    const uint32_t rel = (uint32_t)(s + (int32_t)off);
                    if (off < len && rel < uW) {                 // (an empty field has offset 255)
                        bool marker = false;
                        if (!dual) {
                            const uint32_t wsel = off >> 5;
                            const uint32_t skw = wsel == 0 ? x[1] : (wsel == 1 ? x[2] : (wsel == 2 ? o4 : (wsel == 3 ? o5 : o6)));
                            marker = ((skw >> (off & 31u)) & 1u) != 0u;
                        }
                        if (marker) atomicOr(&pres[__umul24(mm >> 5, uW) + rel], 1u << (mm & 31));
                        else if (PACKED) atomicAdd(&cnt[__umul24(mm * 2u + (code >> 1), uW) + rel], 1u << (16 * (code & 1u)));
                        else atomicAdd(&cnt[__umul24(mm * 4u + code, uW) + rel], 1u);
                    }
                }
                return;
            }
            if (SEGS) {
                // read segments (see k_pileup_dense): lane q of a quad walks the words of its quarter of the record; the mm
                // level is the record's, so a word's ten counters are ten consecutive columns of the level's rows
                const uint32_t q = (uint32_t)tid & 3u;
                const uint32_t hdr = quad_first(x[0]);
                const uint32_t mm = hdr >> 24;
                if (((hdr >> 16) & 0xFFu) == 0) return;                 // padding record / a slot the wave did not load
                if (mm >= (uint32_t)M) { bad_mm = 1; return; }
                const int32_t r0 = (int32_t)(gb[u] + (hdr & 0xFFFFu) - w0) + (int32_t)(q * 40u) - 10;
                const uint32_t rowb = __umul24(mm * (PACKED ? 2u : 4u), (uint32_t)W);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t r = r0 + 10 * k;
                    const uint32_t wv = (k == 0 && q == 0) ? SEG_SKIPW : x[k];
                    if (wv == SEG_SKIPW || (uint32_t)(r + 9) >= (uint32_t)(W + 9)) continue;
#pragma unroll
                    for (int j = 0; j < 10; j++) {
                        const uint32_t code = __builtin_amdgcn_ubfe(wv, 3 * j, 3);
                        const uint32_t rel = (uint32_t)(r + j);
                        if (rel >= (uint32_t)W) continue;
                        if (code < 4u) {
                            if (PACKED) atomicAdd(&cnt[rowb + __umul24(code >> 1, (uint32_t)W) + rel], 1u << (16 * (code & 1)));
                            else atomicAdd(&cnt[rowb + __umul24(code, (uint32_t)W) + rel], 1u);
                        } else if (code == 5u) {                        // a base that is not A/C/T/G: the level is present here
                            atomicOr(&pres[__umul24(mm >> 5, (uint32_t)W) + rel], 1u << (mm & 31));
                        }
                    }
                }
                return;
            }
#pragma unroll
            for (int h = 0; h < (COMPACT ? 4 : 2); h++) {
                uint32_t rel, base, mm;
                if (COMPACT) { rel = (x[h] & 0xFFFFu) + (gb[u] - w0); base = (x[h] >> 24) & 7u; mm = (x[h] >> 16) & 0xFFu; }
                else { rel = x[2 * h] - w0; base = (x[2 * h + 1] >> 16) & 0xFFu; mm = x[2 * h + 1] & 0xFFFFu; }
                if (rel >= (uint32_t)W) continue;
                if (mm >= (uint32_t)M) { bad_mm = 1; continue; }
                if (base < 4) {
                    if (PACKED) atomicAdd(&cnt[__umul24(mm * 2 + (base >> 1), (uint32_t)W) + rel], 1u << (16 * (base & 1)));
                    else atomicAdd(&cnt[__umul24(mm * 4 + base, (uint32_t)W) + rel], 1u);
                } else if (!COMPACT || base != 7u) {            // 7 = padding record of the compact stream
                    atomicOr(&pres[__umul24(mm >> 5, (uint32_t)W) + rel], 1u << (mm & 31));
                }
            }
        };
        for (uint32_t i0 = lo; i0 < hi; i0 += 4 * nthr) {
            issue_one(2, i0); issue_one(3, i0);
            count_slot(0); count_slot(1);
            if (i0 + 4 * nthr < hi) { issue_one(0, i0 + 4 * nthr); issue_one(1, i0 + 4 * nthr); }
            count_slot(2); count_slot(3);
        }
        if (bad_mm) flag_or(a, ISX_FLAG_MM_RANGE);
        __syncthreads();
        ISX_TS(2);
        ISX_ARGS_FRESH();
        if (!linkage) prefetch_window(w + grid);
        if (DREC) {
            // ---- materialise: per level  covered = prefix sum of the difference row; observed = covered - skipped; count of the
            //      reference's base = observed - the exceptions counted at the position.  A wave owns chunks of 64 consecutive positions
            //      (chunk = wave, wave + waves: W <= 2 x block, so two at most), a lane ONE position of a chunk: the prefix inside a chunk is
            //      one DPP scan, the chunks' totals go through LDS (ctot, in the deferred queue: idle until the level loop) and are scanned
            //      by every wave for itself.  Levels in groups of 8: their loads and scans are independent chains.
            uint32_t *ctot = queue;             // [8][32]
            const int lane_m = tid & 63;
            const int nwv = nthr >> 6, NCH = (W + 63) >> 6;
            const uint32_t uW = (uint32_t)W;
            int cc[2], pp[2];
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                cc[sl] = __builtin_amdgcn_readfirstlane((tid >> 6) + sl * nwv);
                pp[sl] = 64 * cc[sl] + lane_m;
            }
            for (int m0 = 0; m0 < M; m0 += 8) {
                const int mc = min(8, M - m0);
                uint32_t x[2][8];
                int32_t inc[2][8];
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    if (cc[sl] >= NCH) continue;            // (uniform)
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        x[sl][i] = 0;
                        if (i < mc && pp[sl] < W) x[sl][i] = aux[__umul24((uint32_t)(PACKED ? m0 + i : 2 * (m0 + i) + 1), uW) + (uint32_t)pp[sl]];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (i >= mc) continue;              // (uniform)
                        inc[sl][i] = (int32_t)wave_scan_incl(PACKED ? (uint32_t)((int32_t)x[sl][i] >> 16) : x[sl][i]);
                        if (lane_m == 63) ctot[32 * i + cc[sl]] = (uint32_t)inc[sl][i];
                    }
                }
                __syncthreads();
                uint32_t lbits[2] = {0, 0};     // levels of this group present at the lane's positions
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (i >= mc) continue;                  // (uniform)
                    const int m = m0 + i;
                    const uint32_t t = lane_m < NCH ? ctot[32 * i + lane_m] : 0u;
                    const uint32_t before = wave_scan_incl(t) - t;          // lane c: sum of the totals of the chunks before chunk c
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) {
                        if (cc[sl] >= NCH) continue;        // (uniform)
                        const int32_t run = (int32_t)__builtin_amdgcn_readlane((int)before, cc[sl]) + inc[sl][i];
                        const int p = pp[sl];
                        if (p >= W || run == 0) continue;                   // no read of this level covers the position
                        const uint32_t nskip = PACKED ? (x[sl][i] & 0xFFFFu) : aux[__umul24((uint32_t)(2 * m), uW) + (uint32_t)p];
                        const uint32_t observed = (uint32_t)run - nskip;
                        if (!observed) continue;
                        lbits[sl] |= 1u << i;
                        const uint32_t r = rcode[sl];
                        if (r < 4u) {
                            if (PACKED) {
                                const uint32_t a01 = cnt[(m * 2) * W + p], a23 = cnt[(m * 2 + 1) * W + p];
                                const uint32_t vref = observed - ((a01 & 0xFFFFu) + (a01 >> 16) + (a23 & 0xFFFFu) + (a23 >> 16));
                                if (vref) cnt[(m * 2 + (int)(r >> 1)) * W + p] = ((r >> 1) ? a23 : a01) + (vref << (16 * (r & 1u)));
                            } else {
                                const uint32_t e4 = cnt[(m * 4) * W + p] + cnt[(m * 4 + 1) * W + p] + cnt[(m * 4 + 2) * W + p] + cnt[(m * 4 + 3) * W + p];
                                if (observed != e4) cnt[(m * 4 + (int)r) * W + p] += observed - e4;
                            }
                        }
                    }
                }
                // the group's levels into the positions' presence words (beside the bits the non-ACGT markers set during the stream)
#pragma unroll
                for (int sl = 0; sl < 2; sl++)
                    if (lbits[sl]) pres[(m0 >> 5) * W + pp[sl]] |= lbits[sl] << (m0 & 31);
                if (m0 + 8 < M) __syncthreads();            // (uniform) the next group's totals overwrite ctot
            }
            __syncthreads();
            ISX_ARGS_FRESH();
        }
        ISX_TS(3);
        const uint32_t CW = (uint32_t)a.slab;   // entry slab of this window: [w * CW, (w + 1) * CW)
        const uint64_t slab0 = (uint64_t)w * CW;
        const int lane = tid & 63;

        // ---- update_snp_table's `for mm in sorted(MMcounts)`, level-major across the wave ----
        // Every lane owns a position; all lanes walk the mm levels together, so the entries of one
        // level are ballot-compacted into CONSECUTIVE slots of the window's slab (coalesced 32-byte
        // entry stores, no per-position entry count pass, no global atomic).  Positions with SNV rows
        // (rare) go to the row queue and are finished after the window-level allocations.
        uint32_t *rowq = queue + 2 * QCAP;      // [rqcap][4]: p | any<<16 | cry<<17 | mask<<20, row off, site off | nlev<<24, slev off
        // mode 1 (rare, queued positions only): write the SNV rows and the site's level rows
        auto emit_rows = [&](int p, uint32_t gpos, uint32_t row_at, uint32_t cry_in, uint32_t slev_at) {
            const int ref_base = ref_at(a, gpos);
            uint32_t cum[4] = {0, 0, 0, 0};
            uint32_t rows = 0, nl = 0;
            for (int m = 0; m < M; m++) {
                uint32_t l[4];
#pragma unroll
                for (int k = 0; k < 4; k++) l[k] = rd(m, k, p);
                const uint32_t present = l[0] | l[1] | l[2] | l[3] | ((pres[(m >> 5) * W + p] >> (m & 31)) & 1u);
                if (!present) continue;
#pragma unroll
                for (int k = 0; k < 4; k++) cum[k] += l[k];
                if (slev_at != 0xFFFFFFFFu) {
                    isx_slev sl;
                    sl.mm = (uint16_t)m; sl.pad = 0;
                    sl.cnt[0] = l[0]; sl.cnt[1] = l[1]; sl.cnt[2] = l[2]; sl.cnt[3] = l[3];
                    a.slev[slev_at + nl] = sl;
                }
                nl++;
                const uint32_t total = cum[0] + cum[1] + cum[2] + cum[3];
                const SiteCall sc = call_level(a, thr_lds, cum, total, ref_base, true);
                if (sc.snp < 0) continue;
                isx_snv r;
                r.gpos = gpos; r.mm = (uint16_t)m;
                r.con_base = (uint8_t)sc.snp; r.var_base = (uint8_t)sc.var;
                r.allele_count = (uint8_t)sc.morphia; r.cls = (uint8_t)sc.cls;
                r.cryptic = (uint8_t)cry_in;                        // position-level flag (p2c map)
                r.ref_base = (uint8_t)ref_base;
                r.cnt[0] = cum[0]; r.cnt[1] = cum[1]; r.cnt[2] = cum[2]; r.cnt[3] = cum[3];
                a.snv[row_at + rows] = r;
                rows++;
            }
        };
        // SPARSE, pass A: which levels every position has -> level indices.  A lane's two positions p = tid, tid + nthr; position order
        // is (j, tid) order, so the prefix sum runs over j = 0 first.  One global atomic hands the window its range of level slots.
        uint32_t lmask[2] = {0, 0}, loff[2] = {0, 0};
        uint32_t lev_base = 0;
        bool lev_ok = true;
        if (SPARSE) {
            uint32_t *wtot2 = rowq;                                 // [2][16] per-wave totals (the row queue is idle until the level loop)
            uint32_t nl2[2], inc2[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int p = tid + j * nthr;
                const uint32_t gpos = w0 + p;
                uint32_t mk = 0;
                if (p < W && gpos < a.n_pos && !(dbg & 2)) {
                    mk = pres[p];                                   // (M <= 32: one presence word)
                    if (!DREC)                                      // (DREC: the materialise phase left the complete mask there)
                    for (int m = 0; m < M; m++) {
                        uint32_t any4;
                        if (PACKED) any4 = cnt[(m * 2) * W + p] | cnt[(m * 2 + 1) * W + p];
                        else any4 = cnt[(m * 4) * W + p] | cnt[(m * 4 + 1) * W + p] | cnt[(m * 4 + 2) * W + p] | cnt[(m * 4 + 3) * W + p];
                        mk |= (any4 ? 1u : 0u) << m;
                    }
                    if (a.lev_mask_bytes == 1) reinterpret_cast<uint8_t *>(a.lev_mask)[gpos] = (uint8_t)mk;
                    else if (a.lev_mask_bytes == 2) reinterpret_cast<uint16_t *>(a.lev_mask)[gpos] = (uint16_t)mk;
                    else reinterpret_cast<uint32_t *>(a.lev_mask)[gpos] = mk;
                }
                lmask[j] = mk;
                nl2[j] = (uint32_t)__popc(mk);
                inc2[j] = wave_scan_incl(nl2[j]);
                if (lane == 63) wtot2[j * 16 + (tid >> 6)] = inc2[j];
            }
            __syncthreads();
            uint32_t tot0 = 0, before0 = 0, tot1 = 0, before1 = 0;
            const int nwv = nthr >> 6, wv = tid >> 6;
            for (int k = 0; k < nwv; k++) {
                const uint32_t x0 = wtot2[k], x1 = wtot2[16 + k];
                tot0 += x0; tot1 += x1;
                if (k < wv) { before0 += x0; before1 += x1; }
            }
            loff[0] = before0 + inc2[0] - nl2[0];
            loff[1] = tot0 + before1 + inc2[1] - nl2[1];
            if (tid == 0) {
                const uint32_t nw = tot0 + tot1;
                const uint32_t at = nw ? cur_add(a, CUR_ENT_TOTAL, nw) : 0u;
                a.lev_win_off[w] = at;
                bool fits = at + nw <= a.cap_lev;
                if (!fits) flag_or(a, ISX_FLAG_CAP_ENTRIES);
                scratch[S_ENT_BASE] = at;
                scratch[S_ENT_TOT] = fits ? 1u : 0u;
            }
            __syncthreads();
            lev_base = scratch[S_ENT_BASE];
            lev_ok = scratch[S_ENT_TOT] != 0u;
        }
        ISX_TS(4);
        uint32_t n_cl = 0, n_rr = 0;                                // SPARSE: this lane's queued clonT / clonTR values (list slots are sized from them)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int p = tid + j * nthr;
            const uint32_t gpos = w0 + p;
            const bool valid = p < W && gpos < a.n_pos && !(dbg & 2);
            const int ref_base = valid ? (int)ref_at(a, gpos) : 4;
            uint32_t cum[4] = {0, 0, 0, 0};
            uint32_t any = 0, cry = 0, rows = 0, nlev = 0, mask = 0;
            for (int m = 0; m < M; m++) {
                uint32_t l[4] = {0, 0, 0, 0};
                bool present = false;
                if (SPARSE) {
                    present = ((lmask[j] >> m) & 1u) != 0 && lev_ok;
                    if (present) {
#pragma unroll
                        for (int k = 0; k < 4; k++) l[k] = rd(m, k, p);
                    }
                } else if (valid) {
#pragma unroll
                    for (int k = 0; k < 4; k++) l[k] = rd(m, k, p);
                    present = (l[0] | l[1] | l[2] | l[3] | ((pres[(m >> 5) * W + p] >> (m & 31)) & 1u)) != 0;
                }
                const unsigned long long bal = __ballot(present);
                if (bal == 0) continue;                             // wave-uniform
                uint64_t ei;
                if (SPARSE) {
                    if (!present) continue;
                    ei = (uint64_t)(lev_base + loff[j] + (uint32_t)__popc(lmask[j] & ((1u << m) - 1u)));
                    const uint32_t cov_m = l[0] + l[1] + l[2] + l[3];
                    if (a.lev_cov_bytes == 1) reinterpret_cast<uint8_t *>(a.lev_cov)[ei] = (uint8_t)min(cov_m, 255u);
                    else reinterpret_cast<uint16_t *>(a.lev_cov)[ei] = (uint16_t)min(cov_m, 65535u);
                    if (cov_m >= a.sat_thr) {
                        const uint32_t k = cur_add(a, CUR_SAT, 1u);
                        if (k < a.cap_sat) a.sat[k] = make_uint2((uint32_t)ei, cov_m);
                    }
                } else {
                    uint32_t wbase = 0;
                    const int first = __ffsll((long long)bal) - 1;
                    if (lane == first) wbase = atomicAdd(&scratch[S_ENT_TOT], (uint32_t)__popcll(bal));
                    wbase = __shfl(wbase, first);
                    if (!present) continue;
                    const uint32_t slot_w = wbase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (slot_w < CW) ei = slab0 + slot_w;
                    else {                                              // slab full (more than CW / W levels per position on average)
                        const uint32_t o = cur_add(a, CUR_ENTRIES, 1u);
                        if (o >= a.cap_ovf) { flag_or(a, ISX_FLAG_CAP_ENTRIES); continue; }
                        ei = a.ovf0 + o;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) cum[k] += l[k];         // mm_counts_to_counts(MMcounts, mm)
                const uint32_t total = cum[0] + cum[1] + cum[2] + cum[3];
                const SiteCall sc = call_level(a, thr_lds, cum, total, ref_base, false);
                float cl = __builtin_nanf("");
                const bool want_r = a.min_cov_r > 0 && (int64_t)total >= (int64_t)a.min_cov_r;
                bool want_c = false;
                if ((int64_t)total >= (int64_t)a.min_cov) {
                    const uint32_t mx = max(max(cum[0], cum[1]), max(cum[2], cum[3]));
                    if (mx == total) cl = 1.0f;                     // (s/s)^2 + 0 + 0 + 0
                    else want_c = true;
                }
                float clr = __builtin_nanf("");
                if (want_c || want_r) {
                    const uint32_t qs = atomicAdd(&scratch[S_NQ], 1u);
                    if (qs < QCAP && ei < 0xFFFFFFFFull) {
                        queue[qs * 2 + 0] = (uint32_t)ei;
                        queue[qs * 2 + 1] = ((uint32_t)m << 16) | (uint32_t)p | (want_c ? 1u << 30 : 0u) | (want_r ? 1u << 31 : 0u);
                        if (SPARSE) { n_cl += want_c ? 1u : 0u; n_rr += want_r ? 1u : 0u; }
                    } else {                                        // queue full: inline
                        if (want_c) cl = (float)clonality(cum, total);
                        if (want_r) clr = rarefied_clonality(a, cum, gpos, (uint32_t)m);
                        if (SPARSE) {                               // (its own list slots: rare)
                            if (want_c) { const uint32_t k = cur_add(a, CUR_CLON, 1u); if (k < a.cap_clon) a.clon_list[k] = make_uint2((uint32_t)ei, __float_as_uint(cl)); }
                            if (want_r) { const uint32_t k = cur_add(a, CUR_RARE, 1u); if (k < a.cap_rare) a.rare[k] = make_uint2((uint32_t)ei, __float_as_uint(clr)); }
                        }
                    }
                }
                if (!(dbg & 64) && (!SPARSE || a.entries)) {
                    uint4 *dst = reinterpret_cast<uint4 *>(&a.entries[ei]);
                    dst[0] = make_uint4(gpos, (uint32_t)m, l[0], l[1]);                     // gpos | mm,flags | cnt[0..1]
                    dst[1] = make_uint4(l[2], l[3], __float_as_uint(cl), __float_as_uint(clr));
                }
                nlev++;
                if (sc.snp == -2) continue;
                if (sc.snp != -1) {
                    rows++;
                    if (sc.morphia >= 2) { any = 1; mask |= (1u << sc.snp) | (1u << sc.var); }
                    else if (sc.morphia == 1 && any) cry = 1;
                } else if (any) {
                    cry = 1;
                }
            }
            if (rows) {
                if (any && linkage) {                               // cum == counts over ALL levels here
                    maskl[p] = (uint8_t)mask;
                    slabc[p] = atomicAdd(&scratch[S_NAO], masked_sum(cum, mask));
                }
                const uint32_t qi = atomicAdd(&scratch[S_ROW_RANK], 1u);
                if (qi < (uint32_t)a.rqcap) {
                    const uint32_t r_off = atomicAdd(&scratch[S_ROWS], rows);
                    const uint32_t s_off = any ? atomicAdd(&scratch[S_SITES], 1u) : 0u;
                    const uint32_t v_off = any ? atomicAdd(&scratch[S_SLEV], nlev) : 0u;
                    rowq[qi * 4 + 0] = (uint32_t)p | (any << 16) | (cry << 17) | (mask << 20);
                    rowq[qi * 4 + 1] = r_off;
                    rowq[qi * 4 + 2] = s_off | (nlev << 24);
                    rowq[qi * 4 + 3] = v_off;
                } else {                                            // row queue full: this position allocates by itself
                    const uint32_t row_at = cur_add(a, CUR_SNV, rows);
                    uint32_t slev_at = 0xFFFFFFFFu;
                    bool fine = row_at + rows <= a.cap_snv;
                    if (!fine) flag_or(a, ISX_FLAG_CAP_SNV);
                    if (any && fine) {
                        const uint32_t sa = cur_add(a, CUR_SITES, 1u);
                        flag_or(a, ISX_FLAG_SITES_LOOSE);            // (outside the window's own range: the site table takes the device-wide sort)
                        slev_at = cur_add(a, CUR_SLEV, nlev);
                        if (sa >= a.cap_sites || slev_at + nlev > a.cap_slev) { flag_or(a, ISX_FLAG_CAP_SITES); fine = false; }
                        else {
                            isx_site ss;
                            ss.gpos = gpos; ss.entry_off = slev_at; ss.n_levels = (uint16_t)nlev;
                            ss.mask = (uint8_t)mask; ss.pad = 0;
                            a.sites[sa] = ss;
                        }
                    }
                    if (fine) emit_rows(p, gpos, row_at, cry, slev_at);
                }
            }
        }
        __syncthreads();
        ISX_TS(5);
        ISX_ARGS_FRESH();
        const uint32_t n_ent = scratch[S_ENT_TOT], nrows = scratch[S_ROWS], nsites = scratch[S_SITES], nao = scratch[S_NAO],
                       nslev = scratch[S_SLEV], nrq = min(scratch[S_ROW_RANK], (uint32_t)a.rqcap);
        if (!SPARSE && tid == 0) {
            a.win_nent[w] = min(n_ent, CW);
            my_entries += n_ent;                                    // per-workgroup total, published once at the end
        }
        if (tid == 64 && nrows) scratch[S_ROW_BASE] = cur_add(a, CUR_SNV, nrows);
        if (tid == 128 % nthr) {
            uint32_t sb = 0;
            if (nsites) { sb = cur_add(a, CUR_SITES, nsites); scratch[S_SITE_BASE] = sb; }
            if (a.win_site_cnt) { a.win_site_base[w] = sb; a.win_site_cnt[w] = nsites; }       // the window's sites lie side by side: k_site_order sorts window by window
        }
        if (tid == 192 % nthr && nao) scratch[S_AO_BASE] = cur_add(a, CUR_AO, nao);
        if (tid == 256 % nthr && nslev) scratch[S_SLEV_BASE] = cur_add(a, CUR_SLEV, nslev);
        // ---- deferred clonalities: calculate_clonality (snv_utilities.py:225-231) in fp64, densely packed ----
        const uint32_t nq = min(scratch[S_NQ], QCAP);
        bool clon_fit = true, rare_fit = true;
        if (SPARSE) {
            // list slots of the window's queued values: the lanes' counts summed per wave, one LDS atomic a wave, one global atomic a list
            const uint32_t wc = wave_scan_incl(n_cl), wr = wave_scan_incl(n_rr);
            if (lane == 63) { if (wc) atomicAdd(&scratch[S_NCLON], wc); if (wr) atomicAdd(&scratch[S_NRARE], wr); }
            __syncthreads();
            const uint32_t nclon = scratch[S_NCLON], nrare = scratch[S_NRARE];
            if (tid == 320 % nthr && nclon) scratch[S_CLON_BASE] = cur_add(a, CUR_CLON, nclon);
            if (tid == 384 % nthr && nrare) scratch[S_RARE_BASE] = cur_add(a, CUR_RARE, nrare);
            if (nclon | nrare) __syncthreads();                     // (uniform)
            clon_fit = scratch[S_CLON_BASE] + nclon <= a.cap_clon;  // a list that outgrew its table: the host sees the cursor and repeats the pass
            rare_fit = scratch[S_RARE_BASE] + nrare <= a.cap_rare;
        }
        for (uint32_t q = tid; q < nq; q += nthr) {
            const uint32_t pm = queue[q * 2 + 1];
            const int p = (int)(pm & 0xFFFFu), mq = (int)((pm >> 16) & 0x3FFFu);
            uint32_t c[4] = {0, 0, 0, 0};
            for (int m = 0; m <= mq; m++) {             // mm_counts_to_counts(MMcounts, mm)
#pragma unroll
                for (int k = 0; k < 4; k++) c[k] += rd(m, k, p);
            }
            if (pm & (1u << 30)) {
                const float v = (float)clonality(c, c[0] + c[1] + c[2] + c[3]);
                if (!SPARSE || a.entries) a.entries[queue[q * 2]].clon = v;
                if (SPARSE && clon_fit) a.clon_list[scratch[S_CLON_BASE] + atomicAdd(&scratch[S_CLON_RANK], 1u)] = make_uint2(queue[q * 2], __float_as_uint(v));
            }
            if (pm & (1u << 31)) {
                const float v = rarefied_clonality(a, c, w0 + p, (uint32_t)mq);
                if (!SPARSE || a.entries) a.entries[queue[q * 2]].clon_rarefied = v;
                if (SPARSE && rare_fit) a.rare[scratch[S_RARE_BASE] + atomicAdd(&scratch[S_RARE_RANK], 1u)] = make_uint2(queue[q * 2], __float_as_uint(v));
            }
        }
        if (nrows) __syncthreads();             // uniform: bases from the atomics above
        ISX_TS(6);
        const uint32_t row_base = scratch[S_ROW_BASE], site_base = scratch[S_SITE_BASE], ao_base = scratch[S_AO_BASE],
                       slev_base = scratch[S_SLEV_BASE];
        bool ok = true;
        if (nrows && (row_base + nrows > a.cap_snv || site_base + nsites > a.cap_sites || ao_base + nao > a.cap_ao ||
                      slev_base + nslev > a.cap_slev)) {
            if (tid == 0) flag_or(a, row_base + nrows > a.cap_snv ? ISX_FLAG_CAP_SNV
                                            : ao_base + nao > a.cap_ao ? ISX_FLAG_CAP_AO : ISX_FLAG_CAP_SITES);
            ok = false;
        }
        // ---- SNV rows / SNP sites of the queued positions (snv_utilities.py:107-133) ----
        for (uint32_t q = tid; q < (ok ? nrq : 0u); q += nthr) {
            const uint32_t w0q = rowq[q * 4 + 0];
            const int p = (int)(w0q & 0xFFFFu);
            const uint32_t any = (w0q >> 16) & 1u, cry = (w0q >> 17) & 1u, mask = (w0q >> 20) & 0xFu;
            const uint32_t slev_at = any ? slev_base + rowq[q * 4 + 3] : 0xFFFFFFFFu;
            emit_rows(p, w0 + p, row_base + rowq[q * 4 + 1], cry, slev_at);
            if (any) {
                isx_site ss;
                ss.gpos = w0 + p; ss.entry_off = slev_at;
                ss.n_levels = (uint16_t)(rowq[q * 4 + 2] >> 24);
                ss.mask = (uint8_t)mask; ss.pad = 0;
                a.sites[site_base + (rowq[q * 4 + 2] & 0xFFFFFFu)] = ss;
            }
        }
        ISX_TS(7);
        if (linkage) {
            if (ok && nao) {
                __syncthreads();                // every wave is done with the counters: they become the stage
                if (DREC) allele_pass_delta(a, cur_lo, cur_hi, w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
                else if (SEGS) allele_pass_segs(a, cur_lo, cur_hi, w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
                else allele_pass(a, cur_lo << (RSH - 1), cur_hi << (RSH - 1), w0, W, maskl, slabc, ao_base, lds + a.stage_off, tid, nthr);
            }
            prefetch_window(w + grid);
        }
        ISX_TS(8);
        __syncthreads();
        ISX_TS(9);
        ISX_ARGS_FRESH();
    }
    if (!SPARSE && tid == 0 && my_entries) cur_add(a, CUR_ENT_TOTAL, my_entries);
}
#undef a
#undef ISX_ARGS_FRESH


// ---------------------------------------------------------------------------------------------
// Position order without sorting.  k_pileup_dense hands out table slots window by window as the windows reach the cursors, so the
// SNV rows, SNP sites, the clonality list and the clonTR list come out grouped by window but with the windows (and the entries
// inside one) in no particular order.  A window is a range of positions and every entry of these tables has its own position
// (one mm bin: one row per position), so the ordered tables are a GATHER: k_win_scan = exclusive prefix of the per-window counts
// in window order; k_win_gather = one workgroup per window, rank of an entry = set bits below its position in the window's
// occupancy bitmap.  Replaces four sorts per batch (rocprim's radix sort of a few 10^5 keys is 5-8 kernel launches each).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_win_scan(const uint32_t *__restrict__ win_rec, uint32_t *__restrict__ win_out, int n_win,
                                                   uint32_t *state, uint32_t epoch)
{
    // One workgroup per chunk of 1024 windows (round 6; one workgroup walking all chunks took 73 us on a 117 Mbp batch): a thread loads ONE
    // window's record (two 16-byte loads, coalesced over the wave), the four counts are scanned over the chunk -- DPP inside a wave, per-wave
    // totals through LDS --, the chunk's totals are published in `state` (8 words a chunk: totals | epoch) and every workgroup adds up the
    // totals of the chunks before it once they carry this launch's epoch (workgroups are dispatched in index order: the ones waited for run).
    __shared__ uint32_t wsum[4][16], pre[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
    const uint4 *rec4 = reinterpret_cast<const uint4 *>(win_rec);
    uint4 *out4 = reinterpret_cast<uint4 *>(win_out);
    const int w = g * 1024 + tid;
    uint32_t c[4] = {0, 0, 0, 0};
    if (w < n_win) {
        const uint4 lo = rec4[2 * (size_t)w], hi = rec4[2 * (size_t)w + 1];
        c[0] = lo.y; c[1] = lo.w; c[2] = hi.y; c[3] = hi.w;
    }
    uint32_t inc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        inc[k] = wave_scan_incl(c[k]);
        if (lane == 63) wsum[k][wave] = inc[k];
    }
    if (tid < 4) pre[tid] = 0;
    __syncthreads();
    uint32_t tot[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t off = inc[k] - c[k], t = 0;
        for (int j = 0; j < 16; j++) { const uint32_t x = wsum[k][j]; if (j < wave) off += x; t += x; }
        tot[k] = t;
        inc[k] = off;
    }
    if (tid == 0) {
        uint32_t *st = state + 8 * (size_t)g;
#pragma unroll
        for (int k = 0; k < 4; k++) __hip_atomic_store(&st[k], tot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st[4], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t add[4] = {0, 0, 0, 0};
    for (int cpre = tid; cpre < g; cpre += 1024) {
        uint32_t *sp = state + 8 * (size_t)cpre;
        while (__hip_atomic_load(&sp[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
#pragma unroll
        for (int k = 0; k < 4; k++) add[k] += __hip_atomic_load(&sp[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < g) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (add[k]) atomicAdd(&pre[k], add[k]);
    }
    __syncthreads();
    if (w < n_win) out4[w] = make_uint4(pre[0] + inc[0], pre[1] + inc[1], pre[2] + inc[2], pre[3] + inc[3]);
}

struct GatherArgs {
    const uint32_t *win_rec, *win_out;
    const isx_snv *snv_raw; isx_snv *snv;
    const isx_site *sites_raw; isx_site *sites;
    const uint2 *clon_raw; uint2 *clon;
    const uint2 *rare_raw; uint2 *rare;
    int W;
};

// exclusive prefix of the set bits of a window's occupancy bitmap (256 words = 8192 positions), per word
__device__ __forceinline__ void bitmap_prefix(const uint32_t *bm, uint32_t *pre, uint32_t *tmp4, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t c = (uint32_t)__popc(bm[tid]);
    const uint32_t v = wave_scan_incl(c);
    if (lane == 63) tmp4[wave] = v;
    __syncthreads();
    uint32_t off = v - c;
    for (int j = 0; j < wave; j++) off += tmp4[j];
    pre[tid] = off;
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_win_gather(const GatherArgs g)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    const uint32_t *wr = g.win_rec + 8 * (size_t)w, *wo = g.win_out + 4 * (size_t)w;
    const uint32_t n_rows = wr[1], n_sites = wr[3], n_clon = wr[5], n_rare = wr[7];
    if (!(n_rows | n_sites | n_clon | n_rare)) return;
    __shared__ uint32_t bm_rows[256], pre_rows[256], bm[256], pre[256], tmp4[4];
    const uint32_t w0 = (uint32_t)w * (uint32_t)g.W;
    auto rank_of = [](const uint32_t *bmx, const uint32_t *prex, uint32_t p) {
        return prex[p >> 5] + (uint32_t)__popc(bmx[p >> 5] & ((1u << (p & 31u)) - 1u));
    };
    bm_rows[tid] = 0;
    __syncthreads();
    if (n_rows | n_sites) {                     // rows first: a site's entry_off is the index of ITS row in the ordered table
        for (uint32_t i = tid; i < n_rows; i += 256) { const uint32_t p = g.snv_raw[wr[0] + i].gpos - w0; atomicOr(&bm_rows[p >> 5], 1u << (p & 31u)); }
        __syncthreads();
        bitmap_prefix(bm_rows, pre_rows, tmp4, tid);
        for (uint32_t i = tid; i < n_rows; i += 256) {
            const isx_snv r = g.snv_raw[wr[0] + i];
            g.snv[wo[0] + rank_of(bm_rows, pre_rows, r.gpos - w0)] = r;
        }
    }
    if (n_sites) {
        bm[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n_sites; i += 256) { const uint32_t p = g.sites_raw[wr[2] + i].gpos - w0; atomicOr(&bm[p >> 5], 1u << (p & 31u)); }
        __syncthreads();
        bitmap_prefix(bm, pre, tmp4, tid);
        for (uint32_t i = tid; i < n_sites; i += 256) {
            isx_site st = g.sites_raw[wr[2] + i];
            const uint32_t p = st.gpos - w0;
            st.entry_off = wo[0] + rank_of(bm_rows, pre_rows, p);
            g.sites[wo[1] + rank_of(bm, pre, p)] = st;
        }
    }
#pragma unroll
    for (int which = 0; which < 2; which++) {
        const uint32_t n = which ? n_rare : n_clon;
        if (!n) continue;                       // (uniform)
        const uint2 *src = (which ? g.rare_raw : g.clon_raw) + wr[which ? 6 : 4];
        uint2 *dst = (which ? g.rare : g.clon) + wo[which ? 3 : 2];
        __syncthreads();
        bm[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += 256) { const uint32_t p = src[i].x - w0; atomicOr(&bm[p >> 5], 1u << (p & 31u)); }
        __syncthreads();
        bitmap_prefix(bm, pre, tmp4, tid);
        for (uint32_t i = tid; i < n; i += 256) { const uint2 e = src[i]; dst[rank_of(bm, pre, e.x - w0)] = e; }
    }
}

}  // namespace

size_t pileup_lds_bytes(int W, int M, int qcap, int rqcap, int linkage, int packed, int block, int segs, int *stage_off, int *dlt_off)
{
    size_t words, cnt_words;
    const int pad = segs == 64 ? ISX_SEG_PAD : ISX_DENSE_PAD;       // segs: 0 = observation records, 64 = segment records, 32 = reference-delta records
    if (dlt_off) *dlt_off = 0;
    const bool pkl = M == 1 && segs == 32 && packed;    // reference-delta records with 16-bit counters: two counter rows + the queue row
    if (M == 1) { cnt_words = (size_t)(pkl ? 2 : 4) * (W + pad); words = cnt_words + (size_t)(W + pad) + S_N + S_RNG + THR_LDS / 2; }
    else {
        cnt_words = (size_t)M * (packed ? 2 : 4) * W;
        if (segs == 32) cnt_words += (size_t)M * (packed ? 1 : 2) * W;      // reference-delta records: the levels' skipped-columns / coverage-difference rows
        words = cnt_words + (size_t)((M + 31) / 32) * W + S_N + (size_t)qcap * 2 + (size_t)rqcap * 4 + THR_LDS / 2;
    }
    size_t bytes = words * sizeof(uint32_t);
    if (stage_off) *stage_off = 0;
    if (linkage) {
        bytes += (size_t)W * 5;                 // slabc[W] + maskl[W]
        const size_t stage_words = (size_t)(block / 64) * 128 + (segs ? (size_t)W / 32 + 2 : 0);   // allele pass: 64 two-word entries per wave (+ the segment walk's site bitmap)
        if (cnt_words < stage_words) {          // small windows: the counters cannot host the stage
            bytes = (bytes + 15) & ~(size_t)15;
            if (stage_off) *stage_off = (int)(bytes / 4);
            bytes += stage_words * 4;
        }
    }
    if (segs == 32 && M == 1) {                 // the coverage-difference row + 16 per-wave totals of its prefix sum
        bytes = (bytes + 15) & ~(size_t)15;     // (packed: the differences live in the queue row; the totals + one reference code per position)
        if (dlt_off) *dlt_off = (int)(bytes / 4);
        bytes += pkl ? 16 * 4 + (((size_t)W + 15) & ~(size_t)15) : ((size_t)(W + pad) + 16) * 4;
    }
    return bytes;
}

// ev_start / ev_stop bracket exactly this dispatch (hipExtLaunchKernel: the packet's own start / end time
// stamps, what rocprofv3 reports too) -- no separate event-record packets around the kernel
struct LaunchCfg { int block; size_t lds; int grid; hipStream_t s; hipEvent_t ev_start, ev_stop; };

template <class K>
static void launch_one(K kernel, const PileupArgs &a, const LaunchCfg &l)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds);
    hipExtLaunchKernelGGL(kernel, dim3(l.grid), dim3(l.block), (uint32_t)l.lds, l.s, l.ev_start, l.ev_stop, 0u, a);
}

void launch_pileup(const PileupArgs &a, int block, size_t lds, int grid, int packed, hipStream_t s, hipEvent_t ev_start,
                   hipEvent_t ev_stop)
{
    const LaunchCfg l{block, lds, grid, s, ev_start, ev_stop};
    const int sel = (a.enable_linkage != 0 ? 1 : 0) | (a.rec32 ? 2 : 0) | (packed ? 4 : 0);
    if (a.M > 1 && a.drec) {
        switch ((sel & 5) | (a.lev_cov ? 2 : 0)) {
        case 0: launch_one(k_pileup_mm<false, false, true, true, false, true>, a, l); break;
        case 1: launch_one(k_pileup_mm<false, true, true, true, false, true>, a, l); break;
        case 2: launch_one(k_pileup_mm<false, false, true, true, true, true>, a, l); break;
        case 3: launch_one(k_pileup_mm<false, true, true, true, true, true>, a, l); break;
        case 4: launch_one(k_pileup_mm<true, false, true, true, false, true>, a, l); break;
        case 5: launch_one(k_pileup_mm<true, true, true, true, false, true>, a, l); break;
        case 6: launch_one(k_pileup_mm<true, false, true, true, true, true>, a, l); break;
        default: launch_one(k_pileup_mm<true, true, true, true, true, true>, a, l); break;
        }
    } else if (a.M > 1 && a.seg && a.lev_cov) {
        switch (sel & 5) {
        case 0: launch_one(k_pileup_mm<false, false, true, true, true>, a, l); break;
        case 1: launch_one(k_pileup_mm<false, true, true, true, true>, a, l); break;
        case 4: launch_one(k_pileup_mm<true, false, true, true, true>, a, l); break;
        default: launch_one(k_pileup_mm<true, true, true, true, true>, a, l); break;
        }
    } else if (a.M > 1 && a.seg) {
        switch (sel & 5) {
        case 0: launch_one(k_pileup_mm<false, false, true, true>, a, l); break;
        case 1: launch_one(k_pileup_mm<false, true, true, true>, a, l); break;
        case 4: launch_one(k_pileup_mm<true, false, true, true>, a, l); break;
        default: launch_one(k_pileup_mm<true, true, true, true>, a, l); break;
        }
    } else if (a.M > 1) {
        switch (sel) {
        case 0: launch_one(k_pileup_mm<false, false, false>, a, l); break;
        case 1: launch_one(k_pileup_mm<false, true, false>, a, l); break;
        case 2: launch_one(k_pileup_mm<false, false, true>, a, l); break;
        case 3: launch_one(k_pileup_mm<false, true, true>, a, l); break;
        case 4: launch_one(k_pileup_mm<true, false, false>, a, l); break;
        case 5: launch_one(k_pileup_mm<true, true, false>, a, l); break;
        case 6: launch_one(k_pileup_mm<true, false, true>, a, l); break;
        default: launch_one(k_pileup_mm<true, true, true>, a, l); break;
        }
    } else {
        const bool link = a.enable_linkage != 0;
        if (a.seg) { if (link) launch_one(k_pileup_dense<true, 64>, a, l); else launch_one(k_pileup_dense<false, 64>, a, l); return; }
        if (a.drec && packed) { if (link) launch_one(k_pileup_dense<true, 32, true>, a, l); else launch_one(k_pileup_dense<false, 32, true>, a, l); return; }
        if (a.drec) { if (link) launch_one(k_pileup_dense<true, 32>, a, l); else launch_one(k_pileup_dense<false, 32>, a, l); return; }
#ifndef ISX_NO_PK16
        if (a.rec16 && a.W <= ISX_PK16_MAX_W) { if (link) launch_one(k_pileup_dense<true, 2, true>, a, l); else launch_one(k_pileup_dense<false, 2, true>, a, l); }
        else
#endif
        if (a.rec16) { if (link) launch_one(k_pileup_dense<true, 2>, a, l); else launch_one(k_pileup_dense<false, 2>, a, l); }
        else if (a.rec32) { if (link) launch_one(k_pileup_dense<true, 4>, a, l); else launch_one(k_pileup_dense<false, 4>, a, l); }
        else { if (link) launch_one(k_pileup_dense<true, 8>, a, l); else launch_one(k_pileup_dense<false, 8>, a, l); }
    }
}

__global__ void k_extract_gpos(const uint2 *rec, const uint32_t *rec32, const uint32_t *gbase, uint32_t *gpos, uint16_t *gpos16,
                               const uint32_t *base16, uint32_t base16_records, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g;
    if (rec32) { const uint32_t x = rec32[i]; g = x == ISX_PAD32 ? ISX_SENTINEL : gbase[i / ISX_GROUP] + (x & 0xFFFFu); }
    else g = rec[i].x;
    if (gpos16) gpos16[i] = g == ISX_SENTINEL ? (uint16_t)0xFFFFu : (uint16_t)(g - base16[i / base16_records]);
    else gpos[i] = g;
}

void launch_extract_gpos(const uint2 *rec, const uint32_t *rec32, const uint32_t *gbase, uint32_t *gpos, uint16_t *gpos16,
                         const uint32_t *base16, uint32_t base16_records, uint64_t n_rec, hipStream_t s)
{
    hipLaunchKernelGGL(k_extract_gpos, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s, rec, rec32, gbase, gpos, gpos16,
                       base16, base16_records, n_rec);
}

void launch_win_order(const uint32_t *win_rec, uint32_t *win_out, int n_win, int W, const isx_snv *snv_raw, isx_snv *snv, const isx_site *sites_raw,
                      isx_site *sites, const uint2 *clon_raw, uint2 *clon, const uint2 *rare_raw, uint2 *rare, uint32_t *scan_state, uint32_t epoch, hipStream_t s)
{
    if (n_win <= 0) return;
    hipLaunchKernelGGL(k_win_scan, dim3((n_win + 1023) / 1024), dim3(1024), 0, s, win_rec, win_out, n_win, scan_state, epoch);
    GatherArgs g{win_rec, win_out, snv_raw, snv, sites_raw, sites, clon_raw, clon, rare_raw, rare, W};
    hipLaunchKernelGGL(k_win_gather, dim3(n_win), dim3(256), 0, s, g);
}

void launch_publish_state(const PileupArgs &a, uint32_t epoch, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish_state, dim3(1), dim3(64), 0, s, a.cursors, a.host_state, epoch);
}
