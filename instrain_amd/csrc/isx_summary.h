// isx_summary.h -- host-side interface of the per-scaffold summary pass (isx_summary.hip)
#pragma once
#include <functional>

#include "isx_internal.h"

struct SummaryBuffers {
    uint32_t *cov = nullptr;        // cumulative coverage over levels <= mm, per flat position
    float *cv = nullptr, *cr = nullptr;     // clonality / rarefied clonality of the highest level <= mm
    uint32_t *k_u32 = nullptr;
    float *k_f32 = nullptr;
    uint32_t *seg_off = nullptr, *seg_be = nullptr;
    int64_t *bounds = nullptr;
    void *acc = nullptr;
    double *med = nullptr;
    isx_scaffold_level *rows = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    int n_seg = -1;
    size_t cap_pos = 0;             // positions the position-sized arrays hold (a pipe slot sees batches of different sizes)
    void release();
    void fit_positions(size_t n_pos);
};

struct SummaryIn {
    hipStream_t stream;
    hipEvent_t *ev;                 // 2 events
    uint32_t n_pos;
    int n_scaffolds, M;
    const int64_t *scaffold_bounds; // host, [n_scaffolds + 1]
    // dense path
    const uint4 *counts;            // NULL: a pipe slot without a count table -> cov16 + the exact values of saturated positions
    const uint16_t *cov16;
    const uint2 *sat;
    uint32_t n_sat;
    const float *clon, *clon_r;
    // mm path
    const isx_entry *entries;
    const uint32_t *win_nent;
    uint32_t slab, n_win, n_ovf;
    uint64_t ovf0;
};

// (position, value) pairs ordered by position: a 32-bit radix sort of the 8-byte entries on their low word; *temp grows on demand
int sort_pairs_by_position(hipStream_t s, const uint2 *in, uint2 *out, size_t n, void **temp, size_t *temp_bytes);

int run_summary(const SummaryIn &in, SummaryBuffers &B, isx_scaffold_level *host_out, float *ms);
int run_genome_summary(const SummaryIn &in, SummaryBuffers &B, int n_genomes, const int32_t *genome_first, int mask_edges,
                       isx_genome_level *host_out, float *ms);

struct CompareBuffers {
    uint32_t *cov_a = nullptr, *cov_b = nullptr;
    float *scratch_f = nullptr;
    int64_t *bounds = nullptr;
    void *acc_a = nullptr, *acc_b = nullptr;
    unsigned long long *both = nullptr;     // [4 * n_seg]: both, either, consensus SNPs, population SNPs
    isx_compare_level *rows = nullptr;
    // SNP-table half (readComparer.py:205-290)
    uint64_t *keys = nullptr;               // [4 * cap_snv]: sorted A, sorted B, 2 x sort input
    uint32_t *idx = nullptr;                // same layout
    void *cand = nullptr;                   // candidate rows (mm independent verdicts)
    isx_compare_snp *snp_rows = nullptr;    // emitted (position, mm) rows
    uint32_t *cursors = nullptr;            // [0] n_cand, [1] n_snp_rows, [2..2+n_seg) scaffold failed
    void *temp = nullptr;
    size_t temp_bytes = 0;
    size_t cap_pos = 0, cap_seg = 0, cap_rows = 0, cap_snv = 0, cap_snp_rows = 0;
    uint32_t n_snp_rows = 0;                // rows of the last isx_compare_scaffolds
    void release();
};

struct CompareSnpIn {                       // nullptr lut = coverage half only
    const uint8_t *lut = nullptr;           // 255 = coverage not in the null model -> fallback
    int32_t lut_n = 0, fallback = 0;
    double min_freq = 0.05;
    const isx_snv *snv_a = nullptr, *snv_b = nullptr;
    uint32_t n_a = 0, n_b = 0;
};

int run_compare(const SummaryIn &a, const SummaryIn &b, uint32_t min_cov, const CompareSnpIn &snp, CompareBuffers &B,
                isx_compare_level *host_out, float *ms);

// the mm path's entry table (window slabs + overflow) compacted and ordered by (gpos, mm) on the device, then brought to
// host_out: by one hipMemcpyAsync, or by `copier` (device source, host destination, bytes, stream) when the caller has a
// faster way to pageable memory (a pipe's pinned staging + its host threads)
typedef std::function<int(const void *, void *, size_t, hipStream_t)> EntryCopier;
// soa != NULL: the shrunk hand-back instead (isx_pipe_fetch_entries_shrunk): four host columns of n_entries 4-byte values
struct EntrySoa { uint32_t *gpos, *mm_cov; float *clon, *clon_rarefied; };
int fetch_entries_sorted(hipStream_t s, const isx_entry *entries, const uint32_t *win_nent, uint32_t slab, uint32_t n_win,
                         uint32_t n_ovf, uint64_t n_entries, isx_entry *host_out, const EntryCopier *copier = nullptr,
                         const EntrySoa *soa = nullptr);
