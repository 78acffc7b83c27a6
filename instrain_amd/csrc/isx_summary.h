// isx_summary.h -- host-side interface of the per-scaffold summary pass (isx_summary.hip)
#pragma once
#include "isx_internal.h"

struct SummaryBuffers {
    uint32_t *cov = nullptr;        // cumulative coverage over levels <= mm, per flat position
    float *cv = nullptr, *cr = nullptr;     // clonality / rarefied clonality of the highest level <= mm
    uint32_t *k_u32 = nullptr;
    float *k_f32 = nullptr;
    uint32_t *seg_off = nullptr;
    int64_t *bounds = nullptr;
    void *acc = nullptr;
    double *med = nullptr;
    isx_scaffold_level *rows = nullptr;
    void *temp = nullptr;
    size_t temp_bytes = 0;
    int n_seg = -1;
    void release();
};

struct SummaryIn {
    hipStream_t stream;
    hipEvent_t *ev;                 // 2 events
    uint32_t n_pos;
    int n_scaffolds, M;
    const int64_t *scaffold_bounds; // host, [n_scaffolds + 1]
    // dense path
    const uint4 *counts;
    const float *clon, *clon_r;
    // mm path
    const isx_entry *entries;
    const uint32_t *win_nent;
    uint32_t slab, n_win, n_ovf;
    uint64_t ovf0;
};

int run_summary(const SummaryIn &in, SummaryBuffers &B, isx_scaffold_level *host_out, float *ms);

struct CompareBuffers {
    uint32_t *cov_a = nullptr, *cov_b = nullptr;
    float *scratch_f = nullptr;
    int64_t *bounds = nullptr;
    void *acc_a = nullptr, *acc_b = nullptr;
    unsigned long long *both = nullptr;
    isx_compare_level *rows = nullptr;
    void release();
};

int run_compare(const SummaryIn &a, const SummaryIn &b, uint32_t min_cov, CompareBuffers &B, isx_compare_level *host_out, float *ms);
