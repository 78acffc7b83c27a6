// isx_internal.h -- shared between the HIP translation units of libinstrain_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/instrain_amd.h"

#define ISX_CHUNK 1024              // observation directory granule (records)
#define ISX_PAD 2048                // the record stream is padded to a multiple of this (k_allele_obs tile)
#define ISX_SENTINEL 0xFFFFFFFFu    // gpos of padding records (never inside a window)

void isx_set_error(const std::string &msg);

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            isx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            return ISX_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// device error flag bits (dev_flags[0])
#define ISX_FLAG_MM_RANGE 1u
#define ISX_FLAG_CAP_ENTRIES 2u
#define ISX_FLAG_CAP_SNV 4u
#define ISX_FLAG_CAP_SITES 8u
#define ISX_FLAG_CAP_AO 16u
#define ISX_FLAG_CAP_INCR 32u
#define ISX_FLAG_CAP_LD 64u

// cursors (dev_cursors[i], uint32)
enum { CUR_ENTRIES = 0, CUR_SNV = 1, CUR_SITES = 2, CUR_AO = 3, CUR_N = 8 };

// SNP site record: a position where update_snp_table returned anySNP (snv_utilities.py:129-133).
// Holds what linkage needs later: the `bases` set and where the per-level counts live.
struct isx_site {
    uint32_t gpos;
    uint32_t entry_off;     // mm path: index of the first entry of this position; dense path: unused
    uint16_t n_levels;      // mm path: number of entries (levels present)
    uint8_t mask;           // `bases` set, bit b = base b
    uint8_t pad;
};

// allele observation = one update_linked_reads append (linkage.py:281)
struct isx_ao {
    uint32_t pair;
    uint32_t site;          // flat position when written by the pileup kernel; then the rank of the
                            // site in the position-sorted site table (k_ao_rank)
    uint32_t obs_idx;       // arrival order (orders the two mates of a self pair)
    uint16_t mm;
    uint8_t base;
    uint8_t pad;
};

struct PileupArgs {
    const uint2 *rec;           // packed isx_obs, padded to a multiple of ISX_CHUNK with sentinels
    const uint2 *win_range;     // per window: [lo, hi) in records (multiples of ISX_CHUNK)
    const uint8_t *ref;
    const uint32_t *pair;       // read-pair id per record (linkage only)
    const uint16_t *thr;        // lut_n entries: folded presence threshold per coverage (build_thresholds)
    int32_t lut_n, fallback;
    uint32_t n_pos;
    int32_t W, logW, M;
    int32_t n_win;
    int32_t min_cov;
    int32_t debug_mode;
    int32_t qcap;               // deferred-clonality queue capacity (entries)
    double min_freq;
    // outputs
    uint4 *counts;              // dense path (M == 1): [n_pos]
    float *clon;                // dense path: [n_pos]
    isx_entry *entries;         // mm path
    uint32_t cap_entries;
    isx_snv *snv;
    uint32_t cap_snv;
    isx_site *sites;
    uint32_t cap_sites;
    isx_ao *ao;                 // allele observations, exactly sized slabs per SNP site
    uint32_t cap_ao;
    int32_t enable_linkage;
    uint32_t *cursors;
    uint32_t *flags;
};

void launch_pileup(const PileupArgs &a, int block, size_t lds, int grid_dense, hipStream_t s);
size_t pileup_lds_bytes(int W, int M, int qcap, int linkage);

struct LinkageBuffers;      // defined in isx_linkage.hip
