// isx_linkage.h -- host-side interface of the linkage pipeline (isx_linkage.hip)
#pragma once
#include "isx_internal.h"

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;     // elements
};

struct LinkageBuffers {
    DevBuf<uint32_t> site_keys, site_keys2, site_gpos, site_split;
    DevBuf<isx_site> sites_sorted;
    DevBuf<isx_ao> ao2;
    DevBuf<uint32_t> ao_key, ao_key2;
    DevBuf<uint32_t> incr_cnt, incr_off;
    DevBuf<uint64_t> keys, keys2, ukeys;
    DevBuf<uint32_t> ucnt, n_runs, rows_per, row_off;
    DevBuf<isx_ld> ld;
    DevBuf<uint8_t> temp;
    void release();
};

struct LinkageIn {
    hipStream_t stream;
    hipEvent_t *ev;             // 6 events: start, sites, allele, group, incr, ld
    uint64_t n_pairs;           // 0 = unknown
    isx_ao *ao;                 // written by the pileup kernel (site field = flat position)
    uint32_t n_ao;
    const isx_site *sites;      // unsorted, from k_pileup_call
    uint32_t n_sites;
    const isx_entry *entries;   // mm path
    const uint4 *counts;        // dense path
    const int64_t *split_bounds;
    int n_splits;
    int M;
    int min_snp;
};

struct LinkageOut {
    uint64_t n_ao = 0, n_increments = 0, n_edges = 0, n_ld = 0;
};

int run_linkage(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out);
