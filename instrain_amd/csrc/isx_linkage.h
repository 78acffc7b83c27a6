// isx_linkage.h -- host-side interface of the linkage pipeline (isx_linkage.hip)
#pragma once
#include <vector>
#include "isx_internal.h"

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;     // elements
};

constexpr size_t LD_HEAD = 1;               // rows of ld_block in front of the LD rows (>= 64 bytes of state words)
constexpr size_t LD_PREFIX_ROWS = 2047;     // rows that come home with the state words ((LD_HEAD + rows) * 72 bytes: a multiple of 16, one copy kernel)

struct DenseSplit {         // one split of the dense MFMA path
    uint64_t xt_off;        // byte offset of its column-major X^T block
    uint32_t rpad;          // rows (read pairs), padded to 128
    uint32_t ctiles;        // column tiles of 32 (4 columns per site)
    uint32_t first_site, n_sites, first_row;
    uint32_t tile0;         // index of its tile (0, 0) in the tile counters (tiles ordered I, then J >= I)
};
struct DenseTile { uint32_t slot, I, J, pad; };     // a 128 x 128 column block (I <= J, units of 4 tiles)

struct LinkageBuffers {
    DevBuf<uint32_t> site_keys, site_keys2, site_gpos, site_split;
    DevBuf<isx_site> sites_sorted;
    DevBuf<isx_ao> ao2;
    DevBuf<uint32_t> ao_key, ao_key2;
    DevBuf<uint32_t> incr_cnt, incr_off;
    DevBuf<uint64_t> keys, keys2, ukeys;
    DevBuf<uint32_t> ucnt, n_runs, rows_per, row_off;
    DevBuf<isx_ld> ld_block;     // [LD_HEAD rows: the bucket chain's state words][the LD rows]
    DevBuf<isx_ld> ld;           // a view of ld_block behind its head (cap = rows it can hold; ld_block owns the memory)
    DevBuf<uint8_t> temp;
    // bucket chain (isx_linkage.hip): per-pair chains of the allele observations, a bucket of pair increments per first site
    DevBuf<uint64_t> chain_head;
    DevBuf<uint32_t> next, site_cnt, site_off, site_cur, site_nu, site_rows, site_row_off, site_list1, site_list2, win_off, win_list;
    DevBuf<uint64_t> edge_list;  // (site1, index of the edge's first key in the site's bucket) as uint2
    uint32_t chain_epoch = 0;
    std::vector<isx_ld> h_ld;    // what the chain's one read-back brought: state words + the first rows
    // dense path
    DevBuf<uint64_t> key64, key64b;
    DevBuf<uint32_t> head, row_id, first_row, first_site, split_slot, tile_cnt, tile_off, vals, vals2;
    DevBuf<DenseSplit> dsplits;
    DevBuf<DenseTile> dtiles;
    DevBuf<uint8_t> xt;
    void release();
};

struct LinkageIn {
    hipStream_t stream;
    hipEvent_t *ev;             // 6 events: start, sites, allele, group, incr, ld
    hipEvent_t *ev_mfma;        // 2 events around k_dense_gemm<false> (dense path)
    uint64_t n_pairs;           // 0 = unknown
    isx_ao *ao;                 // written by the pileup kernel (site field = flat position)
    uint32_t n_ao;
    const isx_site *sites;      // from the pileup kernel: unsorted, or ...
    uint32_t n_sites;
    bool sites_ordered = false; // ... already in position order (dense path: k_win_gather) -- the site sort is skipped
    const uint32_t *win_site_base = nullptr, *win_site_cnt = nullptr;   // ... or (mm path) window by window: [n_win] first site / sites of every window
    int n_win = 0;              //     (k_site_order puts the table in position order without a device-wide sort)
    const isx_slev *slev;       // mm path: per-level counts of the SNP sites
    const isx_snv *snv;         // dense path: the SNV rows (isx_site::entry_off indexes them)
    const int64_t *split_bounds;
    int n_splits;
    int M;
    int min_snp;
    int mode;                   // 1 sparse, 2 dense MFMA (n_mm_bins == 1)
    Philox philox;              // stream of the rarefied LD columns
};

struct LinkageOut {
    uint64_t n_ao = 0, n_increments = 0, n_edges = 0, n_ld = 0;
    const isx_ld *ld_host = nullptr;    // bucket chain: the first n_ld_host rows are on the host already (valid until the buffers' next run)
    uint64_t n_ld_host = 0;
    int chain = 0;                      // 1 sorted chain (rocPRIM sorts), 2 dense MFMA, 3 bucket chain
    uint64_t dense_tiles = 0, dense_bytes = 0, dense_macs = 0;   // dense path only
};

int run_linkage(const LinkageIn &in, LinkageBuffers &B, LinkageOut &out);
