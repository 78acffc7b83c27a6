// isx_batch.h -- the context / resident-batch objects behind the C ABI, shared by isx_api.hip (create / run /
// fetch) and isx_pipe.hip (the streaming hand-over: reusable batch slots fed through pinned staging).
#pragma once
#include <vector>

#include "isx_internal.h"
#include "isx_linkage.h"
#include "isx_summary.h"

namespace isxenc { class HostPool; }

struct isx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // passes (pileup kernel + cursor publication) run on one of two streams, alternating by batch: consecutive
    // batches' kernels are in different queues, so the next one's workgroups move in as the current one's retire
    // (no kernel-boundary gap); everything else of a batch runs on `stream` after its pass is known to be complete
    hipStream_t pstream[2] = {nullptr, nullptr};
    uint32_t side_mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ... and those CUs as a queue mask (all zero: no reserve)
    int pass_cus = 256;                 // CUs the pass queues may use (256 minus what ISX_PASS_CU_RESERVE keeps free for the finishers' short kernels)
    struct isx_batch *unpublished[2] = {nullptr, nullptr};   // per pass stream: batch whose last pass has no publication enqueued yet
    unsigned n_created = 0;
    unsigned n_pipes = 0;               // pipes ever created on this context
    uint8_t *d_lut = nullptr;
    std::vector<int32_t> h_lut;
    int32_t lut_n = 0, fallback = 0;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
};

// a queue for what is NOT a pass (copy-in, finishers' chains, copy-out): on the reserved CUs when the context keeps some (isx_api.hip)
hipError_t isx_side_stream_create(isx_ctx *c, hipStream_t *s);

struct isx_batch {
    isx_ctx *ctx = nullptr;
    isx_params prm{};
    int64_t n_pos = 0, n_obs = 0;
    int64_t cap_pos = 0, cap_obs = 0;   // pipe slots: what the device buffers were sized for (0 = this batch's own n_pos / n_obs)
    size_t slab_region = 0;             // pipe slots, mm path: entries set aside for the window slabs (0 = n_win * slab)
    bool arena = false;                 // pipe slots: the input arrays live in the slot's arena, not in own allocations
    bool segs = false;                  // the stream is read segments (d_seg / d_drec), not observation records
    bool drec = false;                  // ... as 32-byte reference-delta records (one mm bin; d_drec) instead of 64-byte segment records
    uint64_t n_rec = 0;         // padded
    uint64_t n_pairs = 0;
    int32_t n_splits = 0;
    int W = 0, logW = 0, M = 1, n_win = 0, block = 1024, grid = 0, packed = 0;
    size_t lds = 0;
    // device
    uint2 *d_rec = nullptr;             // wide stream (8-byte isx_obs) -- or:
    uint32_t *d_rec32 = nullptr, *d_gbase = nullptr;    // compact stream (4-byte records + one position base per 256)
    uint16_t *d_rec16 = nullptr;                        // short stream (n_mm_bins == 1: 2-byte records, base per 512)
    uint4 *d_seg = nullptr;                             // read-segment stream (64-byte records, base per 16; d_pair per record)
    uint4 *d_drec = nullptr;                            // reference-delta stream (32-byte records, base per 32; d_pair per record)
    int dlt_off = 0;
    uint32_t *d_pair = nullptr, *d_gpos = nullptr, *d_cbase = nullptr;
    uint2 *d_pair_runs = nullptr;       // pipe slots: pair ids as runs (PileupArgs::pair_runs) instead of d_pair
    uint32_t *d_run_index = nullptr;
    uint32_t n_runs = 0;
    uint16_t *d_gpos16 = nullptr;
    int gpos16_shift = 7;
    uint8_t *d_ref = nullptr;
    int ref_packed = 0;                 // pipe slots: 2 = a 2-bit plane (+ d_ref_n: bit plane of the positions that are not A/C/T/G), see PileupArgs
    uint8_t *d_ref_n = nullptr;
    uint2 *d_win = nullptr;
    uint16_t *d_thr = nullptr;
    int qcap = 1024, rqcap = 0, stage_off = 0;
    bool in_flight = false, publish_enqueued = true, tim_pending = false;
    int ps = 0;                                           // which pass stream of the context
    int64_t *d_bounds = nullptr;
    uint4 *d_counts = nullptr;
    float *d_clon = nullptr;
    float *d_clon_r = nullptr;       // rarefied clonality of the dense path [n_pos]
    uint16_t *d_cov16 = nullptr;     // pipe slots (dense path): coverage per position for the shrunk hand-back
    uint2 *d_rare = nullptr;         // pipe slots (dense path): sparse clonTR list
    size_t cap_rare = 0;
    uint32_t n_rare = 0, n_sat = 0;  // of the last pass: list entries (may exceed cap_rare: list unusable), saturated positions
    // pipe slots (dense path), the hand-back of a shallow batch: 1-byte coverage, exact (gpos, coverage) of the positions beyond
    // 255 / 65535, clonality as a sparse (gpos, value) list -- sorted by position on the device before it leaves
    uint8_t *d_cov8 = nullptr;
    uint2 *d_sat = nullptr, *d_clon_list = nullptr, *d_clon_sorted = nullptr;
    size_t cap_sat = 0, cap_clon = 0;
    uint32_t n_clon = 0;
    int spin_us = 2000;                 // finish_pass: how long the finisher polls the epoch word before it falls back to a stream wait (a pipe
                                        // whose caller gave it few host threads -- a rank of a multi-GPU job on a shared CPU quota -- polls briefly)
    bool nib_pass = false;              // the last pass really wrote the 4-bit plane (nib_out asked for, and the kernel is the packed reference-delta one)
    bool nib_out = false;               // this pass: coverage as the 4-bit plane (in d_cov8) + 16-bit rows of the windows beyond 15 (in d_cov16): lean slots
    uint32_t *d_cov_row_win = nullptr;  // ... and the window of every such row
    size_t cap_cov_row_win = 0;
    uint32_t n_cov_rows = 0;
    bool sparse_out = false, cov8_out = false;   // this pass: write the sparse clonality list / the 1-byte coverage
    bool rare_dense = true;             // this pass writes the dense clonTR array (a lean slot's shallow batch: the list alone; finish_slot repeats the pass otherwise)
    bool lean = false, clon_dense = false;       // lean slot (isx_pipe_params.lean_output): the dense clonality array / the 16-bit coverage
                                                 // are written only when a batch needs them (clon_dense: its clonality list did not fit)
    // mm path, level-sparse hand-back (pipe slots with n_mm_bins <= 32 and no want_counts; PileupArgs::lev_*): per position the mask of its
    // levels, per present level its coverage in lev_cov_bytes (1 for a batch shallower than 64x, else 2), the window's first level index;
    // d_clon_list / d_rare / d_sat then hold (level index, value).  d_entries is a FLAT table indexed by level (handed to the consumers of
    // the slab layout as "no slabs + n_entries overflow entries": entry_wins / n_ovf) -- and absent in a lean slot.
    bool lev_sparse = false;
    void *d_lev_mask = nullptr, *d_lev_cov = nullptr;
    uint32_t *d_lev_win_off = nullptr;
    size_t cap_lev = 0;
    int lev_mask_bytes = 0, lev_cov_bytes = 1;
    isx_entry *d_entries = nullptr;  // mm path: [n_win][slab] slabs, then cap_ovf overflow entries
    uint32_t *d_win_nent = nullptr;
    uint32_t *d_win_site_base = nullptr, *d_win_site_cnt = nullptr;    // k_pileup_mm with linkage: a window's range in the site table
    size_t cap_win_sites = 0;
    bool sites_loose = false;           // the last pass allocated a site outside its window's range (ISX_FLAG_SITES_LOOSE)
    isx_slev *d_slev = nullptr;
    uint32_t slab = 0;
    size_t cap_ovf = 0, cap_slev = 0;
    isx_snv *d_snv = nullptr;
    isx_site *d_sites = nullptr;
    // dense path: the kernel fills these in the order the windows reach the cursors; k_win_gather then writes d_snv / d_sites /
    // d_rare / d_clon_sorted in position order (launch_pass) -- everything downstream reads ordered tables and sorts nothing
    isx_snv *d_snv_raw = nullptr;
    isx_site *d_sites_raw = nullptr;
    uint2 *d_rare_raw = nullptr;
    uint32_t *d_win_rec = nullptr, *d_win_out = nullptr;
    size_t cap_snv_raw = 0, cap_sites_raw = 0, cap_rare_raw = 0, cap_win = 0;
    bool ordered = false;            // the last pass left its tables ordered
    isx_ao *d_ao = nullptr;
    uint32_t *d_cursors = nullptr, *d_flags = nullptr;   // one allocation: cursors[CUR_N] | flags[4]
    uint32_t *h_state = nullptr;                          // mapped pinned mirror, written by k_publish_state
    uint32_t *d_host_state = nullptr;                     // device address of h_state
    uint32_t base[CUR_N] = {};                            // cursor values at the start of the next run
    uint32_t epoch = 0;                                   // run counter, echoed by k_publish_state
    size_t cap_entries = 0, cap_snv = 0, cap_sites = 0, cap_ao = 0;
    LinkageBuffers L;
    const isx_ld *ld_host = nullptr;    // bucket chain: the first n_ld_host LD rows, already on the host (L.h_ld)
    size_t n_ld_host = 0;
    uint32_t n_ao_pass = 0;             // allele observations of the last pass (finish_pass_sizes -> finish_pass_link)
    int link_chain = 0;                 // which chain the last pass took (LinkageOut::chain)
    SummaryBuffers S;
    CompareBuffers C;
    hipEvent_t ev_sum[2] = {};
    hipEvent_t ev[10] = {};
    bool ran = false;
    uint32_t n_ovf = 0;
    isx_sizes sizes{};
    isx_timings tim{};
};


// windows of the entry table's slab region as its consumers see it (fetch_entries_sorted, the summaries): a level-sparse slot keeps a flat
// table, i.e. no slabs and sizes.n_entries "overflow" entries
static inline uint32_t entry_wins(const isx_batch *b) { return b->lev_sparse ? 0u : (uint32_t)b->n_win; }

// ---- shared between isx_api.hip and isx_pipe.hip ----
std::vector<uint16_t> build_thresholds(const std::vector<int32_t> &lut, int32_t fallback, double min_freq);

// window size of a batch of n_pos positions (0 = params.window unset -> auto), see isx_batch_create
void batch_pick_block(isx_batch *b);
int batch_window_for(const isx_batch *b, int64_t n_pos, bool packed);

// window -> record range directory from the per-chunk position ranges (prefix-max / suffix-min); returns
// the longest record range of a window
uint64_t build_window_directory(const uint32_t *cmin, const uint32_t *cmax, const uint8_t *cany, uint64_t n_chunks, int W,
                                int64_t n_pos, std::vector<uint2> &win, uint32_t chunk = ISX_CHUNK);
uint64_t build_window_directory_mt(isxenc::HostPool &pool, const uint32_t *cmin, const uint32_t *cmax, const uint8_t *cany, uint64_t n_chunks, int W,
                                   int64_t n_pos, std::vector<uint2> &win, uint32_t chunk, std::vector<uint32_t> &pmax, std::vector<uint32_t> &smin);

// derived launch geometry (LDS bytes, persistent grid, row-queue capacity, entry slab) once b->W / b->packed are set
int batch_set_geometry(isx_batch *b);

extern "C" {
int launch_pass(isx_batch *b);
int finish_pass(isx_batch *b, uint32_t *cap_flags, hipStream_t link_stream = nullptr);
int finish_pass_sizes(isx_batch *b, uint32_t *cap_flags, hipStream_t link_stream = nullptr);
int finish_pass_link(isx_batch *b, hipStream_t link_stream = nullptr);
// grow the tables named by cap_flags (x4 up to their hard bounds); the pass must then be repeated
int batch_grow_tables(isx_batch *b, uint32_t cap_flags);
}
