// isx_internal.h -- shared between the HIP translation units of libinstrain_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/instrain_amd.h"

#define ISX_PAD32 0x0700FFFFu          // compact padding record: base code 7 (never counted), mm 0
#define ISX_GROUP16 512              // short stream: records per position base (one wave-wide 16-byte load of 8 records per lane)
#define ISX_GROUP 256                // compact stream: records per position base (one wave-wide 16-byte load)
#define ISX_CHUNK 1024              // observation directory granule (records)
#define ISX_DENSE_PAD 8             // k_pileup_dense: extra words per counter row (junk columns of the packed 16-bit decode)
// read-segment stream (include/instrain_amd.h isx_segs; seg_encode.h): 64-byte records, 16 per position base
#define ISX_SEG_GROUP 16
#define ISX_SEG_PAD 32              // k_pileup_dense on segments: extra words per counter row = ISX_SEG_LM columns before position 0 of
#define ISX_SEG_LM 16               // the window + 16 after its last one: a 10-base word that straddles a window edge needs no per-base test
#define ISX_PK16_MAX_W 3264         // ... whose byte offsets (5 (W + 8) + 7) * 4 must stay below 65536
#define ISX_PAD 2048                // the record stream is padded to a multiple of this (whole directory chunks, 16-byte loads)
#define ISX_SENTINEL 0xFFFFFFFFu    // gpos of padding records (never inside a window)
// The compact streams (2- / 4-byte records) are followed by this many bytes of padding records (and the group bases by
// ISX_TAIL_GROUPS entries): k_pileup_dense streams them with unconditional 16-byte loads, a workgroup's last round may run
// this far past its window's range -- what it reads there is either beyond the window (dropped) or padding.
#define ISX_TAIL_BYTES 65536
#define ISX_TAIL_GROUPS 256

void isx_set_error(const std::string &msg);

// Device memory for buffers that are created and dropped again and again with similar sizes (the linkage stages' scratch,
// the tables of a pipe's slots): freed blocks are kept per size class and handed out again, so a hipMalloc -- which takes
// the process' address-space lock and stalls for as long as another thread is giving gigabytes back to the system -- only
// happens the first time.  isx_dev_trim() (isx_ctx_destroy) really frees what is cached.
hipError_t isx_dev_malloc(void **p, size_t bytes);
void isx_dev_free(void *p);
// the same for pinned host memory (a pipe's staging arenas: pinning and unpinning a few hundred MB takes tens of ms each
// way, more than a small job's device work), at most 8 GiB (an eighth of the host's RAM) kept; device blocks: a sixth of the device's memory, per device
hipError_t isx_pin_malloc(void **p, size_t bytes);
void isx_pin_free(void *p);
void isx_dev_trim();        // both caches

// Device -> PINNED host memory.  Pieces below 1 MiB (ISX_D2H_DMA_MIN) leave by a copy kernel: hipMemcpyAsync serves both directions of
// this stack's copies from one queue, so a small copy-out waited behind the next batch's 100 MB copy-in (profiles/r03_stream_ab.md), and
// pinned host memory is mapped into the device's address space.  Position-sized tables go by hipMemcpyAsync: a kernel that streams
// megabytes into host memory fills the memory system's queues with PCIe writes, and every other kernel's HBM traffic and the copy-in
// DMA wait behind them (round 4: whole-database pass 100 -> 66 ms).
hipError_t isx_copy_to_host(void *hdst_pinned, const void *dsrc, size_t bytes, hipStream_t stream);
// up to ISX_COPY_JOBS tables to pinned host memory in one launch (16-byte aligned both ends; the last piece is copied whole: both sides have the room)
#define ISX_COPY_JOBS 8
struct isx_copy_job { void *dst; const void *src; size_t bytes; };
hipError_t isx_copy_multi_to_host(const isx_copy_job *jobs, int n, hipStream_t stream);
// the same with the route given: by_kernel = the copy kernels whatever the size (what a small batch hands back, isx_pipe.hip's level tables)
hipError_t isx_copy_to_host_route(void *hdst_pinned, const void *dsrc, size_t bytes, hipStream_t stream, bool by_kernel);
// Small read-backs into ANY host memory (table sizes, the last element of a scan, a few hundred rows): the same kernel route
// through a pinned scratch of the calling thread (1 MiB) -- a 4-byte hipMemcpyAsync queues behind whatever 100 MB copy-in the DMA
// engine is busy with (up to ~2 ms each, several per batch).  isx_read_back enqueues, isx_read_sync waits for the stream and
// delivers the values; larger requests than the scratch has room for go to hipMemcpyAsync.
// rows [0, min(*cursor - base, cap_rows)) of a device table into pinned host memory: the count is read on the device, so the
// copy can be enqueued behind the kernel that produces the rows without the host knowing how many there will be -- and without
// copying a fixed-size prefix of mostly unused rows (up to 15 bytes beyond the last row are copied: both tables are larger)
hipError_t isx_copy_rows_to_host(void *hdst_pinned, const void *dsrc, const uint32_t *d_cursor, uint32_t base, uint32_t row_bytes,
                                 size_t cap_rows, hipStream_t stream);
// inside a range the caller registered with isx_host_register (pinned: the copy engine reads it where it lies)
bool isx_host_is_registered(const void *ptr, size_t bytes);
hipError_t isx_read_back(void *host_dst, const void *dsrc, size_t bytes, hipStream_t stream);
hipError_t isx_read_sync(hipStream_t stream);
// batching: between isx_read_batch(true) and the calling thread's next isx_read_sync the copies of isx_read_back are not launched one by one but
// together, by that isx_read_sync (one kernel) -- for sources that do not change in between (a pipe's finisher: window lists, exact-coverage rows, the
// linkage stages' state words + rows).  isx_read_batch(false) / isx_read_drop end it.
void isx_read_batch(bool on);
// waits that sleep between polls (isx_api.hip): what every host thread of the library waits for the device with
hipError_t isx_wait_event(hipEvent_t e);
hipError_t isx_wait_stream(hipStream_t s);
void isx_read_drop();       // forget the calling thread's pending read-backs (every failing HIP_TRY does: their destinations may be
                            // stack variables of the function that is about to return)
// hipMalloc outside the caches, with the trim-and-retry of isx_dev_malloc when the device is full of cached blocks
hipError_t isx_raw_dev_malloc(void **p, size_t bytes);
template <class T> static inline hipError_t isx_raw_dev_malloc(T **p, size_t bytes) { return isx_raw_dev_malloc(reinterpret_cast<void **>(p), bytes); }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            isx_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                  \
            isx_read_drop();                                                                   \
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
#define ISX_FLAG_SITES_LOOSE 128u      // k_pileup_mm: a site was allocated outside its window's range (row queue full): the window site table does not cover the batch

// cursors (dev_cursors[i], uint32)
enum { CUR_ENTRIES = 0 /* mm path: overflow entries */, CUR_SNV = 1, CUR_SITES = 2, CUR_AO = 3, CUR_SLEV = 4,
       CUR_ENT_TOTAL = 5, CUR_RARE = 6 /* dense path: entries of the sparse clonTR list */,
       CUR_SAT = 7 /* dense path: positions whose coverage reaches sat_thr (entries of the exact-coverage list) */,
       CUR_CLON = 8 /* dense path: entries of the sparse clonality list */,
       CUR_COVX = 9 /* dense path, 4-bit coverage plane: windows that also wrote a 16-bit row */, CUR_N = 10 };

// SNP site record: a position where update_snp_table returned anySNP (snv_utilities.py:129-133).
// Holds what linkage needs later: the `bases` set and where the per-level counts live.
struct isx_site {
    uint32_t gpos;
    uint32_t entry_off;     // mm path: index of the site's first isx_slev row; dense path: index of the site's SNV row (its counts)
    uint16_t n_levels;      // mm path: number of levels present
    uint8_t mask;           // `bases` set, bit b = base b
    uint8_t pad;
};

// per-level counts of a SNP site (snv2mm2counts[pos], profile_utilities.py:266), mm path
struct isx_slev {
    uint16_t mm, pad;
    uint32_t cnt[4];
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

// Philox4x32-10 counter-based generator (Salmon et al., SC'11): the reference's two unseeded-random
// outputs (np.random.choice in snv_utilities.py:242 and linkage.py:200) are produced here from
// (seed, position, mm level, draw index) so that a run is reproducible and order-independent.
struct Philox {
    uint32_t k0, k1;
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo)
    {
        const uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32); lo = (uint32_t)p;
    }
    __host__ __device__ inline void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t *out) const
    {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            const uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// np.random.choice(4, n_draws, p=w/sum(w)) as numpy does it: cdf = cumsum(p) / cdf[-1], uniform
// draws, searchsorted(side='right'); returns the counts of the 4 categories.
__host__ __device__ inline void rarefy4(const Philox &ph, uint32_t c0, uint32_t c1, uint32_t domain, const double *p,
                                        int n_draws, uint32_t *rc)
{
    double cdf[4];
    cdf[0] = p[0]; cdf[1] = cdf[0] + p[1]; cdf[2] = cdf[1] + p[2]; cdf[3] = cdf[2] + p[3];
    const double tot = cdf[3];
    cdf[0] /= tot; cdf[1] /= tot; cdf[2] /= tot; cdf[3] /= tot;
    rc[0] = rc[1] = rc[2] = rc[3] = 0;
    for (int d = 0; d < n_draws; d += 4) {
        uint32_t r[4];
        ph(c0, c1, (uint32_t)(d >> 2), domain, r);
        for (int j = 0; j < 4 && d + j < n_draws; j++) {
            const double u = (double)r[j] * (1.0 / 4294967296.0);
            const int idx = (u >= cdf[0]) + (u >= cdf[1]) + (u >= cdf[2]);
            rc[0] += idx == 0; rc[1] += idx == 1; rc[2] += idx == 2; rc[3] += idx == 3;
        }
    }
}

struct PileupArgs {
    const uint2 *rec;           // wide stream: packed isx_obs, padded to a multiple of ISX_PAD with sentinels; or ...
    const uint32_t *rec32;      // ... compact stream (rec == NULL): delta:16 | mm:8 | base:3 per record, position =
    const uint32_t *gbase;      //     gbase[record / 256] + delta; padding records are ISX_PAD32; or ...
    const uint16_t *rec16;      // ... short stream (n_mm_bins == 1 only; rec, rec32 == NULL): delta:13 | base:3 per record,
                                //     position = gbase[record / 512] + delta; padding records are 0xFFFF
    const uint4 *seg;           // ... read-segment stream (rec, rec32, rec16 == NULL): 64-byte records as four 16-byte quarters, the
                                //     first word of a record = delta:16 | len:8 | mm:8, start = gbase[record / 16] + delta, then 15 words of
                                //     ten 3-bit base codes; `pair` (linkage) is indexed by RECORD
    const uint4 *drec;          // ... reference-delta stream (one mm bin; rec, rec32, rec16, seg == NULL): 32-byte records as two 16-byte halves
                                //     (include/instrain_amd.h ISX_DREC_*), start = gbase[record / 32] + delta; `pair` is indexed by RECORD
    const uint2 *win_range;     // per window: [lo, hi) in records (multiples of ISX_CHUNK; of ISX_SEG_GROUP for the segment stream)
    const uint8_t *ref;         // reference base code per flat position (ref_packed 0) -- or, in a pipe slot, packed: 1 = two codes per byte
    int32_t ref_packed;         // (round 3: position 2 i in the low nibble of byte i), 2 = a 2-bit plane, four positions a byte (A C T G), with
    const uint8_t *ref_n;       // ref_n = bit plane of the positions that are NOT A/C/T/G (NULL: the batch has none): 0.25 / 0.375 B a position
    const uint32_t *pair;       // read-pair id per record (linkage only), or NULL and ...
    const uint2 *pair_runs;     // ... runs of equal pair ids: (first device record, pair id), ascending; run_index[c] = the run
    const uint32_t *run_index;  //     that holds device record 1024 c (pipe slots: ~0.06 B per record over PCIe instead of 4)
    uint32_t n_runs;
    const uint32_t *gpos;       // positions alone, 4 B per record (linkage only): what the allele pass streams ...
    const uint16_t *gpos16;     // ... or, when every 1024-record chunk spans < 65535 positions, 2 B deltas to
    const uint32_t *chunk_base; //     the chunk's / group's lowest position (0xFFFF = padding record); gpos is NULL then
    int32_t gpos16_shift;       //     log2(16-byte loads of 8 deltas per base): 7 = per ISX_CHUNK, 5 = per ISX_GROUP
    const uint16_t *thr;        // lut_n entries: folded presence threshold per coverage (build_thresholds)
    int32_t lut_n, fallback;
    uint32_t n_pos;
    int32_t W, logW, M;
    int32_t n_win;
    int32_t min_cov;
    int32_t debug_mode;
    int32_t qcap;               // deferred-clonality queue capacity (entries)
    int32_t rqcap;              // mm path: row-queue capacity (positions with SNV rows per window)
    int32_t stage_off;          // allele pass: LDS word offset of the per-wave hit stage (0 = aliases the counters)
    int32_t dlt_off;            // reference-delta stream: LDS word offset of the coverage-difference row (pileup_lds_bytes)
    int32_t stripe;             // reference-delta stream, packed rows, no count table: the stripe path of k_pileup_dense (ISX_LAYOUT_NO_STRIPES switches it off)
    int32_t pad, lm;            // dense path: counter row stride = W + pad words, position 0 of the window at column lm
    double min_freq;
    // outputs
    uint4 *counts;              // dense path (M == 1): [n_pos]; NULL = not kept (a pipe slot without want_counts: 16 B/pos less to write)
    float *clon;                // dense path: [n_pos]
    float *clon_r;              // rarefied clonality: dense [n_pos] / mm path [cap_entries]; pre-filled with NaN
    uint16_t *cov16;            // dense path, pipe slots: min(coverage, 65535) per position (NULL = not wanted)
    uint8_t *cov8;              // ... and min(coverage, 255) for the 1-byte hand-back of a shallow batch (NULL = not wanted)
    // ... or, a lean slot's shallow batch (reference-delta records, 16-bit LDS rows): min(coverage, 15), two positions a byte (cov4, low nibble
    // = the even position) -- and, of every window that holds a position beyond 15, the whole window as a 16-bit row: row k of cov_rows
    // (k from CUR_COVX) with cov_row_win[k] = the window.  0.5 B a position home for a metagenome at depth 3 instead of 1
    uint8_t *cov4;
    uint16_t *cov_rows;
    uint32_t *cov_row_win;
    uint32_t cap_cov_rows;      // entries of cov_rows (rows that would reach beyond are not written; the host repeats such a batch with cov8)
    uint2 *sat;                 // ... exact (gpos, coverage) of the positions whose coverage reaches sat_thr (255 with cov8, else 65535)
    uint32_t cap_sat, sat_thr;
    uint2 *clon_list;           // dense path, pipe slots: (gpos, float bits of clonT) of the positions whose clonality is NOT 1.0 (more than
                                // one base observed), unordered: every other position that reaches min_cov has exactly 1.0
    uint32_t cap_clon;
    uint2 *rare;                // dense path, pipe slots: (gpos, float bits of clonTR) of the positions that have one, unordered
    uint32_t cap_rare;
    int32_t min_cov_r;          // rarefied_coverage; <= 0 disables the rarefied output
    uint32_t seed_lo, seed_hi;
    isx_entry *entries;         // mm path: per-window slabs [n_win][slab] then the overflow region
    uint32_t slab;              // entries per window slab
    uint32_t cap_ovf;           // overflow capacity
    uint64_t ovf0;              // index of the overflow region = n_win * slab
    uint32_t *win_nent;         // [n_win] entries used in each slab
    uint32_t *win_site_base, *win_site_cnt;     // [n_win] k_pileup_mm with linkage: where a window's SNP sites lie in `sites` and how many (round 6: position order without a sort)
    isx_slev *slev;             // per-level counts of the SNP sites
    uint32_t cap_slev;
    isx_snv *snv;
    uint32_t cap_snv;
    isx_site *sites;
    uint32_t cap_sites;
    isx_ao *ao;                 // allele observations, exactly sized slabs per SNP site
    uint32_t cap_ao;
    int32_t enable_linkage;
    uint32_t *win_rec;          // dense path: [n_win][8] = (first slot, count) of the window's SNV rows | SNP sites | clonality list | clonTR list
                                // entries (NULL = not wanted): what k_win_gather orders the tables by
    uint32_t *cursors;          // monotonic across runs: slot = atomicAdd(...) - base[...] (no per-run memset)
    uint32_t *flags;
    uint32_t base[CUR_N];       // cursor values when this run started (host copy of the last read-back)
    uint32_t *host_state;       // mapped pinned [CUR_N + 4]: k_publish_state copies cursors | flags here
    // deferred publication of the PREVIOUS pass on this stream (pipelined launches): workgroup 0 copies that
    // pass's cursors to its host state before it starts -- one kernel per step instead of two
    const uint32_t *pub_cursors;
    uint32_t *pub_host_state;
    uint32_t pub_epoch;
    // mm path, level-sparse hand-back (pipe slots with n_mm_bins <= 32 and no want_counts; k_pileup_mm<..., SPARSE>): what
    // shrink_basewise keeps of the (position, mm) levels (profile_utilities.py:337-350) in 1-3 bytes a level instead of a 32-byte entry --
    // which levels a position has, every present level's coverage, and the few clonalities that are not 1.0
    void *lev_mask;             // [n_pos] bit m = level m is present at the position; element width lev_mask_bytes (1, 2 or 4)
    void *lev_cov;              // one element (lev_cov_bytes: 1 or 2) per present (position, level): min(the level's coverage, sat_thr); a window's
                                // levels lie together, position-major (ascending position, then ascending level), from lev_win_off[window] on
    uint32_t *lev_win_off;      // [n_win] index of the window's first level (slots come from CUR_ENT_TOTAL, in the order the windows get there)
    uint32_t cap_lev;           // levels lev_cov (and, when kept, the flat entry table) have room for
    int32_t lev_mask_bytes, lev_cov_bytes;
    // clon_list / rare / sat hold (level index, value) here: the clonT values other than 1.0, every clonTR value, the exact coverage of
    // the levels at or beyond sat_thr; `entries` (NULL in a lean slot) is then the flat table entries[level index]
};

void launch_pileup(const PileupArgs &a, int block, size_t lds, int grid, int packed, hipStream_t s, hipEvent_t ev_start,
                   hipEvent_t ev_stop);        // record format from a.rec16 / a.rec32 / a.rec; the events bracket the dispatch
void launch_publish_state(const PileupArgs &a, uint32_t epoch, hipStream_t s);
// position order of the dense path's tables by window-ordered gather instead of sorting (isx_pileup.hip: k_win_scan, k_win_gather)
void launch_win_order(const uint32_t *win_rec, uint32_t *win_out, int n_win, int W, const isx_snv *snv_raw, isx_snv *snv, const isx_site *sites_raw,
                      isx_site *sites, const uint2 *clon_raw, uint2 *clon, const uint2 *rare_raw, uint2 *rare, uint32_t *scan_state, uint32_t epoch, hipStream_t s);
// scan_state: 8 words per chunk of 1024 windows, zeroed once when allocated; epoch: a value no earlier launch on this state used (not 0)
void launch_extract_gpos(const uint2 *rec, const uint32_t *rec32, const uint32_t *gbase, uint32_t *gpos, uint16_t *gpos16,
                         const uint32_t *base16, uint32_t base16_records, uint64_t n_rec, hipStream_t s);
size_t pileup_lds_bytes(int W, int M, int qcap, int rqcap, int linkage, int packed, int block, int segs, int *stage_off, int *dlt_off = nullptr);

struct LinkageBuffers;      // defined in isx_linkage.hip
