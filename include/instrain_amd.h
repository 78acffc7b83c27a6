/*
 * instrain_amd.h -- C ABI of libinstrain_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the `inStrain profile` hot path.  The reference has no FFI layer
 * (it is pure Python); the seam this library replaces is
 *
 *   profile_split(samfile, scaffold, start, end, split_number, seq, R2M, null_model, **kw)
 *       -> SplitObject          /root/reference/inStrain/profile/profile_utilities.py:115-216
 *
 * called once per split from split_profile_wrapper_groups (profile_utilities.py:92-112).
 * Here MANY splits are profiled by one call: a "batch" is a set of splits laid out in one
 * flat position space (flat position = split offset + position inside the split), so the
 * device sees one long, position-clustered observation stream and >>256 workgroups.
 *
 * Plain C types only; no torch / HIP types cross this boundary.  Every function returns
 * ISX_OK (0) or a negative error code; isx_last_error() gives the message (thread-local).
 * One isx_ctx per GPU; a ctx is not thread-safe, different ctxs are independent.
 *
 * What each group of entry points replaces in the reference:
 *   isx_set_null_model      snv_utilities.py:14-38   generate_snp_model (the dict handed to
 *                                                    every profile_split call)
 *   isx_batch_create        profile_utilities.py:150-153 + 268-286: the pileup iterator's
 *                           visits, already filtered to reads in R2M, as packed records
 *                           (decoded on the host -- isx_bam_* below -- or by the caller)
 *   isx_batch_run           profile_utilities.py:218-266 process_bam_sites (get_base_counts_mm,
 *                           update_covT, snv_utilities.update_snp_table 40-145, call_snv_site,
 *                           calculate_clonality, calc_snp_class), linkage.update_linked_reads
 *                           254-283, calc_mm_SNV_linkage_network 14-44, calculate_ld 46-240
 *   isx_batch_fetch_*       the SplitObject fields (profile_utilities.py:195-211): covT / clonT
 *                           (entries), raw_snp_table (snv rows), raw_linkage_table (ld rows)
 *   isx_bam_*               pysam.AlignmentFile + samfile.pileup(...) with the htslib-1.9 rules
 *                           of profile_utilities.py:150-153, and filter_reads.get_paired_reads /
 *                           filter_scaff2pair2info (filter_reads.py:885-956, 201-260)
 */
#ifndef INSTRAIN_AMD_H
#define INSTRAIN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISX_OK 0
#define ISX_ERR_ARG (-1)        /* bad argument */
#define ISX_ERR_HIP (-2)        /* HIP runtime error / no gfx950 device */
#define ISX_ERR_CAPACITY (-3)   /* an output table outgrew its device buffer */
#define ISX_ERR_MM_RANGE (-4)   /* an observation's mm level >= n_mm_bins */
#define ISX_ERR_IO (-5)         /* BAM / file error */
#define ISX_ERR_STATE (-6)      /* call out of order (e.g. fetch before run) */

/* 4 (round 5): isx_pipe_result grew (coverage4 / cov_rows ...), isx_pipe_params.lean_output reads what was padding, one-mm-bin read
 * batches travel as 32-byte reference-delta records, bit-plane hand-over (isx_read_planes).  A caller compiled against another version
 * must not call in: compare isx_abi_version() with this constant first (instrain_amd/_lib.py does and refuses to load). */
/* 5 (round 6): isx_pipe_result grew again (lev_*: the level-sparse hand-back of mm profiling), isx_levels_expand, ISX_LAYOUT_MM_ENTRIES. */
#define ISX_ABI_VERSION 5

/* Base codes everywhere: 0=A 1=C 2=T 3=G (P2C order, profile_utilities.py:34), 4 = anything else. */

typedef struct isx_ctx isx_ctx;
typedef struct isx_batch isx_batch;
typedef struct isx_bam isx_bam;                /* host BAM front end, see the end of this header */
struct isx_bam_params_s;
struct isx_bam_info_s;

/* One packed pileup observation = one (pileup column, pileup read) visit on which the
 * reference touches its count table.  8 bytes. */
typedef struct {
    uint32_t gpos;      /* flat position in the batch */
    uint16_t mm;        /* R2M[read name]; 0 when mm profiling is skipped */
    uint8_t base;       /* 0..4 */
    uint8_t flags;      /* reserved, 0 */
} isx_obs;

typedef struct {
    int32_t min_cov;            /* -c, default 5   (profile_utilities.py:142) */
    int32_t min_snp;            /* --min_snp, default 20 at the CLI (argumentParser.py:164) */
    double min_freq;            /* -f, default 0.05 */
    int32_t rarefied_coverage;  /* --rarefied_coverage, default 50; <= 0 switches the rarefied outputs off */
    int32_t n_mm_bins;          /* mm levels 0..n_mm_bins-1 may occur; 1 = --skip_mm_profiling */
    int32_t enable_linkage;     /* 0: pileup / SNV only */
    int32_t linkage_mode;       /* 0 auto (= sparse in this version), 1 sparse pair-increment path,
                                 * 2 dense int8 MFMA path (n_mm_bins == 1 only) */
    int32_t window;             /* 0 = auto; positions per workgroup window (multiple of 64) */
    uint64_t seed;              /* counter-based RNG seed for the rarefied outputs */
    int32_t layout;             /* 0 = automatic (production).  ISX_LAYOUT_* bits force one of the resident layouts the
                                 * library otherwise picks itself; every layout gives identical results (tests, A/B timing) */
    int32_t reserved;           /* 0 */
} isx_params;

#define ISX_LAYOUT_WIDE_RECORDS 1       /* 8-byte records (isx_obs as is) instead of the 2- / 4-byte stream */
#define ISX_LAYOUT_NO_SHORT_RECORDS 2   /* n_mm_bins == 1: 4-byte instead of 2-byte records */
#define ISX_LAYOUT_NO_PACKED_COUNTERS 4 /* mm path: u32 instead of packed u16 LDS counters */
#define ISX_LAYOUT_SEG64_RECORDS 8      /* read-level batch with one mm bin: 64-byte segment records (3-bit codes, the round-3 stream)
                                         * instead of the 32-byte reference-delta records */

#define ISX_LAYOUT_MM_ENTRIES 16        /* read-level pipe with mm profiling on: hand the levels back as 32-byte entries in window slabs (the
                                         * round-2 way: isx_pipe_fetch_entries) instead of the level-sparse tables (isx_pipe_result.lev_*) */

#define ISX_LAYOUT_MM_DELTA_RECORDS 32  /* read-level batch with mm profiling on: 32-byte reference-delta records with the pair's mm level in bits
                                         * 24..30 of a segment's header (k_pileup_mm materialises every level's coverage-difference row) instead of
                                         * the 64-byte segment records.  Half the bytes over PCIe; measured (round 6) the kernel is
                                         * no faster -- the per-level prefix sums and the wider LDS rows eat what the shorter stream saves -- and
                                         * the host stager of isx_segs input costs twice the segment records': opt-in, not the default */

#define ISX_LAYOUT_NO_STRIPES 64         /* one-mm-bin reference-delta batches without a count table (pipe slots): the round-4 epilogue -- every position
                                         * gets the reference base's count and walks the first epilogue pass -- instead of the stripe path (a thread owns
                                         * eight positions, coverage straight from the difference row, only positions at min_cov go on) */

/* (position, mm)-present entry: one per mm level present at a position, ascending mm.
 * 32 bytes (two aligned 16-byte device stores).  cnt = counts of THIS level; covT[mm][pos] = sum(cnt);
 * clon = clonT[mm][pos] (float32 of the cumulative-<=mm clonality) or NaN when cumulative coverage
 * < min_cov; clon_rarefied = clonTR[mm][pos] (snv_utilities.py:233-247; Philox-seeded) or NaN when
 * cumulative coverage < rarefied_coverage. */
typedef struct {
    uint32_t gpos;
    uint16_t mm;
    uint16_t flags;
    uint32_t cnt[4];
    float clon;
    float clon_rarefied;
} isx_entry;

/* raw_snp_table row (snv_utilities.py:118-127, 274-290).  28 bytes. */
typedef struct {
    uint32_t gpos;
    uint16_t mm;
    uint8_t con_base, var_base;
    uint8_t allele_count;       /* "morphia" */
    uint8_t cls;                /* 0 AmbiguousReference 1 DivergentSite 2 SNS 3 SNV 4 con_SNV 5 pop_SNV */
    uint8_t cryptic;
    uint8_t ref_base;           /* 0..4 */
    uint32_t cnt[4];            /* cumulative over levels <= mm; position_coverage = sum */
} isx_snv;

/* raw_linkage_table row (linkage.py:230-237 + 67-71).  The *_normalized columns are the
 * reference's rarefied variants (linkage.py:200-228, unseeded np.random.choice there): here they
 * are drawn from a Philox stream keyed by (seed, position_A, position_B, mm). */
typedef struct {
    uint32_t gpos_a, gpos_b;    /* flat positions, gpos_a <= gpos_b */
    uint16_t mm;
    uint8_t allele_A, allele_a, allele_B, allele_b;
    uint16_t pad;
    uint32_t total, countAB, countAb, countaB, countab;
    uint32_t pad2;
    double r2, d_prime;
    double r2_normalized, d_prime_normalized;
} isx_ld;

/* sizes of the result tables of a batch after isx_batch_run */
typedef struct {
    int64_t n_entries;      /* 0 when n_mm_bins == 1 (dense counts/clon arrays instead) */
    int64_t n_snv;
    int64_t n_sites;        /* positions with anySNP (snv_utilities.py:129-133) */
    int64_t n_allele_obs;   /* update_linked_reads appends */
    int64_t n_increments;   /* calc_mm_SNV_linkage_network pair increments */
    int64_t n_edges;        /* graph edges (distinct site pairs, self pairs included) */
    int64_t n_ld;
} isx_sizes;

/* per-kernel device time of the last isx_batch_run, milliseconds (HIP events on the ctx stream) */
typedef struct {
    float pileup_ms;        /* k_pileup_dense / k_pileup_mm: window histogram + SNV call epilogue (+ allele pass with linkage) */
    float sites_ms;         /* site table sort / rank (bucket chain, round 6: k_link_prep) */
    float allele_ms;        /* k_ao_rank: allele observations -> site ranks (bucket chain: + their read pairs' chains, k_ao_chain) */
    float group_ms;         /* group allele observations by pair (bucket chain: both walks of the pairs' chains + the scan between them) */
    float incr_ms;          /* pair increments -> keys, sort, run-length (bucket chain: per-site aggregation + the rows' count and scan) */
    float ld_ms;            /* LD rows */
    float total_ms;
    int32_t pileup_blocks, pileup_threads, pileup_lds_bytes, pileup_window;
    /* dense int8-MFMA linkage path (linkage_mode 2) */
    float mfma_ms;          /* one k_dense_gemm pass over all tiles (the path runs it twice: count, emit) */
    int32_t dense_tiles;    /* useful 32x32 tiles of the upper triangles */
    int64_t dense_macs;     /* int8 multiply-accumulates of those tiles in one pass */
    int64_t dense_bytes;    /* bytes of the X^T blocks */
    int32_t record_bytes;   /* resident observation stream: 4 (compact records) or 8 (isx_obs as is) bytes per record */
    int32_t pad;
} isx_timings;

const char *isx_last_error(void);
int isx_abi_version(void);

int isx_ctx_create(int device_id, isx_ctx **out);
void isx_ctx_destroy(isx_ctx *ctx);
/* Keep `cus_per_xcd` (0..8) compute units of each of the 8 XCDs free of pileup kernels: the context's two pass queues are masked off them
 * and every side queue of its pipes (copy-in, finishers' linkage chains, copy-out) onto them.  A pileup kernel is a persistent grid that
 * fills every CU it may use for its whole run, so without a reserve each short launch of a finisher's chain waits for one to end; with 4
 * (32 of 256 CUs) the whole-database stream of bench.py gains ~8 %, a resident batch loses the CUs' share.  0 (the default; the
 * environment's ISX_PASS_CU_RESERVE overrides the default) = no reserve, pass queues at the highest stream priority instead.  Call it
 * before the context's first batch or pipe.  No counterpart in the reference (its workers are processes: profile/profile_utilities.py). */
int isx_ctx_reserve_cus(isx_ctx *ctx, int cus_per_xcd);

/* lut[c] = minimum count for a base to be "present" at coverage c, or < 0 when the
 * coverage is missing from the model; fallback = model[-1]. */
int isx_set_null_model(isx_ctx *ctx, const int32_t *lut, int64_t n, int32_t fallback);

/*
 * Make a batch resident on the device.
 *   n_pos         flat positions; ref[n_pos] reference base codes (0..4)
 *   split_bounds  [n_splits+1] ascending flat offsets, split_bounds[0]=0, [n_splits]=n_pos;
 *                 linkage never crosses a bound (profile_utilities.py:164,188-189)
 *   obs[n_obs]    packed observations in BAM arrival order (position-clustered)
 *   pair[n_obs]   dense read-pair id of each observation (both mates share it); may be NULL
 *                 when enable_linkage == 0
 * Host buffers may be freed after the call returns.  The observations are re-encoded while they are staged
 * for the upload: 2 bytes per record on the device when n_mm_bins == 1, 4 bytes otherwise (a position delta
 * to the base of the record's group, mm, base code; isx_timings.record_bytes tells which), so a pass reads a
 * quarter / half of what isx_obs occupies on the host.
 */
int isx_batch_create(isx_ctx *ctx, const isx_params *params, int64_t n_pos, const uint8_t *ref,
                     int32_t n_splits, const int64_t *split_bounds, int64_t n_obs,
                     const isx_obs *obs, const uint32_t *pair, isx_batch **out);
void isx_batch_destroy(isx_batch *b);

/* One pass of the hot path over the resident batch (blocking). Re-runnable. */
int isx_batch_run(isx_batch *b);

/* The same pass split for pipelining: isx_batch_launch enqueues the pileup / SNV-calling kernel on the
 * context's stream and returns at once; isx_batch_wait blocks until it is done, runs the linkage stages
 * (they need the table sizes on the host) and makes sizes / fetch available.  isx_batch_run == launch +
 * wait.  Several batches of one context may be in flight (a context keeps two queues and alternates its
 * batches between them), so the next shard's pass is queued next to the current one -- what the reference does with its worker pool
 * (profile_controller.py:243-271), without a launch gap between shards.  A batch has at most one pass in
 * flight; fetch / sizes refuse (ISX_ERR_STATE) until isx_batch_wait has returned. */
int isx_batch_launch(isx_batch *b);
int isx_batch_wait(isx_batch *b);
int isx_batch_sizes(const isx_batch *b, isx_sizes *out);
int isx_batch_timings(const isx_batch *b, isx_timings *out);

/* Results -> caller-allocated host buffers sized from isx_batch_sizes. Tables come back in
 * canonical order: entries (gpos, mm); snv (gpos, mm); ld (gpos_a, gpos_b, mm). */
int isx_batch_fetch_entries(isx_batch *b, isx_entry *out);
/* clon_rarefied (may be NULL): clonTR of the dense path, NaN where coverage < rarefied_coverage */
int isx_batch_fetch_dense(isx_batch *b, uint32_t *counts /* [n_pos][4] */, float *clon /* [n_pos] */,
                          float *clon_rarefied /* [n_pos] */);
int isx_batch_fetch_snv(isx_batch *b, isx_snv *out);
int isx_batch_fetch_ld(isx_batch *b, isx_ld *out);
/* The appends of update_linked_reads (linkage.py:254-283), isx_sizes.n_allele_obs rows in no particular order: read pair `pair`
 * showed base `base` (in the site's `bases` set) at the SNP site at flat position gpos; `order` ranks the appends of one pair at one
 * site (both mates visible at a column: the self pairs of linkage.py:26-30).  read_to_snvs[mm][read] of the reference is a pair's
 * rows of level mm sorted by (gpos, order); mm_to_position_graph its itertools.combinations (profile_utilities.py:205-211: kept
 * with --store_everything).  Needs enable_linkage; valid after isx_batch_run / isx_pipe_collect. */
typedef struct {
    uint32_t pair;
    uint32_t gpos;
    uint32_t order;
    uint16_t mm;
    uint8_t base;
    uint8_t pad;
} isx_allele_obs;
int isx_batch_fetch_allele_obs(isx_batch *b, isx_allele_obs *out);

/* ---- per-scaffold merge summaries (make_coverage_table, profile_utilities.py:425-506) ----
 * One row per (scaffold, mm level): the position-sized aggregates of the cumulative coverage over
 * levels <= mm (mm_counts_to_counts_shrunk :508-532) and of the clonalities of the highest level
 * <= mm (get_basewise_clons :534-546).  `present` = the level has coverage on this scaffold (it is
 * a key of covT).  The SNV-table columns (SNS/SNV counts, ANI) are table-sized and stay on the host. */
typedef struct {
    int64_t nonzero;                /* positions with cumulative coverage > 0 (breadth * length) */
    uint64_t sum_cov, sumsq_cov;    /* exact integer sums -> mean / std / SEM */
    double median_cov;              /* np.median(covs) over ALL positions of the scaffold */
    int64_t counted;                /* positions with a clonality (breadth_minCov * length) */
    double sum_clon, median_clon;   /* nucl_diversity = 1 - sum_clon / counted; ..._median = 1 - median_clon */
    int64_t counted_rarefied;
    double sum_clon_rarefied, median_clon_rarefied;
    int32_t mm, present;
} isx_scaffold_level;

/* out[n_scaffolds][n_mm_bins]; scaffold_bounds[n_scaffolds + 1] ascending flat offsets spanning [0, n_pos].
 * Needs a completed isx_batch_run. device_ms (may be NULL) = device time of the pass. */
int isx_batch_summarize(isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, isx_scaffold_level *out,
                        float *device_ms);

/* ---- genome-level coverage roll-up (genomeUtilities.py:297-365 genomeLevel_coverage_info on
 * generate_genome_coverage_array :932-981; iRep is not part of this) ----
 * A genome = consecutive scaffolds of the batch.  One row per (genome, mm level): the coverage, cumulative over levels
 * <= mm, of all its scaffolds laid end to end with mask_edges positions cut from both ends of every scaffold (a scaffold
 * shorter than twice that drops out): how many positions are left, the exact sums over them and the median.
 * coverage_median = int(median_cov); coverage_std = sqrt(sumsq / n - (sum / n)^2); coverage_SEM = sample std / sqrt(n). */
typedef struct {
    int64_t n;                      /* positions after masking (0: the reference then uses the single value 0) */
    uint64_t sum_cov, sumsq_cov;
    double median_cov;
    int32_t mm, pad;
} isx_genome_level;

/* out[n_genomes][n_mm_bins]; genome_first_scaffold[n_genomes + 1] ascending scaffold indices, [0] = 0, [n_genomes] = n_scaffolds */
int isx_batch_summarize_genomes(isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t n_genomes,
                                const int32_t *genome_first_scaffold, int32_t mask_edges, isx_genome_level *out, float *device_ms);

/* ---- compare: two samples on the same scaffolds (readComparer.py:35-143 compare_scaffold, one pair) ----
 * Two batches over the SAME flat space (same scaffolds laid out identically, same ctx).  One row per
 * (scaffold, mm): positions where both / either sample reach min_cov in the coverage cumulated over
 * levels <= mm (readComparer.py:145-191 calc_mm2overlap); present_x = the level is a key of that
 * sample's covT on the scaffold (the reference evaluates the union of both key sets).
 * consensus_snps / population_snps: rows of the reference's Mdb at that mm with the flag set
 * (readComparer.py:205-290 _calc_SNP_count_alternate, :437-502 _update_overlap_table); -1 when the SNP
 * half was not run (isx_compare_coverage), -2 when the reference itself fails on this scaffold (a SNP row
 * with an N reference base that the other sample lacks makes it look up the column 'N_1' -> KeyError). */
typedef struct {
    int64_t both, either;
    int32_t mm, present_a, present_b, pad;
    int64_t consensus_snps, population_snps;
} isx_compare_level;

/* One row of the reference's Mdb (readComparer.py:221-225 OUT_COLUMNS): a position covered by both
 * samples at `mm` whose highest-mm SNP rows differ in consensus (consensus_snp) and / or share no
 * detectable allele (population_snp).  has_x = sample x has a SNP row there (else its columns are NaN
 * in the reference, 0 here); bases 0..3 = A,C,T,G, 4 = N; position_coverage_x = sum of cnt_x. */
typedef struct {
    uint32_t gpos;
    uint16_t mm;
    uint8_t consensus_snp, population_snp;
    uint8_t has_a, has_b;
    uint8_t con_a, ref_a, var_a, con_b, ref_b, var_b;
    uint32_t cnt_a[4], cnt_b[4];
} isx_compare_snp;

/* out[n_scaffolds][max(n_mm_bins_a, n_mm_bins_b)]; coverage half only */
int isx_compare_coverage(isx_batch *a, isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t min_cov,
                         isx_compare_level *out, float *device_ms);

/* Coverage half + SNP-table half.  min_freq and the ctx's null model play the roles of compare's
 * --min_freq / --fdr (is_present, readComparer.py:306-315).  *n_snp_rows = rows isx_compare_fetch_snps
 * will deliver (sorted by mm, gpos); they stay with batch `a` until its next compare call. */
int isx_compare_scaffolds(isx_batch *a, isx_batch *b, int32_t n_scaffolds, const int64_t *scaffold_bounds, int32_t min_cov,
                          double min_freq, isx_compare_level *out, int64_t *n_snp_rows, float *device_ms);
int isx_compare_fetch_snps(isx_batch *a, isx_compare_snp *out);

/* ---- streaming hand-over: profile a stream of batches, each exactly once ----
 * What production does (the reference's analogue is the command / result queue pair around its worker pool,
 * profile_controller.py:157-193, 243-271): batches arrive from the host, are profiled once and their tables go
 * back.  A pipe owns `depth` resident slots sized for the largest batch, pinned staging for both directions, a
 * persistent pool of host threads and three queues, so that at any time
 *     batch k+1 is encoded into pinned staging by the host threads (8-byte isx_obs -> 2- / 4-byte records),
 *     batch k   is copied in (hipMemcpyAsync) and profiled,
 *     batch k-1's tables are copied out (hipMemcpyAsync) / read by the caller.
 * isx_pipe_submit returns once the batch is encoded and everything else is enqueued (the caller's buffers may be
 * reused); a thread of the pipe finishes every batch as soon as its copy-out has landed (table sizes, growth + a
 * repeated pass when a table was too small, the linkage stages, row sorting) while the caller encodes the next one;
 * isx_pipe_collect blocks until that is done for the batch and hands its tables over; isx_pipe_release gives the
 * slot back.  Tickets count from 0 in submission order; at most `depth` batches may be submitted and not yet released
 * (isx_pipe_submit then returns ISX_ERR_STATE).  Not thread-safe: one caller thread per pipe. */
typedef struct isx_pipe isx_pipe;

typedef struct {
    int64_t max_pos;            /* largest n_pos of a batch */
    int64_t max_obs;            /* largest n_obs of a batch */
    int32_t max_splits;
    int32_t depth;              /* resident slots: 1 = no overlap between batches, 4 is a good default for a stream */
    int32_t host_threads;       /* encoder threads incl. the caller; 0 = automatic (the cpus this process may use, at most 32) */
    int32_t pin_threads;        /* 1: spread the threads over the L3 domains of the GPU's NUMA node (sched_setaffinity) */
    double jump_slack;          /* device records set aside for streams that jump between far-apart positions
                                 * (next contig / genome of a database), as a fraction of max_obs; 0 = 0.25 */
    int32_t want_counts;        /* n_mm_bins == 1: 1 = also hand back the per-base count table and the dense clonTR array
                                 * (16 + 4 more bytes per position over PCIe: the reference keeps them only with
                                 * --store_everything, profile_utilities.py:205-211); 0 = the shrunk tables below */
    int32_t ring_kib;           /* pinned staging of a slot's records: 0 = automatic (the whole stream when it is at most
                                 * 512 MiB -- 128 MiB with depth 1 --, otherwise a ring of 2 x 128 MiB -- 2 x 32 MiB --
                                 * through which it leaves in waves while it is being encoded), > 0 = a ring of that
                                 * many KiB (at least 32), < 0 = never a ring */
    int32_t stage_async;        /* read-level pipe: 1 = isx_pipe_submit_reads only queues the batch and returns its ticket; a
                                 * thread of the pipe encodes it into staging and enqueues it (in ticket order) while the caller
                                 * goes on.  `ref` and the isx_segs arrays must then stay valid and unchanged until
                                 * isx_pipe_collect / isx_pipe_release of that ticket; what the encoder finds wrong with
                                 * them (mm range, capacity ...) is reported by isx_pipe_collect */
    int32_t lean_output;        /* n_mm_bins == 1 without want_counts: 1 = a slot's kernel writes ONLY what travels home -- 1- (or 2-) byte
                                 * coverage and the sparse lists -- not the dense clonality array (4 B/pos) nor the 16-bit coverage beside the
                                 * 8-bit one: 1-2 instead of 6-7 bytes a position written.  A batch whose clonality list does not fit is
                                 * passed again with the dense array.  isx_batch_summarize* (which read those arrays on the device) are then
                                 * refused on the slot's batch.  (This field sits in what was padding: the struct's size is unchanged.) */
    int64_t max_segs;           /* > 0: a READ-LEVEL pipe (isx_pipe_submit_reads / isx_pipe_submit_bam hand over read segments,
                                 * see isx_segs below; isx_pipe_submit is refused): the largest n_seg of a batch.  max_obs then
                                 * only bounds the linkage tables (0 = 150 x max_segs) */
} isx_pipe_params;

/* one entry of the sparse clonTR table (positions whose coverage reaches rarefied_coverage) */
typedef struct {
    uint32_t gpos;
    float clon_rarefied;
} isx_rare;

/* exact coverage of a position whose 16- / 8-bit hand-back entry saturated */
typedef struct {
    uint32_t gpos;
    uint32_t coverage;
} isx_sat;

typedef struct {
    int64_t ticket;
    int64_t n_pos, n_obs;
    isx_sizes sizes;
    /* n_mm_bins == 1: tables in pinned host memory, valid until isx_pipe_release.  What shrink_basewise keeps
     * (profile_utilities.py:337-350) is coverage, clonality and the few rarefied clonalities, so that is what
     * crosses PCIe: 6 bytes per position instead of 24 (round 3: 1-2 bytes + short lists, see coverage8 / clon_sparse below).
     * The tables leave the device through a copy kernel into the pinned block, not through the DMA engine. */
    const uint16_t *coverage16; /* [n_pos] covT = min(sum of the four counts, 65535); n_saturated positions hold 65535
                                 * (their exact counts: isx_batch_fetch_dense on `batch`, or want_counts) */
    const float *clon;          /* [n_pos] clonT, NaN below min_cov */
    const isx_rare *rare;       /* [n_rare] clonTR as a sparse table, ascending gpos; NULL when more than n_pos / 8
                                 * positions have one (a deep sample): then clon_rarefied holds the dense array */
    int64_t n_rare, n_saturated;
    const uint32_t *counts;     /* [n_pos][4], NULL unless want_counts */
    const float *clon_rarefied; /* [n_pos] dense clonTR (NaN = none); NULL unless want_counts or rare == NULL */
    const isx_snv *snv;         /* [sizes.n_snv], canonical (gpos, mm) order; both modes */
    /* the slot itself: isx_batch_fetch_entries / isx_batch_fetch_ld / isx_batch_summarize / isx_compare_* may be
     * called on it until isx_pipe_release (n_mm_bins > 1: the entry table is fetched this way) */
    isx_batch *batch;
    /* where the time of this batch went, milliseconds */
    float encode_ms;            /* host threads: isx_obs / isx_segs -> records in pinned staging */
    float h2d_ms, kernel_ms, d2h_ms;    /* device-side durations (HIP events) */
    float collect_wait_ms;      /* host time isx_pipe_collect spent waiting */
    int32_t record_bytes;       /* 2 or 4 (observation records), 64 (read segments) */
    int32_t encode_passes;      /* 1; 2 when the stream jumped more often than the slot's slack allowed */
    int64_t h2d_bytes, d2h_bytes;
    const isx_ld *ld;           /* [sizes.n_ld] the LD rows when linkage is enabled (else NULL), reference order;
                                 * valid until isx_pipe_release */
    /* n_mm_bins == 1 without want_counts: the position-sized tables shrink further.
     * coverage16 == NULL -> coverage8 (a shallow batch: mean depth below 16).
     * clon == NULL -> clon_sparse: clonT[p] is exactly 1.0 at every position whose coverage reaches min_cov, EXCEPT the listed
     * positions (more than one base observed there); below min_cov there is none (NaN).  clon (the dense array) comes instead
     * when more than half of the positions would be listed. */
    const uint8_t *coverage8;   /* [n_pos] min(covT, 255); exact values of the positions at 255 or beyond: `saturated` */
    const isx_rare *clon_sparse;/* [n_clon] (gpos, clonT != 1.0), ascending gpos */
    int64_t n_clon;
    const isx_sat *saturated;   /* [n_saturated] (gpos, exact coverage) of the positions whose coverage16 / coverage8 entry
                                 * saturated, unordered; NULL when the list outgrew the pipe's room (isx_batch_fetch_dense then) */
    /* lean slots (isx_pipe_params.lean_output), a batch of mean depth below 6 in reference-delta records: coverage16 == coverage8 == NULL ->
     * coverage4: min(covT, 15) of two positions a byte (low nibble = the even position), and every window of cov_window positions
     * (window w = positions [w * cov_window, (w + 1) * cov_window)) that holds a position beyond 15 ALSO as a whole row of 16-bit
     * values: row k of cov_rows is window cov_row_window[k] (rows in no particular order; a position beyond 65534: `saturated`).
     * Half a byte per position over PCIe for a metagenome at depth 3 instead of one. */
    const uint8_t *coverage4;   /* [(n_pos + 1) / 2] */
    const uint16_t *cov_rows;   /* [n_cov_rows][cov_window] (the last window's row: entries beyond n_pos are 0) */
    const uint32_t *cov_row_window; /* [n_cov_rows] */
    int64_t n_cov_rows;
    int32_t cov_window, pad_cov;
    /* n_mm_bins in 2..32 (mm profiling on, the reference's default: argumentParser.py:131) in a read-level pipe without want_counts: what
     * shrink_basewise keeps of the (position, mm) levels (profile_utilities.py:337-350; covT = the level's own coverage, :288-295; clonT /
     * clonTR of the counts up to the level, snv_utilities.py:85-104) comes home LEVEL-SPARSE instead of as 32-byte entries --
     *   lev_mask     [n_pos] elements of lev_mask_bytes (1, 2, 4): bit m = level m is present at the position (also a level made present
     *                by a base that is not A/C/T/G: coverage 0, profile_utilities.py:279-285)
     *   lev_cov      [n_lev] elements of lev_cov_bytes (1, or 2 for a deep batch): the level's coverage, saturating at 255 / 65535 -- the
     *                levels of window w (positions [w * lev_window, (w + 1) * lev_window)) are the elements from lev_win_off[w] on, in
     *                (position, mm) order; the windows' ranges are disjoint but in no particular order
     *   lev_sat      [n_lev_sat] (index into lev_cov, exact coverage) of the saturated elements
     *   lev_clon     [n_lev_clon] (index, clonT): every level whose cumulative coverage reaches min_cov has clonT exactly 1.0 EXCEPT the
     *                listed ones (more than one base observed up to that level); below min_cov there is none (NaN)
     *   lev_rare     [n_lev_rare] (index, clonTR) of the levels that have one (cumulative coverage >= rarefied_coverage)
     * 1-3 bytes per level over PCIe instead of 32 (16 with isx_pipe_fetch_entries_shrunk).  isx_levels_expand makes the four columns of
     * isx_pipe_fetch_entries_shrunk from them on the host.  A lean slot (isx_pipe_params.lean_output) writes nothing else; a plain
     * slot also keeps the 32-byte entries on the device (isx_batch_summarize, isx_pipe_fetch_entries).  NULL / 0 otherwise. */
    const void *lev_mask;
    const void *lev_cov;
    const uint32_t *lev_win_off;
    const isx_rare *lev_clon, *lev_rare;    /* .gpos = the level's index into lev_cov */
    const isx_sat *lev_sat;                 /* .gpos = the level's index into lev_cov */
    int64_t n_lev, n_lev_clon, n_lev_rare, n_lev_sat;
    int32_t lev_mask_bytes, lev_cov_bytes, lev_window, n_lev_windows, lev_min_cov, pad_lev;
    /* a 64-bit checksum over the bytes of the SNV rows and the LD rows of this batch as they lie in `snv` / `ld`, made by the pipe's finisher
     * thread when the rows have landed (order-sensitive; the same rows give the same value): a caller that streams many batches can compare
     * every batch's tables with a verified pass without reading them (bench.py check_timed) */
    uint64_t rows_checksum;
} isx_pipe_result;

/* The level-sparse tables of a collected batch (isx_pipe_result.lev_*) as the four columns of isx_pipe_fetch_entries_shrunk: n_lev values
 * each in (gpos, mm) order -- gpos; mm_cov = mm << 24 | the level's coverage; clon; clon_rarefied (NaN = none; may be NULL when n_lev_rare == 0).  Host work only (no GPU
 * call), on host_threads threads.  ISX_ERR_CAPACITY when a level's coverage reaches 2^24; ISX_ERR_STATE when the result holds no such
 * tables or they are inconsistent. */
int isx_levels_expand(const isx_pipe_result *r, int32_t host_threads, uint32_t *gpos, uint32_t *mm_cov, float *clon, float *clon_rarefied);

int isx_pipe_create(isx_ctx *ctx, const isx_params *params, const isx_pipe_params *pp, isx_pipe **out);
void isx_pipe_destroy(isx_pipe *p);
/* same arguments as isx_batch_create */
int isx_pipe_submit(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                    int64_t n_obs, const isx_obs *obs, const uint32_t *pair, int64_t *ticket);
/* The same, fed by the BAM front end (isx_bam_* below; the file must be scanned and filtered): references `refs`
 * (ascending ids) are expanded straight into the slot's pinned staging -- neither the 8-byte records nor (read-level
 * pipe) the read segments of the batch ever exist as a whole.  split_bounds == NULL: the front end's own iterate_splits geometry.  ref[n_pos] = base codes of
 * the batch's references laid end to end.  info (may be NULL) receives the batch's counts.  `bam` must stay open until the
 * batch has been collected (its reads go back to the handle then and are freed with it). */
int isx_pipe_submit_bam(isx_pipe *p, isx_bam *bam, const struct isx_bam_params_s *bp, const int32_t *refs, int32_t n_refs,
                        const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds, struct isx_bam_info_s *info,
                        int64_t *ticket);
int isx_pipe_collect(isx_pipe *p, int64_t ticket, isx_pipe_result *out);
int isx_pipe_release(isx_pipe *p, int64_t ticket);
/* n_mm_bins > 1: the entry table of a collected batch ([sizes.n_entries], (gpos, mm) order) -- what
 * isx_batch_fetch_entries(result.batch, out) returns, moved through the slot's pinned staging by the pipe's host threads */
int isx_pipe_fetch_entries(isx_pipe *p, int64_t ticket, isx_entry *out);
/* The same table shrunk to what shrink_basewise keeps of a (position, mm) level (profile_utilities.py:337-350): covT is the
 * level's coverage, not its four counts.  Four columns of sizes.n_entries values in (gpos, mm) order: gpos; mm_cov = mm << 24 |
 * sum of the level's four counts; clon; clon_rarefied (NaN = none) -- 16 bytes an entry over PCIe instead of 32.
 * ISX_ERR_CAPACITY when a level's coverage reaches 2^24 or an mm level 256 (fetch the full entries then). */
int isx_pipe_fetch_entries_shrunk(isx_pipe *p, int64_t ticket, uint32_t *gpos, uint32_t *mm_cov, float *clon, float *clon_rarefied);


/* ---- read-level hand-over: the per-base expansion of the pileup happens on the device ----
 * Instead of one 8-byte record per (pileup column, pileup read) visit the host ships READ SEGMENTS: a segment = up to
 * ISX_SEG_BASES consecutive reference positions covered by one M / = / X run of a read's CIGAR (a longer run is cut into
 * several segments; an insertion / deletion / skip ends a segment), with one 3-bit code per position:
 *     0..3 = A, C, T, G observed with quality >= min_base_quality after htslib's overlap resolution
 *     4    = nothing to count here (low quality, masked by the mate's overlap, position outside the scaffold, padding)
 *     5    = a base that is not A/C/T/G but passes the quality filter (it makes its mm level "present" at the position,
 *            profile_utilities.py:279-285, without being counted); 6, 7 = reserved, treated as 4
 * packed ten per 32-bit word (base j of the segment: word j / 10, bits 3 (j % 10) .. + 2; the top two bits are 0; unused
 * slots hold code 4).  ~0.43 bytes per base instead of 8 (isx_obs) on the host and 2 / 4 on the device, and the host never
 * walks single bases into records.  Replaces exactly what isx_obs replaces: the visits of
 * samfile.pileup(...) + get_base_counts_mm (profile_utilities.py:150-153, 268-286).
 * Segments arrive in BAM order (ascending read start; a read's segments follow each other), i.e. position-clustered. */
#define ISX_SEG_BASES 150
#define ISX_SEG_WORDS 15
#define ISX_SEG_SKIP_WORD 0x24924924u   /* ten codes 4 */
/* The device stream of a read-level batch with ONE mm bin is made of reference-delta records (since round 4): what differs from
 * the reference travels, not the bases.  32 bytes per record, 32 records per group (one wave-wide 16-byte load) sharing a
 * 32-bit position base.  A record is one of two kinds, told apart by bit 31 of its word 0 (ISX_DREC_DUAL):
 *  - a DUAL record: two segments that have no skipped column, 16 bytes each (the common case: a read that aligns without
 *    deletion / low-quality base / N is one such segment) --
 *      word 0      delta:16 | len:8 | 0x80      start = gbase[record / 32] + delta; len 0 = this half is empty
 *      word 1      dense read-pair id (0 without linkage)
 *      words 2, 3  exceptions 0-2, 3-5: three (offset:8 | base:2) fields in bits 0..29 each; an exception = an observed base
 *                  that differs from the reference code at its position (every observed base where the reference is not
 *                  A/C/T/G); empty field = 0x3FF
 *    (the second half, words 4-7, alike; a half filled later never precedes, in the segments' order, the record's first);
 *  - a FULL record: one segment with its plane of skipped columns --
 *      word 0      delta:16 | len:8 | 0         len 0 = padding record
 *      words 1-2   skip bits of columns 0..63   bit set = no observation at that column (below min_qual, deletion, ref-skip,
 *      words 4-6   skip bits of columns 64..159                                        a base that is not A/C/T/G); bits >= len are 0
 *      word 3      exceptions 0-2
 *      word 7      dense read-pair id
 * A segment with more exceptions than its kind holds (ISX_DREC_EXC / ISX_DREC_EXC_FULL) travels as several pieces.  16 bytes per
 * 150-base read that matches the reference's alignment columns -- 0.11 bytes per base instead of 0.43, pair id included; the kernel
 * adds +1 / -1 at a segment's ends to a coverage-difference row, one LDS atomic per skipped column and one per exception, and
 * rebuilds the reference base's count from the prefix sum. */
#define ISX_DREC_WORDS 8
#define ISX_DREC_EXC 6          /* exceptions of a dual record's half */
#define ISX_DREC_EXC_FULL 3     /* ... of a full record */
#define ISX_DREC_DUAL 0x80000000u
#define ISX_DREC_NO_EXC 0x3FFFFFFFu

typedef struct {
    int64_t n_seg;
    const uint32_t *gpos;       /* [n_seg] flat position of the segment's first base */
    const uint8_t *len;         /* [n_seg] 1 .. ISX_SEG_BASES positions; gpos + len <= n_pos */
    const uint8_t *mm;          /* [n_seg] R2M[read name] (< n_mm_bins); NULL = 0 (--skip_mm_profiling) */
    const uint32_t *pair;       /* [n_seg] dense read-pair id (both mates share it); NULL unless linkage is enabled */
    const uint32_t *bases;      /* [n_seg][ISX_SEG_WORDS] packed codes */
} isx_segs;

/* isx_batch_create / isx_pipe_submit with read segments instead of observations; results are identical to handing over
 * the observations the segments stand for */
int isx_batch_create_reads(isx_ctx *ctx, const isx_params *params, int64_t n_pos, const uint8_t *ref, int32_t n_splits,
                           const int64_t *split_bounds, const isx_segs *segs, isx_batch **out);
int isx_pipe_submit_reads(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                          const isx_segs *segs, int64_t *ticket);

/* Staged batches -- the zero-copy hand-over.  isx_pipe_stage_reads does ALL the host work of isx_pipe_submit_reads (comparison with the
 * reference / record encoding on the pipe's threads, reference packing, window directory) once, into a pinned image owned by the
 * returned isx_wire; isx_pipe_submit_wire then only enqueues the DMA copies from that image + the pass + the copy-out: no
 * staging threads, no host pass over the reads.  A caller that decodes its reads ahead of the device (the reference's workers decode
 * their BAM region before they profile it, profile_utilities.py:150-153) stages every batch as it is decoded; a rank of a
 * multi-GPU node needs no encoder threads in its submit loop.  A wire belongs to the pipe it was staged for, may be submitted any
 * number of times, and is freed by the caller (after the last batch submitted from it has been collected).
 * isx_wire_bytes = what one submit copies over PCIe. */
typedef struct isx_wire isx_wire;
int isx_pipe_stage_reads(isx_pipe *p, int64_t n_pos, const uint8_t *ref, int32_t n_splits, const int64_t *split_bounds,
                         const isx_segs *segs, isx_wire **out);
int isx_pipe_submit_wire(isx_pipe *p, const isx_wire *wire, int64_t *ticket);
int64_t isx_wire_bytes(const isx_wire *wire);
void isx_wire_free(isx_wire *wire);
/* Keep a staged batch's reference planes in device memory: later submits of the wire bring in only what belongs to the sample (records,
 * directory, bounds).  The reference of a database is the same for every sample profiled against it -- the reference program holds its
 * fasta in memory for the whole run (profile_controller.py:415-433) -- and is a quarter of a shallow metagenome batch's copy-in.  The
 * device copy lives until isx_wire_free. */
int isx_wire_keep_reference(isx_pipe *p, isx_wire *wire);

/* ---- read-level hand-over as BIT PLANES (round 5): what a decoder holds anyway, and what the host stager wants ----
 * The same read segments as isx_segs, one 64-byte line per segment instead of fifteen words of 3-bit codes:
 *     words 0-4 (uint64)  the 2-bit base code (A C T G = 0 1 2 3, P2C order) of column j at bits 2 (j % 32) of word j / 32
 *                         (a BAM's 4-bit seq nibbles map straight onto it); columns that are not observed may hold anything
 *     words 5-7           bit j % 64 of word 5 + j / 64 set = column j is NOT observed: base quality below min_base_quality after
 *                         htslib's overlap resolution, a base that is not A/C/T/G, padding (codes 4 / 5 of isx_segs.  With one mm
 *                         bin a non-ACGT base has no effect on any table: profile_utilities.py:279-285 only makes its mm level
 *                         "present", snv_utilities.py:85-104 then sees zero counts); bits from column len on are ignored
 * and the reference as it travels to the device: a 2-bit plane, four positions a byte (position p at bits 2 (p % 4) of byte p / 4,
 * anything that is not A/C/T/G as 0) + a bit plane marking the positions that are not A/C/T/G (NULL = there are none).
 * Why: the stager of isx_pipe_submit_reads unpacks 150 codes to bytes and compares them with reference bytes; from planes the
 * observed-and-different columns are one XOR of five words against the funnel-shifted reference plane (32 columns a step), the skip
 * plane is copied into the record as it is, and the reference plane is copied, not packed.  Pipes whose batches travel as
 * reference-delta records: one mm bin (--database_mode / --skip_mm_profiling), or -- round 6 -- n_mm_bins > 1 with ISX_LAYOUT_MM_DELTA_RECORDS
 * (isx_read_planes.mm; otherwise ISX_ERR_STATE -- hand isx_segs over).  Replaces the same reference code as
 * isx_segs: the visits of samfile.pileup(...) + get_base_counts_mm (profile_utilities.py:150-153, 268-286).  Tables are
 * byte-identical to those of the isx_segs the planes stand for (tests/test_gpu_planes.py). */
#define ISX_PLANE_WORDS 8
typedef struct {
    int64_t n_seg;
    const uint32_t *gpos;       /* [n_seg] flat position of the segment's first column, BAM order */
    const uint8_t *len;         /* [n_seg] 1 .. ISX_SEG_BASES columns; gpos + len <= n_pos */
    const uint32_t *pair;       /* [n_seg] dense read-pair id; NULL unless linkage is enabled */
    const uint64_t *planes;     /* [n_seg][ISX_PLANE_WORDS] */
    /* round 6 (ABI 5) -- mm profiling on (a pipe with n_mm_bins > 1 and ISX_LAYOUT_MM_DELTA_RECORDS): */
    const uint8_t *mm;          /* [n_seg] R2M[read name] (< n_mm_bins, <= 127); NULL = 0.  A base that is not A/C/T/G but passes the quality
                                 * filter makes its pair's level present at the column (profile_utilities.py:279-285): its column is marked NOT
                                 * observed like any other, its 2-bit code is 1 (0 at every other column that is not observed), and bit 63 of the
                                 * line's word 7 is set when the line holds such a column (the stager's common path tests that bit alone).  With
                                 * one mm bin all of this is ignored, as before. */
} isx_read_planes;

typedef struct {
    const uint8_t *plane2;      /* [(n_pos + 3) / 4] */
    const uint8_t *nplane;      /* [(n_pos + 7) / 8] or NULL */
    uint64_t key;               /* 0, or the caller's name for THIS reference content (a batch of a database: the same scaffolds in the same
                                 * order): the pipe keeps the planes of a key on the device after their first trip (isx_pipe_submit_planes)
                                 * and later batches that carry the key -- of any sample -- bring in only their reads.  The reference program
                                 * holds its fasta in memory for the whole run the same way (profile_controller.py:415-433).  The planes
                                 * must still be passed (the stager compares against them on the host) and must be the key's content */
} isx_ref_planes;

/* host helpers (no GPU needed).  isx_pack_ref_planes: reference codes (0..3 = A C T G, anything else = not a base) -> the two planes
 * (plane2 [(n_pos + 3) / 4], nplane [(n_pos + 7) / 8], both always written); *has_n = whether nplane marks any position.
 * isx_planes_from_segs: isx_segs -> planes[n_seg][8] (gpos / len / pair are shared with the input as they are). */
int isx_pack_ref_planes(const uint8_t *ref, int64_t n_pos, int32_t host_threads, uint8_t *plane2, uint8_t *nplane, int32_t *has_n);
int isx_planes_from_segs(const isx_segs *segs, int32_t host_threads, uint64_t *planes);
/* isx_pack_reads with planes as output (same arguments otherwise; seg_planes [cap_seg][8]) */
int isx_pack_read_planes(int64_t n_reads, const int64_t *ref_start, const int64_t *clip_lo, const int64_t *clip_hi,
                         const uint32_t *cigar, const int64_t *cigar_off, const char *seq, const uint8_t *qual, const int64_t *seq_off,
                         const uint32_t *pair, int32_t min_base_quality, int64_t cap_seg, uint32_t *seg_gpos,
                         uint8_t *seg_len, uint32_t *seg_pair, uint64_t *seg_planes, int64_t *n_seg);
/* isx_pipe_submit_reads / isx_pipe_stage_reads from planes.  The stager's work per batch is the XOR pass + a copy of the reference
 * planes into pinned staging; with isx_pipe_params.stage_async the pipe's own stager thread does it while the previous batch's DMA runs. */
int isx_pipe_submit_planes(isx_pipe *p, int64_t n_pos, const isx_ref_planes *ref, int32_t n_splits, const int64_t *split_bounds,
                           const isx_read_planes *reads, int64_t *ticket);
int isx_pipe_stage_planes(isx_pipe *p, int64_t n_pos, const isx_ref_planes *ref, int32_t n_splits, const int64_t *split_bounds,
                          const isx_read_planes *reads, isx_wire **out);
/* device memory a pipe may spend on resident references (isx_ref_planes.key), in MiB; 0 = the default (4096), < 0 = keep nothing.  Keys
 * beyond the budget travel every time.  Entries live until isx_pipe_destroy. */
int isx_pipe_set_reference_budget(isx_pipe *p, int64_t mib);
/* A caller's long-lived host arrays registered for the copy engine (round 6): what profile_controller.py keeps in memory for the whole run -- the
 * fasta, here its 2-bit / non-ACGT planes -- can be pinned ONCE (hipHostRegister; like loading the fasta, outside any batch) and is then copied to
 * the device straight from where it lies: isx_pipe_submit_planes skips the copy of the reference planes into its pinned staging (a tenth of the
 * stager's time on a whole-database pass) whenever ref->plane2 (and ref->nplane, if given) lie inside a registered range.  The planes still travel
 * with every batch (unlike isx_ref_planes.key, which keeps them on the device).  The memory must stay valid and registered until every batch that
 * was submitted with it has been collected; isx_host_unregister before freeing it.  ISX_ERR_HIP when the range cannot be pinned (RLIMIT_MEMLOCK),
 * ISX_ERR_ARG for a range that shares a page with a registered one (pinning works on pages: give every registered array pages of its own) / was
 * never registered.  Thread-safe. */
int isx_host_register(const void *ptr, int64_t bytes);
int isx_host_unregister(const void *ptr);
/* the stager on its own (no GPU needed), like isx_encode_delta: planes + reference planes -> 32-byte reference-delta records */
int isx_encode_planes(const isx_read_planes *reads, const isx_ref_planes *ref, int64_t n_pos, int32_t host_threads, int32_t slack_groups,
                      int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, int64_t *n_rec, int64_t *need_slack);
/* the same with mm profiling on: reads->mm rides in the records' headers, marked non-ACGT columns become exceptions at skipped columns --
 * the records isx_encode_delta makes of the segments (and their mm) the planes stand for */
int isx_encode_planes_mm(const isx_read_planes *reads, const isx_ref_planes *ref, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int32_t slack_groups,
                         int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, int64_t *n_rec, int64_t *need_slack);

/* Host helper for a caller that decodes the BAM itself (e.g. a pysam loop over samfile.fetch()): reads -> segments.
 * Per read r: flat position of its reference start ref_start[r] (may be negative relative to the scaffold when the
 * caller lays scaffolds end to end: [clip_lo[r], clip_hi[r]) is the scaffold's range in flat space -- columns outside
 * are truncated like the reference's pileup does), its CIGAR cigar[cigar_off[r] .. cigar_off[r + 1]) in BAM encoding
 * (len << 4 | op, op = MIDNSHP=X), its bases seq[seq_off[r] ..] as ASCII and qualities qual[seq_off[r] ..] (raw phred, after
 * any overlap resolution), mm[r] (NULL = 0) and pair[r] (NULL = none).  Output arrays hold cap_seg segments (*n_seg is
 * set; ISX_ERR_CAPACITY when they are too small -- isx_count_read_segs gives the exact number). */
int isx_count_read_segs(int64_t n_reads, const uint32_t *cigar, const int64_t *cigar_off, const int64_t *ref_start,
                        const int64_t *clip_lo, const int64_t *clip_hi, int64_t *n_seg);
int isx_pack_reads(int64_t n_reads, const int64_t *ref_start, const int64_t *clip_lo, const int64_t *clip_hi,
                   const uint32_t *cigar, const int64_t *cigar_off, const char *seq, const uint8_t *qual, const int64_t *seq_off,
                   const uint8_t *mm, const uint32_t *pair, int32_t min_base_quality, int64_t cap_seg, uint32_t *seg_gpos,
                   uint8_t *seg_len, uint8_t *seg_mm, uint32_t *seg_pair, uint32_t *seg_bases, int64_t *n_seg);

/* The read-level pipe's host-side staging on its own (no GPU needed): segments -> device record stream.
 *   device record = 16 words: delta:16 | len:8 | mm:8, then the 15 payload words; records come in groups of 16 (one
 *   wave-wide 16-byte load) with one 32-bit position base per group, start = gbase[record / 16] + delta; a group is
 *   closed early and padded with empty records (len 0, payload ISX_SEG_SKIP_WORD) where the stream jumps >= 65536
 *   positions; *n_rec is a multiple of 16.  rec holds cap_rec records, gbase cap_rec / 16 bases, pair_out (NULL with
 *   segs->pair == NULL) cap_rec ids. */
int isx_encode_segs(const isx_segs *segs, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int64_t cap_rec, uint32_t *rec,
                    uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec);
/* records (a multiple of 16) the segment starts gpos[n_seg] need in the device stream: the encoder's own counting pass -- a group
 * closes after 16 segments or where its starts would span more than 65 535 positions, i.e. every few segments on a sparse
 * (low-coverage) stream.  What cap_rec of isx_encode_segs must be at least; < 0 on a bad argument. */
int64_t isx_seg_records_needed(const uint32_t *gpos, int64_t n_seg, int32_t host_threads);
/* the same through the pipe's staging ring (what a read-level pipe does when a batch's records exceed 96 / 256 MiB): waves of at
 * most ring_records / 2 records (ring_records: a multiple of 32, 0 = no ring) are written into a private ring and every finished
 * wave is copied to its place in rec -- the copy standing in for the DMA engine */
int isx_encode_segs_ring(const isx_segs *segs, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int64_t cap_rec, int64_t ring_records,
                         uint32_t *rec, uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec);

/* The same staging for a one-mm-bin pipe (no GPU needed): segments + the reference codes ref[n_pos] -> 32-byte reference-delta
 * records (ISX_DREC_* above) in groups of 32.  The segments are cut into tasks of 4096; every task's region holds the groups its
 * segments need if none of them has a skipped column (64 to a group) + slack_groups spare ones (>= 1) for full records and the pieces
 * of segments with many exceptions, unused groups are padding.  ISX_ERR_CAPACITY with *need_slack > slack_groups: encode again with
 * that many (a pipe sizes every task's region exactly on its second attempt); otherwise cap_rec (isx_delta_records_needed) is too
 * small.  pair_out is not written (the read-pair ids, segs->pair, travel inside the records) and may be NULL.  ring_records as in
 * isx_encode_segs_ring (a multiple of 64). */
int isx_encode_delta(const isx_segs *segs, const uint8_t *ref, int64_t n_pos, int32_t n_mm_bins, int32_t host_threads, int32_t slack_groups,
                     int64_t cap_rec, int64_t ring_records, uint32_t *rec, uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec, int64_t *need_slack);
int64_t isx_delta_records_needed(const uint32_t *gpos, int64_t n_seg, int32_t host_threads, int32_t slack_groups);

/* The pipe's host-side encoder on its own (no GPU needed): obs[n_obs] -> resident record stream.
 *   record_bytes 2: delta:13 | base:3, groups of 512 records; 4: delta:16 | mm:8 | base:3 (<< 24), groups of 256;
 *   position = gbase[record / group] + delta; padding records 0xFFFF / 0x0700FFFF; a group never spans
 *   >= 8191 / 65535 positions (the stream is cut and padded where it jumps), *n_rec is a multiple of 2048.
 * rec / gbase / pair_out (NULL with pair == NULL) hold cap_rec records / cap_rec / group bases / cap_rec ids. */
int isx_encode_obs(const isx_obs *obs, const uint32_t *pair, int64_t n_obs, int64_t n_pos, int32_t record_bytes,
                   int32_t host_threads, double slack, int64_t cap_rec, void *rec, uint32_t *gbase, uint32_t *pair_out,
                   int64_t *n_rec, int32_t *passes);
/* The same through the pipe's staging ring (what a pipe does when a batch's records exceed its ring): the encoder
 * writes waves of at most ring_records / 2 records (ring_records: a multiple of 4096; pair must be NULL) into a private
 * ring and every finished wave is copied to its place in rec -- the copy standing in for the DMA engine. */
int isx_encode_obs_ring(const isx_obs *obs, const uint32_t *pair, int64_t n_obs, int64_t n_pos, int32_t record_bytes,
                        int32_t host_threads, double slack, int64_t cap_rec, int64_t ring_records, void *rec,
                        uint32_t *gbase, uint32_t *pair_out, int64_t *n_rec, int32_t *passes);

/* ---- BGZF blocks inflated on the device ----
 * The reference reads its BAM through pysam / htslib (filter_reads.py:885-956, profile_utilities.py:150-153): bgzf.c hands every
 * block of at most 64 KiB to zlib's inflate.  The blocks are independent raw deflate streams (RFC 1951), so the device decodes one
 * per lane, all blocks of a file side by side (csrc/isx_inflate.hip).
 *   isx_bgzf_index            walks the block headers of a BGZF image (nothing is inflated): where each block's deflate stream lies,
 *                             how long it and its inflated bytes are (ISIZE), where those go in the inflated stream
 *   isx_bgzf_inflate_device   copies the compressed span to the device, inflates, copies the inflated bytes to `out` (host memory);
 *                             *kernel_ms = the kernel alone.  ISX_ERR_IO names the first block that is not a valid deflate stream of
 *                             ISIZE bytes (the CRC32 of a block is not checked)
 *   isx_bgzf_inflate_host     the same decoder on the calling thread (no GPU: what the CPU tests pin against zlib) */
typedef struct {
    int64_t in_off;             /* offset of the block's deflate stream in the image */
    int32_t in_len;             /* its length in bytes */
    int32_t out_len;            /* ISIZE: inflated bytes (0 .. 65536) */
    int64_t out_off;            /* where they go: the sum of the ISIZEs before this block */
} isx_bgzf_block;
int isx_bgzf_index(const uint8_t *image, int64_t n_bytes, int64_t cap_blocks, isx_bgzf_block *blocks, int64_t *n_blocks, int64_t *out_bytes);
int isx_bgzf_inflate_device(isx_ctx *ctx, const uint8_t *image, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks,
                            uint8_t *out, int64_t out_bytes, float *kernel_ms);
int isx_bgzf_inflate_host(const uint8_t *image, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, int64_t out_bytes);
/* The BAM front end's host decoder on the calling thread (csrc/fast_inflate.h: table driven, about twice zlib's inflate on BAM blocks;
 * inside the front end zlib decodes whatever this one refuses) -- here without the fallback, for tests and tools/inflate_rate.py. */
int isx_bgzf_inflate_fast(const uint8_t *image, int64_t n_bytes, const isx_bgzf_block *blocks, int64_t n_blocks, uint8_t *out, int64_t out_bytes);

/* ---- host-side BAM front end (BGZF/BAM decode, read-pair filter, htslib-1.9 pileup rules) ----
 * Two passes over the (memory-mapped, compressed) file, like the reference, none of which holds the file's reads:
 *   isx_bam_scan         filter_reads.get_paired_reads for every reference (filter_reads.py:885-956): per
 *                        (scaffold, read name) the NM sum / mapq / length / insert record; per read 4 bytes
 *   isx_bam_filter       paired_read_filter + filter_scaff2pair2info (filter_reads.py:471-532, 201-260) with the
 *                        file's own median insert -- or a median handed in (one sample spread over several files)
 *   isx_bam_set_r2m      instead: the controller's own sR2M[scaffold] (profile_controller.py:415-433)
 *   isx_bam_expand_refs  samfile.pileup(...) of profile_utilities.py:150-153 for a SUBSET of the references (one
 *                        batch / one GPU's shard): only the BGZF blocks holding them are inflated again
 * isx_bam_expand = all three over the whole file. */
typedef struct isx_bam_params_s {
    double min_read_ani;        /* -l, 0.95 */
    int32_t min_mapq;           /* -1 */
    double max_insert_relative; /* 3 */
    int32_t min_insert;         /* 50 */
    int32_t min_base_quality;   /* 30 (profile_utilities.py:153) */
    int32_t skip_mm;            /* --skip_mm_profiling */
    int32_t window_length;      /* 10000 (fasta.py:56-73 iterate_splits) */
    int32_t pairing_filter;     /* 0 paired_only (default), 1 non_discordant, 2 all_reads (filter_reads.py:499-525) */
} isx_bam_params;

typedef struct isx_bam_info_s {
    int32_t n_refs;             /* references of the file (scan / filter) or of the batch (expand) */
    int32_t n_splits;
    int64_t n_reads;
    int64_t n_pos;              /* sum of reference lengths = flat space (of the file / of the batch) */
    int64_t n_obs;
    int64_t n_pairs;            /* dense pair ids handed out */
    int64_t unfiltered_pairs, filtered_pairs;
    int64_t filtered_bases;     /* sum of query lengths of filtered pairs ("Gbp profiled" numerator) */
    double median_insert;
    int32_t max_mm;
    int32_t pad;
    int64_t unfiltered_reads, unfiltered_singletons, filtered_singletons;   /* read_report tallies (filter_reads.py:487-495, 372-377) */
    int64_t n_segs;             /* read segments of the batch (isx_bam_segment_refs / isx_pipe_submit_bam on a read-level pipe), else 0 */
} isx_bam_info;

int isx_bam_open(const char *path, isx_bam **out);     /* header + BGZF block index; nothing else is inflated */
void isx_bam_close(isx_bam *bam);           /* a large handle is given back to the system on a thread of its own */
void isx_bam_close_wait(isx_bam *bam);      /* ... or before this returns (a caller that closes on its own helper thread) */
int isx_bam_set_threads(isx_bam *bam, int32_t threads); /* 0 = automatic (2 x the container's cpu quota, at most 64) */
int isx_bam_ref(const isx_bam *bam, int32_t i, const char **name, int64_t *length, int64_t *flat_offset);
/* names[offs[i] .. offs[i+1]) = i-th priority read (--priority_reads, filter_reads.py:428-469); used by the next isx_bam_filter */
int isx_bam_set_priority_reads(isx_bam *bam, int64_t n, const char *names, const int64_t *offs);
int isx_bam_scan(isx_bam *bam, isx_bam_info *info /* may be NULL */);
/* Pass 1 over share `part` of `n_parts` of the file (one share per rank of a multi-GPU run): the handle owns the references
 * whose first read lies in its share of the segments -- complete pair tables for those, every other reference looks empty
 * (isx_bam_ref_counts).  The file-wide median insert is the caller's to combine: isx_bam_insert_sizes of every share (an
 * all-gather) -> isx_bam_filter(median_insert); non_discordant / all_reads additionally need isx_bam_set_cross_names (below).
 * isx_bam_scan == share 0 of 1. */
int isx_bam_scan_part(isx_bam *bam, int32_t part, int32_t n_parts, isx_bam_info *info /* may be NULL */);
/* The scaffolds that exist for the read filter: the reference builds its pair table from the scaffolds of the fasta only
 * (filter_reads.py:63-77, 157-178), so the median insert (:213-217), the read_report tallies and the cross-scaffold name
 * look-ups of non_discordant / all_reads ignore every other reference of the BAM.  refs[n] reference ids; n == 0 = all
 * (a BAM mapped to exactly the fasta).  Applies to isx_bam_insert_sizes and isx_bam_filter. */
int isx_bam_set_wanted_refs(isx_bam *bam, const int32_t *refs, int32_t n);
/* insert sizes of the two-read pairs: out may be NULL to ask for *n only (a median across files / ranks) */
int isx_bam_insert_sizes(isx_bam *bam, int64_t *out, int64_t cap, int64_t *n);
int isx_bam_filter(isx_bam *bam, const isx_bam_params *p, double median_insert /* NaN = this file's own */, isx_bam_info *info);
/* non_discordant / all_reads over SHARES of a file (filter_reads.py:497-532 look every read name up across all scaffolds):
 * isx_bam_pair_keys hands out, per pair entry of this handle in table order, two independent 64-bit hashes of the name,
 * the reference it sits on and info4 = (nm, mapq, length, reads) as the scan found them (reads == 0: not an entry for
 * the filter); h1 == NULL asks for *n only.  The caller finds the names held by more than one scaffold over all shares and
 * tells every share about its own entries among them with isx_bam_set_cross_names: entry[i] = index into the table,
 * occurrences[i] = how many scaffolds of the file hold the name (>= 2), info4[4 i ..] = the entry's merged info after
 * _merge_info in header order (used by all_reads only).  isx_bam_filter then accepts those modes on a share.
 * isx_bam_filter_insert_sizes: the inserts of what went through paired_read_filter (the values the median is taken over)
 * after a first isx_bam_filter run -- gathered over the shares they give the file's median for the final run. */
int isx_bam_pair_keys(isx_bam *bam, uint64_t *h1, uint64_t *h2, int32_t *tid, int64_t *info4, int64_t cap, int64_t *n);
int isx_bam_set_cross_names(isx_bam *bam, int64_t n, const int64_t *entry, const int64_t *occurrences, const int64_t *info4);
int isx_bam_filter_insert_sizes(isx_bam *bam, int64_t *out, int64_t cap, int64_t *n);
int isx_bam_set_r2m(isx_bam *bam, int32_t ref, int64_t n, const char *names, const int64_t *offs, const int32_t *mm /* NULL = 0 */);
/* pairs with more than `cap` mismatches are piled up at mm level `cap` from now on (the device bins 128 levels: a caller that meets a
 * pair beyond that clamps and warns instead of failing the call); isx_bam_r2m still reports the true values */
int isx_bam_set_mm_cap(isx_bam *bam, int32_t cap);
/* Round 6: mm levels of any size, exactly (the reference bins any mm, profile_utilities.py:268-286; its tables depend on the ORDER of the
 * levels alone -- counts are cumulated over the levels <= mm, :297-312).  isx_bam_mm_levels: the distinct mm values of the kept pairs,
 * ascending (*n = how many there are; the first min(*n, cap) are written).  isx_bam_set_mm_levels: from now on a pair travels with the RANK
 * of its mm in `levels` (n_mm_bins = n; the caller maps the ranks in the tables back to the values); n = 0 switches the ranks off.  With
 * more than 128 distinct values isx_bam_set_mm_cap(127) still merges the ranks beyond. */
int isx_bam_mm_levels(const isx_bam *bam, int32_t *levels, int32_t cap, int32_t *n);
int isx_bam_set_mm_levels(isx_bam *bam, const int32_t *levels, int32_t n);
/* names of the read pairs of the batch prepared last (isx_bam_expand_refs / isx_bam_segment_refs / isx_pipe_submit_bam) by dense pair
 * id, i.e. by the ids isx_allele_obs.pair carries: names[offs[i] .. offs[i + 1]) (names / offs may be NULL to ask for the sizes).
 * Only while the handle still has the names (no isx_bam_drop_names before the batch was prepared): --store_everything. */
int isx_bam_batch_pair_names(const isx_bam *bam, int64_t *n, int64_t *name_bytes, char *names, int64_t *offs);
/* the reference's Rdic[scaffold] as the filter left it (read pair -> mm, controller.py:274-281): sizes first (names NULL),
 * then names[name_bytes], offs[n + 1], mm[n] */
int isx_bam_r2m(const isx_bam *bam, int32_t ref, int64_t *n, int64_t *name_bytes, char *names, int64_t *offs, int32_t *mm);
int isx_bam_drop_names(isx_bam *bam);                   /* frees the read names (no filter / set_r2m call may follow) */
/* per reference: records in the file, pairs that passed the filter (the reference's s2p, profile_controller.py:441) */
int isx_bam_ref_counts(const isx_bam *bam, int64_t *reads, int64_t *filtered_pairs);
/* overlap resolution + expansion of the given references (ascending ids = file order), laid end to end (the batch's
 * flat space); results stay in the handle until the next expand (isx_bam_copy / isx_bam_view) */
int isx_bam_expand_refs(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, isx_bam_info *info);
/* the same batch as READ SEGMENTS (isx_segs above) instead of observations -- what isx_pipe_submit_bam hands a read-level
 * pipe: no per-base walk on the host.  info->n_obs = the columns the segments cover (an upper bound of the observations);
 * isx_bam_copy_segs copies them out (any pointer may be NULL): gpos / len / mm / pair [n_seg], bases [n_seg][15],
 * split_bounds[n_splits + 1], split_ref[n_splits] */
int isx_bam_segment_refs(isx_bam *bam, const isx_bam_params *p, const int32_t *refs, int32_t n_refs, isx_bam_info *info, int64_t *n_seg);
int isx_bam_copy_segs(const isx_bam *bam, uint32_t *gpos, uint8_t *len, uint8_t *mm, uint32_t *pair, uint32_t *bases,
                      int64_t *split_bounds, int32_t *split_ref);
/* the segments of isx_bam_segment_refs as bit planes (isx_read_planes.planes: [n_seg][ISX_PLANE_WORDS]) -- what isx_pipe_submit_bam
 * hands a one-mm-bin pipe's stager, straight from the records' 4-bit seq */
int isx_bam_copy_read_planes(const isx_bam *bam, uint64_t *planes);
/* the re-pileup of SNV pooling (polymorpher.py:287-293: samfile.pileup(scaffold, start, stop, truncate=True)): only the columns
 * [start, stop) of one reference, from the reads overlapping them (only the BGZF blocks that can hold such reads are touched);
 * gpos stays the position on the reference */
int isx_bam_expand_region(isx_bam *bam, const isx_bam_params *p, int32_t ref, int64_t start, int64_t stop, isx_bam_info *info);
/* scan + filter + expansion of every reference of the file */
int isx_bam_expand(isx_bam *bam, const isx_bam_params *p, isx_bam_info *info);
/* copy out: obs[n_obs], pair[n_obs], split_bounds[n_splits+1], split_ref[n_splits] */
int isx_bam_copy(const isx_bam *bam, isx_obs *obs, uint32_t *pair, int64_t *split_bounds, int32_t *split_ref);
/* Zero-copy alternative to isx_bam_copy for the two large arrays: pointers into the handle, valid until the next
 * expand / isx_bam_close (isx_batch_create / isx_pipe_submit copy from them). */
int isx_bam_view(const isx_bam *bam, const isx_obs **obs, const uint32_t **pair);

#ifdef __cplusplus
}
#endif
#endif
