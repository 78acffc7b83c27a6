"""ctypes binding of libinstrain_amd.so (include/instrain_amd.h).

There is NO CPU fallback: if the HIP extension is missing or no MI355X is visible the calls
raise IsxError.  Nothing here imports `oracle/`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libinstrain_amd.so")
if os.environ.get("ISX_LIB"):            # A/B timing of two in-tree builds (tools/); never a fallback
    LIB_PATH = os.path.abspath(os.environ["ISX_LIB"])

BGZF_BLOCK_DT = np.dtype([("in_off", np.int64), ("in_len", np.int32), ("out_len", np.int32), ("out_off", np.int64)])
OBS_DT = np.dtype([("gpos", "<u4"), ("mm", "<u2"), ("base", "u1"), ("flags", "u1")])
ENTRY_DT = np.dtype([("gpos", "<u4"), ("mm", "<u2"), ("flags", "<u2"), ("cnt", "<u4", (4,)), ("clon", "<f4"),
                     ("clon_rarefied", "<f4")])
SNV_DT = np.dtype([("gpos", "<u4"), ("mm", "<u2"), ("con_base", "u1"), ("var_base", "u1"),
                   ("allele_count", "u1"), ("cls", "u1"), ("cryptic", "u1"), ("ref_base", "u1"),
                   ("cnt", "<u4", (4,))])
LD_DT = np.dtype([("gpos_a", "<u4"), ("gpos_b", "<u4"), ("mm", "<u2"), ("allele_A", "u1"), ("allele_a", "u1"),
                  ("allele_B", "u1"), ("allele_b", "u1"), ("pad", "<u2"), ("total", "<u4"), ("countAB", "<u4"),
                  ("countAb", "<u4"), ("countaB", "<u4"), ("countab", "<u4"), ("pad2", "<u4"),
                  ("r2", "<f8"), ("d_prime", "<f8"), ("r2_normalized", "<f8"), ("d_prime_normalized", "<f8")])
AO_DT = np.dtype([("pair", "<u4"), ("gpos", "<u4"), ("order", "<u4"), ("mm", "<u2"), ("base", "u1"), ("pad", "u1")])
assert AO_DT.itemsize == 16
assert OBS_DT.itemsize == 8 and ENTRY_DT.itemsize == 32 and SNV_DT.itemsize == 28 and LD_DT.itemsize == 72


class Params(C.Structure):
    _fields_ = [("min_cov", C.c_int32), ("min_snp", C.c_int32), ("min_freq", C.c_double),
                ("rarefied_coverage", C.c_int32), ("n_mm_bins", C.c_int32), ("enable_linkage", C.c_int32),
                ("linkage_mode", C.c_int32), ("window", C.c_int32), ("seed", C.c_uint64), ("layout", C.c_int32),
                ("reserved", C.c_int32)]


LAYOUT_WIDE_RECORDS, LAYOUT_NO_SHORT_RECORDS, LAYOUT_NO_PACKED_COUNTERS, LAYOUT_SEG64_RECORDS = 1, 2, 4, 8          # NO_PACKED_COUNTERS also keeps reference-delta batches on 32-bit LDS counters


class PipeParams(C.Structure):
    _fields_ = [("max_pos", C.c_int64), ("max_obs", C.c_int64), ("max_splits", C.c_int32), ("depth", C.c_int32),
                ("host_threads", C.c_int32), ("pin_threads", C.c_int32), ("jump_slack", C.c_double),
                ("want_counts", C.c_int32), ("ring_kib", C.c_int32), ("stage_async", C.c_int32), ("lean_output", C.c_int32),
                ("max_segs", C.c_int64)]


SEG_BASES, SEG_WORDS, SEG_SKIP_WORD = 150, 15, 0x24924924


class Segs(C.Structure):
    """isx_segs: SoA view of a batch of read segments (pointers into numpy arrays the caller keeps alive)"""
    _fields_ = [("n_seg", C.c_int64), ("gpos", C.c_void_p), ("len", C.c_void_p), ("mm", C.c_void_p), ("pair", C.c_void_p),
                ("bases", C.c_void_p)]


PLANE_WORDS = 8
LAYOUT_MM_ENTRIES = 16          # isx_params.layout bits (include/instrain_amd.h ISX_LAYOUT_*)
LAYOUT_MM_DELTA_RECORDS = 32
ABI_VERSION = 5             # ISX_ABI_VERSION of include/instrain_amd.h this binding was written against


class ReadPlanes(C.Structure):
    """isx_read_planes: the same segments as bit planes, one 64-byte line each"""
    _fields_ = [("n_seg", C.c_int64), ("gpos", C.c_void_p), ("len", C.c_void_p), ("pair", C.c_void_p), ("planes", C.c_void_p), ("mm", C.c_void_p)]


class RefPlanes(C.Structure):
    """isx_ref_planes: the reference as it travels (2-bit plane + bit plane of the positions that are not A/C/T/G, or NULL)"""
    _fields_ = [("plane2", C.c_void_p), ("nplane", C.c_void_p), ("key", C.c_uint64)]


RARE_DT = np.dtype([("gpos", "<u4"), ("clon_rarefied", "<f4")])
CLON_DT = np.dtype([("gpos", "<u4"), ("clon", "<f4")])          # isx_pipe_result.clon_sparse (isx_rare's layout)
SAT_DT = np.dtype([("gpos", "<u4"), ("coverage", "<u4")])


class Sizes(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_entries", "n_snv", "n_sites", "n_allele_obs", "n_increments",
                                         "n_edges", "n_ld")]


class Timings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("pileup_ms", "sites_ms", "allele_ms", "group_ms", "incr_ms", "ld_ms",
                                         "total_ms")] + \
               [(n, C.c_int32) for n in ("pileup_blocks", "pileup_threads", "pileup_lds_bytes", "pileup_window")] + \
               [("mfma_ms", C.c_float), ("dense_tiles", C.c_int32), ("dense_macs", C.c_int64), ("dense_bytes", C.c_int64),
                ("record_bytes", C.c_int32), ("pad", C.c_int32)]


class PipeResult(C.Structure):
    _fields_ = [("ticket", C.c_int64), ("n_pos", C.c_int64), ("n_obs", C.c_int64), ("sizes", Sizes),
                ("coverage16", C.c_void_p), ("clon", C.c_void_p), ("rare", C.c_void_p), ("n_rare", C.c_int64),
                ("n_saturated", C.c_int64), ("counts", C.c_void_p), ("clon_rarefied", C.c_void_p), ("snv", C.c_void_p),
                ("batch", C.c_void_p),
                ("encode_ms", C.c_float), ("h2d_ms", C.c_float), ("kernel_ms", C.c_float), ("d2h_ms", C.c_float),
                ("collect_wait_ms", C.c_float), ("record_bytes", C.c_int32), ("encode_passes", C.c_int32),
                ("h2d_bytes", C.c_int64), ("d2h_bytes", C.c_int64), ("ld", C.c_void_p),
                ("coverage8", C.c_void_p), ("clon_sparse", C.c_void_p), ("n_clon", C.c_int64), ("saturated", C.c_void_p),
                ("coverage4", C.c_void_p), ("cov_rows", C.c_void_p), ("cov_row_window", C.c_void_p), ("n_cov_rows", C.c_int64),
                ("cov_window", C.c_int32), ("pad_cov", C.c_int32),
                # mm profiling on, level-sparse hand-back (include/instrain_amd.h isx_pipe_result.lev_*)
                ("lev_mask", C.c_void_p), ("lev_cov", C.c_void_p), ("lev_win_off", C.c_void_p), ("lev_clon", C.c_void_p), ("lev_rare", C.c_void_p),
                ("lev_sat", C.c_void_p), ("n_lev", C.c_int64), ("n_lev_clon", C.c_int64), ("n_lev_rare", C.c_int64), ("n_lev_sat", C.c_int64),
                ("lev_mask_bytes", C.c_int32), ("lev_cov_bytes", C.c_int32), ("lev_window", C.c_int32), ("n_lev_windows", C.c_int32),
                ("lev_min_cov", C.c_int32), ("pad_lev", C.c_int32), ("rows_checksum", C.c_uint64)]


class BamParams(C.Structure):
    _fields_ = [("min_read_ani", C.c_double), ("min_mapq", C.c_int32), ("max_insert_relative", C.c_double),
                ("min_insert", C.c_int32), ("min_base_quality", C.c_int32), ("skip_mm", C.c_int32),
                ("window_length", C.c_int32), ("pairing_filter", C.c_int32)]


PAIRING_FILTERS = {"paired_only": 0, "non_discordant": 1, "all_reads": 2}


class BamInfo(C.Structure):
    _fields_ = [("n_refs", C.c_int32), ("n_splits", C.c_int32), ("n_reads", C.c_int64), ("n_pos", C.c_int64),
                ("n_obs", C.c_int64), ("n_pairs", C.c_int64), ("unfiltered_pairs", C.c_int64),
                ("filtered_pairs", C.c_int64), ("filtered_bases", C.c_int64), ("median_insert", C.c_double),
                ("max_mm", C.c_int32), ("pad", C.c_int32), ("unfiltered_reads", C.c_int64),
                ("unfiltered_singletons", C.c_int64), ("filtered_singletons", C.c_int64), ("n_segs", C.c_int64)]


SCAFFOLD_LEVEL_DT = np.dtype([("nonzero", "<i8"), ("sum_cov", "<u8"), ("sumsq_cov", "<u8"), ("median_cov", "<f8"),
                              ("counted", "<i8"), ("sum_clon", "<f8"), ("median_clon", "<f8"),
                              ("counted_rarefied", "<i8"), ("sum_clon_rarefied", "<f8"),
                              ("median_clon_rarefied", "<f8"), ("mm", "<i4"), ("present", "<i4")])
assert SCAFFOLD_LEVEL_DT.itemsize == 88
GENOME_LEVEL_DT = np.dtype([("n", "<i8"), ("sum_cov", "<u8"), ("sumsq_cov", "<u8"), ("median_cov", "<f8"), ("mm", "<i4"), ("pad", "<i4")])
assert GENOME_LEVEL_DT.itemsize == 40


COMPARE_LEVEL_DT = np.dtype([("both", "<i8"), ("either", "<i8"), ("mm", "<i4"), ("present_a", "<i4"),
                             ("present_b", "<i4"), ("pad", "<i4"), ("consensus_snps", "<i8"), ("population_snps", "<i8")])
assert COMPARE_LEVEL_DT.itemsize == 48
COMPARE_SNP_DT = np.dtype([("gpos", "<u4"), ("mm", "<u2"), ("consensus_snp", "u1"), ("population_snp", "u1"),
                           ("has_a", "u1"), ("has_b", "u1"), ("con_a", "u1"), ("ref_a", "u1"), ("var_a", "u1"),
                           ("con_b", "u1"), ("ref_b", "u1"), ("var_b", "u1"), ("cnt_a", "<u4", (4,)), ("cnt_b", "<u4", (4,))])
assert COMPARE_SNP_DT.itemsize == 48


ERR_CAPACITY = -3          # ISX_ERR_CAPACITY (include/instrain_amd.h)


class IsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libinstrain_amd error %d: %s" % (code, msg))
        self.code = code


SYMBOLS = ["isx_last_error", "isx_abi_version", "isx_ctx_create", "isx_ctx_destroy", "isx_ctx_reserve_cus", "isx_set_null_model",
           "isx_batch_create", "isx_batch_create_reads", "isx_batch_destroy", "isx_batch_run", "isx_batch_launch", "isx_batch_wait", "isx_batch_sizes", "isx_batch_timings",
           "isx_batch_fetch_entries", "isx_batch_fetch_dense", "isx_batch_fetch_snv", "isx_batch_fetch_ld", "isx_batch_fetch_allele_obs",
           "isx_batch_summarize", "isx_batch_summarize_genomes", "isx_compare_coverage", "isx_compare_scaffolds", "isx_compare_fetch_snps",
           "isx_pipe_create", "isx_pipe_destroy", "isx_pipe_submit", "isx_pipe_submit_reads", "isx_pipe_stage_reads", "isx_pipe_submit_wire", "isx_wire_bytes", "isx_wire_free", "isx_wire_keep_reference", "isx_pipe_submit_bam", "isx_encode_segs", "isx_encode_segs_ring", "isx_seg_records_needed", "isx_encode_delta", "isx_delta_records_needed", "isx_count_read_segs", "isx_pack_reads", "isx_pipe_collect", "isx_pipe_release", "isx_pipe_fetch_entries", "isx_pipe_fetch_entries_shrunk", "isx_levels_expand", "isx_encode_obs", "isx_encode_obs_ring",
           "isx_pack_ref_planes", "isx_planes_from_segs", "isx_pack_read_planes", "isx_pipe_submit_planes", "isx_pipe_stage_planes", "isx_encode_planes", "isx_encode_planes_mm", "isx_pipe_set_reference_budget", "isx_host_register", "isx_host_unregister",
           "isx_bgzf_index", "isx_bgzf_inflate_device", "isx_bgzf_inflate_host", "isx_bgzf_inflate_fast",
           "isx_bam_open", "isx_bam_close", "isx_bam_close_wait", "isx_bam_set_threads", "isx_bam_ref", "isx_bam_set_priority_reads", "isx_bam_scan", "isx_bam_scan_part",
           "isx_bam_insert_sizes", "isx_bam_set_wanted_refs", "isx_bam_pair_keys", "isx_bam_set_cross_names", "isx_bam_filter_insert_sizes", "isx_bam_filter", "isx_bam_set_r2m", "isx_bam_r2m", "isx_bam_drop_names", "isx_bam_batch_pair_names", "isx_bam_set_mm_cap", "isx_bam_mm_levels", "isx_bam_set_mm_levels", "isx_bam_ref_counts",
           "isx_bam_expand_refs", "isx_bam_segment_refs", "isx_bam_copy_segs", "isx_bam_copy_read_planes", "isx_bam_expand_region", "isx_bam_expand", "isx_bam_copy", "isx_bam_view"]

_lib = None


def load():
    """Load the in-tree shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IsxError(-2, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.isx_last_error.restype = C.c_char_p
    lib.isx_abi_version.restype = C.c_int
    if lib.isx_abi_version() != ABI_VERSION:        # structs and record formats changed between versions: never call into another one
        raise IsxError(-6, "%s has ABI version %d, this package binds version %d: rebuild it (python -c 'import __graft_entry__ as g; g.build()')"
                       % (LIB_PATH, lib.isx_abi_version(), ABI_VERSION))
    lib.isx_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.isx_ctx_destroy.argtypes = [vp]
    lib.isx_ctx_reserve_cus.argtypes = [vp, C.c_int]
    lib.isx_bgzf_index.argtypes = [vp, i64, i64, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.isx_bgzf_inflate_device.argtypes = [vp, vp, i64, vp, i64, vp, i64, C.POINTER(C.c_float)]
    lib.isx_bgzf_inflate_host.argtypes = [vp, i64, vp, i64, vp, i64]
    lib.isx_bgzf_inflate_fast.argtypes = [vp, i64, vp, i64, vp, i64]
    lib.isx_ctx_destroy.restype = None
    lib.isx_set_null_model.argtypes = [vp, vp, i64, i32]
    lib.isx_batch_create.argtypes = [vp, C.POINTER(Params), i64, vp, i32, vp, i64, vp, vp, C.POINTER(vp)]
    lib.isx_batch_create_reads.argtypes = [vp, C.POINTER(Params), i64, vp, i32, vp, C.POINTER(Segs), C.POINTER(vp)]
    lib.isx_pipe_submit_reads.argtypes = [vp, i64, vp, i32, vp, C.POINTER(Segs), C.POINTER(i64)]
    lib.isx_pipe_stage_reads.argtypes = [vp, i64, vp, i32, vp, C.POINTER(Segs), C.POINTER(vp)]
    lib.isx_pipe_submit_wire.argtypes = [vp, vp, C.POINTER(i64)]
    lib.isx_wire_bytes.argtypes = [vp]
    lib.isx_wire_bytes.restype = i64
    lib.isx_wire_free.argtypes = [vp]
    lib.isx_wire_keep_reference.argtypes = [vp, vp]
    lib.isx_wire_free.restype = None
    lib.isx_encode_segs.argtypes = [C.POINTER(Segs), i64, i32, i32, i64, vp, vp, vp, C.POINTER(i64)]
    lib.isx_encode_segs_ring.argtypes = [C.POINTER(Segs), i64, i32, i32, i64, i64, vp, vp, vp, C.POINTER(i64)]
    lib.isx_encode_delta.argtypes = [C.POINTER(Segs), vp, i64, i32, i32, i32, i64, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.isx_delta_records_needed.argtypes = [vp, i64, i32, i32]
    lib.isx_delta_records_needed.restype = i64
    lib.isx_seg_records_needed.argtypes = [vp, i64, i32]
    lib.isx_seg_records_needed.restype = i64
    lib.isx_count_read_segs.argtypes = [i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    lib.isx_pack_reads.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, C.POINTER(i64)]
    lib.isx_batch_destroy.argtypes = [vp]
    lib.isx_batch_destroy.restype = None
    lib.isx_batch_run.argtypes = [vp]
    lib.isx_batch_launch.argtypes = [vp]
    lib.isx_batch_wait.argtypes = [vp]
    lib.isx_batch_sizes.argtypes = [vp, C.POINTER(Sizes)]
    lib.isx_batch_timings.argtypes = [vp, C.POINTER(Timings)]
    for f in ("isx_batch_fetch_entries", "isx_batch_fetch_snv", "isx_batch_fetch_ld", "isx_batch_fetch_allele_obs"):
        getattr(lib, f).argtypes = [vp, vp]
    lib.isx_batch_fetch_dense.argtypes = [vp, vp, vp, vp]
    lib.isx_batch_summarize.argtypes = [vp, i32, vp, vp, C.POINTER(C.c_float)]
    lib.isx_batch_summarize_genomes.argtypes = [vp, i32, vp, i32, vp, i32, vp, C.POINTER(C.c_float)]
    lib.isx_compare_coverage.argtypes = [vp, vp, i32, vp, i32, vp, C.POINTER(C.c_float)]
    lib.isx_compare_scaffolds.argtypes = [vp, vp, i32, vp, i32, C.c_double, vp, C.POINTER(i64), C.POINTER(C.c_float)]
    lib.isx_compare_fetch_snps.argtypes = [vp, vp]
    lib.isx_pipe_create.argtypes = [vp, C.POINTER(Params), C.POINTER(PipeParams), C.POINTER(vp)]
    lib.isx_pipe_destroy.argtypes = [vp]
    lib.isx_pipe_destroy.restype = None
    lib.isx_pipe_submit.argtypes = [vp, i64, vp, i32, vp, i64, vp, vp, C.POINTER(i64)]
    lib.isx_pipe_submit_bam.argtypes = [vp, vp, C.POINTER(BamParams), vp, i32, vp, i32, vp, C.POINTER(BamInfo), C.POINTER(i64)]
    lib.isx_pipe_collect.argtypes = [vp, i64, C.POINTER(PipeResult)]
    lib.isx_pipe_release.argtypes = [vp, i64]
    lib.isx_pipe_fetch_entries.argtypes = [vp, i64, vp]
    lib.isx_pipe_fetch_entries_shrunk.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.isx_levels_expand.argtypes = [C.POINTER(PipeResult), i32, vp, vp, vp, vp]
    lib.isx_encode_obs.argtypes = [vp, vp, i64, i64, i32, i32, C.c_double, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i32)]
    lib.isx_encode_obs_ring.argtypes = [vp, vp, i64, i64, i32, i32, C.c_double, i64, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i32)]
    lib.isx_bam_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.isx_bam_close.argtypes = [vp]
    lib.isx_bam_close.restype = None
    lib.isx_bam_close_wait.argtypes = [vp]
    lib.isx_bam_close_wait.restype = None
    lib.isx_bam_expand.argtypes = [vp, C.POINTER(BamParams), C.POINTER(BamInfo)]
    lib.isx_bam_set_threads.argtypes = [vp, i32]
    lib.isx_bam_set_priority_reads.argtypes = [vp, i64, C.c_char_p, vp]
    lib.isx_bam_scan.argtypes = [vp, C.POINTER(BamInfo)]
    lib.isx_bam_scan_part.argtypes = [vp, i32, i32, C.POINTER(BamInfo)]
    lib.isx_bam_set_wanted_refs.argtypes = [vp, vp, i32]
    lib.isx_bam_insert_sizes.argtypes = [vp, vp, i64, C.POINTER(i64)]
    lib.isx_bam_filter_insert_sizes.argtypes = [vp, vp, i64, C.POINTER(i64)]
    lib.isx_bam_pair_keys.argtypes = [vp, vp, vp, vp, vp, i64, C.POINTER(i64)]
    lib.isx_bam_set_cross_names.argtypes = [vp, i64, vp, vp, vp]
    lib.isx_bam_filter.argtypes = [vp, C.POINTER(BamParams), C.c_double, C.POINTER(BamInfo)]
    lib.isx_bam_set_r2m.argtypes = [vp, i32, i64, C.c_char_p, vp, vp]
    lib.isx_bam_drop_names.argtypes = [vp]
    lib.isx_bam_r2m.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), vp, vp, vp]
    lib.isx_bam_batch_pair_names.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), vp, vp]
    lib.isx_bam_set_mm_cap.argtypes = [vp, i32]
    lib.isx_bam_mm_levels.argtypes = [vp, vp, i32, vp]
    lib.isx_bam_set_mm_levels.argtypes = [vp, vp, i32]
    lib.isx_bam_ref_counts.argtypes = [vp, vp, vp]
    lib.isx_bam_expand_region.argtypes = [vp, C.POINTER(BamParams), i32, i64, i64, C.POINTER(BamInfo)]
    lib.isx_bam_expand_refs.argtypes = [vp, C.POINTER(BamParams), vp, i32, C.POINTER(BamInfo)]
    lib.isx_bam_segment_refs.argtypes = [vp, C.POINTER(BamParams), vp, i32, C.POINTER(BamInfo), C.POINTER(i64)]
    lib.isx_bam_copy_segs.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.isx_bam_copy_read_planes.argtypes = [vp, vp]
    lib.isx_pack_ref_planes.argtypes = [vp, i64, i32, vp, vp, C.POINTER(i32)]
    lib.isx_planes_from_segs.argtypes = [C.POINTER(Segs), i32, vp]
    lib.isx_pack_read_planes.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, C.POINTER(i64)]
    lib.isx_pipe_submit_planes.argtypes = [vp, i64, C.POINTER(RefPlanes), i32, vp, C.POINTER(ReadPlanes), C.POINTER(i64)]
    lib.isx_pipe_set_reference_budget.argtypes = [vp, i64]
    lib.isx_host_register.argtypes = [vp, i64]
    lib.isx_host_unregister.argtypes = [vp]
    lib.isx_pipe_stage_planes.argtypes = [vp, i64, C.POINTER(RefPlanes), i32, vp, C.POINTER(ReadPlanes), C.POINTER(vp)]
    lib.isx_encode_planes.argtypes = [C.POINTER(ReadPlanes), C.POINTER(RefPlanes), i64, i32, i32, i64, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.isx_encode_planes_mm.argtypes = [C.POINTER(ReadPlanes), C.POINTER(RefPlanes), i64, i32, i32, i32, i64, i64, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.isx_bam_ref.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i64)]
    lib.isx_bam_copy.argtypes = [vp, vp, vp, vp, vp]
    lib.isx_bam_view.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise IsxError(rc, load().isx_last_error().decode(errors="replace"))
