"""Mirror of the re-pileup used by SNV pooling:

    inStrain.polymorpher.extract_SNVS_from_bam(bam_loc, R2M, positions, scaffold)   polymorpher.py:275-316
    get_pooling_counts                                                                polymorpher.py:312-315

The reference opens a second pileup iterator with the same htslib arguments as profile_split and
returns, for every requested position, mm_counts_to_counts(get_base_counts_mm(column, R2M)) = the
A,C,T,G counts over ALL mm levels.  Here that is one dense (n_mm_bins == 1) pass of the same
k_pileup_dense kernel over the whole BAM followed by a gather at the requested positions.
"""
import numpy as np

from .. import engine


def extract_SNVS_from_bam(bam_loc, R2M, positions, scaffold, ctx=None, null_model=None, **kwargs):
    """-> {position: np.array([A, C, T, G])} like the reference (zeros where nothing is piled up).

    R2M is accepted for signature compatibility; the read-pair filter is recomputed by the C++ front
    end from the same flags (min_read_ani, min_mapq, max_insert_relative, min_insert), which yields the
    R2M the reference would have stored for this BAM."""
    positions = [int(p) for p in positions]
    if len(positions) == 0:
        return {}
    own = ctx is None
    if own:
        ctx = engine.Context(int(kwargs.get('device', 0)))
    if null_model is not None:
        from .snv_utilities import null_model_lut
        ctx.set_null_model(*null_model_lut(null_model))
    bf = engine.BamFile(bam_loc)
    try:
        obs, pair, bounds, sref = bf.expand(min_read_ani=kwargs.get('min_read_ani', 0.95), min_mapq=kwargs.get('min_mapq', -1),
                                            max_insert_relative=kwargs.get('max_insert_relative', 3),
                                            min_insert=kwargs.get('min_insert', 50), skip_mm=True)
        refs = {n: (ln, off) for n, ln, off in bf.refs()}
        if scaffold not in refs:
            raise ValueError("scaffold {0} is not in the .bam file {1}!".format(scaffold, bam_loc))
        n_pos = int(bf.info["n_pos"])
    finally:
        bf.close()
    b = engine.Batch(ctx, np.zeros(n_pos, dtype=np.uint8), [0, n_pos], obs, None, n_mm_bins=1, enable_linkage=False,
                     rarefied_coverage=0)
    try:
        b.run()
        counts = b.fetch()["counts"]
    finally:
        b.close()
        if own:
            ctx.close()
    ln, off = refs[scaffold]
    out = {}
    for p in positions:
        out[p] = counts[off + p].astype(np.int64) if 0 <= p < ln else np.zeros(4, dtype=np.int64)
    return out
