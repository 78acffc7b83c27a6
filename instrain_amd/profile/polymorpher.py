"""Mirror of the re-pileup used by SNV pooling:

    inStrain.polymorpher.extract_SNVS_from_bam(bam_loc, R2M, positions, scaffold)   polymorpher.py:275-316
    get_pooling_counts                                                                polymorpher.py:312-315

The reference opens a second pileup iterator with the same htslib arguments as profile_split and
returns, for every requested position, mm_counts_to_counts(get_base_counts_mm(column, R2M)) = the
A,C,T,G counts over ALL mm levels.  Here that is one dense (n_mm_bins == 1) pass of the same
k_pileup_dense kernel over the region the reference piles up, followed by a gather at the requested positions.
"""
import numpy as np

from .. import engine


def extract_SNVS_from_bam(bam_loc, R2M, positions, scaffold, ctx=None, null_model=None, **kwargs):
    """-> {position: np.array([A, C, T, G])} like the reference (zeros where nothing is piled up).

    Like the reference (polymorpher.py:287-293) only the columns [min(positions) - 1, max(positions) + 1) of the scaffold
    are piled up, from the reads that overlap them.  R2M: the scaffold's {pair: mm} (or a set of pair names) the way the
    reference hands it over -- exactly those read pairs are counted; None = the built-in read filter with the flags in
    kwargs (min_read_ani, min_mapq, max_insert_relative, min_insert, pairing_filter)."""
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
        names = [n for n, _, _ in bf.refs()]
        if scaffold not in names:
            raise ValueError("scaffold {0} is not in the .bam file {1}!".format(scaffold, bam_loc))
        tid = names.index(scaffold)
        ln = bf.refs()[tid][1]
        lo, hi = max(min(positions) - 1, 0), min(max(positions) + 1, ln)
        fkw = dict(min_read_ani=kwargs.get('min_read_ani', 0.95), min_mapq=kwargs.get('min_mapq', -1),
                   max_insert_relative=kwargs.get('max_insert_relative', 3), min_insert=kwargs.get('min_insert', 50),
                   pairing_filter=kwargs.get('pairing_filter', 'paired_only'))
        bf.scan()
        if R2M is None:
            bf.filter(**fkw)
        else:
            bf.set_r2m(tid, list(R2M), None)
        if hi <= lo:
            obs = np.zeros(0, dtype=engine.OBS_DT)
        else:
            obs, _, _, _ = bf.expand_region(tid, lo, hi, skip_mm=True, **fkw)
    finally:
        bf.close()
    out = {}
    n_pos = max(hi - lo, 1)
    if len(obs):
        obs = obs.copy()
        obs["gpos"] -= lo                                   # the device batch covers the region only
        b = engine.Batch(ctx, np.zeros(n_pos, dtype=np.uint8), [0, n_pos], obs, None, n_mm_bins=1, enable_linkage=False,
                         rarefied_coverage=0)
        try:
            b.run()
            counts = b.fetch()["counts"]
        finally:
            b.close()
    else:
        counts = np.zeros((n_pos, 4), dtype=np.uint32)
    if own:
        ctx.close()
    for p in positions:
        out[p] = counts[p - lo].astype(np.int64) if lo <= p < hi else np.zeros(4, dtype=np.int64)
    return out
