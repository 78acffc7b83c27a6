"""Mirror of the per-pair body of `inStrain compare`:

    inStrain.readComparer.calc_mm2overlap(covT1, covT2, min_cov)                 readComparer.py:145-191
    inStrain.readComparer._calc_SNP_count_alternate(SNPtable1, SNPtable2, ...)   readComparer.py:205-290
    inStrain.readComparer._update_overlap_table(...)                             readComparer.py:437-502
    inStrain.readComparer.compare_scaffold (one sample pair)                     readComparer.py:35-143

Two samples profiled on the same scaffolds (two resident batches over the same flat space); the
position-sized work (cumulate covT over mm, threshold at min_cov, intersect / unite) and the SNP-table
comparison (highest-mm row per position, merge of the two samples' rows, consensus / population
verdicts, masking with the covered-in-both set per mm) run on the device (isx_compare_coverage /
isx_compare_scaffolds); the per-mm bookkeeping and the column naming here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def calc_mm2overlap(batch1, batch2, scaffold_bounds, min_cov=5):
    """-> (list over scaffolds of {mm: n_covered_in_both}, list of {mm: coverage}, device ms).
    Levels = union of the two samples' covT keys on that scaffold, like the reference; coverage =
    len(coveredInBoth) / len(coveredInEither) (0 when nothing is covered)."""
    sb = np.ascontiguousarray(scaffold_bounds, dtype=np.int64)
    M = max(batch1.n_mm_bins, batch2.n_mm_bins)
    out = np.zeros((len(sb) - 1, M), dtype=_lib.COMPARE_LEVEL_DT)
    ms = C.c_float(0)
    check(batch1.lib.isx_compare_coverage(batch1.h, batch2.h, len(sb) - 1, sb.ctypes.data, int(min_cov), out.ctypes.data, C.byref(ms)))
    mm2overlap, mm2coverage = [], []
    for rows in out:
        o, c = {}, {}
        for r in rows:
            if r["present_a"] or r["present_b"]:
                o[int(r["mm"])] = int(r["both"])
                c[int(r["mm"])] = r["both"] / r["either"] if r["either"] > 0 else 0
        mm2overlap.append(o)
        mm2coverage.append(c)
    return mm2overlap, mm2coverage, ms.value


BASES = np.array(["A", "C", "T", "G", "N"])


def compare_scaffolds(batch1, batch2, scaffold_bounds, scaffold_names=None, name1="sample1", name2="sample2",
                      min_cov=5, min_freq=0.05, store_mismatch_locations=False):
    """One sample pair of compare_scaffold (readComparer.py:80-121) for every scaffold of the flat space.
    -> (Cdb rows: list of dicts with the reference's columns, ascending (scaffold, mm);
        Mdb: structured array of mismatch locations (COMPARE_SNP_DT + 'scaffold' index, 'position') or None;
        device ms).
    A scaffold on which the reference itself fails (KeyError on an N reference base, see
    include/instrain_amd.h) is reported as {'scaffold': ..., 'failed': True} like its results=None."""
    sb = np.ascontiguousarray(scaffold_bounds, dtype=np.int64)
    n_scaf = len(sb) - 1
    names = list(scaffold_names) if scaffold_names is not None else list(range(n_scaf))
    M = max(batch1.n_mm_bins, batch2.n_mm_bins)
    out = np.zeros((n_scaf, M), dtype=_lib.COMPARE_LEVEL_DT)
    ms, n_rows = C.c_float(0), C.c_int64(0)
    check(batch1.lib.isx_compare_scaffolds(batch1.h, batch2.h, n_scaf, sb.ctypes.data, int(min_cov), float(min_freq),
                                           out.ctypes.data, C.byref(n_rows), C.byref(ms)))
    table, failed = [], set()
    for i, rows in enumerate(out):
        mLen = int(sb[i + 1] - sb[i])
        if (rows["consensus_snps"] == -2).any():
            table.append({"scaffold": names[i], "failed": True})
            failed.add(i)
            continue
        for r in rows:
            if not (r["present_a"] or r["present_b"]):
                continue
            bases = int(r["both"])
            snps, popsnps = int(r["consensus_snps"]), int(r["population_snps"])
            table.append({"mm": int(r["mm"]), "scaffold": names[i], "name1": name1, "name2": name2,
                          "coverage_overlap": r["both"] / r["either"] if r["either"] > 0 else 0,
                          "compared_bases_count": bases, "percent_genome_compared": bases / mLen, "length": mLen,
                          "consensus_SNPs": snps, "population_SNPs": popsnps,
                          "conANI": (bases - snps) / bases if bases else np.nan,
                          "popANI": (bases - popsnps) / bases if bases else np.nan})
    mdb = None
    if store_mismatch_locations:
        raw = np.zeros(n_rows.value, dtype=_lib.COMPARE_SNP_DT)
        if n_rows.value:
            check(batch1.lib.isx_compare_fetch_snps(batch1.h, raw.ctypes.data))
        scaf = np.searchsorted(sb, raw["gpos"].astype(np.int64), side="right") - 1
        if failed:
            ok = ~np.isin(scaf, sorted(failed))
            raw, scaf = raw[ok], scaf[ok]
        mdb = {"raw": raw, "scaffold": scaf, "position": raw["gpos"].astype(np.int64) - sb[scaf]}
    return table, mdb, ms.value
