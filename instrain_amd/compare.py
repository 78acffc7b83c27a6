"""Mirror of the coverage-overlap step of `inStrain compare`:

    inStrain.readComparer.calc_mm2overlap(covT1, covT2, min_cov)      readComparer.py:145-191

Two samples profiled on the same scaffolds (two resident batches over the same flat space); the
position-sized work (cumulate covT over mm, threshold at min_cov, intersect / unite) runs on the
device (isx_compare_coverage), the per-mm bookkeeping here.
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
