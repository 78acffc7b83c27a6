"""
oracle/compare.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of inStrain.readComparer.calc_mm2overlap
(/root/reference/inStrain/readComparer.py:145-191): cumulate each sample's covT over ascending mm,
threshold at min_cov, intersect / unite.  Input: the oracle's `entries` tables of the two samples
on one scaffold.  Pinned by tests/golden/compare_*.npz (outputs of the reference's own function).
"""
import numpy as np


def calc_mm2overlap(entries1, entries2, length, min_cov=5):
    def level_sums(e):
        s = e["cnt"].sum(axis=1)
        return e["pos"][s > 0], e["mm"][s > 0], s[s > 0]
    p1, m1, s1 = level_sums(entries1)
    p2, m2, s2 = level_sums(entries2)
    mms = sorted(set(int(x) for x in m1) | set(int(x) for x in m2))      # union of the covT keys
    cov1 = np.zeros(length, dtype=np.int64)
    cov2 = np.zeros(length, dtype=np.int64)
    mm2overlap, mm2coverage = {}, {}
    for mm in mms:
        np.add.at(cov1, p1[m1 == mm], s1[m1 == mm])
        np.add.at(cov2, p2[m2 == mm], s2[m2 == mm])
        t1, t2 = cov1 >= min_cov, cov2 >= min_cov
        both, either = np.nonzero(t1 & t2)[0], int((t1 | t2).sum())
        mm2overlap[mm] = set(int(x) for x in both)
        mm2coverage[mm] = len(both) / either if either > 0 else 0
    return mm2overlap, mm2coverage
