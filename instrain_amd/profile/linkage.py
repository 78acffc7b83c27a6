"""The --store_everything extras of the reference's linkage step, rebuilt on the host from the device's allele observations.

    read_to_snvs          update_linked_reads           /root/reference/inStrain/profile/linkage.py:254-283
    mm_to_position_graph  calc_mm_SNV_linkage_network   linkage.py:14-44

The device profiles linkage from the same appends (one row per read pair x SNP site x base in the site's `bases` set,
isx_batch_fetch_allele_obs); no downstream table reads these two objects -- the reference keeps them on the split only with
store_everything (profile_utilities.py:205-211) -- so they are made on demand, in plain Python, per split.
"""
import itertools
from collections import defaultdict

import numpy as np

BASES = "ACTG"          # P2C order (profile_utilities.py:34)


def _dlist():
    return defaultdict(list)


class SortedAlleleObs:
    """A batch's allele observations sorted ONCE by position: every split then cuts its rows with two bisections instead of masking the
    whole batch's array (a 64 Mbp batch has thousands of splits and millions of rows: O(splits x rows) otherwise)."""

    def __init__(self, ao):
        order = np.argsort(ao["gpos"], kind="stable")
        self.rows = ao[order]
        self.gpos = np.ascontiguousarray(self.rows["gpos"]).astype(np.int64)

    def cut(self, lo, hi):
        a, b = np.searchsorted(self.gpos, [int(lo), int(hi)])
        return self.rows[a:b]


def read_to_snvs_of_split(ao, lo, hi, pair_names=None):
    """Allele observations of one split -> read_to_snvs: mm -> read name -> ["position:base", ...] with positions relative to
    the split's start (RelPosition, profile_utilities.py:244, 263-265), a read's entries in column order and, inside a column,
    in the order the pileup visited the mates.  ao: the batch's rows (engine.Batch.fetch_allele_obs) or a SortedAlleleObs of them;
    [lo, hi): the split's range of flat positions; pair_names: dense pair id -> read-pair name (None: the id itself, as "r<id>" has
    no meaning here)."""
    out = defaultdict(_dlist)
    if isinstance(ao, SortedAlleleObs):
        a = ao.cut(lo, hi)
    else:
        g = ao["gpos"].astype(np.int64)
        a = ao[np.flatnonzero((g >= lo) & (g < hi))]
    if not len(a):
        return out
    order = np.lexsort((a["order"], a["gpos"], a["pair"], a["mm"]))
    a = a[order]
    rel = a["gpos"].astype(np.int64) - int(lo)
    for mm, pair, r, b in zip(a["mm"].tolist(), a["pair"].tolist(), rel.tolist(), a["base"].tolist()):
        name = pair if pair_names is None else pair_names[pair]
        out[mm][name].append("%d:%s" % (r, BASES[b]))
    return out


def calc_mm_SNV_linkage_network(read_to_snvs, scaff=False):
    """linkage.py:14-44: every pair of a read's SNV entries (itertools.combinations, list order) adds one to
    G[p1][p2]['mm2combo2counts'][mm]["b1:b2"]; returns a networkx Graph like the reference (networkx is one of the
    reference's own requirements)."""
    import networkx as nx
    G = nx.Graph()
    for mm, tread2snvs in read_to_snvs.items():
        for read in tread2snvs:
            for e1, e2 in itertools.combinations(tread2snvs[read], 2):
                s1, b1 = e1.split(":")
                s2, b2 = e2.split(":")
                p1, p2 = int(s1), int(s2)
                if not G.has_edge(p1, p2):
                    G.add_edge(p1, p2)
                    G[p1][p2]['mm2combo2counts'] = {}
                d = G[p1][p2]['mm2combo2counts'].setdefault(mm, {})
                c = b1 + ":" + b2
                d[c] = d.get(c, 0) + 1
    return G


def flatten(read_to_snvs, G, name_to_id=int):
    """canonical flat arrays of the two objects (what tests/golden/linkage_extras.npz stores):
    rts = rows (mm, read id, index in the read's list, position, base code) sorted; graph = rows (p1, p2, mm, b1, b2, count) with
    p1 / p2 as the combo was counted (the first element of a combination comes first), sorted"""
    rts = []
    for mm, reads in read_to_snvs.items():
        for name, lst in reads.items():
            for k, e in enumerate(lst):
                p, b = e.split(":")
                rts.append((int(mm), name_to_id(name), k, int(p), BASES.index(b)))
    gr = []
    for p1, p2, d in G.edges(data=True):
        for mm, combos in d['mm2combo2counts'].items():
            for c, n in combos.items():
                b1, b2 = c.split(":")
                a, b = (p1, p2) if p1 <= p2 else (p2, p1)       # a combination's first element is the earlier column
                gr.append((int(a), int(b), int(mm), BASES.index(b1), BASES.index(b2), int(n)))
    rts = np.array(sorted(rts), dtype=np.int64).reshape(-1, 5)
    gr = np.array(sorted(gr), dtype=np.int64).reshape(-1, 6)
    return rts, gr
