"""
oracle/py_columns.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A per-column PYTHON restatement of the reference's split worker, shaped like the reference itself: one pass over
the pileup columns of a split, per column a dict mm -> numpy array of four counts filled read by read, numpy /
Python arithmetic per (position, mm level), per-read lists of SNV alleles and a dict-of-dicts linkage network.  It is
what SURVEY.md section 8(d) asks to be timed next to the C port as the "reference-like" CPU baseline
(bench.py cpu_baseline_python: multiprocessing over splits like profile_controller.py:243-271), and a second,
independent restatement the C port is checked against (tests/test_oracle_golden.py).

Restates, in the reference's order of evaluation:
  process_bam_sites            /root/reference/inStrain/profile/profile_utilities.py:218-266
  get_base_counts_mm           profile_utilities.py:268-286
  update_covT                  profile_utilities.py:288-295
  mm_counts_to_counts          profile_utilities.py:297-312
  update_snp_table             /root/reference/inStrain/profile/snv_utilities.py:40-145
  call_snv_site                snv_utilities.py:147-196
  calc_snp_class               snv_utilities.py:198-223 (is_present: readComparer.py:307-316)
  calculate_clonality          snv_utilities.py:225-231
  generate_snp_table           snv_utilities.py:274-290
  update_linked_reads          /root/reference/inStrain/profile/linkage.py:254-283
  calc_mm_SNV_linkage_network  linkage.py:14-44
  calculate_ld ... _calc_ld_single  linkage.py:46-240
The two unseeded-random outputs (clonTR, *_normalized) are left out like in oracle_core.c.

Input: the packed observation stream of one split (one record per (pileup column, pileup read) visit on which the
reference touches its table), exactly what oracle.profile_split takes; output: the same structured arrays.
"""
from collections import defaultdict
from itertools import combinations

import numpy as np

from .oracle import ENTRY_DT, LD_DT, SNV_DT

P2C = {'A': 0, 'C': 1, 'T': 2, 'G': 3}
C2P = "ACTG"


def _min_bases(null_model, total):
    """the count a base needs at this coverage; coverages missing from the model use model[-1] (snv_utilities.py:174-177)"""
    return null_model[total] if total in null_model else null_model[-1]


def call_snv_site(counts, ref_base, null_model, min_cov, min_freq):
    """-> (None | -1 | consensus base index, number of bases present) -- snv_utilities.py:147-196"""
    total = int(counts.sum())
    if total < min_cov:
        return None, 0
    need = _min_bases(null_model, total)
    present = 0
    for c in counts:
        if c >= need and float(c) / total >= min_freq:
            present += 1
    top = int(np.argmax(counts))
    if present > 1:
        return top, present
    if present == 1 and C2P[top] != ref_base:
        return top, present
    if present == 0:
        return top, present
    return -1, present


def calculate_clonality(counts):
    """snv_utilities.py:225-231: sum of squared base frequencies, fp64 in source order"""
    s = float(counts.sum())
    prob = (float(counts[0]) / s) * (float(counts[0]) / s) + (float(counts[1]) / s) * (float(counts[1]) / s) \
        + (float(counts[2]) / s) * (float(counts[2]) / s) + (float(counts[3]) / s) * (float(counts[3]) / s)
    return prob


def calc_snp_class(con, ref, var, counts, allele_count, null_model, min_freq):
    """index into oracle.CLASSES -- snv_utilities.py:198-223"""
    if ref not in P2C:
        return 0
    if allele_count == 0:
        return 1
    if allele_count == 1:
        return 2
    if C2P[con] == ref:
        return 3
    if C2P[var] == ref:
        return 4
    total = int(counts.sum())
    c = counts[P2C[ref]]
    if c >= _min_bases(null_model, total) and float(c) / total >= min_freq:
        return 4
    return 5


def major_minor(counts):
    """the two most frequent bases, ties in A, C, T, G order (a stable descending sort, linkage.py:133-136)"""
    order = sorted(range(4), key=lambda k: counts[k], reverse=True)
    return order[0], order[1]


def profile_split(pos, base, mm, pair, seq, start, null_model, min_cov=5, min_freq=0.05, min_snp=20):
    """One split, column by column.  null_model: dict {coverage: count, -1: fallback} (generate_snp_model).
    Returns the dict oracle.profile_split returns."""
    mLen = len(seq)
    # the pileup iterator: columns in position order, the reads of a column in arrival order
    columns = defaultdict(list)
    for i in range(len(pos)):
        rp = int(pos[i]) - start
        if 0 <= rp < mLen:                          # truncate=True
            columns[rp].append((int(pair[i]), int(mm[i]), int(base[i])))

    entries, snv_rows = [], []
    read_to_snvs = defaultdict(lambda: defaultdict(list))    # mm -> read -> [(position, base)]
    snv2mm2counts = {}
    for rp in sorted(columns):
        # get_base_counts_mm: the level is created before the base is looked at
        table = {}
        for name, m, b in columns[rp]:
            if m not in table:
                table[m] = np.zeros(4, dtype=np.int64)
            if b < 4:
                table[m][b] += 1
        ref_base = seq[rp]
        any_snp, cryptic = False, False
        bases = set()
        first_row = len(snv_rows)
        counts = np.zeros(4, dtype=np.int64)
        for m in sorted(table):                     # update_covT + update_snp_table, levels ascending
            counts = counts + table[m]              # mm_counts_to_counts(MMcounts, mm)
            snp, morphia = call_snv_site(counts, ref_base, null_model, min_cov, min_freq)
            clon = np.float32(calculate_clonality(counts)) if counts.sum() >= min_cov else np.float32(np.nan)
            entries.append((rp + start, m, tuple(int(x) for x in table[m]), clon))
            if snp is None:
                continue
            if snp != -1:
                rest = list(counts)
                rest[snp] = 0
                var = rest.index(max(rest))
                snv_rows.append([rp + start, m, tuple(int(x) for x in counts), P2C.get(ref_base, 4), snp, var, morphia,
                                 calc_snp_class(snp, ref_base, var, counts, morphia, null_model, min_freq), 0,
                                 int(counts.sum())])
                if morphia >= 2:
                    any_snp = True
                    bases.add(snp)
                    bases.add(var)
                elif morphia == 1 and any_snp:
                    cryptic = True
            elif any_snp:
                cryptic = True
        if cryptic:
            for r in snv_rows[first_row:]:
                r[8] = 1
        if any_snp:                                 # update_linked_reads
            for name, m, b in columns[rp]:
                if b < 4 and b in bases:
                    read_to_snvs[m][name].append((rp, b))
            snv2mm2counts[rp] = {m: table[m].copy() for m in table}

    # calc_mm_SNV_linkage_network: every pair of SNV alleles seen on one read (pair), per mm level
    graph = defaultdict(lambda: defaultdict(lambda: defaultdict(int)))     # (p1, p2) -> mm -> (b1, b2) -> n
    n_increments = 0
    for m, reads in read_to_snvs.items():
        for name, snvs in reads.items():
            for (p1, b1), (p2, b2) in combinations(snvs, 2):
                graph[(p1, p2)][m][(b1, b2)] += 1
                n_increments += 1

    def cum_counts(p, m):
        out = np.zeros(4, dtype=np.int64)
        for k, v in snv2mm2counts[p].items():
            if k <= m:
                out = out + v
        return out

    ld_rows = []
    for (p1, p2) in sorted(graph):
        combo = defaultdict(int)
        for m in sorted(graph[(p1, p2)]):           # counts cumulate over ascending mm
            for k, v in graph[(p1, p2)][m].items():
                combo[k] += v
            if m not in snv2mm2counts[p1] or m not in snv2mm2counts[p2]:
                continue
            cA, cB = cum_counts(p1, m), cum_counts(p2, m)
            if int(cA.sum() + cB.sum()) < min_snp:
                continue
            A, a = major_minor(cA)
            B, b = major_minor(cB)
            if cA[A] == 0 or cA[a] == 0 or cB[B] == 0 or cB[b] == 0:
                continue
            AB, Ab, aB, ab = combo[(A, B)], combo[(A, b)], combo[(a, B)], combo[(a, b)]
            total = AB + Ab + aB + ab
            if not total > min_snp:
                continue
            t = float(total)
            fAB, fAb, faB, fab = AB / t, Ab / t, aB / t, ab / t
            fA, fa, fB, fb = fAB + fAb, fab + faB, fAB + faB, fab + fAb
            linkD = fAB - fA * fB
            r2 = float('nan') if (fa == 0 or fA == 0 or fB == 0 or fb == 0) else linkD * linkD / (fA * fa * fB * fb)
            linkd = fab - fa * fb
            dp = float('nan')
            if linkd < 0:
                dp = linkd / max([-fA * fB, -fa * fb])
            elif linkD > 0:
                dp = linkd / min([fA * fb, fa * fB])
            ld_rows.append((p1 + start, p2 + start, m, abs(p2 - p1), total, AB, Ab, aB, ab, A, a, B, b, r2, dp))

    E = np.zeros(len(entries), dtype=ENTRY_DT)
    for i, (p, m, c, cl) in enumerate(entries):
        E[i] = (p, m, c, cl)
    S = np.zeros(len(snv_rows), dtype=SNV_DT)
    for i, r in enumerate(snv_rows):
        S[i] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9])
    L = np.zeros(len(ld_rows), dtype=LD_DT)
    for i, r in enumerate(ld_rows):
        L[i] = r
    return {"entries": E, "snv": S, "ld": L, "n_edges": len(graph), "n_increments": n_increments}
