"""
oracle/compare.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the per-scaffold, per-sample-pair body of inStrain.readComparer.compare_scaffold
(/root/reference/inStrain/readComparer.py):
  calc_mm2overlap            :145-191  cumulate each sample's covT over ascending mm, threshold at
                                       min_cov, intersect / unite
  _calc_SNP_count_alternate  :205-290  + call_con_snps :292-304, is_present :306-315,
                                       call_pop_snps :317-371
  _update_overlap_table      :437-502
Input: the oracle's `entries` / `snv` tables of the two samples on one scaffold.  Pinned by
tests/golden/compare_*.npz (outputs of the reference's own functions, tests/golden/make_golden.py).
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


def _is_present(count, total, lut, fallback, min_freq):
    """readComparer.py:306-315 is_present: model[total] if total in model else model[-1]"""
    total = int(total)
    min_bases = int(lut[total]) if 0 <= total < len(lut) and lut[total] >= 0 else fallback
    return (count >= min_bases) and ((float(count) / total) >= min_freq)


def compare_snp_tables(snv1, snv2, mm2overlap, lut, fallback, min_freq=0.05):
    """readComparer.py:205-290 _calc_SNP_count_alternate + 292-304 call_con_snps + 317-371
    call_pop_snps on the oracle's SNV tables of one scaffold (the cumulative_snv_table is the
    raw table, profile_utilities.py:579-596; compare sorts it by mm, compare_utils.py:124-138, so
    drop_duplicates(keep='last') keeps every position's HIGHEST-mm row -- at every compared mm).
    -> list of (mm, position, consensus_SNP, population_SNP) for rows with either flag, sorted."""
    def last_rows(s):
        o = np.lexsort((s["mm"], s["pos"]))
        s = s[o]
        keep = np.r_[s["pos"][1:] != s["pos"][:-1], True] if len(s) else np.zeros(0, bool)
        return {int(r["pos"]): r for r in s[keep]}
    r1, r2 = last_rows(snv1), last_rows(snv2)
    verdict = {}
    for p in sorted(set(r1) | set(r2)):
        a, b = r1.get(p), r2.get(p)
        if a is None or b is None:
            x = b if a is None else a                      # the sample that has the row
            con = x["con_base"] != x["ref_base"]
            ref = int(x["ref_base"])
            if not 0 <= ref < 4:
                raise KeyError("N_%d" % (2 if a is None else 1))    # the reference's own failure
            pop = not _is_present(int(x["cnt"][ref]), x["position_coverage"], lut, fallback, min_freq)
        else:
            con = a["con_base"] != b["con_base"]
            if not con:
                pop = False
            elif _is_present(int(b["cnt"][a["con_base"]]), b["position_coverage"], lut, fallback, min_freq):
                pop = False
            elif _is_present(int(a["cnt"][b["con_base"]]), a["position_coverage"], lut, fallback, min_freq):
                pop = False
            elif a["allele_count"] > 1 and b["allele_count"] > 1 and a["var_base"] == b["var_base"]:
                pop = False
            else:
                pop = True
        if con or pop:
            verdict[p] = (bool(con), bool(pop))
    rows = []
    for mm in sorted(mm2overlap):
        covs = mm2overlap[mm]
        rows += [(mm, p, c, q) for p, (c, q) in verdict.items() if p in covs]
    return rows


def overlap_table(mm2overlap, mm2coverage, snp_rows, length):
    """readComparer.py:437-502 _update_overlap_table -> list of dict rows ascending mm"""
    out = []
    for mm in sorted(mm2overlap):
        bases = len(mm2overlap[mm])
        snps = sum(1 for r in snp_rows if r[0] == mm and r[2])
        pops = sum(1 for r in snp_rows if r[0] == mm and r[3])
        out.append({"mm": mm, "coverage_overlap": mm2coverage[mm], "compared_bases_count": bases,
                    "percent_genome_compared": bases / length, "length": length,
                    "consensus_SNPs": snps, "population_SNPs": pops,
                    "conANI": (bases - snps) / bases if bases else np.nan,
                    "popANI": (bases - pops) / bases if bases else np.nan})
    return out
