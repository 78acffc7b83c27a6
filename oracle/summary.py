"""
oracle/summary.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the reference's per-scaffold merge summaries:

  make_coverage_table          /root/reference/inStrain/profile/profile_utilities.py:425-506
  mm_counts_to_counts_shrunk   profile_utilities.py:508-532
  get_basewise_clons           profile_utilities.py:534-546
  estimate_breadth             profile_utilities.py:548-555
  calc_snps                    /root/reference/inStrain/profile/snv_utilities.py:249-272

Input: the oracle's own per-split tables of ONE scaffold (entries / snv structured arrays of
oracle/oracle.py, positions absolute on the scaffold).  Pinned by the stored
cumulative_scaffold_table of the reference's sars_cov_2 run (tests/test_oracle_golden.py).
"""
import numpy as np

from .oracle import CLASSES


def estimate_breadth(coverage):
    return (-1) * np.exp(-1 * ((0.883) * coverage)) + 1


def coverage_table(entries, snv, length, clon_r=None):
    """-> list of dict rows, one per mm level that is a key of covT (every level some column created, also one whose
    only read showed a non-ACGT base: update_covT stores 0 and shrink_basewise keeps the key), ascending, with the
    reference's column names."""
    lvl_sum = entries["cnt"].sum(axis=1)
    cov_levels = sorted(set(int(m) for m in entries["mm"]))                  # covT keys after shrink_basewise
    rows = []
    for mm in cov_levels:
        k = entries["mm"] <= mm
        covs = np.zeros(length, dtype=np.float64)                           # fill_zeros=lengt, float64 Series
        np.add.at(covs, entries["pos"][k], lvl_sum[k].astype(np.float64))
        nonzeros = int(np.count_nonzero(covs))

        def clons_upto(values):
            # get_basewise_clons: p2c.update(clonT[m].to_dict()) for ascending m -> python dict order
            p2c = {}
            for m in sorted(set(int(x) for x in entries["mm"][k])):
                sel = (entries["mm"] == m) & ~np.isnan(values)
                o = np.argsort(entries["pos"][sel], kind="stable")
                for p, v in zip(entries["pos"][sel][o], values[sel][o]):
                    p2c[int(p)] = float(np.float32(v))     # Series.to_dict() boxes to python floats ->
            return list(p2c.values())                       # np.mean / np.median below run in float64

        clons = clons_upto(entries["clon"])
        rclons = clons_upto(clon_r) if clon_r is not None else []
        counted = len(clons)
        row = {"length": length, "breadth": nonzeros / length, "coverage": np.mean(covs),
               "coverage_median": int(np.median(covs)), "coverage_std": np.std(covs),
               "coverage_SEM": float(np.std(covs, ddof=1) / np.sqrt(length)) if length > 1 else float("nan")}
        if counted:
            row["nucl_diversity"] = 1 - np.mean(clons)
            row["nucl_diversity_median"] = 1 - np.median(clons)
        else:
            row["nucl_diversity"] = row["nucl_diversity_median"] = np.nan
        if len(rclons):
            row["nucl_diversity_rarefied"] = 1 - np.mean(rclons)
            row["nucl_diversity_rarefied_median"] = 1 - np.median(rclons)
        else:
            row["nucl_diversity_rarefied"] = row["nucl_diversity_rarefied_median"] = np.nan
        row["breadth_minCov"] = counted / length
        row["breadth_rarefied"] = len(rclons) / length
        row["breadth_expected"] = estimate_breadth(row["coverage"])
        # calc_snps: rows with mm' <= mm, per position the one with the highest mm'
        s = snv[snv["mm"] <= mm]
        if len(s):
            o = np.lexsort((s["mm"], s["pos"]))
            s = s[o]
            last = np.r_[s["pos"][1:] != s["pos"][:-1], True]
            s = s[last]
        cls = np.array(CLASSES)[s["cls"]] if len(s) else np.array([], dtype=str)
        row["divergent_site_count"] = len(s)
        row["SNS_count"] = int((s["allele_count"] == 1).sum()) if len(s) else 0
        row["SNV_count"] = int((s["allele_count"] > 1).sum()) if len(s) else 0
        row["consensus_divergent_sites"] = int(np.isin(cls, ["SNS", "con_SNV", "pop_SNV"]).sum())
        row["population_divergent_sites"] = int(np.isin(cls, ["SNS", "pop_SNV"]).sum())
        if counted == 0:
            row["conANI_reference"] = row["popANI_reference"] = 0
        else:
            row["conANI_reference"] = (counted - row["consensus_divergent_sites"]) / counted
            row["popANI_reference"] = (counted - row["population_divergent_sites"]) / counted
        row["mm"] = mm
        rows.append(row)
    return rows


def genome_coverage_rows(covT, s2l, genome2scaffolds, mms, mask_edges=100):
    """genomeLevel_coverage_info (/root/reference/inStrain/genomeUtilities.py:297-365) without iRep, on
    generate_genome_coverage_array (:932-981): per genome and mm the coverage of all its scaffolds laid end to end with
    `mask_edges` positions cut from both ends of every scaffold (a scaffold shorter than twice that drops out), cumulative
    over levels <= mm -> median (int), SEM (ddof 1), std (ddof 0).
    covT: scaffold -> {mm: (positions, values)} (the shrunk covT); s2l: scaffold -> length; rows in the reference's order
    (genomes in dict order, mms ascending)."""
    rows = []
    for genome, scaffolds in genome2scaffolds.items():
        scaffolds = [s for s in scaffolds if s in s2l]
        for mm in mms:
            arrs = []
            for sc in scaffolds:
                ln = int(s2l[sc])
                cov = np.zeros(ln, dtype=np.float64)                # mm_counts_to_counts_shrunk(fill_zeros=slen); absent scaffold -> NaN -> fillna(0)
                for m, (pos, val) in covT.get(sc, {}).items():
                    if int(m) <= int(mm):
                        np.add.at(cov, np.asarray(pos, dtype=np.int64), np.asarray(val, dtype=np.float64))
                if mask_edges:
                    cov = cov[mask_edges:ln - mask_edges] if ln >= 2 * mask_edges else cov[:0]
                arrs.append(cov)
            covs = np.concatenate(arrs) if arrs else np.zeros(0)
            if len(covs) == 0:
                covs = np.zeros(1)                                  # pd.Series([0]) in the reference
            n = len(covs)
            sem = float(np.std(covs, ddof=1) / np.sqrt(n)) if n > 1 else float("nan")       # scipy.stats.sem
            rows.append({"mm": int(mm), "genome": genome, "coverage_median": int(np.median(covs)),
                         "coverage_SEM": sem, "coverage_std": float(np.std(covs))})
    return rows
