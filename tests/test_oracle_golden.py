"""Pins the C oracle (oracle/oracle_core.c) against golden vectors produced by the
reference's own Python (tests/golden/make_golden.py) -- CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle
from tests import util

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(util.GOLD, "synth_*.npz"))) + ["c3_split"]


def run_oracle_case(g, lut, fb):
    return oracle.profile_split(g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]),
                                lut, fb, min_cov=int(g["p_min_cov"]), min_freq=float(g["p_min_freq"]),
                                min_snp=int(g["p_min_snp"]))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_vectors(name):
    lut, fb = util.load_lut()
    g = util.load_case(name)
    res = run_oracle_case(g, lut, fb)
    got = util.canon_from_struct(res)
    exp = util.canon_from_golden(g)
    util.assert_same(got, exp, float_tol=0.0, what=name)     # same IEEE ops in the same order -> identical
    assert res["n_edges"] == int(g["n_edges"])


def test_cases_exist():
    assert len(CASES) >= 8


@pytest.mark.parametrize("name", ["synth_mm4", "synth_selfpairs", "synth_ambig", "synth_skipmm_ld", "synth_lowcov"])
def test_python_column_restatement_matches_reference_vectors(name):
    """oracle/py_columns.py (the per-column Python loop bench.py times as the reference-like CPU baseline) gives the
    reference's tables too -- a second restatement, independent of the C port"""
    from oracle import py_columns
    lut, fb = util.load_lut()
    nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
    nm[-1] = fb
    g = util.load_case(name)
    res = py_columns.profile_split(g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), nm,
                                   min_cov=int(g["p_min_cov"]), min_freq=float(g["p_min_freq"]), min_snp=int(g["p_min_snp"]))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=0.0, what=name)
    assert res["n_edges"] == int(g["n_edges"])
    c = run_oracle_case(g, lut, fb)
    assert res["n_increments"] == c["n_increments"]


# ---------------------------------------------------------------------------------------
# The reference's stored golden run (sars_cov_2 .IS folder, inStrain 1.2.4 + real pysam)
# ---------------------------------------------------------------------------------------
def iterate_splits(sLen, W):
    """profile/fasta.py:56-73"""
    n = sLen // W + 1
    cl = int(sLen / n)
    out, s, e = [], 0, 0
    for i in range(n):
        if i + 1 == n:
            out.append((s, sLen - 1))
        else:
            e += cl
            out.append((s, e - 1))
            s += cl
    return out


def read_fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">")).upper()


def sars_golden_tables():
    import pandas as pd
    S = pd.read_csv(os.path.join(util.GOLD, "sars_cov_2_raw_snp_table.csv.gz")).rename(
        columns={"refBase": "ref_base", "conBase": "con_base", "varBase": "var_base",
                 "baseCoverage": "position_coverage"})      # test/tests/test_utils.py:165-195
    L = pd.read_csv(os.path.join(util.GOLD, "sars_cov_2_raw_linkage_table.csv.gz"))
    return (S.sort_values(["position", "mm"]).reset_index(drop=True),
            L.sort_values(["position_A", "position_B", "mm"]).reset_index(drop=True))


def check_against_sars_golden(snv, ld, float_tol):
    """snv / ld: concatenated structured arrays (oracle field names), absolute positions."""
    S, L = sars_golden_tables()
    o = np.lexsort((snv["mm"], snv["pos"]))
    snv = snv[o]
    assert len(snv) == len(S) == 707
    assert (snv["pos"] == S["position"].values).all() and (snv["mm"] == S["mm"].values).all()
    for i, b in enumerate("ACTG"):
        assert (snv["cnt"][:, i] == S[b].values).all(), b
    assert (util.BASES[snv["con_base"]] == S["con_base"].values).all()
    assert (util.BASES[snv["var_base"]] == S["var_base"].values).all()
    assert (util.BASES[snv["ref_base"]] == S["ref_base"].values).all()
    assert (snv["allele_count"] == S["allele_count"].values).all()
    assert (snv["cryptic"].astype(bool) == S["cryptic"].values).all()
    assert (snv["position_coverage"] == S["position_coverage"].values).all()
    o = np.lexsort((ld["mm"], ld["pos_b"], ld["pos_a"]))
    ld = ld[o]
    assert len(ld) == len(L) == 2228
    for a, b in [("pos_a", "position_A"), ("pos_b", "position_B"), ("mm", "mm"), ("distance", "distance"),
                 ("total", "total"), ("cAB", "countAB"), ("cAb", "countAb"), ("caB", "countaB"), ("cab", "countab")]:
        assert (ld[a] == L[b].values).all(), b
    for k in ["allele_A", "allele_a", "allele_B", "allele_b"]:
        assert (util.BASES[ld[k]] == L[k].values).all(), k
    for a, b in [("r2", "r2"), ("d_prime", "d_prime")]:
        x, y = ld[a], L[b].values.astype(float)
        assert (np.isnan(x) == np.isnan(y)).all(), b
        m = ~np.isnan(x)
        assert np.max(np.abs(x[m] - y[m])) <= float_tol, (b, np.max(np.abs(x[m] - y[m])))
    assert (ld["distance"] == 0).sum() == 16        # self pairs are real


def test_oracle_vs_stored_sars_golden():
    lut, fb = util.load_lut()
    z = np.load(os.path.join(util.GOLD, "sars_cov_2_obs.npz"))
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    snv, ld = [], []
    for s, e in iterate_splits(len(seq), 10000):
        r = oracle.profile_split(z["pos"], z["base"], z["mm"], z["pair"], seq[s:e + 1], s, lut, fb,
                                 min_cov=5, min_freq=0.05, min_snp=20)
        snv.append(r["snv"]); ld.append(r["ld"])
    check_against_sars_golden(np.concatenate(snv), np.concatenate(ld), float_tol=1e-9)


def test_iterate_splits_sweep():
    rows = np.load(os.path.join(util.GOLD, "iterate_splits.npy"))
    for L in np.unique(rows[:, 0]):
        for W in (1000, 10000):
            exp = rows[(rows[:, 0] == L) & (rows[:, 1] == W)][:, 3:5]
            got = np.array(iterate_splits(int(L), W))
            assert (got == exp).all(), (L, W)


def test_bam_py_reproduces_sars_observations():
    """BGZF/BAM decode + read filter + htslib-1.9 overlap rules (oracle/bam_py.py) regenerate the
    committed packed observations; read-filter tallies equal the stored read_report."""
    import pandas as pd
    from oracle import bam_py
    refs, reads = bam_py.read_bam(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    assert refs == [("MT039887.1", 29879)] and len(reads) == 28913
    p2i = bam_py.get_paired_reads(reads, 0)
    r2m, tallies = bam_py.filter_pairs({"MT039887.1": p2i})
    t = tallies["MT039887.1"]
    rr = pd.read_csv(os.path.join(util.GOLD, "sars_cov_2_read_report.csv.gz"), comment="#")
    row = rr[rr["scaffold"] == "MT039887.1"].iloc[0]
    assert t["filtered_pairs"] == int(row["filtered_pairs"]) == 13124
    assert t["unfiltered_pairs"] == int(row["unfiltered_pairs"])
    assert t["median_insert"] == 267.0
    bam_py.resolve_overlaps(reads, 0)
    pos, base, mm, pair, _ = bam_py.expand_observations(reads, 0, r2m["MT039887.1"])
    z = np.load(os.path.join(util.GOLD, "sars_cov_2_obs.npz"))
    assert len(pos) == len(z["pos"]) == 3717600
    assert (pos == z["pos"]).all() and (base == z["base"]).all() and (mm == z["mm"]).all() and (pair == z["pair"]).all()


# mapping of the stored v1.2.4 column names to the current ones (test/tests/test_utils.py:160-195)
OLD2NEW = {"median_cov": "coverage_median", "std_cov": "coverage_std", "mean_microdiversity": "nucl_diversity",
           "median_microdiversity": "nucl_diversity_median", "unmaskedBreadth": "breadth_minCov",
           "expected_breadth": "breadth_expected", "SNPs": "divergent_site_count", "Reference_SNPs": "SNS_count",
           "consensus_SNPs": "consensus_divergent_sites", "population_SNPs": "population_divergent_sites",
           "conANI": "conANI_reference", "popANI": "popANI_reference"}


def check_coverage_table_vs_sars_golden(rows, float_tol):
    """rows: list of dicts with the CURRENT column names, one per mm level (any order)."""
    import pandas as pd
    g = pd.read_csv(os.path.join(util.GOLD, "sars_cov_2_cumulative_scaffold_table.csv.gz")).rename(columns=OLD2NEW)
    got = pd.DataFrame(rows).sort_values("mm").reset_index(drop=True)
    g = g.sort_values("mm").reset_index(drop=True)
    assert len(got) == len(g) == 26 and (got["mm"].values == g["mm"].values).all()
    for c in ["length", "coverage_median", "divergent_site_count", "SNS_count", "consensus_divergent_sites",
              "population_divergent_sites"]:
        assert (got[c].values.astype(np.int64) == g[c].values.astype(np.int64)).all(), c
    assert (got["SNV_count"].values == (g["BiAllelic_SNPs"] + g["MultiAllelic_SNPs"]).values).all()
    assert ((got["length"] - (got["breadth"] * got["length"]).round().astype(int)).values == g["bases_w_0_coverage"].values).all()
    for c in ["breadth", "coverage", "coverage_std", "nucl_diversity", "nucl_diversity_median", "breadth_minCov",
              "breadth_expected", "conANI_reference", "popANI_reference"]:
        a, b = got[c].values.astype(float), g[c].values.astype(float)
        assert np.max(np.abs(a - b)) <= float_tol * max(1.0, np.max(np.abs(b))), (c, np.max(np.abs(a - b)))


def test_oracle_coverage_table_vs_stored_golden():
    """oracle/summary.py (make_coverage_table restatement) on the oracle's own tables == the 26 rows
    of the reference's stored cumulative_scaffold_table"""
    from oracle import summary
    lut, fb = util.load_lut()
    z = np.load(os.path.join(util.GOLD, "sars_cov_2_obs.npz"))
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    ent, snv = [], []
    for s, e in iterate_splits(len(seq), 10000):
        r = oracle.profile_split(z["pos"], z["base"], z["mm"], z["pair"], seq[s:e + 1], s, lut, fb)
        ent.append(r["entries"]); snv.append(r["snv"])
    rows = summary.coverage_table(np.concatenate(ent), np.concatenate(snv), len(seq))
    check_coverage_table_vs_sars_golden(rows, float_tol=1e-12)


@pytest.mark.parametrize("name", ["compare_a", "compare_b", "compare_c", "compare_d"])
def test_oracle_compare_overlap_vs_reference_vectors(name):
    """oracle/compare.py vs outputs of the reference's own calc_mm2overlap (readComparer.py:145-191),
    _calc_SNP_count_alternate (:205-290) and _update_overlap_table (:437-502)"""
    from oracle import compare
    lut, fb = util.load_lut()
    g = util.load_case(name)
    seq = str(g["seq"])
    ra = oracle.profile_split(g["a_pos"], g["a_base"], g["a_mm"], g["a_pair"], seq, 0, lut, fb)
    rb = oracle.profile_split(g["b_pos"], g["b_base"], g["b_mm"], g["b_pair"], seq, 0, lut, fb)
    ea, eb = ra["entries"], rb["entries"]
    o, c = compare.calc_mm2overlap(ea, eb, len(seq), min_cov=5)
    assert sorted(o) == list(g["mm"])
    assert [len(o[m]) for m in g["mm"]] == list(g["both"])
    assert np.max(np.abs(np.array([c[m] for m in g["mm"]]) - g["coverage"])) == 0.0
    assert sorted(o[int(g["mm"][-1])]) == list(g["pos_in_both_last"])
    rows = compare.compare_snp_tables(ra["snv"], rb["snv"], o, lut, fb, min_freq=0.05)
    assert [r[0] for r in rows] == list(g["m_mm"]) and [r[1] for r in rows] == list(g["m_position"])
    assert [r[2] for r in rows] == list(g["m_consensus_SNP"]) and [r[3] for r in rows] == list(g["m_population_SNP"])
    t = compare.overlap_table(o, c, rows, len(seq))
    assert [r["consensus_SNPs"] for r in t] == list(g["t_consensus_SNPs"])
    assert [r["population_SNPs"] for r in t] == list(g["t_population_SNPs"])
    for k in ("conANI", "popANI", "percent_genome_compared"):
        np.testing.assert_array_equal(np.array([r[k] for r in t], dtype=np.float64), g["t_" + k])


def _genome_inputs():
    g = np.load(os.path.join(util.GOLD, "genome_coverage_inputs.npz"))
    scaffolds = [str(s) for s in g["scaffolds"]]
    s2l = dict(zip(scaffolds, (int(x) for x in g["lengths"])))
    covT = {}
    for si, mm, pos, val in g["cov"]:
        d = covT.setdefault(scaffolds[int(si)], {}).setdefault(int(mm), ([], []))
        d[0].append(int(pos)); d[1].append(int(val))
    g2s = {}
    for genome, sc in g["genome_of"]:
        g2s.setdefault(str(genome), []).append(str(sc))
    return covT, s2l, g2s, [int(m) for m in g["mms"]]


def test_genome_level_coverage_rollup_vs_reference():
    """oracle/summary.genome_coverage_rows against the reference's genomeLevel_coverage_info (genomeUtilities.py:297-365)
    on a synthetic covT: masked scaffold edges, a scaffold shorter than the mask, one without coverage, one absent"""
    import pandas as pd
    from oracle import summary
    covT, s2l, g2s, mms = _genome_inputs()
    rows = summary.genome_coverage_rows(covT, s2l, g2s, mms, mask_edges=100)
    ref = pd.read_csv(os.path.join(util.GOLD, "genome_coverage.csv"))
    assert len(rows) == len(ref) == 12
    for r, (_, e) in zip(rows, ref.iterrows()):
        assert r["mm"] == e["mm"] and r["genome"] == e["genome"] and r["coverage_median"] == e["coverage_median"], (r, dict(e))
        for k in ("coverage_SEM", "coverage_std"):
            assert (np.isnan(r[k]) and np.isnan(e[k])) or abs(r[k] - e[k]) <= 1e-9 * max(1.0, abs(e[k])), (k, r, dict(e))
