"""GPU parity for the per-pair body of `inStrain compare`: isx_compare_coverage /
isx_compare_scaffolds vs golden vectors produced by the reference's own calc_mm2overlap
(readComparer.py:145-191), _calc_SNP_count_alternate (:205-290) and _update_overlap_table (:437-502)."""
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from instrain_amd import engine
    c = engine.Context(0)
    lut, fb = util.load_lut()
    c.set_null_model(lut, fb)
    yield c
    c.close()


def _batch(ctx, seq_codes, bounds, pos, base, mm, pair, n_mm=None):
    from instrain_amd import engine
    n_mm = int(mm.max()) + 1 if n_mm is None else n_mm
    b = engine.Batch(ctx, seq_codes, bounds, engine.pack_obs(pos.astype(np.uint32), base, mm), pair.astype(np.uint32),
                     n_mm_bins=n_mm, enable_linkage=False)
    b.run()
    return b


@pytest.mark.parametrize("name", ["compare_a", "compare_b"])
def test_overlap_vs_reference_vectors(ctx, name):
    from instrain_amd import compare, engine
    g = util.load_case(name)
    codes = engine.encode_seq(str(g["seq"]))
    b1 = _batch(ctx, codes, [0, len(codes)], g["a_pos"], g["a_base"], g["a_mm"], g["a_pair"])
    b2 = _batch(ctx, codes, [0, len(codes)], g["b_pos"], g["b_base"], g["b_mm"], g["b_pair"])
    mm2overlap, mm2coverage, ms = compare.calc_mm2overlap(b1, b2, [0, len(codes)], min_cov=5)
    b1.close(); b2.close()
    assert sorted(mm2overlap[0]) == list(g["mm"])
    assert [mm2overlap[0][m] for m in g["mm"]] == list(g["both"])
    assert np.max(np.abs(np.array([mm2coverage[0][m] for m in g["mm"]]) - g["coverage"])) < 1e-15
    assert ms > 0


@pytest.mark.parametrize("name", ["compare_a", "compare_b", "compare_c", "compare_d"])
def test_snp_compare_vs_reference_vectors(ctx, name):
    from instrain_amd import compare, engine
    g = util.load_case(name)
    codes = engine.encode_seq(str(g["seq"]))
    b1 = _batch(ctx, codes, [0, len(codes)], g["a_pos"], g["a_base"], g["a_mm"], g["a_pair"])
    b2 = _batch(ctx, codes, [0, len(codes)], g["b_pos"], g["b_base"], g["b_mm"], g["b_pair"])
    table, mdb, ms = compare.compare_scaffolds(b1, b2, [0, len(codes)], min_cov=5, min_freq=0.05, store_mismatch_locations=True)
    # twice: the buffers of the first call are reused
    table2, mdb2, _ = compare.compare_scaffolds(b1, b2, [0, len(codes)], min_cov=5, min_freq=0.05, store_mismatch_locations=True)
    b1.close(); b2.close()
    assert [r["mm"] for r in table] == list(g["mm"])
    assert [r["compared_bases_count"] for r in table] == list(g["both"])
    assert [r["consensus_SNPs"] for r in table] == list(g["t_consensus_SNPs"])
    assert [r["population_SNPs"] for r in table] == list(g["t_population_SNPs"])
    for k in ("conANI", "popANI", "percent_genome_compared"):
        np.testing.assert_array_equal(np.array([r[k] for r in table], dtype=np.float64), g["t_" + k])
    np.testing.assert_array_equal(np.array([r["coverage_overlap"] for r in table]), g["coverage"])
    raw = mdb["raw"]
    assert list(raw["mm"]) == list(g["m_mm"]) and list(mdb["position"]) == list(g["m_position"])
    assert list(raw["consensus_snp"].astype(bool)) == list(g["m_consensus_SNP"])
    assert list(raw["population_snp"].astype(bool)) == list(g["m_population_SNP"])
    assert table2 == table and (mdb2["raw"] == raw).all()
    assert ms > 0


def test_snp_compare_multi_scaffold_vs_oracle(ctx):
    """three scaffolds (one without any SNP row), dense sample vs mm sample, against oracle/compare.py"""
    from instrain_amd import compare, engine
    from oracle import compare as ocompare, oracle
    from tests.test_gpu_parity import _random_split
    lut, fb = util.load_lut()
    seq, p1, b1_, m1, r1 = _random_split(511, 6000, 30, 1, 50)
    _, p2, b2_, m2, r2 = _random_split(512, 6000, 24, 4, 50)
    k1, k2 = ~((p1 >= 2000) & (p1 < 2600)), ~((p2 >= 2000) & (p2 < 2300))     # a gap around the middle scaffold
    p1, b1_, m1, r1 = p1[k1], b1_[k1], m1[k1], r1[k1]
    p2, b2_, m2, r2 = p2[k2], b2_[k2], m2[k2], r2[k2]
    codes = engine.encode_seq(seq)
    sb = np.array([0, 2000, 2300, 6000])
    A = _batch(ctx, codes, sb, p1, b1_, m1, r1, n_mm=1)
    B = _batch(ctx, codes, sb, p2, b2_, m2, r2, n_mm=4)
    table, mdb, _ = compare.compare_scaffolds(A, B, sb, min_cov=5, min_freq=0.05, store_mismatch_locations=True)
    A.close(); B.close()
    got_rows = sorted(zip(mdb["scaffold"].tolist(), mdb["raw"]["mm"].tolist(), mdb["position"].tolist(),
                          mdb["raw"]["consensus_snp"].astype(bool).tolist(), mdb["raw"]["population_snp"].astype(bool).tolist()))
    exp_rows, exp_table = [], []
    for i, (s, e) in enumerate(zip(sb[:-1], sb[1:])):
        def split(p, b, m, r):
            k = (p >= s) & (p < e)
            return oracle.profile_split(p[k] - s, b[k], m[k], r[k], seq[s:e], 0, lut, fb)
        ra, rb = split(p1, b1_, m1 * 0, r1), split(p2, b2_, m2, r2)
        o, c = ocompare.calc_mm2overlap(ra["entries"], rb["entries"], int(e - s), min_cov=5)
        try:
            rows = ocompare.compare_snp_tables(ra["snv"], rb["snv"], o, lut, fb, min_freq=0.05)
        except KeyError:                    # the reference's own failure mode (N reference base)
            exp_table.append((i, None, None, None, None))
            continue
        exp_rows += [(i, mm, p, cc, q) for mm, p, cc, q in rows]
        for t in ocompare.overlap_table(o, c, rows, int(e - s)):
            exp_table.append((i, t["mm"], t["compared_bases_count"], t["consensus_SNPs"], t["population_SNPs"]))
    assert got_rows == sorted(exp_rows)
    assert [(r["scaffold"], r.get("mm"), r.get("compared_bases_count"), r.get("consensus_SNPs"), r.get("population_SNPs"))
            for r in table] == exp_table
    assert len(exp_rows) > 0


def test_overlap_dense_vs_mm_and_multi_scaffold(ctx):
    """a dense (skip-mm) sample against an mm sample over three scaffolds, vs plain numpy"""
    from instrain_amd import compare, engine
    from tests.test_gpu_parity import _random_split
    seq, p1, b1_, m1, r1 = _random_split(501, 7000, 14, 1, 50)
    _, p2, b2_, m2, r2 = _random_split(502, 7000, 9, 4, 50)
    codes = engine.encode_seq(seq)
    sb = np.array([0, 1000, 1001, 7000])
    A = _batch(ctx, codes, sb, p1, b1_, m1, r1, n_mm=1)
    B = _batch(ctx, codes, sb, p2, b2_, m2, r2, n_mm=4)
    mm2overlap, mm2coverage, _ = compare.calc_mm2overlap(A, B, sb, min_cov=5)
    A.close(); B.close()
    cov1 = np.bincount(p1[b1_ < 4], minlength=7000)
    for i, (s, e) in enumerate(zip(sb[:-1], sb[1:])):
        c2 = np.zeros(7000, dtype=np.int64)
        keysA = {0} if cov1[s:e].any() else set()
        for m in range(4):
            k = (m2 == m) & (b2_ < 4)
            lvl = np.bincount(p2[k], minlength=7000)
            c2 += lvl
            present = bool(lvl[s:e].any()) or (m in keysA)
            assert (m in mm2overlap[i]) == present, (i, m)
            if not present:
                continue
            t1, t2 = cov1[s:e] >= 5, c2[s:e] >= 5
            assert mm2overlap[i][m] == int((t1 & t2).sum())
            either = int((t1 | t2).sum())
            assert abs(mm2coverage[i][m] - ((t1 & t2).sum() / either if either else 0)) < 1e-15
