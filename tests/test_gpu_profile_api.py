"""GPU tests written the way the reference's own tests read (test/tests/test_profile.py): run the
profile entry point on a BAM + FASTA, then assert on DataFrames."""
import os

import numpy as np
import pandas as pd
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sars_profile():
    import instrain_amd.profile as prof
    from tests.test_oracle_golden import read_fasta
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    splits = prof.profile_bam(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"), s2s={"MT039887.1": seq},
                              null_model=model, min_cov=5, min_freq=0.05, min_snp=20, min_read_ani=0.95)
    assert sorted(splits) == ["MT039887.1.0", "MT039887.1.1", "MT039887.1.2"]
    return splits


def test_split_objects_have_the_reference_fields(sars_profile):
    S = sars_profile["MT039887.1.1"]
    for att in ["scaffold", "split_number", "bam", "length", "raw_snp_table", "raw_linkage_table", "covT", "clonT",
                "clonTR", "min_freq", "log"]:       # profile_utilities.py:195-214
        assert hasattr(S, att), att
    assert S.length == 9959 and S.split_number == 1
    assert list(S.raw_snp_table.columns) == ['scaffold', 'position', 'ref_base', 'A', 'C', 'T', 'G', 'con_base',
                                              'var_base', 'mm', 'allele_count', 'class', 'cryptic', 'position_coverage']
    assert S.covT[0].dtype == np.int32 and S.clonT[0].dtype == np.float32      # shrink_basewise dtypes


def test_tables_equal_stored_golden(sars_profile):
    """like test_profile_16: every row of raw_snp_table / raw_linkage_table vs the stored run
    (random *_normalized columns excluded, test_profile.py:896-900)"""
    from tests.test_oracle_golden import sars_golden_tables
    gS, gL = sars_golden_tables()
    S = pd.concat([s.raw_snp_table for s in sars_profile.values()]).sort_values(["position", "mm"]).reset_index(drop=True)
    L = pd.concat([s.raw_linkage_table for s in sars_profile.values()]).sort_values(["position_A", "position_B", "mm"]).reset_index(drop=True)
    assert len(S) == len(gS) and len(L) == len(gL)
    for c in ["scaffold", "position", "ref_base", "A", "C", "T", "G", "con_base", "var_base", "mm", "allele_count",
              "cryptic", "position_coverage"]:
        assert (S[c].values == gS[c].values).all(), c
    for c in ["total", "countAB", "countAb", "countaB", "countab", "allele_A", "allele_a", "allele_B", "allele_b",
              "distance", "position_A", "position_B", "mm", "scaffold"]:
        assert (L[c].values == gL[c].values).all(), c
    for c in ["r2", "d_prime"]:
        a, b = L[c].values.astype(float), gL[c].values.astype(float)
        assert (np.isnan(a) == np.isnan(b)).all() and np.nanmax(np.abs(a - b)) <= 1e-6, c


@pytest.mark.parametrize("skip_mm", [False, True])
def test_profile_bam_through_the_staging_ring(sars_profile, skip_mm):
    """the whole seam with the slot's records staged through a small pinned ring (what a pipe does by itself for batches of
    hundreds of MB): same SplitObjects as with the whole stream pinned"""
    import instrain_amd.profile as prof
    from tests.test_oracle_golden import read_fasta
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    kw = dict(s2s={"MT039887.1": seq}, null_model=model, min_cov=5, min_freq=0.05, min_snp=20, min_read_ani=0.95,
              skip_mm_profiling=skip_mm)
    bam = os.path.join(util.GOLD, "sars_cov_2.sorted.bam")
    a = sars_profile if not skip_mm else prof.profile_bam(bam, **kw)
    b = prof.profile_bam(bam, staging_ring_kib=64, **kw)
    assert sorted(a) == sorted(b)
    for k in a:
        A, B = a[k], b[k]
        assert A.raw_snp_table.equals(B.raw_snp_table) and A.raw_linkage_table.equals(B.raw_linkage_table), k
        assert sorted(A.covT) == sorted(B.covT)
        for mm in A.covT:
            assert A.covT[mm].equals(B.covT[mm]) and A.clonT[mm].equals(B.clonT[mm]), (k, mm)


def test_covT_vs_snv_table_coverage(sars_profile):
    """the reference's test_profile_13 (test_profile.py:726-750): cumulative covT == position_coverage"""
    for S in sars_profile.values():
        for _, row in S.raw_snp_table.iterrows():
            cov = sum(int(ser.get(row["position"], 0)) for mm, ser in S.covT.items() if mm <= row["mm"])
            assert cov == row["position_coverage"], (row["position"], row["mm"])


def test_coverage_summary_equals_stored_cumulative_table(sars_profile):
    """per-mm mean coverage / breadth recomputed from covT == the stored cumulative_scaffold_table"""
    g = pd.read_csv(os.path.join(util.GOLD, "sars_cov_2_cumulative_scaffold_table.csv.gz"))
    L = 29879
    dense = {}
    for S in sars_profile.values():
        for mm, ser in S.covT.items():
            dense.setdefault(mm, np.zeros(L, dtype=np.int64))[ser.index.values] += ser.values
    cum = np.zeros(L, dtype=np.int64)
    for mm in sorted(dense):
        cum = cum + dense[mm]
        row = g[g["mm"] == mm].iloc[0]
        assert abs(cum.mean() - row["coverage"]) < 1e-9
        assert abs((cum > 0).sum() / L - row["breadth"]) < 1e-12
        assert int(np.median(cum)) == int(row["median_cov"])


def test_missing_scaffold_follows_failure_convention():
    """like test_profile_17 / profile_utilities.py:104-111, 154-156: a scaffold that cannot be profiled is logged with
    the reference's SplitException line and dropped"""
    import instrain_amd.profile as prof
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    logs = []
    fdb = pd.DataFrame({"scaffold": ["other"], "split_number": [0], "start": [0], "end": [3]})
    out = prof.profile_bam(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"), fdb, None, None, s2s={"other": "ACGT"},
                           null_model=model, logs=logs)
    assert out == {} and len(logs) == 1 and "FAILURE SplitException other 0" in logs[0]


def _messy(tmp_path, seed=9, n_pairs=4000):
    from tests import bamwriter
    refs = [("scafA", 2500), ("scafB", 700), ("scafC", 3100), ("scafD", 1500)]
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = {n: "".join(rng.choice(list("ACGT"), ln)) for n, ln in refs}
    path = str(tmp_path / "messy.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(seed, refs, n_pairs))
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    return refs, seqs, path, model


def _same_split(a, b):
    for att in ("scaffold", "split_number", "length"):
        assert getattr(a, att) == getattr(b, att)
    for att in ("raw_snp_table", "raw_linkage_table"):
        x, y = getattr(a, att), getattr(b, att)
        assert len(x) == len(y)
        if len(x):
            cols = [c for c in x.columns if not c.endswith("_normalized")]
            pd.testing.assert_frame_equal(x[cols].reset_index(drop=True), y[cols].reset_index(drop=True))
    for att in ("covT", "clonT"):
        x, y = getattr(a, att), getattr(b, att)
        assert sorted(x) == sorted(y), att
        for m in x:
            pd.testing.assert_series_equal(x[m], y[m])


def test_profile_bam_honours_fasta_db_and_per_scaffold_failures(tmp_path):
    """fasta_db picks the scaffolds and their splits (profile_controller.py:415-433); a scaffold without a usable sequence
    is dropped with its SplitException lines while the others are profiled (profile_utilities.py:100-111); small batch
    budgets cut the run into several device batches with identical results"""
    import instrain_amd.profile as prof
    refs, seqs, path, model = _messy(tmp_path)
    kw = dict(null_model=model, min_cov=5, min_freq=0.05, min_snp=10, min_read_ani=0.9, window_length=1000)
    full = prof.profile_bam(path, s2s=seqs, **kw)
    assert len(full) == sum(ln // 1000 + 1 for _, ln in refs)
    # (a) subset + custom splits of scafC (two uneven splits instead of iterate_splits' four)
    fdb = pd.DataFrame({"scaffold": ["scafA"] * 3 + ["scafC"] * 2, "split_number": [0, 1, 2, 0, 1],
                        "start": [0, 833, 1666, 0, 1000], "end": [832, 1665, 2499, 999, 3099]})
    sub = prof.profile_bam(path, fdb, None, None, s2s=seqs, **kw)
    assert sorted(sub) == ["scafA.0", "scafA.1", "scafA.2", "scafC.0", "scafC.1"]
    for k in ("scafA.0", "scafA.1", "scafA.2"):
        _same_split(sub[k], full[k])
    assert sub["scafC.1"].length == 2100
    n_full = sum(len(full["scafC.%d" % i].raw_snp_table) for i in range(4))
    assert len(sub["scafC.0"].raw_snp_table) + len(sub["scafC.1"].raw_snp_table) == n_full        # SNV rows do not depend on the cut
    cov_full = sum(int(s.sum()) for i in range(4) for s in full["scafC.%d" % i].covT.values())
    assert sum(int(s.sum()) for i in range(2) for s in sub["scafC.%d" % i].covT.values()) == cov_full
    # (b) scafB has no sequence, scafD's differs in length: both dropped with their lines, A and C unaffected
    logs = []
    bad = dict(seqs)
    del bad["scafB"]
    bad["scafD"] = bad["scafD"][:-1]
    fdb_all = pd.DataFrame([(n, i, s, e) for n, ln in refs for i, (s, e) in enumerate(prof.profile_utilities.iterate_splits(ln, 1000))],
                           columns=["scaffold", "split_number", "start", "end"])
    part = prof.profile_bam(path, fdb_all, None, None, s2s=bad, logs=logs, **kw)
    assert sorted(part) == sorted(k for k in full if k.startswith(("scafA", "scafC")))
    assert len(logs) == 1 + 2 and sum("scafB" in l for l in logs) == 1 and sum("scafD" in l for l in logs) == 2
    for k in part:
        _same_split(part[k], full[k])
    # (c) tiny budgets: one scaffold per batch, the pipe grows as needed
    many = prof.profile_bam(path, s2s=seqs, batch_positions=3000, batch_observations=10_000, **kw)
    assert sorted(many) == sorted(full)
    for k in full:
        _same_split(many[k], full[k])


def test_profile_bam_honours_the_controllers_r2m(tmp_path):
    """sR2M as the controller hands it over (controller.py:274-281): the built-in filter's own R2M gives the same
    profile; a reduced / re-levelled R2M gives exactly the profile of those read pairs; sets mean skip_mm_profiling"""
    import instrain_amd.profile as prof
    from instrain_amd import engine
    from oracle import bam_py, oracle
    from tests.test_oracle_golden import iterate_splits
    refs, seqs, path, model = _messy(tmp_path, seed=12)
    lut, fb = util.load_lut()
    kw = dict(null_model=model, min_cov=5, min_freq=0.05, min_snp=10, window_length=1000)
    full = prof.profile_bam(path, s2s=seqs, min_read_ani=0.9, **kw)
    bam = engine.BamFile(path)
    bam.scan(); bam.filter(min_read_ani=0.9)
    r2m = {name: bam.r2m(t) for t, (name, _, _) in enumerate(bam.refs())}
    bam.close()
    same = prof.profile_bam(path, None, r2m, None, s2s=seqs, **kw)
    assert sorted(same) == sorted(full)
    for k in full:
        _same_split(same[k], full[k])
    # half of scafC's pairs, every mm + 1, nothing on the other scaffolds -> oracle on exactly those pairs
    keep = {n: m + 1 for n, m in list(r2m["scafC"].items())[::2]}
    fdb = pd.DataFrame([("scafC", i, s, e) for i, (s, e) in enumerate(iterate_splits(3100, 1000))],
                       columns=["scaffold", "split_number", "start", "end"])
    got = prof.profile_bam(path, fdb, {"scafC": keep}, None, s2s=seqs, **kw)
    rrefs, rr = bam_py.read_bam(path)
    bam_py.resolve_overlaps(rr, 2)
    pos, base, mm, pr, _ = bam_py.expand_observations(rr, 2, keep, ref_len=3100)
    n_rows = 0
    for i, (s, e) in enumerate(iterate_splits(3100, 1000)):
        exp = oracle.profile_split(pos, base, mm, pr, seqs["scafC"][s:e + 1], s, lut, fb, min_cov=5, min_freq=0.05, min_snp=10)
        S = got["scafC.%d" % i]
        o = np.lexsort((exp["snv"]["mm"], exp["snv"]["pos"]))
        es = exp["snv"][o]
        g = S.raw_snp_table.sort_values(["position", "mm"]) if len(S.raw_snp_table) else S.raw_snp_table
        assert len(g) == len(es)
        if len(es):
            assert (g["position"].values == es["pos"]).all() and (g["mm"].values == es["mm"]).all()
            assert (g[["A", "C", "T", "G"]].values == es["cnt"]).all()
        lv = exp["entries"]["cnt"].sum(axis=1)
        for m in set(int(x) for x in exp["entries"]["mm"]):
            k = (exp["entries"]["mm"] == m) & (lv > 0)
            ser = S.covT[m].sort_index()
            assert (ser.index.values == np.sort(exp["entries"]["pos"][k])).all()
        assert min(S.covT) >= 1                                # every level was raised by one
        n_rows += len(es)
    assert n_rows > 10
    # a set of names = --skip_mm_profiling
    got = prof.profile_bam(path, fdb, {"scafC": set(keep)}, None, s2s=seqs, skip_mm_profiling=True, **kw)
    assert all(list(S.covT) == [0] for S in got.values())
    tot = sum(int(S.covT[0].sum()) for S in got.values())
    assert tot == int((base < 4).sum()) or tot == len(pos) - int((base >= 4).sum())


def test_store_everything_keeps_the_count_table(tmp_path):
    import instrain_amd.profile as prof
    refs, seqs, path, model = _messy(tmp_path, seed=13, n_pairs=1500)
    S = prof.profile_bam(path, s2s=seqs, null_model=model, min_read_ani=0.9, window_length=1000, skip_mm_profiling=True,
                         store_everything=True)["scafA.1"]
    pc = S.pileup_counts                                       # profile_utilities.py:205-211
    assert pc.shape == (S.length, 4)
    cov = pc.sum(axis=1)
    ser = S.covT[0]
    assert (cov[ser.index.values - 833] == ser.values).all() and (cov > 0).sum() == len(ser)


def test_device_coverage_table_equals_stored_golden():
    """make_coverage_table with the device aggregates (isx_batch_summarize) vs all 26 rows of the
    reference's stored cumulative_scaffold_table"""
    import instrain_amd.profile as prof
    from tests.test_oracle_golden import check_coverage_table_vs_sars_golden, read_fasta
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    tabs = {}
    prof.profile_bam(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"), s2s={"MT039887.1": seq}, null_model=model,
                     min_cov=5, min_freq=0.05, min_snp=20, scaffold_tables=tabs)
    t = tabs["MT039887.1"]
    assert list(t["mm"]) == list(range(26))
    check_coverage_table_vs_sars_golden(t.to_dict("records"), float_tol=1e-9)


@pytest.mark.parametrize("skip_mm", [True, False])
def test_device_summary_big_and_small_scaffolds(skip_mm):
    """a 200 kbp scaffold (sorted by the device-wide radix sort) next to small ones (segmented sort):
    medians / sums per (scaffold, mm) vs plain numpy on the fetched tables"""
    from instrain_amd import engine, synth
    lut, fb = util.load_lut()
    ctx = engine.Context(0)
    ctx.set_null_model(lut, fb)
    w = synth.make_workload(genome_len=300_000, coverage=14, n_sites=600, seed=21, skip_mm=skip_mm, err=0.004)
    M = w["n_mm_bins"]
    sb = np.array([0, 200_000, 200_500, 300_000])
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=M, rarefied_coverage=12, seed=4)
    b.run()
    res = b.fetch()
    lv, _ = b.summarize(sb)
    b.close()
    ctx.close()
    n_pos = 300_000
    if M == 1:
        ent = engine.dense_to_entries(res["counts"], res["clon"])
        ent["clon_rarefied"] = res["clon_r"][ent["gpos"]]
    else:
        ent = res["entries"]
    cov = np.zeros(n_pos, dtype=np.int64)
    cv = np.full(n_pos, np.nan); cr = np.full(n_pos, np.nan)
    assert M >= (1 if skip_mm else 3)
    for m in range(M):
        e = ent[ent["mm"] == m]
        cov[e["gpos"]] += e["cnt"].sum(axis=1).astype(np.int64)
        k = ~np.isnan(e["clon"]); cv[e["gpos"][k]] = e["clon"][k].astype(np.float64)
        k = ~np.isnan(e["clon_rarefied"]); cr[e["gpos"][k]] = e["clon_rarefied"][k].astype(np.float64)
        for i, (s0, s1) in enumerate(zip(sb[:-1], sb[1:])):
            d = lv[i, m]
            c = cov[s0:s1]
            assert d["nonzero"] == np.count_nonzero(c) and d["sum_cov"] == c.sum() and d["sumsq_cov"] == (c * c).sum()
            assert d["median_cov"] == np.median(c), (i, m)
            for vals, cnt, med, sm in ((cv[s0:s1], "counted", "median_clon", "sum_clon"),
                                       (cr[s0:s1], "counted_rarefied", "median_clon_rarefied", "sum_clon_rarefied")):
                v = vals[~np.isnan(vals)]
                assert d[cnt] == len(v), (i, m, cnt)
                if len(v):
                    assert d[med] == np.median(v.astype(np.float32)).astype(np.float64) or abs(d[med] - np.median(v)) < 1e-7, (i, m, med)
                    assert abs(d[sm] - v.sum()) < 1e-6 * max(1.0, len(v))


@pytest.mark.parametrize("mm_levels", [1, 5])
def test_device_summary_vs_oracle_multi_scaffold(mm_levels):
    """three scaffolds of unequal length in one batch: every aggregate vs oracle/summary.py"""
    from instrain_amd import engine
    from oracle import oracle, summary
    from tests.test_gpu_parity import _random_split
    lut, fb = util.load_lut()
    ctx = engine.Context(0)
    ctx.set_null_model(lut, fb)
    seq, pos, base, mm, pair = _random_split(400 + mm_levels, 9000, 45, mm_levels, 200)
    sb = np.array([0, 2500, 2700, 9000])
    b = engine.Batch(ctx, engine.encode_seq(seq), sb, engine.pack_obs(pos.astype(np.uint32), base, mm), pair.astype(np.uint32),
                     n_mm_bins=mm_levels, rarefied_coverage=30, seed=9)
    b.run()
    res = b.fetch()
    lv, ms = b.summarize(sb)
    b.close()
    ctx.close()
    assert ms > 0
    ent_all = res["entries"] if mm_levels > 1 else engine.dense_to_entries(res["counts"], res["clon"])
    clon_r_all = res["clon_r"] if mm_levels > 1 else res["clon_r"][ent_all["gpos"]]
    for i, (s, e) in enumerate(zip(sb[:-1], sb[1:])):
        r = oracle.profile_split(pos, base, mm, pair, seq[s:e], int(s), lut, fb)
        k = (ent_all["gpos"] >= s) & (ent_all["gpos"] < e)
        cr = np.full(len(r["entries"]), np.nan, dtype=np.float32)
        # the rarefied values are the product's own (random in the reference): align them by (pos, mm)
        key = lambda p, m: p.astype(np.int64) * 1000 + m
        idx = {kk: j for j, kk in enumerate(key(r["entries"]["pos"], r["entries"]["mm"]))}
        for kk, v in zip(key(ent_all["gpos"][k], ent_all["mm"][k]), clon_r_all[k]):
            cr[idx[kk]] = v
        e_rel = r["entries"].copy()
        e_rel["pos"] -= int(s)
        s_rel = r["snv"].copy()
        s_rel["pos"] -= int(s)
        rows = {row["mm"]: row for row in summary.coverage_table(e_rel, s_rel, int(e - s), clon_r=cr)}
        L = float(e - s)
        for m in range(mm_levels):
            d = lv[i, m]
            assert bool(d["present"]) == (m in rows), (i, m)
            if m not in rows:
                continue
            o = rows[m]
            assert d["nonzero"] == round(o["breadth"] * L) and d["counted"] == round(o["breadth_minCov"] * L)
            assert abs(d["sum_cov"] / L - o["coverage"]) < 1e-9 and int(d["median_cov"]) == o["coverage_median"]
            var = d["sumsq_cov"] / L - (d["sum_cov"] / L) ** 2
            assert abs(np.sqrt(var) - o["coverage_std"]) < 1e-8
            if d["counted"]:
                assert abs((1 - d["sum_clon"] / d["counted"]) - o["nucl_diversity"]) < 1e-9
                assert abs((1 - d["median_clon"]) - o["nucl_diversity_median"]) < 1e-9
            assert d["counted_rarefied"] == round(o["breadth_rarefied"] * L)
            if d["counted_rarefied"]:
                assert abs((1 - d["sum_clon_rarefied"] / d["counted_rarefied"]) - o["nucl_diversity_rarefied"]) < 1e-9
                assert abs((1 - d["median_clon_rarefied"]) - o["nucl_diversity_rarefied_median"]) < 1e-9


def test_snv_pooling_repileup_counts():
    """extract_SNVS_from_bam (polymorpher.py:275-316): counts over all mm levels at a position list
    == the oracle's level counts summed over mm, zeros for uncovered positions"""
    from instrain_amd.profile import polymorpher
    from oracle import oracle
    from tests.test_oracle_golden import iterate_splits, read_fasta, sars_golden_tables
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    gS, _ = sars_golden_tables()
    positions = sorted(set(int(p) for p in gS["position"])) + [0, 29878]
    got = polymorpher.extract_SNVS_from_bam(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"), None, positions,
                                            "MT039887.1", null_model=model)
    z = np.load(os.path.join(util.GOLD, "sars_cov_2_obs.npz"))
    exp = np.zeros((29879, 4), dtype=np.int64)
    k = z["base"] < 4
    np.add.at(exp, (z["pos"][k], z["base"][k]), 1)
    assert set(got) == set(positions)
    for p in positions:
        assert (got[p] == exp[p]).all(), p
    # the highest-mm golden row of a position can never exceed the pooled counts
    top = gS.sort_values("mm").drop_duplicates("position", keep="last")
    for _, r in top.iterrows():
        assert (np.array([r["A"], r["C"], r["T"], r["G"]]) <= got[int(r["position"])]).all()


def test_end_to_end_messy_bam_vs_oracle(tmp_path):
    """BAM with indels / clips / overlapping mates / three scaffolds -> profile_bam (C++ front end + HIP
    kernels) == oracle (Python front end + C oracle), split by split"""
    import instrain_amd.profile as prof
    from oracle import bam_py, oracle
    from tests import bamwriter, prod
    from tests.test_oracle_golden import iterate_splits
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    refs = [("scafA", 2500), ("scafB", 700), ("scafC", 3100)]
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = {n: "".join(rng.choice(list("ACGT"), ln)) for n, ln in refs}
    path = str(tmp_path / "messy.bam")
    reads = bamwriter.random_reads(9, refs, 4000)
    bamwriter.write_bam(path, refs, reads)
    splits = prof.profile_bam(path, s2s=seqs, null_model=model, min_cov=5, min_freq=0.05, min_snp=10, min_read_ani=0.9,
                              window_length=1000)
    rrefs, rr = bam_py.read_bam(path)
    p2i = {r[0]: bam_py.get_paired_reads(rr, t) for t, r in enumerate(refs)}
    r2m, _ = bam_py.filter_pairs(p2i, min_read_ani=0.9)
    n_rows = n_ld = 0
    for t, (name, ln) in enumerate(refs):
        bam_py.resolve_overlaps(rr, t)
        pos, base, mm, pr, _ = bam_py.expand_observations(rr, t, r2m[name])
        for i, (s, e) in enumerate(iterate_splits(ln, 1000)):
            exp = oracle.profile_split(pos, base, mm, pr, seqs[name][s:e + 1], s, lut, fb, min_cov=5, min_freq=0.05, min_snp=10)
            S = splits["%s.%d" % (name, i)]
            assert S.length == e - s + 1
            got_snv = S.raw_snp_table
            o = np.lexsort((exp["snv"]["mm"], exp["snv"]["pos"]))
            es = exp["snv"][o]
            assert len(got_snv) == len(es)
            if len(es):
                g = got_snv.sort_values(["position", "mm"])
                assert (g["position"].values == es["pos"]).all() and (g["mm"].values == es["mm"]).all()
                for k, b in enumerate("ACTG"):
                    assert (g[b].values == es["cnt"][:, k]).all()
                assert (g["con_base"].values == util.BASES[es["con_base"]]).all()
                assert (g["class"].values == util.CLASSES[es["cls"]]).all()
                assert (g["cryptic"].values == es["cryptic"].astype(bool)).all()
            L = S.raw_linkage_table
            o = np.lexsort((exp["ld"]["mm"], exp["ld"]["pos_b"], exp["ld"]["pos_a"]))
            el = exp["ld"][o]
            assert len(L) == len(el)
            if len(el):
                L = L.sort_values(["position_A", "position_B", "mm"])
                assert (L["position_A"].values == el["pos_a"]).all() and (L["total"].values == el["total"]).all()
                assert (L["countAB"].values == el["cAB"]).all() and (L["countab"].values == el["cab"]).all()
                a, b = L["r2"].values.astype(float), el["r2"]
                assert (np.isnan(a) == np.isnan(b)).all() and (np.nanmax(np.abs(a - b)) if (~np.isnan(a)).any() else 0) <= 1e-6
            # covT per mm
            lv = exp["entries"]["cnt"].sum(axis=1)
            for m in set(int(x) for x in exp["entries"]["mm"][lv > 0]):
                k = (exp["entries"]["mm"] == m) & (lv > 0)
                ser = S.covT[m].sort_index()
                assert (ser.index.values == np.sort(exp["entries"]["pos"][k])).all()
                assert (ser.values == lv[k][np.argsort(exp["entries"]["pos"][k])]).all()
            n_rows += len(es); n_ld += len(el)
    assert n_rows > 50


SHARDED_WORKER = '''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch
torch.cuda.set_device(0)
import torch.distributed as dist
from instrain_amd import dist as idist
from tests import util
rank, local, world = idist.init_from_env(backend="gloo")
path, out = sys.argv[1], sys.argv[2]
z = np.load(sys.argv[3], allow_pickle=True)
seqs = {str(k): str(v) for k, v in zip(z["names"], z["seqs"])}
lut, fb = util.load_lut()
model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
model[-1] = fb
kw = dict(min_cov=5, min_freq=0.05, min_snp=10, min_read_ani=0.9, window_length=1000, device=0)
if len(sys.argv) > 4:
    kw["pairing_filter"] = sys.argv[4]
splits, tables, load = idist.profile_bam_sharded(path, seqs, model, rank, world, **kw)
loads = [None] * world
dist.all_gather_object(loads, (load, len(splits)))
if rank == 0:
    np.savez(out, snv=tables["snv"], ld=tables["ld"], summary=tables["summary"], loads=np.array([l[0] for l in loads]),
             n_splits=np.array([l[1] for l in loads]))
dist.barrier()
dist.destroy_process_group()
'''


def test_one_bam_profiled_by_two_ranks_equals_one_rank(tmp_path):
    """scaffolds of one BAM LPT-sharded over two ranks (both on GPU 0, gloo rendezvous): the SNV / linkage / summary
    tables gathered on rank 0 equal the single-process profile"""
    import subprocess
    import sys
    import instrain_amd.profile as prof
    from instrain_amd import engine
    from tests import bamwriter
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    refs = [("s%d" % i, ln) for i, ln in enumerate([4000, 900, 12500, 700, 2600, 5100])]
    rng = np.random.Generator(np.random.PCG64(8))
    seqs = {n: "".join(rng.choice(list("ACGT"), ln)) for n, ln in refs}
    path = str(tmp_path / "shard.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(43, refs[:5], 7000))
    np.savez(str(tmp_path / "seqs.npz"), names=np.array(list(seqs)), seqs=np.array(list(seqs.values())))
    script = tmp_path / "w.py"
    script.write_text(SHARDED_WORKER % repo)
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
    r = util.run_group([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29578", str(script), path, out, str(tmp_path / "seqs.npz")],
                       env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    g = np.load(out)
    assert (g["n_splits"] > 0).all() and g["n_splits"].sum() == sum(ln // 1000 + 1 for _, ln in refs)
    assert g["loads"].max() / g["loads"].mean() < 1.6                      # LPT balance on the reference's cost estimate
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    one = prof.profile_bam(path, s2s=seqs, null_model=model, min_cov=5, min_freq=0.05, min_snp=10, min_read_ani=0.9, window_length=1000)
    tid = {n: i for i, (n, _) in enumerate(refs)}
    rows = []
    lrows = []
    for S in one.values():
        t = S.raw_snp_table
        for r_ in t.itertuples():
            rows.append((tid[r_.scaffold], r_.position, r_.mm, r_.A, r_.C, r_.T, r_.G, r_.allele_count))
        for r_ in S.raw_linkage_table.itertuples():
            lrows.append((tid[r_.scaffold], r_.position_A, r_.position_B, r_.mm, r_.total, r_.countAB))
    got = sorted((int(x["tid"]), int(x["gpos"]), int(x["mm"]), int(x["cnt"][0]), int(x["cnt"][1]), int(x["cnt"][2]), int(x["cnt"][3]),
                  int(x["allele_count"])) for x in g["snv"])
    assert got == sorted(rows) and len(rows) > 100
    gotl = sorted((int(x["tid"]), int(x["gpos_a"]), int(x["gpos_b"]), int(x["mm"]), int(x["total"]), int(x["countAB"])) for x in g["ld"])
    assert gotl == sorted(lrows)
    assert sorted(set(int(t) for t in g["summary"]["tid"])) == [0, 1, 2, 3, 4]


RCCL_WORKER = '''
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from instrain_amd import dist as idist
from instrain_amd._lib import SNV_DT
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)          # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
a = np.arange(100003, dtype=np.int64) * 7
got = idist.all_gather_concat(a)                                       # lengths + padded payload through ncclAllGather on device tensors
assert got.dtype == a.dtype and (got == a).all()
assert len(idist.all_gather_concat(np.zeros(0, np.uint8))) == 0
snv = np.zeros(1234, dtype=SNV_DT); snv["gpos"] = np.arange(1234)
out = idist.gather_tables({"snv": snv, "ld": np.zeros(0, np.int32)}, dst=0)
assert (out["snv"]["gpos"] == snv["gpos"]).all() and len(out["ld"]) == 0
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_collectives_over_rccl_world_of_one(tmp_path):
    """the exchange steps of the multi-GPU path through the RCCL backend itself (one GPU here, so a world of one: RCCL refuses two
    ranks on one device; ISX_DIST_FORCE makes the world of one take the collective route instead of the short cut)"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % repo)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29585", ISX_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


@pytest.mark.parametrize("mode", ["all_reads", "non_discordant"])
def test_sharded_profile_with_cross_scaffold_filters(tmp_path, mode):
    """non_discordant / all_reads with every rank scanning only its share (dist.resolve_cross_names: read-name hashes
    all-gathered, repeated names resolved): the gathered SNV / linkage tables equal the single-process profile of the file"""
    import subprocess
    import sys
    import instrain_amd.profile as prof
    from tests import bamwriter
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "cross.bam")
    refs = bamwriter.cross_scaffold_bam(path, 91, triple=(mode == "all_reads"))
    rng = np.random.Generator(np.random.PCG64(9))
    seqs = {n: "".join(rng.choice(list("ACGT"), ln)) for n, ln in refs}
    np.savez(str(tmp_path / "seqs.npz"), names=np.array(list(seqs)), seqs=np.array(list(seqs.values())))
    script = tmp_path / "w.py"
    script.write_text(SHARDED_WORKER % repo)
    out = str(tmp_path / "gathered.npz")
    port = "29581" if mode == "all_reads" else "29582"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = util.run_group([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", port, str(script), path, out, str(tmp_path / "seqs.npz"), mode],
                       env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    g = np.load(out)
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    one = prof.profile_bam(path, s2s=seqs, null_model=model, min_cov=5, min_freq=0.05, min_snp=10, min_read_ani=0.9, window_length=1000,
                           pairing_filter=mode)
    tid = {n: i for i, (n, _) in enumerate(refs)}
    rows, lrows = [], []
    for S in one.values():
        for r_ in S.raw_snp_table.itertuples():
            rows.append((tid[r_.scaffold], r_.position, r_.mm, r_.A, r_.C, r_.T, r_.G, r_.allele_count))
        for r_ in S.raw_linkage_table.itertuples():
            lrows.append((tid[r_.scaffold], r_.position_A, r_.position_B, r_.mm, r_.total, r_.countAB))
    got = sorted((int(x["tid"]), int(x["gpos"]), int(x["mm"]), int(x["cnt"][0]), int(x["cnt"][1]), int(x["cnt"][2]), int(x["cnt"][3]),
                  int(x["allele_count"])) for x in g["snv"])
    assert got == sorted(rows) and len(rows) > 100
    gotl = sorted((int(x["tid"]), int(x["gpos_a"]), int(x["gpos_b"]), int(x["mm"]), int(x["total"]), int(x["countAB"])) for x in g["ld"])
    assert gotl == sorted(lrows)


@pytest.mark.parametrize("mm_levels", [1, 4])
def test_genome_level_rollup_vs_oracle(mm_levels):
    """isx_batch_summarize_genomes + profile.genome_utilities against oracle/summary.genome_coverage_rows (itself pinned
    by the reference's genomeLevel_coverage_info, tests/test_oracle_golden.py): three genomes of consecutive scaffolds, one
    scaffold shorter than the 2 x 100 masked positions, one without reads, a genome of 300 kbp (device-wide sort path)"""
    from instrain_amd import engine, synth
    from instrain_amd.profile import genome_utilities as gu
    from oracle import summary
    lens = [5200, 150, 2600, 300_000, 900, 1301, 7000]
    names = ["s%d" % i for i in range(len(lens))]
    genome_first = [0, 3, 5, 7]                       # g0 = s0..s2, g1 = s3..s4, g2 = s5..s6
    genomes = ["g0", "g1", "g2"]
    sb = np.r_[0, np.cumsum(lens)].astype(np.int64)
    n_pos = int(sb[-1])
    rng = np.random.Generator(np.random.PCG64(7 + mm_levels))
    # observations: coverage ~ 12 on every scaffold but s4 (no reads); mm levels random
    pos = []
    for i, ln in enumerate(lens):
        if i == 4:
            continue
        pos.append(sb[i] + rng.integers(0, ln, size=12 * ln))
    pos = np.sort(np.concatenate(pos)).astype(np.uint32)
    base = rng.integers(0, 4, len(pos)).astype(np.uint8)
    mm = rng.integers(0, mm_levels, len(pos))
    ctx = engine.Context(0)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    ref = rng.integers(0, 4, n_pos).astype(np.uint8)
    b = engine.Batch(ctx, ref, sb, engine.pack_obs(pos, base, mm), None, n_mm_bins=mm_levels, enable_linkage=False)
    b.run()
    lv, ms = b.summarize_genomes(sb, genome_first, mask_edges=100)
    got = gu.genome_level_rows(lv, genomes, mms=[0, 1, 3, 9])
    b.close(); ctx.close()
    # the same from the observations with the oracle's restatement
    covT = {}
    for i, nme in enumerate(names):
        k = (pos >= sb[i]) & (pos < sb[i + 1])
        covT[nme] = {}
        for m in range(mm_levels):
            p, c = np.unique(pos[k & (mm == m)] - sb[i], return_counts=True)
            covT[nme][m] = (p, c)
    g2s = {g: names[genome_first[j]:genome_first[j + 1]] for j, g in enumerate(genomes)}
    exp = summary.genome_coverage_rows(covT, dict(zip(names, lens)), g2s, [0, 1, 3, 9], mask_edges=100)
    assert len(got) == len(exp) == 12
    for (_, r), e in zip(got.iterrows(), exp):
        assert r["mm"] == e["mm"] and r["genome"] == e["genome"] and r["coverage_median"] == e["coverage_median"], (dict(r), e)
        for k in ("coverage_SEM", "coverage_std"):
            assert (np.isnan(r[k]) and np.isnan(e[k])) or abs(r[k] - e[k]) <= 1e-9 * max(1.0, abs(e[k])), (k, dict(r), e)
    assert lv["n"][0, 0] == (5200 - 200) + 0 + (2600 - 200) and lv["n"][1, 0] == (300_000 - 200) + (900 - 200)


EXTRA_CASES = ["synth_mm4", "synth_m1", "synth_skipmm", "synth_selfpairs", "synth_offset", "synth_ambig"]


@pytest.mark.parametrize("reads", [None, "reassembled"])
@pytest.mark.parametrize("name", EXTRA_CASES)
def test_store_everything_extras_equal_the_references(name, reads):
    """read_to_snvs (update_linked_reads, linkage.py:254-283) and mm_to_position_graph (calc_mm_SNV_linkage_network, :14-44) rebuilt
    from the device's allele observations == the objects the reference itself built on the same split
    (tests/golden/linkage_extras.npz, written by make_golden.py from the imported reference): every read's list entry for entry,
    every edge's mm / allele-combination count"""
    from instrain_amd import engine
    from instrain_amd.profile import linkage
    g = util.load_case(name)
    gold = np.load(os.path.join(util.GOLD, "linkage_extras.npz"))
    seq, start = str(g["seq"]), int(g["start"])
    pos = g["pos"].astype(np.int64)
    sel = (pos >= start) & (pos < start + len(seq))
    mm = g["mm"][sel]
    obs = engine.pack_obs((pos[sel] - start).astype(np.uint32), g["base"][sel], mm)
    pair = g["pair"][sel].astype(np.uint32)
    ctx = engine.Context(0)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    src, pr = (obs, pair) if reads is None else (util.reassemble_segs(obs["gpos"], obs["base"], obs["mm"], pair), None)
    b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], src, pr, n_mm_bins=int(mm.max()) + 1 if len(mm) else 1,
                     min_cov=int(g["p_min_cov"]), min_freq=float(g["p_min_freq"]), min_snp=int(g["p_min_snp"]))
    b.run()
    ao = b.fetch_allele_obs()
    assert len(ao) == b.sizes()["n_allele_obs"]
    n_edges = b.sizes()["n_edges"]
    b.close()
    ctx.close()
    rts = linkage.read_to_snvs_of_split(ao, 0, len(seq))
    G = linkage.calc_mm_SNV_linkage_network(rts)
    got_rts, got_gr = linkage.flatten(rts, G)
    assert (got_rts == gold[name + "_rts"]).all() and got_rts.shape == gold[name + "_rts"].shape
    assert (got_gr == gold[name + "_graph"]).all() and got_gr.shape == gold[name + "_graph"].shape
    assert G.number_of_edges() == n_edges == int(g["n_edges"])


def test_store_everything_on_a_bam_names_the_reads(tmp_path):
    """profile_bam(store_everything=True): SplitObjects carry read_to_snvs keyed by READ NAME, mm_to_position_graph and
    pileup_counts (profile_utilities.py:205-211), and their log is the reference's WorkerLog pair (:133-134, 212-214)"""
    import instrain_amd.profile as prof
    from instrain_amd import engine
    from tests.test_oracle_golden import read_fasta
    lut, fb = util.load_lut()
    model = {int(i): int(v) for i, v in enumerate(lut) if v >= 0}
    model[-1] = fb
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    bam = os.path.join(util.GOLD, "sars_cov_2.sorted.bam")
    st = {}
    splits = prof.profile_bam(bam, s2s={"MT039887.1": seq}, null_model=model, min_cov=5, min_freq=0.05, min_snp=20,
                              min_read_ani=0.95, store_everything=True, stats=st)
    bf = engine.BamFile(bam)
    bf.scan()
    bf.filter(min_read_ani=0.95)
    r2m = bf.r2m(0)
    bf.close()
    n_entries = n_edges = 0
    for key, S in splits.items():
        rts, G = S.read_to_snvs, S.mm_to_position_graph
        assert S.pileup_counts.shape == (S.length, 4)
        for mm, reads in rts.items():
            for name, lst in reads.items():
                assert r2m[name] == mm                              # the key is the pair's name, its level the filter's mm
                p = [int(e.split(":")[0]) for e in lst]
                assert p == sorted(p) and 0 <= p[0] and p[-1] < S.length
                n_entries += len(lst)
        n_edges += G.number_of_edges()
        words = S.log.split()
        assert words[0] == "WorkerLog" and words[1] == "SplitProfile" and words[2] == key and words[3] == "start"
        assert words[7] == "WorkerLog" and words[10] == "end" and float(words[12]) >= float(words[5]) and int(words[13]) == os.getpid()
    assert n_entries > 1000
    assert n_edges == 963                                           # the stored run's linkage network


def _same_split_mm(a, b, to_a):
    """_same_split with b's mm levels renamed by to_a (b was profiled with order-preserving stand-ins for a's mm values)"""
    for att in ("scaffold", "split_number", "length"):
        assert getattr(a, att) == getattr(b, att)
    for att in ("raw_snp_table", "raw_linkage_table"):
        x, y = getattr(a, att), getattr(b, att)
        assert len(x) == len(y)
        if len(x):
            y = y.copy()
            y["mm"] = [to_a[int(m)] for m in y["mm"]]
            cols = [c for c in x.columns if not c.endswith("_normalized")]
            pd.testing.assert_frame_equal(x[cols].reset_index(drop=True), y[cols].reset_index(drop=True))
    for att in ("covT", "clonT"):
        x, y = getattr(a, att), getattr(b, att)
        assert sorted(x) == sorted(to_a[int(m)] for m in y), att
        for m in y:
            pd.testing.assert_series_equal(x[to_a[int(m)]], y[m])


def test_pairs_beyond_128_mm_levels_are_binned_exactly(tmp_path, caplog):
    """a controller's R2M with pairs of 200..249 mismatches (long reads, a low --min_read_ani).  The reference bins any mm
    (profile_utilities.py:268-286) and its tables depend on the ORDER of the levels alone (:297-312): the pairs travel with the rank of
    their mm among the values that occur (isx_bam_set_mm_levels) and the tables come back under the real values -- the same profile as an
    R2M whose values are small order-preserving stand-ins, level for level, with no warning and under strict=True.  More than 128
    DIFFERENT values: the pairs beyond the 128th are counted at it, with a warning, every SplitObject says so (mm_clamped = that value) --
    the same profile as an R2M that says that value -- and strict=True refuses (ValueError)"""
    import logging
    import instrain_amd.profile as prof
    from instrain_amd import engine
    refs, seqs, path, model = _messy(tmp_path, seed=14, n_pairs=2500)
    kw = dict(null_model=model, min_cov=5, min_freq=0.05, min_snp=10, window_length=1000, s2s=seqs)
    bam = engine.BamFile(path)
    bam.scan(); bam.filter(min_read_ani=0.9)
    r2m = {name: bam.r2m(t) for t, (name, _, _) in enumerate(bam.refs())}
    bam.close()
    far = {s: dict(d) for s, d in r2m.items()}
    k = 0
    for s in far:
        for name in list(far[s])[::7]:
            far[s][name] = 200 + (k % 50)
            k += 1
    assert k > 50
    values = sorted({v for d in far.values() for v in d.values()})
    assert 50 < len(values) <= 128 and values[-1] == 249
    rank = {v: i for i, v in enumerate(values)}
    small = {s: {n: rank[v] for n, v in d.items()} for s, d in far.items()}            # order-preserving stand-ins below 128
    with caplog.at_level(logging.WARNING):
        a = prof.profile_bam(path, None, far, None, strict=True, **kw)
    assert not any("mismatches" in r.message for r in caplog.records)
    b = prof.profile_bam(path, None, small, None, strict=True, **kw)
    assert sorted(a) == sorted(b) and len(a) > 5
    for key in a:
        _same_split_mm(a[key], b[key], values)
        assert a[key].mm_clamped is None
    assert max(max(S.covT) for S in a.values() if len(S.covT)) == 249
    # more than 128 different values
    wide = {s: dict(d) for s, d in r2m.items()}
    k = 0
    for s in wide:
        for name in list(wide[s])[::5]:
            wide[s][name] = 300 + 2 * (k % 150)
            k += 1
    values = sorted({v for d in wide.values() for v in d.values()})
    assert len(values) > 128
    cap = values[127]
    at_cap = {s: {n: min(v, cap) for n, v in d.items()} for s, d in wide.items()}
    caplog.clear()
    with caplog.at_level(logging.WARNING):
        a = prof.profile_bam(path, None, wide, None, **kw)
    assert any("are counted at that level" in r.message for r in caplog.records)
    with pytest.raises(ValueError, match="strict=True refuses"):
        prof.profile_bam(path, None, wide, None, strict=True, **kw)
    b = prof.profile_bam(path, None, at_cap, None, strict=True, **kw)
    assert sorted(a) == sorted(b) and len(a) > 5
    for key in a:
        _same_split(a[key], b[key])
        assert a[key].mm_clamped == cap and b[key].mm_clamped is None
    assert max(max(S.covT) for S in a.values() if len(S.covT)) == cap
    # a call that fails as a whole: the exception itself with strict, a logged failure and a partial dict without
    with pytest.raises(Exception):
        prof.profile_bam(str(tmp_path / "no_such.bam"), None, None, None, strict=True, **kw)
    assert prof.profile_bam(str(tmp_path / "no_such.bam"), None, None, None, **kw) == {}
