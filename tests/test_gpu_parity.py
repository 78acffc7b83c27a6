"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C ABI vs
(a) golden vectors from the reference's own Python, (b) the reference's stored sars_cov_2 run,
(c) the C oracle on seeded random inputs, (d) size-independent properties.

Bars: integer / categorical tables bit-exact; clonality float32 bit-exact; r2 / D' within 1e-6
(north_star) -- in practice identical, the same IEEE fp64 operations run in the same order."""
import glob
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(util.GOLD, "synth_*.npz")))
TOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    from instrain_amd import engine
    c = engine.Context(0)
    lut, fb = util.load_lut()
    c.set_null_model(lut, fb)
    yield c
    c.close()


def _params(g):
    return dict(min_cov=int(g["p_min_cov"]), min_freq=float(g["p_min_freq"]), min_snp=int(g["p_min_snp"]))


@pytest.mark.parametrize("name", CASES)
def test_golden_vectors(ctx, name):
    from tests import prod
    g = util.load_case(name)
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name)
    assert res["n_edges"] == int(g["n_edges"])


@pytest.mark.parametrize("name", ["synth_m1", "synth_skipmm"])
def test_dense_path_equals_mm_path(ctx, name):
    """n_mm_bins == 1 (dense output) and n_mm_bins == 4 with all mm == 0 give identical tables."""
    from tests import prod
    g = util.load_case(name)
    a = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), n_mm_bins=1, **_params(g))
    b = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), n_mm_bins=4, **_params(g))
    util.assert_same(util.canon_from_struct(a), util.canon_from_struct(b), float_tol=0.0, what=name)


@pytest.mark.parametrize("window", [64, 128, 1024])
def test_window_size_invariance(ctx, window):
    from tests import prod
    g = util.load_case("synth_dense")
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), window=window, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what="window%d" % window)


@pytest.mark.parametrize("window", [3264, 3328, 4096])
def test_short_record_decode_variants(ctx, window):
    """one mm bin, 2-byte records: up to W = 3264 two records are decoded at a time with packed 16-bit ALU ops (byte
    offsets of the counters must fit 16 bits), wider windows take the one-at-a-time decode; same tables either way"""
    from tests import prod
    g = util.load_case("synth_m1_ld")
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), window=window, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what="window%d" % window)


def test_stored_sars_golden_from_bam(ctx):
    """BAM -> C++ front end -> kernels -> tables == the reference's stored run (3 splits in one batch)."""
    from instrain_amd import engine
    from tests import prod
    from tests.test_oracle_golden import check_against_sars_golden, read_fasta
    bam = engine.BamFile(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    obs, pair, bounds, sref = bam.expand()
    assert bam.info["filtered_pairs"] == 13124 and bam.info["n_obs"] == 3717600
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    b = engine.Batch(ctx, engine.encode_seq(seq), bounds, obs, pair, n_mm_bins=bam.info["max_mm"] + 1,
                     min_cov=5, min_freq=0.05, min_snp=20)
    b.run()
    res = prod.to_oracle_layout(b.fetch(), lambda g: g.astype(np.int64))
    sizes = b.sizes()
    b.close()
    bam.close()
    check_against_sars_golden(res["snv"], res["ld"], float_tol=TOL)
    assert sizes["n_edges"] == 963 and sizes["n_increments"] == 23319


def _random_split(seed, mLen, depth, mm_levels, n_sites):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(util.GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)         # only its synthetic generator is used (no reference import)
    return mg.synth_case(seed=seed, mLen=mLen, depth=depth, mm_levels=mm_levels, n_sites=n_sites,
                         p_other=0.02, ref_ambig=5, self_pairs=0.2)


@pytest.mark.parametrize("seed,mLen,depth,mm_levels,n_sites", [(101, 3000, 60, 6, 120), (102, 9000, 25, 1, 200),
                                                                (103, 1500, 300, 12, 100), (104, 5000, 40, 33, 150)])
def test_random_splits_vs_oracle(ctx, seed, mLen, depth, mm_levels, n_sites):
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(seed, mLen, depth, mm_levels, n_sites)
    exp = oracle.profile_split(pos, base, mm, pair, seq, 0, lut, fb)
    got = prod.run_split(ctx, pos, base, mm, pair, seq, 0)
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="seed%d" % seed)
    assert got["n_edges"] == exp["n_edges"] and got["sizes"]["n_increments"] == exp["n_increments"]


def test_randomized_parameter_sweep_vs_oracle(ctx):
    """48 random small splits with random thresholds, depths, mm-level counts, non-ACGT rates, windows and
    linkage modes: every table must equal the oracle's (integers exact, floats to TOL)."""
    import importlib.util
    from oracle import oracle
    from tests import prod
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(util.GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    lut, fb = util.load_lut()
    rng = np.random.Generator(np.random.PCG64(20250927))
    n_rows = 0
    for case in range(48):
        mLen = int(rng.choice([37, 64, 130, 500, 1200, 3000]))
        depth = int(rng.choice([3, 8, 20, 60, 200]))
        mml = int(rng.choice([1, 1, 2, 5, 12, 40]))
        kw = dict(seed=1000 + case, mLen=mLen, start=int(rng.choice([0, 17, 123456])), depth=depth,
                  read_len=int(rng.choice([20, 60, 150])), n_sites=int(min(mLen // 6, rng.choice([0, 3, 20, 120]))),
                  mm_levels=mml, p_other=float(rng.choice([0.0, 0.02, 0.25])), ref_ambig=int(rng.choice([0, 0, 5])),
                  self_pairs=float(rng.choice([0.05, 0.5])), err=float(rng.choice([0.0, 0.01, 0.08])),
                  af_lo=float(rng.choice([0.02, 0.2])), af_hi=float(rng.choice([0.5, 1.0])))
        kw["read_len"] = min(kw["read_len"], max(8, mLen // 2))
        kw["ref_ambig"] = min(kw["ref_ambig"], mLen // 8)
        par = dict(min_cov=int(rng.choice([1, 3, 5, 10])), min_freq=float(rng.choice([0.01, 0.05, 0.2, 0.5])),
                   min_snp=int(rng.choice([2, 5, 20])))
        seq, pos, base, mm, pair = mg.synth_case(**kw)
        if len(pos) == 0:
            continue
        exp = oracle.profile_split(pos, base, mm, pair, seq, kw["start"], lut, fb, **par)
        extra = dict(window=int(rng.choice([0, 64, 192])))
        if mml == 1 and rng.random() < 0.5:
            extra["linkage_mode"] = 2                                  # dense MFMA path
        got = prod.run_split(ctx, pos, base, mm, pair, seq, kw["start"], n_mm_bins=mml, **par, **extra)
        what = "case %d %r %r %r" % (case, kw, par, extra)
        util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what=what)
        assert got["n_edges"] == exp["n_edges"] and got["sizes"]["n_increments"] == exp["n_increments"], what
        n_rows += len(exp["snv"]) + len(exp["ld"])
    assert n_rows > 2000


def test_multi_split_batch_vs_per_split_oracle(ctx):
    """Several splits in one flat batch: linkage must not cross split bounds; pair ids shared across bounds."""
    from instrain_amd import engine
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(77, 6000, 50, 4, 240)
    bounds = np.array([0, 1500, 1501, 4000, 6000])
    b = engine.Batch(ctx, engine.encode_seq(seq), bounds, engine.pack_obs(pos.astype(np.uint32), base, mm),
                     pair.astype(np.uint32), n_mm_bins=4)
    b.run()
    got = prod.to_oracle_layout(b.fetch(), lambda g: g.astype(np.int64))
    b.close()
    exp = {"entries": [], "snv": [], "ld": []}
    for s, e in zip(bounds[:-1], bounds[1:]):
        r = oracle.profile_split(pos, base, mm, pair, seq[s:e], int(s), lut, fb)
        for k in exp:
            exp[k].append(r[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="multi")


@pytest.mark.parametrize("mml,wide", [(1, False), (3, False), (1, True), (3, True)])
def test_stream_with_a_jump(ctx, mml, wide, monkeypatch):
    """two islands of reads 250 kbp apart.  Compact stream: the group of 256 records that would hold both is cut
    and padded (device record index != input record index from there on; pair ids follow).  Wide stream
    (forced): the 1024-record chunk spans >= 65535 positions, so the allele pass streams 4-byte positions."""
    layout = 1 if wide else 0            # ISX_LAYOUT_WIDE_RECORDS
    from instrain_amd import engine
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq1, p1, b1, m1, r1 = _random_split(301, 700, 40, mml, 40)
    seq2, p2, b2, m2, r2 = _random_split(302, 900, 40, mml, 50)
    n1 = len(p1) - (len(p1) % 1024) - 200                     # the boundary falls inside a chunk
    p1, b1, m1, r1 = p1[:n1], b1[:n1], m1[:n1], r1[:n1]
    gap = 250_000
    seq = seq1 + "A" * (gap - len(seq1)) + seq2
    pos = np.concatenate([p1, p2 + gap]); base = np.concatenate([b1, b2]); mm = np.concatenate([m1, m2])
    pair = np.concatenate([r1, r2 + int(r1.max()) + 1])
    bounds = np.array([0, len(seq1), gap, len(seq)])
    b = engine.Batch(ctx, engine.encode_seq(seq), bounds, engine.pack_obs(pos.astype(np.uint32), base, mm),
                     pair.astype(np.uint32), n_mm_bins=mml, layout=layout)
    b.run()
    assert b.timings()["record_bytes"] == (8 if wide else (2 if mml == 1 else 4))      # one mm bin: 2-byte records
    got = prod.to_oracle_layout(b.fetch(), lambda g: g.astype(np.int64))
    b.close()
    exp = {"entries": [], "snv": [], "ld": []}
    for s, e in zip(bounds[:-1], bounds[1:]):
        r = oracle.profile_split(pos, base, mm, pair, seq[s:e], int(s), lut, fb)
        for k in exp:
            exp[k].append(r[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    assert len(exp["ld"]) > 50
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="wide chunk")


def test_database_like_stream_with_many_jumps(ctx):
    """7 copies of one profiled region 100 kbp apart in the flat space (a database of genomes, most of it
    uncovered): every copy's tables must equal the oracle's for the region; the compact stream is cut at every
    jump (threaded cut pass: > 4096 input groups)"""
    from instrain_amd import engine
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(401, 3000, 60, 3, 150)
    K, step = 7, 100_000
    n_pairs = int(pair.max()) + 1
    P = np.concatenate([pos + k * step for k in range(K)])
    B = np.tile(base, K); M = np.tile(mm, K)
    R = np.concatenate([pair + k * n_pairs for k in range(K)])
    assert len(P) > 4096 * 256
    ref = np.zeros(K * step, dtype=np.uint8)
    bounds = []
    for k in range(K):
        ref[k * step:k * step + len(seq)] = engine.encode_seq(seq)
        bounds += [k * step, k * step + len(seq)]
    bounds.append(K * step)
    b = engine.Batch(ctx, ref, np.array(bounds), engine.pack_obs(P.astype(np.uint32), B, M), R.astype(np.uint32), n_mm_bins=3)
    b.run()
    assert b.timings()["record_bytes"] == 4
    got = prod.to_oracle_layout(b.fetch(), lambda g: g.astype(np.int64))
    b.close()
    exp = oracle.profile_split(pos, base, mm, pair, seq, 0, lut, fb)
    ce = util.canon_from_struct(exp)
    assert len(exp["ld"]) > 100
    for k in range(K):
        sub = {}
        for name, key in (("entries", "pos"), ("snv", "pos"), ("ld", "pos_a")):
            t = got[name][(got[name][key] >= k * step) & (got[name][key] < (k + 1) * step)].copy()
            for f in (("pos",) if name != "ld" else ("pos_a", "pos_b")):
                t[f] -= k * step
            sub[name] = t
        util.assert_same(util.canon_from_struct(sub), ce, float_tol=TOL, what="copy %d" % k)


@pytest.mark.parametrize("skip_mm", [True, False])
def test_jumpy_stream_all_formats_agree(ctx, skip_mm, monkeypatch):
    """a 240 kbp workload torn into 3 kbp islands 9-70 kbp apart (jumps in most record groups, some below and
    some above the 13-bit / 16-bit delta limits): the 2-byte / 4-byte streams with their cut-and-pad layout
    must give exactly what the plain 8-byte stream gives"""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=240_000, coverage=40, n_sites=2400, seed=31, skip_mm=skip_mm, err=0.003, af_lo=0.2)
    rng = np.random.Generator(np.random.PCG64(7))
    isl = 3000
    gaps = rng.choice([0, 9_000, 70_000], size=240_000 // isl)
    shift = np.cumsum(gaps) - gaps[0]
    obs = w["obs"].copy()
    obs["gpos"] = obs["gpos"] + shift[obs["gpos"] // isl].astype(np.uint32)
    n_pos = int(240_000 + shift[-1])
    ref = np.zeros(n_pos, dtype=np.uint8)
    idx = np.arange(240_000)
    ref[idx + shift[idx // isl]] = w["ref_codes"]
    bounds = np.unique(np.r_[0, (np.arange(0, 240_000, isl) + shift), n_pos])
    M = w["n_mm_bins"]
    res = {}
    for name, layout in (("compact", 0), ("wide", 1), ("four", 2)):      # ISX_LAYOUT_WIDE_RECORDS / _NO_SHORT_RECORDS
        b = engine.Batch(ctx, ref, bounds, obs, w["pair"], n_mm_bins=M, seed=3, min_snp=5, layout=layout)
        b.run()
        res[name] = (b.timings()["record_bytes"], b.fetch(), b.sizes())
        b.close()
    assert res["compact"][0] == (2 if M == 1 else 4) and res["wide"][0] == 8 and res["four"][0] == 4
    assert res["wide"][2]["n_ld"] > 200 and res["wide"][2]["n_snv"] > 500
    for name in ("compact", "four"):
        assert res[name][2] == res["wide"][2]
        for k, a in res["wide"][1].items():
            g = res[name][1][k]
            if a.dtype.names:
                for f in a.dtype.names:
                    np.testing.assert_array_equal(g[f], a[f], err_msg="%s %s.%s" % (name, k, f))
            else:
                np.testing.assert_array_equal(g, a, err_msg="%s %s" % (name, k))


@pytest.mark.parametrize("name", ["synth_mm4", "synth_m1", "synth_dense", "synth_ambig"])
def test_wide_record_stream_equals_golden(ctx, name, monkeypatch):
    """ISX_LAYOUT_WIDE_RECORDS forces the 8-byte stream (isx_obs as is) that the library otherwise only falls back to"""
    from tests import prod
    g = util.load_case(name)
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), layout=1, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name)


@pytest.mark.parametrize("name", ["synth_m1", "synth_skipmm"])
def test_one_mm_bin_with_4_byte_records_equals_golden(ctx, name, monkeypatch):
    """n_mm_bins == 1 normally takes the 2-byte stream; ISX_LAYOUT_NO_SHORT_RECORDS keeps the 4-byte one (k_pileup_dense<*, 4>)"""
    from instrain_amd import engine
    from tests import prod
    g = util.load_case(name)
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"] * 0, g["pair"], str(g["seq"]), int(g["start"]), n_mm_bins=1, layout=2, **_params(g))
    assert res["sizes"]["n_snv"] > 0
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name)


def test_record_stream_choice(ctx):
    """compact 4-byte records normally; the 8-byte stream when a record does not fit (here: an mm level >= 256,
    which is also out of range for any legal n_mm_bins <= 128 and must stay a loud error)"""
    from instrain_amd import engine
    seq, pos, base, mm, pair = _random_split(321, 1500, 30, 3, 60)
    b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], engine.pack_obs(pos.astype(np.uint32), base, mm),
                     pair.astype(np.uint32), n_mm_bins=3)
    b.run()
    assert b.timings()["record_bytes"] == 4
    b.close()
    b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], engine.pack_obs(pos.astype(np.uint32), base, mm * 0),
                     pair.astype(np.uint32), n_mm_bins=1)
    b.run()
    assert b.timings()["record_bytes"] == 2               # one mm bin: delta:13 | base:3
    b.close()
    mm2 = mm.copy()
    mm2[::97] = 300
    b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], engine.pack_obs(pos.astype(np.uint32), base, mm2),
                     pair.astype(np.uint32), n_mm_bins=3)
    with pytest.raises(engine.IsxError) as e:
        b.run()
    assert e.value.code == -4
    b.close()


def test_many_levels_per_site_grow_the_level_rows(ctx):
    """100 mm bins, deep, SNP sites at one position in six: the per-site level rows (first sized for 8 levels a
    site) overflow while the site table is already at its bound -- only the level rows must grow"""
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(611, 500, 500, 100, 80)
    exp = oracle.profile_split(pos, base, mm, pair, seq, 0, lut, fb)
    got = prod.run_split(ctx, pos, base, mm, pair, seq, 0, n_mm_bins=100)
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="levels")
    assert len(exp["snv"]) > 4000


def test_empty_and_ragged(ctx):
    from instrain_amd import engine
    from tests import prod
    z = np.zeros(0, dtype=np.int64)
    res = prod.run_split(ctx, z, z.astype(np.uint8), z, z, "ACGTACGTNN", 0, n_mm_bins=3)
    assert len(res["entries"]) == 0 and len(res["snv"]) == 0 and len(res["ld"]) == 0
    # a single deep column at the last position of a 1-position-short window
    n = 70000
    pos = np.full(n, 130)
    base = (np.arange(n) % 3 == 0).astype(np.uint8)       # A / C mix
    res = prod.run_split(ctx, pos, base, np.zeros(n, int), np.arange(n), "A" * 131, 0, n_mm_bins=1, window=64)
    assert len(res["entries"]) == 1 and res["entries"]["cnt"][0].sum() == n
    assert len(res["snv"]) == 1 and res["snv"]["allele_count"][0] == 2


def test_mm_out_of_range_is_loud(ctx):
    from instrain_amd import engine
    obs = engine.pack_obs(np.array([1, 2], dtype=np.uint32), np.array([0, 1], dtype=np.uint8), np.array([0, 9]))
    b = engine.Batch(ctx, engine.encode_seq("ACGTACGT"), [0, 8], obs, np.array([0, 1], dtype=np.uint32), n_mm_bins=4)
    with pytest.raises(engine.IsxError) as e:
        b.run()
    assert e.value.code == -4
    b.close()


def test_pipelined_launch_wait_equals_run(ctx):
    """isx_batch_launch / isx_batch_wait: two batches in flight give what blocking runs give; state errors are loud"""
    from instrain_amd import engine
    from tests import prod
    cases = []
    for seed, mml, link in ((201, 1, True), (202, 5, False), (203, 3, True)):
        seq, pos, base, mm, pair = _random_split(seed, 4000, 40, mml, 160)
        b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], engine.pack_obs(pos.astype(np.uint32), base, mm),
                         pair.astype(np.uint32), n_mm_bins=mml, enable_linkage=link)
        b.run()
        cases.append((b, b.fetch(), b.sizes()))
    for rep in range(3):
        for b, _, _ in cases:
            b.launch()
        with pytest.raises(engine.IsxError) as e:
            cases[0][0].launch()                                 # one pass in flight per batch
        assert e.value.code == -6
        with pytest.raises(engine.IsxError):
            cases[1][0].sizes()                                  # nothing to read before wait
        for b, exp, sz in reversed(cases):                       # collected in another order than launched
            b.wait()
            got = b.fetch()
            assert b.sizes() == sz
            for k in exp:
                if exp[k].dtype.names:
                    for f in exp[k].dtype.names:
                        np.testing.assert_array_equal(got[k][f], exp[k][f], err_msg="%s.%s" % (k, f))
                else:
                    np.testing.assert_array_equal(got[k], exp[k], err_msg=k)
    with pytest.raises(engine.IsxError) as e:
        cases[0][0].wait()
    assert e.value.code == -6
    for b, _, _ in cases:
        b.close()


def test_full_size_properties(ctx):
    """BASELINE configs[1] at FULL size (5 Mbp, 20x, 9e7 observations): properties that do not need the
    oracle -- sum of level counts == number of ACGT observations (a checksum of checksums: also per
    split); re-run idempotence; covT vs SNV-table coverage consistency (reference test_profile_13)."""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=5_000_000, coverage=20, n_sites=5000, seed=2, skip_mm=True)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1)
    b.run()
    r1 = b.fetch()
    b.run()
    r2 = b.fetch()
    b.close()
    assert int(r1["counts"].sum()) == int((w["obs"]["base"] < 4).sum())
    per_split = np.add.reduceat(r1["counts"].sum(axis=1).astype(np.int64), w["split_bounds"][:-1])
    exp_split = np.bincount(np.searchsorted(w["split_bounds"], w["obs"]["gpos"][w["obs"]["base"] < 4], side="right") - 1,
                            minlength=len(w["split_bounds"]) - 1)
    assert (per_split == exp_split).all()
    for k in r1:
        assert r1[k].tobytes() == r2[k].tobytes(), k
    cov = r1["counts"].sum(axis=1)
    assert (r1["snv"]["cnt"].sum(axis=1) == cov[r1["snv"]["gpos"]]).all()


def _bench_line(r):
    import json
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert len(lines[0]) < 4000                 # the printed line is a digest: the driver's record keeps all of it
    return json.loads(lines[0])


def test_bench_two_ranks_control_flow(tmp_path):
    """bench.py under torch.distributed.run with 2 ranks (both on GPU 0, gloo) -- the N>1 code path
    (per-rank shards of C5, verification pass, barrier, MAX-reduce of the time, SUM of units, final gather) runs and
    prints one JSON line whose value is the whole-job aggregate."""
    import subprocess
    import sys
    repo = os.path.dirname(util.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    env = dict(os.environ, ISX_DIST_BACKEND="gloo", ISX_DEVICE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = util.run_group([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(repo, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--detail", str(tmp_path / "d.json")],
                       env=env, timeout=420)
    j = _bench_line(r)
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["backend"] == "gloo"
    assert j["scaling"] == "strong" and j["value"] > 0 and "final_gather_ms" in j
    assert j["roofline"]["frac"] > 0 and "cpu_baseline" not in j and j["bam_sharded_gbp_per_s"] > 0
    assert "C5" in j["config"]["workload"]
    # the gather is the whole profile of one pass (every batch's SNV + LD rows + per-scaffold summaries), the ranks' own times are in the line
    fg, pr = j["final_gather"], j["per_rank"]
    assert fg["rows"]["snv"] > 0 and fg["rows"]["summary"] > 0 and fg["bytes"] > fg["my_bytes"] > 0
    assert pr["passes_per_step"] == 4 == j["passes_per_step"] and len(pr["pass_ms"]) == 2 and pr["pass_ms_max"] >= pr["pass_ms_min"] > 0
    assert j["c5_staged_replay_gbp_per_s"] > 0


def test_bench_eight_ranks_control_flow(tmp_path):
    """the N = 8 control flow on one GPU (gloo, the ranks share the device): one LPT shard of the database per rank, 16 passes a step,
    per-rank pass times, the final gather of all eight ranks' tables on rank 0"""
    import subprocess
    import sys
    repo = os.path.dirname(util.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ISX_DIST_BACKEND", "ISX_DEVICE")}
    r = util.run_group([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--scale", "0.04", "--depth", "2",
                        "--host-threads", "2", "--only-c5", "--detail", str(tmp_path / "d.json")], env=env, timeout=600)
    j = _bench_line(r)
    assert j["n_gpus"] == 8 and j["world_size_seen"] == 8 and j["value"] > 0 and j["passes_per_step"] == 16
    assert len(j["per_rank"]["pass_ms"]) == 8 and min(j["per_rank"]["batches"]) >= 1
    assert j["final_gather"]["rows"]["snv"] > 0 and j["final_gather_ms"] > 0


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts the two ranks itself; on a box with one
    GPU they share it over gloo (the line says so)"""
    import subprocess
    import sys
    repo = os.path.dirname(util.GOLD.rstrip("/")).rsplit("/tests", 1)[0]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ISX_DIST_BACKEND", "ISX_DEVICE")}
    r = util.run_group([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scale", "0.05",
                        "--only-c5", "--detail", str(tmp_path / "d.json")], env=env, timeout=420)
    j = _bench_line(r)
    assert j["n_gpus"] == 2 and j["world_size_seen"] == 2 and j["value"] > 0
    import torch
    if torch.cuda.device_count() < 2:
        assert j["backend"] == "gloo" and "share" in j["config"]["parallelism"]
    # a launcher that started the wrong number of ranks is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scale", "0.05", "--only-c5"],
                        env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)


@pytest.mark.parametrize("name", ["synth_m1", "synth_skipmm", "synth_m1_ld", "synth_skipmm_ld", "c3_split"])
def test_dense_mfma_linkage_equals_reference(ctx, name):
    """linkage_mode 2 (int8 MFMA X^T X + self pairs) vs the reference golden vectors (M == 1 cases; synth_m1_ld /
    synth_skipmm_ld / c3_split carry 1 714 / 1 141 / 302 LD rows produced by the reference itself)"""
    from tests import prod
    g = util.load_case(name)
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), n_mm_bins=1,
                         linkage_mode=2, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name + "-dense")
    assert res["n_edges"] == int(g["n_edges"])
    if name.endswith("_ld"):
        assert len(res["ld"]) > 1000


def test_c3_split_golden_sparse_path(ctx):
    """one split of the bench's C3 generator (200x, 1 site / 100 bp): tables from the reference's own Python"""
    from tests import prod
    g = util.load_case("c3_split")
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), n_mm_bins=1, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what="c3_split")
    assert res["n_edges"] == int(g["n_edges"]) and len(res["ld"]) == 302


def test_c3_slice_vs_oracle_and_dense_equals_sparse(ctx):
    """BASELINE configs[2] (C3): a 200 kbp slice of the exact bench generator (200x, 1 SNV site / 100 bp) against the
    oracle split by split; the dense int8-MFMA linkage path must equal the sparse one byte for byte"""
    from instrain_amd import engine, synth
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    w = synth.make_workload(genome_len=200_000, coverage=200, n_sites=2000, seed=3, skip_mm=True, af_lo=0.2, af_hi=0.5)
    out = {}
    for mode in (1, 2):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=True,
                         linkage_mode=mode)
        b.run()
        out[mode] = (b.fetch(), b.sizes())
        b.close()
    assert out[1][1] == out[2][1] and out[1][1]["n_ld"] > 3000
    for k in ("counts", "clon", "snv", "ld"):
        a, e = out[1][0][k], out[2][0][k]
        assert a.tobytes() == e.tobytes(), k
    got = prod.to_oracle_layout(out[2][0], lambda g: g.astype(np.int64))
    letters = np.array(list("ACTGN"))
    gpos = w["obs"]["gpos"].astype(np.int64)
    exp = {"entries": [], "snv": [], "ld": []}
    sb = w["split_bounds"]
    for s, e in zip(sb[:-1], sb[1:]):
        o = oracle.profile_split(gpos, w["obs"]["base"], w["obs"]["mm"].astype(np.int64), w["pair"].astype(np.int64),
                                 "".join(letters[w["ref_codes"][s:e]]), int(s), lut, fb)
        for k in exp:
            exp[k].append(o[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="C3 slice")


def test_c3_full_size_properties(ctx):
    """configs[2] in full (5 Mbp x 200x, 50 000 planted sites, 0.9 G observations): size-independent properties.
    Sum of counts == observations; every SNV row's counts == the count table; LD rows stay inside their split,
    position_A <= position_B, counts sum to `total`; sparse and dense linkage agree on every size; a second run
    of the same batch gives identical tables (idempotence)."""
    from instrain_amd import engine, synth
    meta = synth.Metagenome(1, total_read_bp=200 * 5_000_000, seed=3, contigs=1, len_lo=5_000_000, len_hi=5_000_000,
                            abundance_sigma=0.0, min_genome_coverage=0.0, site_frac=0.01, af_lo=0.2, af_hi=0.5)
    w = meta.generate([0])
    assert w["n_pos"] == 5_000_000 and w["n_obs"] > 850_000_000
    sizes, tabs = {}, {}
    for mode in (1, 2):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=True,
                         linkage_mode=mode)
        b.run()
        sizes[mode] = b.sizes()
        f = b.fetch()
        if mode == 1:
            b.run()
            f2 = b.fetch()
            for k in ("counts", "clon", "snv", "ld"):
                cols = [c for c in (f[k].dtype.names or [None]) if c not in ("r2_normalized", "d_prime_normalized")]
                for c in cols:
                    x, y = (f[k][c], f2[k][c]) if c else (f[k], f2[k])
                    assert x.tobytes() == y.tobytes(), ("idempotence", k, c)
        tabs[mode] = f
        b.close()
    assert sizes[1] == sizes[2] and sizes[1]["n_ld"] > 100_000 and sizes[1]["n_snv"] > 45_000
    f = tabs[1]
    assert int(f["counts"].sum(dtype=np.int64)) == w["n_obs"]
    assert (f["snv"]["cnt"] == f["counts"][f["snv"]["gpos"]]).all()
    ld = f["ld"]
    sb = w["split_bounds"]
    sa = np.searchsorted(sb, ld["gpos_a"], side="right")
    assert (sa == np.searchsorted(sb, ld["gpos_b"], side="right")).all() and (ld["gpos_a"] <= ld["gpos_b"]).all()
    assert (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] == ld["total"]).all()
    for c in ("gpos_a", "gpos_b", "total", "countAB", "countAb", "countaB", "countab", "r2", "d_prime"):
        assert tabs[1]["ld"][c].tobytes() == tabs[2]["ld"][c].tobytes(), c


@pytest.mark.parametrize("seed,mLen,depth,n_sites,bounds", [(201, 4000, 120, 200, None), (202, 12000, 60, 500, [0, 3000, 3001, 9000, 12000]),
                                                            (203, 2000, 400, 300, None)])
def test_dense_mfma_linkage_vs_sparse_and_oracle(ctx, seed, mLen, depth, n_sites, bounds):
    """dense vs sparse path on the same batch (multi-split, self pairs, > 32 sites per split) and vs the oracle"""
    from instrain_amd import engine
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(seed, mLen, depth, 1, n_sites)
    bounds = np.array(bounds if bounds else [0, mLen])
    out = {}
    for mode in (1, 2):
        b = engine.Batch(ctx, engine.encode_seq(seq), bounds, engine.pack_obs(pos.astype(np.uint32), base, mm),
                         pair.astype(np.uint32), n_mm_bins=1, linkage_mode=mode)
        b.run()
        out[mode] = (b.fetch(), b.sizes(), b.timings())
        b.close()
    assert out[1][0]["ld"].tobytes() == out[2][0]["ld"].tobytes()
    for k in ("n_edges", "n_ld", "n_increments", "n_allele_obs"):
        assert out[1][1][k] == out[2][1][k], k
    assert out[2][2]["dense_tiles"] > 0 and out[2][2]["dense_macs"] > 0
    exp = {"entries": [], "snv": [], "ld": []}
    for s, e in zip(bounds[:-1], bounds[1:]):
        r = oracle.profile_split(pos, base, mm, pair, seq[s:e], int(s), lut, fb)
        for k in exp:
            exp[k].append(r[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    got = prod.to_oracle_layout(out[2][0], lambda g: g.astype(np.int64))
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="dense%d" % seed)


def test_dense_mode_needs_single_mm_bin(ctx):
    """linkage_mode 2 (the MFMA co-occurrence path) with more than one mm bin is refused when the batch is created"""
    from instrain_amd import engine
    obs = engine.pack_obs(np.arange(8, dtype=np.uint32), np.zeros(8, np.uint8), np.zeros(8, int))
    with pytest.raises(engine.IsxError, match="needs n_mm_bins == 1"):
        engine.Batch(ctx, engine.encode_seq("ACGTACGT"), [0, 8], obs, np.arange(8, dtype=np.uint32), n_mm_bins=2, linkage_mode=2)
    b = engine.Batch(ctx, engine.encode_seq("ACGTACGT"), [0, 8], obs, np.arange(8, dtype=np.uint32), n_mm_bins=1, linkage_mode=2)
    b.run()
    b.close()


@pytest.mark.parametrize("n_mm", [1, 5])
def test_divergent_reference_rows_everywhere(ctx, n_mm):
    """every position differs from the reference (SNS rows at each covered position): overflows the
    per-window row queue of the mm kernel and stresses row allocation of both kernels"""
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    seq, pos, base, mm, pair = _random_split(301, 2500, 40, n_mm, 60)
    rot = {"A": "C", "C": "T", "T": "G", "G": "A", "N": "N"}
    seq2 = "".join(rot[c] for c in seq)                      # a reference that matches (almost) nowhere
    exp = oracle.profile_split(pos, base, mm, pair, seq2, 0, lut, fb)
    assert len(exp["snv"]) > 2000
    got = prod.run_split(ctx, pos, base, mm, pair, seq2, 0, n_mm_bins=n_mm)
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="divergent")


def test_unpacked_counter_variant(ctx, monkeypatch):
    """the u32-counter variant of the mm kernel (taken automatically when a window streams >= 65536
    records) forced through ISX_LAYOUT_NO_PACKED_COUNTERS, against the same golden vectors"""
    from tests import prod
    for name in ("synth_mm4", "synth_dense", "synth_selfpairs"):
        g = util.load_case(name)
        res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), layout=4, **_params(g))
        util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name + "-u32")


def test_deep_window_takes_unpacked_automatically(ctx):
    """> 65535 records in one window: the packed variant is illegal; counts stay exact"""
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    n = 140000
    rng = np.random.Generator(np.random.PCG64(5))
    pos = rng.integers(0, 3, n)                              # three very deep columns
    base = rng.choice(4, n, p=[0.55, 0.25, 0.15, 0.05]).astype(np.uint8)
    mm = rng.integers(0, 3, n)
    pair = np.arange(n)
    exp = oracle.profile_split(pos, base, mm, pair, "AAAC", 0, lut, fb)
    got = prod.run_split(ctx, pos, base, mm, pair, "AAAC", 0, n_mm_bins=3)
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=TOL, what="deep")
    assert got["entries"]["cnt"].sum() == n


def test_tables_grow_when_estimates_are_too_small(ctx, monkeypatch):
    """rows at every position of a >1 Mbp divergent reference exceed the initial SNV-row estimate
    (max(n_pos / 2, 2^20)): the library grows the table and repeats the pass instead of failing"""
    from instrain_amd import engine
    n_pos, depth = 1_300_000, 6
    pos = np.repeat(np.arange(n_pos, dtype=np.uint32), depth)
    base = np.ones(len(pos), dtype=np.uint8)                  # every read says C ...
    obs = engine.pack_obs(pos, base, np.zeros(len(pos), int))
    b = engine.Batch(ctx, np.zeros(n_pos, np.uint8), [0, n_pos], obs, np.arange(len(pos), dtype=np.uint32) // depth,
                     n_mm_bins=1, enable_linkage=False)          # ... the reference says A: SNS everywhere
    b.run()
    s = b.sizes()
    snv = b.fetch()["snv"]
    b.close()
    assert s["n_snv"] == n_pos and (snv["cls"] == 2).all() and (snv["gpos"] == np.arange(n_pos)).all()


def test_rarefied_gate_independent_of_min_cov(ctx):
    """clonTR exists wherever cumulative coverage >= rarefied_coverage (snv_utilities.py:100-102), also where it is
    below min_cov: with rarefied_coverage < min_cov the dense and the mm kernels must mark the same positions"""
    from instrain_amd import engine
    seq, pos, base, mm, pair = _random_split(77, 1500, 14, 1, 20)
    obs = engine.pack_obs(pos.astype(np.uint32), base, mm * 0)
    out = {}
    for M in (1, 3):
        b = engine.Batch(ctx, engine.encode_seq(seq), [0, len(seq)], obs, pair.astype(np.uint32), n_mm_bins=M,
                         min_cov=12, rarefied_coverage=6, seed=9)
        b.run()
        out[M] = b.fetch()
        b.close()
    cov = out[1]["counts"].sum(axis=1)
    have = ~np.isnan(out[1]["clon_r"])
    assert ((cov >= 6) == have).all() and ((cov >= 6) & (cov < 12)).sum() > 50
    e = out[3]["entries"]
    assert (e["gpos"][~np.isnan(e["clon_rarefied"])] == np.flatnonzero(have)).all()
    assert (np.isnan(out[1]["clon"]) == (cov < 12)).all()
