"""The reference's two unseeded-random output families (clonTR: snv_utilities.py:233-247;
r2_normalized / d_prime_normalized: linkage.py:200-228) are excluded from bit parity -- the
reference's own tests delete them (test_profile.py:896-900).  Here they come from a Philox
stream, so they are checked distributionally and for reproducibility."""
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


def _columns(n_pos, a, c, mm=0):
    """n_pos positions, each with `a` A's and `c` C's (one pair per observation)"""
    from instrain_amd import engine
    per = a + c
    pos = np.repeat(np.arange(n_pos, dtype=np.uint32), per)
    base = np.tile(np.r_[np.zeros(a, np.uint8), np.ones(c, np.uint8)], n_pos)
    return engine.pack_obs(pos, base, np.full(len(pos), mm)), np.arange(len(pos), dtype=np.uint32)


@pytest.mark.parametrize("n_mm_bins", [1, 3])
def test_rarefied_clonality_distribution(ctx, n_mm_bins):
    from instrain_amd import engine
    n_pos, a, c, n = 4000, 60, 40, 50
    obs, pair = _columns(n_pos, a, c)
    b = engine.Batch(ctx, np.zeros(n_pos, np.uint8), [0, n_pos], obs, pair, n_mm_bins=n_mm_bins, rarefied_coverage=n,
                     seed=1234, enable_linkage=False)
    b.run()
    r = b.fetch()
    b.run()
    r2 = b.fetch()
    b.close()
    v = r["clon_r"]
    assert len(v) == n_pos and not np.isnan(v).any()
    assert v.tobytes() == r2["clon_r"].tobytes()                       # reproducible
    k = np.arange(n + 1)
    feasible = ((k ** 2 + (n - k) ** 2) / n ** 2).astype(np.float32)
    assert np.isin(v, feasible).all()                                   # a multinomial resample of size 50
    p = np.array([a, c]) / (a + c)
    expect = (p ** 2 + p * (1 - p) / n).sum()                          # E[sum (X_k / n)^2]
    assert abs(v.mean() - expect) < 4e-3, (v.mean(), expect)
    kk = np.round((1 + np.sqrt(np.maximum(0, 2 * v.astype(np.float64) - 1))) / 2 * n)   # count of the major base
    assert abs(kk.mean() / n - 0.6) < 0.01 and 2.5 < kk.std() < 4.5    # Binomial(50, .6): sd 3.46 (folded tail is negligible)
    # exact clonality untouched
    full = r["clon"] if n_mm_bins == 1 else r["entries"]["clon"]
    assert (full == np.float32(0.6 * 0.6 + 0.4 * 0.4)).all()


def test_rarefied_threshold_seed_and_switch(ctx):
    from instrain_amd import engine
    n_pos = 500
    obs, pair = _columns(n_pos, 30, 19)                                 # coverage 49 < 50
    b = engine.Batch(ctx, np.zeros(n_pos, np.uint8), [0, n_pos], obs, pair, n_mm_bins=1, rarefied_coverage=50, enable_linkage=False)
    b.run()
    assert np.isnan(b.fetch()["clon_r"]).all()                          # sum(counts) >= min_covR is required
    b.close()
    obs, pair = _columns(n_pos, 30, 20)
    out = []
    for seed, cov in ((1, 50), (2, 50), (1, 0)):
        b = engine.Batch(ctx, np.zeros(n_pos, np.uint8), [0, n_pos], obs, pair, n_mm_bins=1, rarefied_coverage=cov, seed=seed,
                         enable_linkage=False)
        b.run()
        out.append(b.fetch()["clon_r"])
        b.close()
    assert not np.isnan(out[0]).any() and (out[0] != out[1]).any()      # the seed matters
    assert np.isnan(out[2]).all()                                       # rarefied_coverage <= 0 switches it off


def test_normalized_ld_columns(ctx):
    """perfect linkage (only AB and ab pairs): every resample gives r2_normalized == 1 unless it is
    monomorphic (NaN); D'_normalized == 1 likewise; reproducible; seed-dependent draw pattern."""
    from instrain_amd import engine
    n_sites, depth = 40, 60
    pos, base, pr = [], [], []
    pid = 0
    for s in range(0, n_sites, 2):
        for d in range(depth):
            alt = d % 3 == 0                                            # 1/3 of the pairs carry the alt haplotype
            for p in (10 * s + 5, 10 * s + 15):
                pos.append(p); base.append(1 if alt else 0); pr.append(pid)
            pid += 1
    obs = engine.pack_obs(np.array(pos, np.uint32), np.array(base, np.uint8), np.zeros(len(pos), int))
    res = []
    for seed in (7, 7, 8):
        b = engine.Batch(ctx, np.zeros(10 * n_sites + 20, np.uint8), [0, 10 * n_sites + 20], obs, np.array(pr, np.uint32),
                         n_mm_bins=1, min_snp=20, seed=seed)
        b.run()
        res.append(b.fetch()["ld"])
        b.close()
    ld = res[0]
    assert len(ld) == n_sites // 2 and (ld["r2"] == 1.0).all() and (ld["d_prime"] == 1.0).all()
    rn, dn = ld["r2_normalized"], ld["d_prime_normalized"]
    assert (np.isclose(rn, 1.0, atol=1e-12) | np.isnan(rn)).all() and (~np.isnan(rn)).sum() >= len(ld) - 2
    assert (np.isclose(dn, 1.0, atol=1e-12) | np.isnan(dn)).all()
    assert res[0].tobytes() == res[1].tobytes()
    assert (res[0]["total"] == res[2]["total"]).all()
