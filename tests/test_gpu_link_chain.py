"""The linkage stages' bucket chain (csrc/isx_linkage.hip, round 6: per-pair hash chains, a bucket of pair increments per
first site, an LDS hash table + bitonic sort per site -- eight launches, one host sync) against the sorted chain it
replaces (device-wide radix sorts, ISX_LINK_CHAIN=sorted; still the fallback for a site with more unique keys than the
LDS table holds): same LD rows byte for byte, same sizes.  Both are checked against the reference's golden vectors in
tests/test_gpu_parity.py; here: chain vs chain on larger batches, the fallback, the growth steps.

Reference: calc_mm_SNV_linkage_network linkage.py:14-44, _iterator_ld_sites :78-131, _calc_ld_single :138-196."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from instrain_amd import engine
    lut, fb = util.load_lut()
    c = engine.Context(0)
    c.set_null_model(lut, fb)
    yield c
    c.close()


def _run(ctx, w, n_mm_bins, **kw):
    from instrain_amd import engine
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=n_mm_bins, enable_linkage=True, **kw)
    b.run()
    f, s = b.fetch(), b.sizes()
    b.run()                                     # the same batch again: the chain's tables are reused (epoch-tagged heads)
    f2 = b.fetch()
    b.close()
    assert f["ld"].tobytes() == f2["ld"].tobytes()
    return f, s


def _same(a, b, what):
    fa, sa = a
    fb_, sb = b
    for k in ("n_ld", "n_edges", "n_increments", "n_allele_obs", "n_snv", "n_sites"):
        assert sa[k] == sb[k], (what, k, sa[k], sb[k])
    assert fa["ld"].tobytes() == fb_["ld"].tobytes(), what
    assert fa["snv"].tobytes() == fb_["snv"].tobytes(), what


@pytest.mark.parametrize("skip_mm,cov,sites", [(True, 200, 2000), (False, 60, 1500), (True, 45, 4000)])
def test_bucket_chain_equals_sorted_chain(ctx, monkeypatch, skip_mm, cov, sites):
    """a 200 kbp slice of the C3 generator (deep, dense sites), an mm-on batch, a shallow one"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=200_000, coverage=cov, n_sites=sites, seed=11, skip_mm=skip_mm, af_lo=0.2, af_hi=0.5)
    M = 1 if skip_mm else int(w["obs"]["mm"].max()) + 1
    got = _run(ctx, w, M)
    monkeypatch.setenv("ISX_LINK_CHAIN", "sorted")
    exp = _run(ctx, w, M)
    assert exp[1]["n_ld"] > (1000 if cov >= 60 else 50) and exp[1]["n_edges"] >= exp[1]["n_ld"] // (M + 1)
    _same(got, exp, "bucket vs sorted")


def test_site_with_too_many_keys_takes_the_sorted_chain(ctx, monkeypatch):
    """ISX_LINK_MAXU=2: nearly every site's bucket overflows its table -> the batch falls back, same tables"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=60_000, coverage=80, n_sites=600, seed=5, skip_mm=True, af_lo=0.2, af_hi=0.5)
    monkeypatch.setenv("ISX_LINK_CHAIN", "sorted")
    exp = _run(ctx, w, 1)
    monkeypatch.delenv("ISX_LINK_CHAIN")
    monkeypatch.setenv("ISX_LINK_MAXU", "2")
    got = _run(ctx, w, 1)
    _same(got, exp, "fallback")
    assert exp[1]["n_ld"] > 100


def test_chain_tables_grow(ctx, monkeypatch):
    """the first attempt is told that the buckets hold 16 increments and the row table 4 rows: both growth steps run"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=60_000, coverage=80, n_sites=600, seed=6, skip_mm=True, af_lo=0.2, af_hi=0.5)
    exp = _run(ctx, w, 1)
    monkeypatch.setenv("ISX_LINK_TEST_CAPS", "16,4")
    got = _run(ctx, w, 1)
    _same(got, exp, "growth")
    monkeypatch.setenv("ISX_LINK_TEST_CAPS", "1000000000,4")
    got = _run(ctx, w, 1)
    _same(got, exp, "row growth only")


def test_bucket_chain_vs_oracle_multi_split(ctx):
    """several splits in one batch incl. an empty one and self pairs (both mates over one column), against the oracle"""
    from instrain_amd import synth
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    w = synth.make_workload(genome_len=12_000, coverage=90, n_sites=400, seed=9, skip_mm=True, af_lo=0.2, af_hi=0.5)
    w["split_bounds"] = np.array([0, 3000, 3001, 9000, 12000], dtype=np.int64)
    f, s = _run(ctx, w, 1)
    got = prod.to_oracle_layout(f, lambda g: g.astype(np.int64))
    letters = np.array(list("ACTGN"))
    gpos = w["obs"]["gpos"].astype(np.int64)
    exp = {"entries": [], "snv": [], "ld": []}
    sb = w["split_bounds"]
    for a, e in zip(sb[:-1], sb[1:]):
        o = oracle.profile_split(gpos, w["obs"]["base"], w["obs"]["mm"].astype(np.int64), w["pair"].astype(np.int64),
                                 "".join(letters[w["ref_codes"][a:e]]), int(a), lut, fb)
        for k in exp:
            exp[k].append(o[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=1e-6, what="multi-split")
    assert s["n_ld"] > 100


def test_sparse_pair_ids_take_the_hashed_chains(ctx, monkeypatch):
    """pair ids spread over 2^28 (a caller's own numbering): the chains' heads are a hash table instead of a slot per pair"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=60_000, coverage=80, n_sites=600, seed=7, skip_mm=True, af_lo=0.2, af_hi=0.5)
    w = dict(w)
    w["pair"] = (w["pair"].astype(np.uint64) * 4099 % (1 << 28)).astype(np.uint32)
    assert len(np.unique(w["pair"])) > 0.99 * (int(w["pair"].shape[0]) // 150 // 2)          # (still one id per pair, give or take a collision)
    got = _run(ctx, w, 1)
    monkeypatch.setenv("ISX_LINK_CHAIN", "sorted")
    exp = _run(ctx, w, 1)
    _same(got, exp, "hashed chains")
    assert exp[1]["n_ld"] > 100


def test_batch_without_sites_or_pairs(ctx):
    """no SNP site at all (the chain is not entered), and sites whose reads never see two of them (no increment)"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=30_000, coverage=30, n_sites=0, err=0.0, seed=8, skip_mm=True)
    f, s = _run(ctx, w, 1)
    assert s["n_ld"] == 0 and s["n_edges"] == 0 and len(f["ld"]) == 0
    w = synth.make_workload(genome_len=200_000, coverage=40, n_sites=20, err=0.0, seed=8, skip_mm=True, af_lo=0.3, af_hi=0.5)
    f, s = _run(ctx, w, 1)
    assert s["n_sites"] >= 15 and s["n_ld"] <= 2


@pytest.mark.parametrize("cov,sites,glen", [(60, 1500, 200_000), (150, 2600, 4096)])
def test_mm_site_table_in_position_order_without_a_sort(ctx, monkeypatch, cov, sites, glen):
    """mm profiling on: k_pileup_mm records every window's range in the site table and k_site_order puts the table in position order window by
    window (round 6) -- same rows as through the device-wide sort (ISX_LINK_SITE_SORT=rocprim).  Second case: SNP sites at more than half of the
    positions of a window -- beyond the kernel's row queue, some sites are allocated outside their window's range and the batch takes the sort
    by itself (ISX_FLAG_SITES_LOOSE)"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=glen, coverage=cov, n_sites=sites, seed=13, skip_mm=False, af_lo=0.3, af_hi=0.5)
    M = int(w["obs"]["mm"].max()) + 1
    got = _run(ctx, w, M)
    monkeypatch.setenv("ISX_LINK_SITE_SORT", "rocprim")
    exp = _run(ctx, w, M)
    _same(got, exp, "window table vs sort")
    assert exp[1]["n_ld"] > 500 and exp[1]["n_sites"] > 0.5 * sites
    e = got[0]["entries"] if "entries" in got[0] else None
    assert e is None or e.tobytes() == exp[0]["entries"].tobytes()
