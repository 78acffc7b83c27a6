"""CPU-only: the product's C++ BAM front end (BGZF/BAM decode, read-pair filter, htslib-1.9
overlap rules, expansion) against the oracle's pure-Python restatement and the committed golden
observations; host code only, no GPU."""
import os

import numpy as np
import pytest

from instrain_amd import engine
from tests import util


def test_sars_bam_matches_committed_observations():
    bam = engine.BamFile(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    assert bam.refs() == [("MT039887.1", 29879, 0)]
    obs, pair, bounds, sref = bam.expand()
    i = bam.info
    assert (i["n_reads"], i["unfiltered_pairs"], i["filtered_pairs"], i["median_insert"], i["max_mm"]) == \
           (28913, 13925, 13124, 267.0, 25)
    z = np.load(os.path.join(util.GOLD, "sars_cov_2_obs.npz"))
    assert len(obs) == len(z["pos"]) == 3717600
    assert (obs["gpos"] == z["pos"]).all() and (obs["base"] == z["base"]).all()
    assert (obs["mm"] == z["mm"]).all() and (pair == z["pair"]).all()
    assert list(bounds) == [0, 9959, 19918, 29879] and list(sref) == [0, 0, 0]      # iterate_splits(29879, 10000)
    bam.close()


def test_zero_copy_view_equals_copy():
    a = engine.BamFile(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    b = engine.BamFile(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    o1, p1, b1, s1 = a.expand()
    o2, p2, b2, s2 = b.expand(copy=False)               # views of the handle's arrays (isx_bam_view)
    assert o2.dtype == o1.dtype and len(o2) == len(o1) == 3717600
    assert (o1 == o2).all() and (p1 == p2).all() and (b1 == b2).all() and (s1 == s2).all()
    a.close(); b.close()


def test_small_scaffold_matches_oracle_python():
    """second BAM fixture of the reference's tests (126 bp scaffold, 751 reads): C++ == oracle/bam_py"""
    from oracle import bam_py
    path = os.path.join(util.GOLD, "SmallScaffold.fa.sorted.bam")
    refs, reads = bam_py.read_bam(path)
    p2i = {r[0]: bam_py.get_paired_reads(reads, t) for t, r in enumerate(refs)}
    r2m, tallies = bam_py.filter_pairs(p2i)
    bam = engine.BamFile(path)
    obs, pair, bounds, sref = bam.expand()
    assert [(n, l) for n, l, _ in bam.refs()] == refs
    assert bam.info["filtered_pairs"] == sum(t["filtered_pairs"] for t in tallies.values())
    P, B, M = [], [], []
    off = 0
    for t, (name, ln) in enumerate(refs):
        bam_py.resolve_overlaps(reads, t)
        pos, base, mm, pr, _ = bam_py.expand_observations(reads, t, r2m[name])
        P.append(pos + off); B.append(base); M.append(mm)
        off += ln
    assert (obs["gpos"] == np.concatenate(P)).all()
    assert (obs["base"] == np.concatenate(B)).all() and (obs["mm"] == np.concatenate(M)).all()
    bam.close()


def test_skip_mm_and_errors():
    bam = engine.BamFile(os.path.join(util.GOLD, "SmallScaffold.fa.sorted.bam"))
    obs, pair, bounds, sref = bam.expand(skip_mm=True)
    assert (obs["mm"] == 0).all() and bam.info["max_mm"] == 0
    try:
        bam.expand()
        assert False, "second expand must be refused (qualities were rewritten)"
    except engine.IsxError as e:
        assert e.code == -6
    bam.close()
    try:
        engine.BamFile("/nonexistent.bam")
        assert False
    except engine.IsxError as e:
        assert e.code == -5


def test_random_messy_bam_cpp_equals_oracle_python(tmp_path):
    """synthetic BAMs with indels / clips / ref-skips / =X / overlapping disagreeing mates / flag zoo /
    aux tags of every type: product C++ front end == oracle/bam_py.py, observation for observation"""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafB", 900), ("scafC", 12500)]
    for seed, n_pairs in ((1, 1500), (2, 1500), (3, 7000)):       # > 4096 reads: the threaded extraction / expansion paths
        path = str(tmp_path / ("r%d.bam" % seed))
        reads = bamwriter.random_reads(seed, refs, n_pairs)
        bamwriter.write_bam(path, refs, reads)
        rrefs, rr = bam_py.read_bam(path)
        assert rrefs == refs and len(rr) == len(reads)
        p2i = {r[0]: bam_py.get_paired_reads(rr, t) for t, r in enumerate(refs)}
        r2m, tallies = bam_py.filter_pairs(p2i, min_read_ani=0.9)
        bam = engine.BamFile(path)
        obs, pair, bounds, sref = bam.expand(min_read_ani=0.9, window_length=1000)
        assert bam.info["filtered_pairs"] == sum(t["filtered_pairs"] for t in tallies.values()) > 300
        assert bam.info["unfiltered_pairs"] == sum(t["unfiltered_pairs"] for t in tallies.values())
        P, B, M, R = [], [], [], []
        off = 0
        nid = 0
        for t, (name, ln) in enumerate(refs):
            bam_py.resolve_overlaps(rr, t)
            pos, base, mm, pr, n2i = bam_py.expand_observations(rr, t, r2m[name], ref_len=ln)
            P.append(pos + off); B.append(base); M.append(mm); R.append(pr + nid)
            off += ln
            nid += len(n2i)
        assert len(obs) == sum(len(x) for x in P) > 20000
        assert (obs["gpos"] == np.concatenate(P)).all() and (obs["base"] == np.concatenate(B)).all()
        assert (obs["mm"] == np.concatenate(M)).all() and (pair == np.concatenate(R)).all()
        from instrain_amd import synth
        assert list(bounds) == list(synth.split_bounds_for([r[1] for r in refs], 1000))
        bam.close()


def test_error_paths_and_empty_inputs(tmp_path):
    """truncated file, a read without NM, a BAM with no reads at all: loud errors / empty outputs"""
    from tests import bamwriter
    refs = [("s", 500)]
    reads = bamwriter.random_reads(5, refs, 40)
    good = str(tmp_path / "good.bam")
    bamwriter.write_bam(good, refs, reads)
    raw = open(good, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(engine.IsxError) as e:
        engine.BamFile(cut)
    assert e.value.code == -5
    nonm = str(tmp_path / "nonm.bam")
    rr = [dict(r) for r in reads]
    for r in rr:
        r["nm"] = None
        r["extra_tags"] = False
    bamwriter.write_bam(nonm, refs, rr)
    b = engine.BamFile(nonm)
    with pytest.raises(engine.IsxError) as e:
        b.expand()
    assert e.value.code == -5 and "NM" in str(e.value)
    b.close()
    empty = str(tmp_path / "empty.bam")
    bamwriter.write_bam(empty, refs, [])
    b = engine.BamFile(empty)
    obs, pair, bounds, sref = b.expand()
    assert len(obs) == 0 and len(pair) == 0 and list(bounds) == [0, 500] and b.info["filtered_pairs"] == 0
    o2, p2, _, _ = engine.BamFile(empty).expand(copy=False)
    assert len(o2) == 0 and len(p2) == 0
    b.close()
