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
    o2, p2, _, _ = bam.expand()                       # a second pass re-reads the file: same visits, now with mm
    assert (o2["gpos"] == obs["gpos"]).all() and (o2["base"] == obs["base"]).all() and (p2 == pair).all() and o2["mm"].max() > 0
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


def test_pairing_filter_modes_match_the_reference():
    """paired_only / non_discordant / all_reads, with and without --priority_reads: the scaffold -> {pair: mm}
    dictionaries and read-report tallies that the reference's own paired_read_filter + filter_scaff2pair2info
    produce (tests/golden/make_filter_golden.py) on filter_modes.bam"""
    import json
    g = json.load(open(os.path.join(util.GOLD, "filter_modes.json")))
    bam = engine.BamFile(os.path.join(util.GOLD, "filter_modes.bam"))
    assert [(n, l) for n, l, _ in bam.refs()] == [tuple(r) for r in g["refs"]]
    bam.scan()
    seen = set()
    for case in g["cases"]:
        assert not case.get("keyerror")
        bam.set_priority_reads(g["priority"] if case["priority"] else [])
        info = bam.filter(pairing_filter=case["mode"], **case["params"])
        for k, v in case["tallies"].items():
            assert info[k] == v, (case["mode"], case["priority"], k, info[k], v)
        assert info["median_insert"] == case["median_insert"]
        for t, (name, _, _) in enumerate(bam.refs()):
            assert bam.r2m(t) == case["r2m"].get(name, {}), (case["mode"], case["priority"], name)
        seen.add((case["mode"], case["priority"]))
    assert len(seen) == 6
    # the modes really differ on this file
    by = {(c["mode"], c["priority"]): sum(len(d) for d in c["r2m"].values()) for c in g["cases"]}
    assert by[("paired_only", False)] < by[("paired_only", True)] < by[("non_discordant", False)]
    bam.close()


def test_expand_refs_subsets_and_controller_r2m(tmp_path):
    """pass 2 on subsets of the references == the whole-file expansion cut per reference; and the controller's own
    R2M (isx_bam_set_r2m) instead of the built-in filter: exactly the named pairs, with the given mm"""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafB", 900), ("scafC", 12500), ("empty", 700)]
    path = str(tmp_path / "m.bam")
    reads = bamwriter.random_reads(11, refs[:3], 6000)
    bamwriter.write_bam(path, refs, reads)
    whole = engine.BamFile(path)
    obs, pair, bounds, sref = whole.expand(min_read_ani=0.9, window_length=1000)
    offs = np.r_[0, np.cumsum([r[1] for r in refs])]
    bam = engine.BamFile(path, threads=3)
    bam.scan()
    bam.filter(min_read_ani=0.9)
    reads_per_ref, pairs_per_ref = bam.ref_counts()
    assert reads_per_ref[3] == 0 and pairs_per_ref.sum() == bam.info["filtered_pairs"] == whole.info["filtered_pairs"]
    with pytest.raises(engine.IsxError):
        bam.expand_refs([1, 0])                          # file order only
    for sel in ([2], [0, 1], [3], [0, 1, 2, 3], [2, 3], [0, 2]):
        o, p, b, s = bam.expand_refs(sel, min_read_ani=0.9, window_length=1000)
        exp_parts, at = [], 0
        for t in sel:
            k = (obs["gpos"] >= offs[t]) & (obs["gpos"] < offs[t + 1])
            e = obs[k].copy()
            e["gpos"] = e["gpos"] - offs[t] + at
            exp_parts.append((e, pair[k]))
            at += refs[t][1]
        e = np.concatenate([x[0] for x in exp_parts])
        assert len(o) == len(e) and (o == e).all(), sel
        # pair ids: dense in order of first appearance inside the batch -> same partition of the observations
        ep = np.concatenate([x[1] + 10_000_000 * i for i, x in enumerate(exp_parts)])
        _, a = np.unique(p, return_inverse=True)
        _, c = np.unique(ep, return_inverse=True)
        first_a = np.full(a.max() + 1 if len(a) else 0, -1); first_c = np.full(c.max() + 1 if len(c) else 0, -1)
        assert len(first_a) == len(first_c)
        for arr, first in ((a, first_a), (c, first_c)):
            idx = np.arange(len(arr))[::-1]
            first[arr[::-1]] = idx
        assert (first_a[a] == first_c[c]).all(), sel
        assert b[-1] == at and list(s) == [t for t in sel for _ in range(refs[t][1] // 1000 + 1)]
    # controller-supplied R2M: the oracle's filter result with every mm raised by 2, half of scafC's pairs only
    rrefs, rr = bam_py.read_bam(path)
    p2i = {r[0]: bam_py.get_paired_reads(rr, t) for t, r in enumerate(rrefs)}
    r2m, _ = bam_py.filter_pairs(p2i, min_read_ani=0.9)
    assert bam.r2m(2) == r2m["scafC"] and bam.r2m(0) == r2m["scafA"]
    keep = dict(list(r2m["scafC"].items())[::2])
    bam.set_r2m(2, list(keep), [m + 2 for m in keep.values()])
    o, p, b, s = bam.expand_refs([2], min_read_ani=0.9)
    bam_py.resolve_overlaps(rr, 2)
    pos, base, mm, pr, _ = bam_py.expand_observations(rr, 2, {n: m + 2 for n, m in keep.items()}, ref_len=12500)
    assert len(o) == len(pos) > 1000 and (o["gpos"] == pos).all() and (o["base"] == base).all() and (o["mm"] == mm).all()
    assert bam.info["max_mm"] >= max(keep.values()) + 2
    whole.close(); bam.close()


def test_mm_levels_of_any_size_travel_as_ranks(tmp_path):
    """isx_bam_mm_levels / isx_bam_set_mm_levels (round 6): the reference bins any mm (profile_utilities.py:268-286) and only the ORDER
    of the levels enters its tables (:297-312).  A controller's R2M with values in the hundreds: the distinct values come back
    ascending, and with them set every observation carries the rank of its pair's value -- the very stream of an R2M that holds the
    ranks themselves; switched off again, the values themselves; a cap still merges the ranks beyond it; values that do not ascend
    are refused"""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafC", 9000)]
    path = str(tmp_path / "lv.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(23, refs, 4000))
    bam = engine.BamFile(path, threads=3)
    bam.scan()
    bam.filter(min_read_ani=0.9)
    r2m = bam.r2m(1)
    names = list(r2m)
    big = {n: (700 + 3 * (i % 40) if i % 3 == 0 else int(r2m[n])) for i, n in enumerate(names)}
    bam.set_r2m(1, names, [big[n] for n in names])
    bam.set_r2m(0, [], [])
    bam.scan()
    levels = bam.mm_levels()
    assert list(levels) == sorted(set(big.values())) and levels[-1] == 817 and len(levels) < 128
    rank = {int(v): i for i, v in enumerate(levels)}
    o_val, p_val, _, _ = bam.expand_refs([1], min_read_ani=0.9)
    assert int(o_val["mm"].max()) == 817
    bam.set_mm_levels(levels)
    o_rank, p_rank, _, _ = bam.expand_refs([1], min_read_ani=0.9)
    assert (o_rank["gpos"] == o_val["gpos"]).all() and (o_rank["base"] == o_val["base"]).all() and (p_rank == p_val).all()
    assert (levels[o_rank["mm"].astype(np.int64)] == o_val["mm"]).all() and int(o_rank["mm"].max()) == len(levels) - 1
    # the stream of an R2M that says the ranks
    bam.set_mm_levels([])
    bam.set_r2m(1, names, [rank[big[n]] for n in names])
    o_small, _, _, _ = bam.expand_refs([1], min_read_ani=0.9)
    assert (o_small == o_rank).all()
    # a cap merges the ranks beyond it
    bam.set_r2m(1, names, [big[n] for n in names])
    bam.set_mm_levels(levels)
    bam.set_mm_cap(10)
    o_cap, _, _, _ = bam.expand_refs([1], min_read_ani=0.9)
    assert (o_cap["mm"] == np.minimum(o_rank["mm"], 10)).all()
    with pytest.raises(engine.IsxError, match="ascend"):
        bam.set_mm_levels([3, 3, 9])
    bam.close()


def test_corrupt_inputs_are_errors_not_crashes(tmp_path):
    """truncated / corrupted inflated streams: bounds-checked, reported as ISX_ERR_IO"""
    import struct, zlib
    from tests import bamwriter
    refs = [("s", 3000)]
    reads = bamwriter.random_reads(9, refs, 300)
    good = str(tmp_path / "g.bam")
    bamwriter.write_bam(good, refs, reads)
    import gzip
    raw = gzip.open(good).read()

    def rewrite(data, name):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            for i in range(0, len(data), 60000):
                f.write(bamwriter._bgzf_block(data[i:i + 60000]))
            f.write(bamwriter._bgzf_block(b""))
        return path

    hdr_end = 12 + struct.unpack("<i", raw[4:8])[0]
    hdr_end += 4 + 8 + struct.unpack("<i", raw[hdr_end + 4:hdr_end + 8])[0]
    cases = {"tail.bam": raw[:-11],                                                  # last record cut short
             "badlen.bam": raw[:hdr_end] + struct.pack("<i", 10_000_000) + raw[hdr_end + 4:],       # block_size beyond the file
             "neglen.bam": raw[:hdr_end] + struct.pack("<i", -5) + raw[hdr_end + 4:],
             "lseq.bam": raw[:hdr_end + 20] + struct.pack("<i", 1 << 30) + raw[hdr_end + 24:],       # l_seq beyond the record
             "refs.bam": raw[:8 + struct.unpack("<i", raw[4:8])[0]] + struct.pack("<i", 1 << 28)}    # n_ref without references
    for name, data in cases.items():
        path = rewrite(data, name)
        with pytest.raises(engine.IsxError) as e:
            engine.BamFile(path).expand()
        assert e.value.code == -5, name
    # an aux field that runs past its record
    r2 = [dict(r) for r in reads[:40]]
    path = str(tmp_path / "aux.bam")
    bamwriter.write_bam(path, refs, r2)
    data = bytearray(gzip.open(path).read())
    i = data.find(b"XSZhello")
    assert i > 0
    data[i + 3:i + 9] = b"hellox"                      # the Z string loses its terminator inside the record
    with pytest.raises(engine.IsxError) as e:
        engine.BamFile(rewrite(bytes(data), "aux2.bam")).expand()
    assert e.value.code == -5


def test_record_longer_than_a_segment(tmp_path):
    """a 2.3 Mbp read (one BAM record spanning several 1 MiB segments of the scan) between ordinary pairs: the segments
    inside it hold no record start -- whatever the first-record guesser finds in its bases is overruled by the chain check"""
    from tests import bamwriter
    refs = [("big", 4_000_000)]
    rng = np.random.Generator(np.random.PCG64(3))
    reads = bamwriter.random_reads(31, [("big", 100_000)], 400)
    L = 2_300_000
    long_read = dict(tid=0, pos=150_000, mapq=40, flag=0x1 | 0x40 | 0x2, name="longread", cigar=[("M", L)],
                     seq="".join(rng.choice(list("ACGT"), L)), qual=np.full(L, 40), nm=3, isize=L + 500, extra_tags=False)
    mate = dict(long_read, pos=150_000 + L + 100, flag=0x1 | 0x80 | 0x2, cigar=[("M", 100)], seq="A" * 100, qual=np.full(100, 40),
                nm=0, isize=-(L + 500))
    tail = bamwriter.random_reads(32, [("big", 100_000)], 300)
    for r in tail:
        r["pos"] += 3_000_000
        r["name"] = "t" + r["name"]
    reads = sorted(reads + [long_read, mate] + tail, key=lambda r: (r["tid"], r["pos"]))
    path = str(tmp_path / "long.bam")
    bamwriter.write_bam(path, refs, reads)
    bam = engine.BamFile(path, threads=4)
    info = bam.scan()
    assert info["n_reads"] == len(reads)
    bam.filter(min_read_ani=0.5, max_insert_relative=1e9)
    assert bam.r2m(0).get("longread") == 3
    obs, pair, bounds, sref = bam.expand_refs([0], min_read_ani=0.5, max_insert_relative=1e9)
    k = obs[pair == pair[np.flatnonzero(obs["gpos"] == 150_000)[0]]]
    assert len(k) == L + 100 and (np.diff(k["gpos"][:L].astype(np.int64)) == 1).all() and (k["mm"] == 3).all()
    # the same file through one thread (serial chain) gives the same stream
    b1 = engine.BamFile(path, threads=1)
    o1, p1, _, _ = b1.expand(min_read_ani=0.5, max_insert_relative=1e9)
    assert len(o1) == len(obs) and (o1 == obs).all() and (p1 == pair).all()
    bam.close(); b1.close()


def test_expand_region_equals_the_cut_of_the_whole_reference(tmp_path):
    """samfile.pileup(scaffold, start, stop, truncate=True): the region's columns from the reads that overlap them ==
    the whole reference's stream restricted to those positions (an overlap tweak only ever concerns positions both mates
    cover, so the reads outside the region cannot matter)"""
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafB", 30000)]
    path = str(tmp_path / "r.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(51, refs, 20000))
    bam = engine.BamFile(path, threads=4)
    bam.scan(); bam.filter(min_read_ani=0.9)
    o, p, _, _ = bam.expand_refs([1], min_read_ani=0.9)
    for lo, hi in ((0, 30000), (12000, 12001), (29990, 40000), (5000, 9000), (100, 260)):
        ro, rp, rb, rs = bam.expand_region(1, lo, hi, min_read_ani=0.9)
        k = (o["gpos"] >= lo) & (o["gpos"] < hi)
        assert len(ro) == k.sum() and (ro == o[k]).all(), (lo, hi)
        # pair ids are dense per expansion: same partition of the observations
        _, a = np.unique(rp, return_inverse=True)
        _, c = np.unique(p[k], return_inverse=True)
        fa = np.zeros(a.max() + 1 if len(a) else 0, int); fc = np.zeros(c.max() + 1 if len(c) else 0, int)
        fa[a[::-1]] = np.arange(len(a))[::-1]; fc[c[::-1]] = np.arange(len(c))[::-1]
        assert (fa[a] == fc[c]).all()
    with pytest.raises(engine.IsxError):
        bam.expand_region(1, 10, 10)
    bam.close()


def test_pair_tables_over_several_name_partitions(tmp_path):
    """a reference with more than 16384 reads builds its name -> pair table in several hash partitions (reads bucketed
    by partition first, tables built in parallel, then laid end to end): same pairs, same R2M, same tallies as
    the oracle's single dictionary, for any thread count"""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("small", 3000), ("big", 60_000)]
    reads = bamwriter.random_reads(77, refs[1:], 17_500)
    for r in reads:                                   # random_reads numbered the one reference it was given 0
        r["tid"] = 1
        r["name"] = "b" + r["name"]
    reads = bamwriter.random_reads(78, refs[:1], 200) + reads
    path = str(tmp_path / "parts.bam")
    bamwriter.write_bam(path, refs, reads)
    rrefs, rr = bam_py.read_bam(path)
    p2i = {r[0]: bam_py.get_paired_reads(rr, t) for t, r in enumerate(rrefs)}
    r2m, tallies = bam_py.filter_pairs(p2i, min_read_ani=0.9)
    for threads in (1, 5):
        bam = engine.BamFile(path, threads=threads)
        bam.scan()
        info = bam.filter(min_read_ani=0.9)
        assert info["n_reads"] == len(reads) > 2 * 16384
        assert info["unfiltered_pairs"] == sum(t["unfiltered_pairs"] for t in tallies.values())
        assert info["filtered_pairs"] == sum(t["filtered_pairs"] for t in tallies.values()) > 5000
        assert bam.r2m(1) == r2m["big"] and bam.r2m(0) == r2m["small"]
        bam.close()


@pytest.mark.parametrize("n_parts", [2, 3, 7])
def test_scan_in_shares_equals_the_whole_file_scan(tmp_path, n_parts):
    """isx_bam_scan_part: every share owns the references whose first read lies in its segments; over all shares
    every reference with reads is owned exactly once, its pair table / R2M / expansion are those of the whole-file scan,
    and the insert sizes of all shares together give the whole file's median (the one collective of a multi-rank run)"""
    from tests import bamwriter
    refs = [("s%d" % i, ln) for i, ln in enumerate([4000, 900, 12500, 700, 2600, 5100, 300, 8000])]
    path = str(tmp_path / "shares.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(61, refs[:7], 9000))
    whole = engine.BamFile(path, threads=3)
    whole.scan()
    w_ins = np.sort(whole.insert_sizes())
    w_info = whole.filter(min_read_ani=0.9)
    w_reads, w_pairs = whole.ref_counts()
    o, p, b, s = whole.expand(min_read_ani=0.9)
    offs = np.r_[0, np.cumsum([r[1] for r in refs])]
    owners = np.zeros(len(refs), int)
    all_ins = []
    shares = []
    for part in range(n_parts):
        bam = engine.BamFile(path, threads=2)
        bam.scan(part=(part, n_parts))
        all_ins.append(bam.insert_sizes())
        shares.append(bam)
    all_ins = np.sort(np.concatenate(all_ins))
    assert len(all_ins) == len(w_ins) and (all_ins == w_ins).all()
    median = float(np.median(all_ins))
    assert median == w_info["median_insert"]
    n_pairs = 0
    for part, bam in enumerate(shares):
        with pytest.raises(engine.IsxError):
            bam.filter(min_read_ani=0.9)                        # a share cannot know the file's median insert
        with pytest.raises(engine.IsxError):
            bam.filter(median_insert=median, min_read_ani=0.9, pairing_filter="all_reads")
        info = bam.filter(median_insert=median, min_read_ani=0.9)
        reads, pairs = bam.ref_counts()
        mine = np.flatnonzero(reads)
        owners[mine] += 1
        assert (reads[mine] == w_reads[mine]).all() and (pairs[mine] == w_pairs[mine]).all(), part
        n_pairs += int(pairs.sum())
        for t in mine:
            assert bam.r2m(int(t)) == whole.r2m(int(t)), (part, t)
        if len(mine):
            oo, pp, bb, ss = bam.expand_refs([int(t) for t in mine], min_read_ani=0.9)
            exp, at = [], 0
            for t in mine:
                k = (o["gpos"] >= offs[t]) & (o["gpos"] < offs[t + 1])
                e = o[k].copy()
                e["gpos"] = e["gpos"] - offs[t] + at
                exp.append(e)
                at += refs[t][1]
            exp = np.concatenate(exp)
            assert len(oo) == len(exp) and (oo == exp).all(), part
        bam.close()
    assert (owners[w_reads > 0] == 1).all() and (owners[w_reads == 0] == 0).all(), owners
    assert n_pairs == int(w_pairs.sum())
    whole.close()


def test_share_scan_argument_and_state_errors(tmp_path):
    from tests import bamwriter
    refs = [("a", 3000), ("b", 2000)]
    path = str(tmp_path / "e.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(3, refs, 400))
    bam = engine.BamFile(path, threads=2)
    for part in ((2, 2), (-1, 2), (0, 0)):
        with pytest.raises(engine.IsxError):
            bam.scan(part=part)
    bam.scan(part=(0, 2))
    bam.scan(part=(0, 2))                                # the same share again: a no-op
    with pytest.raises(engine.IsxError):
        bam.scan(part=(1, 2))                            # one handle, one share
    with pytest.raises(engine.IsxError):
        bam.scan()                                       # ... and not the whole file either
    bam.close()
    # more shares than the file has segments: the surplus shares own nothing, the others everything once
    owned = np.zeros(2, int)
    for part in range(40):
        b = engine.BamFile(path, threads=1)
        b.scan(part=(part, 40))
        owned += (b.ref_counts()[0] > 0)
        b.close()
    assert (owned == 1).all()


@pytest.mark.parametrize("mode", ["paired_only", "non_discordant", "all_reads"])
def test_filter_sees_only_the_scaffolds_of_the_fasta(tmp_path, mode):
    """a BAM mapped to a larger database than the fasta: the reference builds its pair table from the fasta's scaffolds only
    (filter_reads.py:63-77, 157-178), so the median insert, the tallies and the cross-scaffold name look-ups ignore the
    other references (isx_bam_set_wanted_refs): the filter's decisions equal those on a BAM that holds only the wanted
    scaffolds' reads -- and, for paired_only, the oracle's restatement of the reference's filter on those scaffolds"""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("in1", 5000), ("out", 6000), ("in2", 3000)]
    reads = bamwriter.random_reads(5, refs, 2500)
    path, path_sub = str(tmp_path / "db.bam"), str(tmp_path / "sub.bam")
    bamwriter.write_bam(path, refs, reads)
    wanted = [0, 2]
    rrefs, rr = bam_py.read_bam(path)
    bamwriter.write_bam(path_sub, refs, [r for r in reads if r["tid"] in wanted])
    a = engine.BamFile(path, threads=2)
    a.scan()
    a.set_wanted_refs(wanted)
    ia = a.filter(min_read_ani=0.9, pairing_filter=mode)
    b = engine.BamFile(path_sub, threads=2)
    b.scan()
    ib = b.filter(min_read_ani=0.9, pairing_filter=mode)
    for k in ("filtered_pairs", "unfiltered_pairs", "unfiltered_reads", "unfiltered_singletons", "filtered_singletons", "median_insert", "filtered_bases"):
        assert ia[k] == ib[k], k
    assert ia["filtered_pairs"] > 100
    for t in range(3):
        assert a.r2m(t) == b.r2m(t)
    assert a.r2m(1) == {}
    assert (np.sort(a.insert_sizes()) == np.sort(b.insert_sizes())).all()
    # without the restriction the other scaffold's pairs count: the restriction matters
    a.set_wanted_refs(None)
    assert len(a.insert_sizes()) > len(b.insert_sizes())
    if mode == "paired_only":
        p2i = {refs[t][0]: bam_py.get_paired_reads(rr, t) for t in wanted}
        r2m, tallies = bam_py.filter_pairs(p2i, min_read_ani=0.9)
        assert ib["filtered_pairs"] == sum(t["filtered_pairs"] for t in tallies.values())
        for t in wanted:
            assert b.r2m(t) == r2m[refs[t][0]]
    a.close()
    b.close()


def test_cigar_in_the_cg_tag_reads_like_the_cigar_itself(tmp_path):
    """a CIGAR of more than 65535 operations is stored as <l_seq>S<ref_len>N + CG:B,I (SAM spec 4.2.2); htslib puts it
    back when it reads the record (bam_tag2cigar), so pysam and the reference never see the placeholder: the same reads
    written both ways must scan, filter and expand identically -- as observations and as read segments"""
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafB", 900), ("scafC", 12500)]
    reads = bamwriter.random_reads(5, refs, 5000)
    plain, tagged = str(tmp_path / "plain.bam"), str(tmp_path / "tagged.bam")
    bamwriter.write_bam(plain, refs, reads)
    bamwriter.write_bam(tagged, refs, reads, cg_every=3)
    out = []
    for path in (plain, tagged):
        bam = engine.BamFile(path)
        obs, pair, bounds, sref = bam.expand(min_read_ani=0.9, window_length=1000)
        info = dict(bam.info)
        seg = bam.segment_refs(np.arange(3), window_length=1000, min_read_ani=0.9)
        out.append((obs, pair, info, seg))
        bam.close()
    (o0, p0, i0, s0), (o1, p1, i1, s1) = out
    assert len(o0) > 20000 and (o0 == o1).all() and (p0 == p1).all()
    assert i0["filtered_pairs"] == i1["filtered_pairs"] > 300 and i0["n_reads"] == i1["n_reads"]
    assert s0[0].n_seg == s1[0].n_seg > 1000
    for f in ("gpos", "len", "mm", "pair", "bases"):
        assert (getattr(s0[0], f) == getattr(s1[0], f)).all(), f


def test_max_depth_of_the_pileup_call(tmp_path):
    """max_depth=100000 (profile_utilities.py:150) as htslib 1.9 applies it: of a run of reads with the SAME start, those pushed
    while the pileup buffer already holds 100 000 reads are dropped -- here a pile of 100 050 reads starting at one position over 5
    older reads that still cover it: 99 995 of the pile are taken, the column is exactly 100 000 deep.  PARITY UNPINNED: the
    reference holds no fixture this deep (and pysam is not in this image); the C++ front end is checked against the oracle's
    restatement of the published htslib rule and against the arithmetic above."""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("deep", 2000)]
    L = 20
    q = np.full(L, 37, np.uint8)
    reads = []

    def pair(name, s1, s2):
        for mate, (s, ms) in enumerate(((s1, s2), (s2, s1))):
            reads.append(dict(tid=0, pos=s, mapq=40, flag=0x1 | 0x2 | (0x40 if mate == 0 else 0x80) | (0x20 if mate == 0 else 0x10),
                              isize=(s2 + L - s1) * (1 if mate == 0 else -1), name=name, cigar=[("M", L)], seq="ACGT" * (L // 4), qual=q, nm=0,
                              mtid=0, mpos=ms))
    for i in range(5):
        pair("old%d" % i, 485 + i, 1100 + i)                    # cover 485 .. 508: still buffered when the pile arrives
    N = 100_050
    for i in range(N):
        pair("pile%d" % i, 500, 1200)
    for i in range(20):
        pair("next%d" % i, 501, 1300 + i)
    reads.sort(key=lambda r: r["pos"])                          # stable: file order inside a start = creation order
    path = str(tmp_path / "deep.bam")
    bamwriter.write_bam(path, refs, reads)
    bam = engine.BamFile(path)
    obs, pair_id, bounds, sref = bam.expand(min_read_ani=0.9, window_length=10000)
    bam.close()
    cov = np.bincount(obs["gpos"], minlength=2000)
    assert cov[500] == 100_000 and cov[499] == 5 and cov[484] == 0           # 5 older + 99 995 of the pile
    # the 20 reads starting at 501 arrive while the pile is still buffered: the first of the run is always pushed (the iterator
    # does not stand on 501 yet), the other 19 meet a full buffer
    assert cov[519] == 99_995 + 1 and cov[520] == 1
    assert cov[1200] == 100_000 and cov[1219] == 100_000                     # the mates: a run of 100 050 over an empty buffer
    # the oracle's restatement, observation for observation
    rrefs, rr = bam_py.read_bam(path)
    r2m, _ = bam_py.filter_pairs({"deep": bam_py.get_paired_reads(rr, 0)}, min_read_ani=0.9)
    assert bam_py.apply_max_depth(rr, 0) == (N - 99_995) + 19 + (N - 100_000)
    bam_py.resolve_overlaps(rr, 0)
    pos, base, mm, pr, _ = bam_py.expand_observations(rr, 0, r2m["deep"], ref_len=2000)
    assert len(pos) == len(obs) and (obs["gpos"] == pos).all() and (obs["base"] == base).all() and (pair_id == pr).all()


def test_max_depth_is_replayed_per_split(tmp_path):
    """The reference opens ONE pileup iterator per split (profile_utilities.py:150-153; splits from fasta.py:56-73), each with its own
    max_depth buffer fed by the reads that overlap the split.  A 25 000-base scaffold at window_length 10 000 has splits [0, 8332],
    [8333, 16665], [16666, 24999].  50 000 reads cover 8310..8329 (split 0 only) and 60 000 more cover 8320..8339 (both splits): split
    0's iterator takes 50 000 of the second pile (its buffer holds the first), split 1's iterator never sees the first pile and takes
    all 60 000 -- the last 10 000 reads of the second pile are present from column 8333 on and absent before it.  (A replay over the
    whole scaffold -- what rounds 4-5 did -- leaves 50 000 at 8333..8339.)  Checked against the arithmetic and against the oracle's
    literal per-split restatement, position by position and base by base.  PARITY UNPINNED (no fixture this deep; no pysam here)."""
    from oracle import bam_py
    from tests import bamwriter
    refs = [("deep", 25000)]
    assert bam_py.iterate_splits(25000, 10000) == [(0, 8332), (8333, 16665), (16666, 24999)]
    L = 20
    q = np.full(L, 37, np.uint8)
    reads = []

    def pair(name, s1, s2, seq="ACGT" * (L // 4)):
        for mate, (s, ms) in enumerate(((s1, s2), (s2, s1))):
            reads.append(dict(tid=0, pos=s, mapq=40, flag=0x1 | 0x2 | (0x40 if mate == 0 else 0x80) | (0x20 if mate == 0 else 0x10),
                              isize=(s2 + L - s1) * (1 if mate == 0 else -1), name=name, cigar=[("M", L)], seq=seq, qual=q, nm=0,
                              mtid=0, mpos=ms))
    for i in range(50_000):
        pair("first%d" % i, 8310, 20000)
    for i in range(60_000):
        pair("second%d" % i, 8320, 20100, seq="TTGCA" * (L // 5))
    for i in range(7):
        pair("plain%d" % i, 8328 + i, 21000 + i)                # arrive while both piles are buffered; some straddle the bound
    reads.sort(key=lambda r: r["pos"])
    path = str(tmp_path / "deep_split.bam")
    bamwriter.write_bam(path, refs, reads)
    bam = engine.BamFile(path)
    obs, pair_id, bounds, sref = bam.expand(min_read_ani=0.9, window_length=10000)
    bam.close()
    assert list(bounds) == [0, 8333, 16666, 25000]
    cov = np.bincount(obs["gpos"], minlength=25000)
    plain = np.zeros(25000, np.int64)
    for i in range(7):
        plain[8328 + i:8328 + i + L] += 1
    assert (cov[8310:8320] == 50_000).all()
    assert (cov[8320:8328] == 100_000).all()                                # 50 000 + the 50 000 of the second pile split 0 takes
    # the plain reads start at 8328..8334 on columns the iterator does not stand on yet (each start is a new run): all taken
    assert (cov[8328:8330] == 100_000 + plain[8328:8330]).all()
    assert (cov[8330:8333] == 50_000 + plain[8330:8333]).all()              # the first pile has ended; split 0 still lacks 10 000 of the second
    assert (cov[8333:8340] == 60_000 + plain[8333:8340]).all()              # split 1's iterator took the whole second pile
    assert (cov[20000:20020] == 50_000).all() and (cov[20100:20120] == 60_000).all()        # the mates: nothing else buffered there
    # the oracle's literal per-split replay
    rrefs, rr = bam_py.read_bam(path)
    r2m, _ = bam_py.filter_pairs({"deep": bam_py.get_paired_reads(rr, 0)}, min_read_ani=0.9)
    pos, base, mm, pr = bam_py.expand_observations_per_split(rr, 0, r2m["deep"], bam_py.iterate_splits(25000, 10000), ref_len=25000)
    assert len(pos) == len(obs)
    want = np.zeros((25000, 5), np.int64)
    np.add.at(want, (pos, base), 1)
    got = np.zeros((25000, 5), np.int64)
    np.add.at(got, (obs["gpos"].astype(np.int64), obs["base"].astype(np.int64)), 1)
    assert (want == got).all()
    # a read pair's observations: the same multiset of per-pair counts
    assert sorted(np.bincount(pr).tolist()) == sorted(np.bincount(pair_id).tolist())
