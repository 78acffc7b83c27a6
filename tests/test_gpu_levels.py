"""GPU: mm profiling on (the reference's default, argumentParser.py:131) through a read-level pipe -- the LEVEL-SPARSE hand-back
(isx_pipe_result.lev_*: level mask per position, one coverage element per present level, lists of the clonalities that are not 1.0)
must say exactly what the 32-byte entries say: covT = a level's own coverage (profile_utilities.py:288-295), clonT / clonTR of the
counts up to the level (snv_utilities.py:85-104), a level made present by a non-ACGT base alone (profile_utilities.py:279-285).
Checked against the one-shot batch (itself pinned by the reference's golden vectors in test_gpu_parity / test_gpu_reads), through
plain slots (entries kept, flat), lean slots (nothing but the level tables) and the round-2 entry slabs (ISX_LAYOUT_MM_ENTRIES)."""
import glob
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

MM_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(util.GOLD, "synth_*.npz"))
                  if int(np.load(p)["mm"].max()) > 0)
MM_ENTRIES = 16     # isx_params.layout: ISX_LAYOUT_MM_ENTRIES
MMDELTA = 32        # ISX_LAYOUT_MM_DELTA_RECORDS: reference-delta records with the mm level in the header instead of segment records
NOPACK = 4          # ISX_LAYOUT_NO_PACKED_COUNTERS


@pytest.fixture(scope="module")
def ctx():
    from instrain_amd import engine
    c = engine.Context(0)
    lut, fb = util.load_lut()
    c.set_null_model(lut, fb)
    yield c
    c.close()


def soa_of_entries(e):
    return (e["gpos"].astype(np.uint32), (e["mm"].astype(np.uint32) << 24) | e["cnt"].sum(axis=1).astype(np.uint32),
            e["clon"].astype(np.float32), e["clon_rarefied"].astype(np.float32))


def same_soa(got, exp, what):
    assert len(got[0]) == len(exp[0]), (what, len(got[0]), len(exp[0]))
    for k, name in enumerate(("gpos", "mm_cov", "clon", "clon_rarefied")):
        assert got[k].tobytes() == exp[k].tobytes(), (what, name, np.flatnonzero(got[k].view(np.uint32) != exp[k].view(np.uint32))[:5])


def through_pipe(ctx, ref, bounds, segs, M, lean=False, layout=0, depth=1, planes=False, **kw):
    """-> (the four columns, the result dict's level tables or None, the full entries or None, snv rows, ld rows)"""
    from instrain_amd import engine
    # (delta records: a segment over a non-ACGT stretch of the reference is many pieces -- room for them, as a caller with such data would give)
    pipe = engine.Pipe(ctx, max_pos=len(ref), max_obs=0, max_segs=max(1, segs.n_seg), max_splits=len(bounds), depth=depth, host_threads=3,
                       pin_threads=False, n_mm_bins=M, lean_output=lean, layout=layout, jump_slack=8.0 if layout & MMDELTA else 0.0, **kw)
    if planes:          # the same reads as bit planes + their mm levels (isx_read_planes.mm): the XOR stager
        t = pipe.submit_planes(engine.RefPlanes.from_codes(ref), bounds, engine.PlaneBatch.from_segs(segs))
    else:
        t = pipe.submit_reads(ref, bounds, segs)
    r = pipe.collect(t, shrunk_entries=True)
    soa = tuple(c.copy() for c in r["entries_soa"])
    lev = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in r["levels"].items()} if "levels" in r else None
    snv, ld = r["snv"].copy(), r["ld"].copy()
    full = None
    if not lean:
        full = pipe.collect(t)["entries"].copy()
    else:
        with pytest.raises(engine.IsxError):        # a lean slot keeps no 32-byte entries
            pipe.collect(t)
    pipe.release(t)
    pipe.close()
    return soa, lev, full, snv, ld


@pytest.mark.parametrize("name", MM_CASES)
def test_golden_vectors_level_sparse(ctx, name):
    """every reference-generated vector with mm > 0: level tables of a plain slot, of a lean slot and the round-2 entry slabs give the
    one-shot batch's entries (pinned against the reference in test_gpu_reads) column by column, bit by bit; SNV / LD rows alike"""
    from instrain_amd import engine
    g = util.load_case(name)
    kw = dict(min_cov=int(g["p_min_cov"]), min_freq=float(g["p_min_freq"]), min_snp=int(g["p_min_snp"]), enable_linkage=True, rarefied_coverage=8, seed=3)
    pos = np.asarray(g["pos"], dtype=np.int64)
    start, seq = int(g["start"]), str(g["seq"])
    sel = (pos >= start) & (pos < start + len(seq))
    mm = np.asarray(g["mm"])[sel]
    M = int(mm.max()) + 1
    segs = util.reassemble_segs((pos[sel] - start).astype(np.uint32), np.asarray(g["base"])[sel], mm, np.asarray(g["pair"])[sel].astype(np.uint32))
    ref = engine.encode_seq(seq)
    b = engine.Batch(ctx, ref, [0, len(ref)], segs, None, n_mm_bins=M, **kw)
    b.run()
    exp = b.fetch()
    b.close()
    want = soa_of_entries(exp["entries"])
    assert np.isfinite(want[3]).any() or int(exp["entries"]["cnt"].sum(axis=1).max()) < 8
    for how, lean, layout in (("plain", False, 0), ("lean", True, 0), ("slabs", False, MM_ENTRIES), ("plain delta records", False, MMDELTA), ("lean delta records", True, MMDELTA),
                              ("lean 32-bit counters", True, NOPACK), ("lean delta records 32-bit counters", True, MMDELTA | NOPACK), ("slabs delta records", False, MM_ENTRIES | MMDELTA)):
        soa, lev, full, snv, ld = through_pipe(ctx, ref, [0, len(ref)], segs, M, lean=lean, layout=layout, planes="delta records" in how and "slabs" not in how, **kw)
        same_soa(soa, want, name + "/" + how)
        assert (lev is None) == (bool(layout & MM_ENTRIES) or M > 32), (name, how)
        if full is not None:
            assert full.tobytes() == exp["entries"].tobytes(), (name, how, "entries")
        assert snv.tobytes() == exp["snv"].tobytes() and ld.tobytes() == exp["ld"].tobytes(), (name, how, "rows")


def test_level_tables_of_a_stream(ctx):
    """several batches in flight through lean slots (windows allot their level ranges in whatever order they get to the cursor): every
    batch's level tables == its one-shot entries; references with positions that are not A/C/T/G; coverage beyond 255 at single
    levels of a shallow batch (the saturation list), a deep batch (two-byte coverage), many mm bins (two- and four-byte masks)"""
    from instrain_amd import engine, synth
    from tests.test_gpu_pipe import small_workload
    ws = []
    for i, (glen, cov) in enumerate(((90_000, 25), (140_000, 40), (60_000, 90), (200_000, 12))):
        w = small_workload(700 + i, glen, cov, False)
        if i == 1:          # a pile of 400 identical reads: one level far beyond 255 in a batch of mean depth 40
            o = w["obs"]
            k = np.flatnonzero((o["gpos"] >= 5000) & (o["gpos"] < 5150) & (o["mm"] == 0))[:150]
            extra = np.tile(o[k], 400)
            w["obs"] = np.concatenate([o[:k[0]], extra, o[k[0]:]])
            w["pair"] = np.concatenate([w["pair"][:k[0]], np.repeat(np.arange(400, dtype=np.uint32) + w["pair"].max() + 1, len(k)), w["pair"][k[0]:]])
        if i == 2:          # spread the pairs over many mm bins
            rng = np.random.Generator(np.random.PCG64(5))
            lv = rng.integers(0, 21, int(w["pair"].max()) + 1).astype(np.uint16)
            o = w["obs"].copy()
            o["mm"] = lv[w["pair"]]
            w["obs"] = o
        w["n_mm_bins"] = int(w["obs"]["mm"].max()) + 1
        w["segs"] = synth.segs_from_obs(w["obs"], w["pair"])
        ws.append(w)
    for use in ([ws[0], ws[1], ws[3]], ws):
        M = max(w["n_mm_bins"] for w in use)
        kw = dict(enable_linkage=False, rarefied_coverage=30, seed=11)
        pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in use), max_obs=0, max_segs=max(w["segs"].n_seg for w in use),
                           max_splits=max(len(w["split_bounds"]) for w in use), depth=3, host_threads=3, pin_threads=False, n_mm_bins=M,
                           lean_output=True, **kw)
        expect = []
        for w in use:
            b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs"], None, n_mm_bins=M, **kw)
            b.run()
            expect.append(b.fetch()["entries"])
            b.close()
        order = list(range(len(use))) + list(range(len(use)))[::-1]          # every batch twice, the slots reused with other batches
        tickets, done, kept = [], 0, []

        def take():
            nonlocal done
            k = order[done]
            r = pipe.collect(tickets[done], shrunk_entries=True, densify=False)
            lev = r["levels"]
            assert lev["mask"].dtype.itemsize == (1 if M <= 8 else 2 if M <= 16 else 4)
            soa = pipe.expand_levels(r)
            same_soa(soa, soa_of_entries(expect[k]), "stream M=%d batch %d" % (M, k))
            if use[k] is ws[1]:
                assert int(expect[k]["cnt"].sum(axis=1).max()) > 400 and len(lev["sat"]) > 10 and lev["cov"].dtype == np.uint8
            assert 50 < len(lev["clon"]) < len(soa[0]) // 4
            own = pipe.levels_copy(r)               # own copies of the level tables (what profile_bam keeps of a batch) ...
            pipe.release(tickets[done])
            kept.append((k, own))                   # ... expanded after the slot has been given back and reused
            done += 1

        for k in order:
            if len(tickets) - done == 3:
                take()
            tickets.append(pipe.submit_reads(use[k]["ref_codes"], use[k]["split_bounds"], use[k]["segs"]))
        while done < len(tickets):
            take()
        pipe.close()
        for k, own in kept:
            same_soa(own.columns(threads=2), soa_of_entries(expect[k]), "kept level tables M=%d batch %d" % (M, k))
            assert own.columns() is own.columns()


def test_deep_batch_two_byte_coverage(ctx):
    """mean depth beyond 64: the coverage stream travels in two bytes"""
    from instrain_amd import engine, synth
    from tests.test_gpu_pipe import small_workload
    w = small_workload(811, 30_000, 300, False)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    M = w["n_mm_bins"]
    kw = dict(enable_linkage=True, min_snp=5, rarefied_coverage=50, seed=2)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], segs, None, n_mm_bins=M, **kw)
    b.run()
    exp = b.fetch()
    b.close()
    for lean in (False, True):
        soa, lev, full, snv, ld = through_pipe(ctx, w["ref_codes"], w["split_bounds"], segs, M, lean=lean, **kw)
        assert lev["cov"].dtype == np.uint16 and len(lev["sat"]) == 0
        same_soa(soa, soa_of_entries(exp["entries"]), "deep lean=%s" % lean)
        assert np.isfinite(soa[3]).sum() > len(soa[0]) // 2            # most levels carry a rarefied clonality here: the list outgrows its pinned room
        assert snv.tobytes() == exp["snv"].tobytes() and ld.tobytes() == exp["ld"].tobytes()


def test_lists_that_outgrow_their_tables_repeat_the_pass(ctx):
    """a small deep batch whose reads disagree with the reference at 3 % of their bases: nearly every level above the first sees more than
    one base, so the list of clonalities that are not 1.0 (and, at rarefied_coverage 20, the clonTR list) is longer than the slot's device
    table was sized for (cap_pos / 4 + 65 536) -- the finisher sees it from the cursors, grows the tables and repeats the pass; the level
    tables still equal the one-shot batch's entries, in the slot's next batch too"""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=40_000, coverage=400, n_sites=200, err=0.03, seed=91, skip_mm=False, max_mm=30)
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    M = w["n_mm_bins"]
    kw = dict(enable_linkage=False, rarefied_coverage=20, seed=4)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], segs, None, n_mm_bins=M, **kw)
    b.run()
    exp = b.fetch()["entries"]
    b.close()
    want = soa_of_entries(exp)
    n_mixed = int((np.isfinite(exp["clon"]) & (exp["clon"] != 1.0)).sum())
    assert n_mixed > 40_000 // 4 + 65_536 + 10_000, n_mixed            # more than the slot's first table holds
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=segs.n_seg, max_splits=len(w["split_bounds"]), depth=1, host_threads=3,
                       pin_threads=False, n_mm_bins=M, lean_output=True, **kw)
    for rnd in range(2):
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs)
        r = pipe.collect(t, shrunk_entries=True, densify=False)
        assert len(r["levels"]["clon"]) == n_mixed
        same_soa(pipe.expand_levels(r), want, "overflowing lists, round %d" % rnd)
        pipe.release(t)
    pipe.close()
