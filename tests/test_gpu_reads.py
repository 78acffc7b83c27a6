"""GPU parity of the READ-LEVEL hand-over (isx_batch_create_reads / isx_pipe_submit_reads): handing over read segments
must give the tables that handing over the observations they stand for gives -- bit for bit -- and therefore the
reference's (golden vectors, stored sars_cov_2 run)."""
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


SEG64 = 8       # isx_params.layout: ISX_LAYOUT_SEG64_RECORDS (one mm bin: the 64-byte segment records instead of reference-delta records)
MMDELTA = 32    # ISX_LAYOUT_MM_DELTA_RECORDS (mm profiling on: reference-delta records with the level in the header instead of segment records)
NOPACK = 4      # ISX_LAYOUT_NO_PACKED_COUNTERS: reference-delta records with 32-bit LDS counters (the path of very deep batches)


@pytest.mark.parametrize("how", ["stream", "reassembled", "stream64", "reassembled64", "stream32u", "reassembled32u", "streamD", "reassembledD", "reassembledDu"])
@pytest.mark.parametrize("name", CASES)
def test_golden_vectors_as_read_segments(ctx, name, how):
    """every reference-generated vector as read segments: 32-byte reference-delta records (difference-array pileup; with several mm bins
    the pair's level rides in the header and k_pileup_mm materialises every level's difference row), with layout SEG64 the 64-byte
    segment records, with NOPACK 32-bit LDS counters"""
    from tests import prod
    g = util.load_case(name)
    kw = _params(g)
    if how.endswith("D") or how.endswith("Du"):         # several mm bins as reference-delta records (round 6), 16-bit and 32-bit LDS counters
        if int(g["mm"].max()) == 0:
            pytest.skip("one mm bin: reference-delta records anyway")
        how, kw = (how[:-1], dict(kw, layout=MMDELTA)) if how.endswith("D") else (how[:-2], dict(kw, layout=MMDELTA | NOPACK))
    elif how.endswith("64") or how.endswith("32u"):
        if int(g["mm"].max()) > 0:
            pytest.skip("several mm bins: segment records by default")
        how, kw = (how[:-2], dict(kw, layout=SEG64)) if how.endswith("64") else (how[:-3], dict(kw, layout=NOPACK))
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), reads=how, **kw)
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what=name + "/" + how)
    assert res["n_edges"] == int(g["n_edges"])


@pytest.mark.parametrize("window", [64, 128, 1024, 3136])
@pytest.mark.parametrize("name", ["synth_dense", "synth_mm4_deep", "synth_m1_ld"])
def test_window_size_invariance(ctx, name, window):
    from tests import prod
    g = util.load_case(name)
    if window > 2048 and int(g["mm"].max()) > 0:
        pytest.skip("the mm kernel's window is at most 2 x its block")
    res = prod.run_split(ctx, g["pos"], g["base"], g["mm"], g["pair"], str(g["seq"]), int(g["start"]), reads="reassembled", window=window, **_params(g))
    util.assert_same(util.canon_from_struct(res), util.canon_from_golden(g), float_tol=TOL, what="%s window%d" % (name, window))


def _tables_equal(a, b, what):
    for k in a:
        if k in ("counts", "clon", "clon_r", "snv", "ld", "entries"):
            x, y = a[k], b[k]
            assert x.shape == y.shape, (what, k, x.shape, y.shape)
            assert x.tobytes() == y.tobytes(), (what, k)


def test_stored_sars_golden_as_read_segments(ctx):
    """BAM -> front end -> observations -> read segments -> kernels == the reference's stored run, and == the observation path"""
    from instrain_amd import engine, synth
    from tests import prod
    from tests.test_oracle_golden import check_against_sars_golden, read_fasta
    bam = engine.BamFile(os.path.join(util.GOLD, "sars_cov_2.sorted.bam"))
    obs, pair, bounds, sref = bam.expand()
    M = bam.info["max_mm"] + 1
    bam.close()
    seq = read_fasta(os.path.join(util.GOLD, "sars_cov_2_MT039887.1.fasta"))
    segs = synth.segs_from_obs(obs, pair)
    assert segs.n_bases >= len(obs)
    kw = dict(n_mm_bins=M, min_cov=5, min_freq=0.05, min_snp=20)
    b = engine.Batch(ctx, engine.encode_seq(seq), bounds, segs, **kw)
    b.run()
    got = b.fetch()
    sizes = b.sizes()
    assert b.timings()["record_bytes"] == 64
    b.close()
    bd = engine.Batch(ctx, engine.encode_seq(seq), bounds, segs, layout=MMDELTA, **kw)       # 26 mm levels as reference-delta records (round 6): the same tables
    bd.run()
    assert bd.timings()["record_bytes"] == 32
    _tables_equal(got, bd.fetch(), "sars mm delta records")
    bd.close()
    res = prod.to_oracle_layout(got, lambda g: g.astype(np.int64))
    check_against_sars_golden(res["snv"], res["ld"], float_tol=TOL)
    assert sizes["n_edges"] == 963 and sizes["n_increments"] == 23319
    b2 = engine.Batch(ctx, engine.encode_seq(seq), bounds, obs, pair, **kw)
    b2.run()
    _tables_equal(got, b2.fetch(), "sars")
    b2.close()


def _mg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(util.GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)         # only its synthetic generator is used (no reference import)
    return mg


def test_randomized_sweep_reads_equal_observations(ctx):
    """random small splits, thresholds, mm-level counts, non-ACGT rates, windows: both hand-overs give the same tables"""
    from instrain_amd import engine, synth
    mg = _mg()
    rng = np.random.Generator(np.random.PCG64(77))
    for it in range(24):
        mLen = int(rng.integers(300, 6000))
        mm_levels = int(rng.choice([1, 1, 2, 5, 17]))
        seq, pos, base, mm, pair = mg.synth_case(seed=1000 + it, mLen=mLen, depth=int(rng.integers(3, 120)), mm_levels=mm_levels,
                                                 n_sites=int(rng.integers(0, 80)), p_other=float(rng.choice([0.0, 0.03])), ref_ambig=int(rng.integers(0, 4)),
                                                 self_pairs=float(rng.choice([0.0, 0.3])))
        kw = dict(n_mm_bins=int(mm.max()) + 1 if len(mm) else 1, min_cov=int(rng.integers(1, 8)), min_freq=float(rng.choice([0.01, 0.05, 0.2])),
                  min_snp=int(rng.integers(1, 25)), window=int(rng.choice([0, 64, 256, 1024] if mm_levels <= 5 else [0, 64, 256])),
                  rarefied_coverage=int(rng.choice([0, 5, 50])))
        obs = engine.pack_obs(pos.astype(np.uint32), base, mm)
        pr = pair.astype(np.uint32)
        ref = engine.encode_seq(seq)
        a = engine.Batch(ctx, ref, [0, mLen], obs, pr, **kw)
        a.run()
        ra = a.fetch()
        sa = a.sizes()
        a.close()
        for segs in (synth.segs_from_obs(obs, pr), util.reassemble_segs(pos, base, mm, pair)):
            for layout in ((0, NOPACK, SEG64) if kw["n_mm_bins"] == 1 else (0, MMDELTA, MMDELTA | NOPACK)):
                b = engine.Batch(ctx, ref, [0, mLen], segs, layout=layout, **kw)
                b.run()
                _tables_equal(ra, b.fetch(), "iteration %d %r layout %d" % (it, kw, layout))
                assert b.sizes() == sa
                assert b.timings()["record_bytes"] == (32 if (kw["n_mm_bins"] == 1 and layout != SEG64) or (layout & MMDELTA) else 64)
                b.close()


def test_non_acgt_base_makes_its_level_present(ctx):
    """code 5 (a base that is not A/C/T/G but passes the filter) creates the (position, mm) entry without a count
    (profile_utilities.py:279-285); code 4 / 6 / 7 do nothing"""
    from instrain_amd import engine
    ref = np.zeros(500, np.uint8)
    codes = np.full((3, 150), 4, np.uint8)
    codes[0, :20] = 0                       # read 0, mm 0: twenty A
    codes[1, 5] = 5                         # read 1, mm 2: one N over position 105
    codes[1, 6] = 6; codes[1, 7] = 7
    codes[2, :10] = 1                       # read 2, mm 1
    segs = engine.SegBatch([100, 100, 300], [20, 8, 10], engine.pack_codes(codes), mm=[0, 2, 1], pair=[0, 1, 2])
    b = engine.Batch(ctx, ref, [0, 500], segs, n_mm_bins=3, min_cov=1, enable_linkage=False)
    b.run()
    e = b.fetch()["entries"]
    b.close()
    key = {(int(r["gpos"]), int(r["mm"])): r["cnt"].tolist() for r in e}
    assert key[(105, 2)] == [0, 0, 0, 0] and key[(105, 0)] == [1, 0, 0, 0]
    assert (106, 2) not in key and (107, 2) not in key
    assert len(e) == 20 + 1 + 10
    # one mm bin: a non-ACGT base is simply not counted
    d = engine.Batch(ctx, ref, [0, 500], engine.SegBatch(segs.gpos, segs.len, segs.bases, None, segs.pair), n_mm_bins=1, min_cov=1, enable_linkage=False)
    d.run()
    c = d.fetch()["counts"]
    d.close()
    assert c[100:120, 0].tolist() == [1] * 20 and c[300:310, 1].tolist() == [1] * 10 and int(c.sum()) == 30


def test_segments_straddling_windows_and_far_jumps(ctx):
    """segments that cross window edges at every offset, a jump of > 65535 positions between two segments (group cut) and
    a segment ending at the last position of the batch"""
    from instrain_amd import engine
    n_pos = 300_000
    rng = np.random.Generator(np.random.PCG64(5))
    starts = np.sort(np.concatenate([rng.integers(0, 3000, 400), rng.integers(200_000, 203_000, 400), [n_pos - 150]])).astype(np.uint32)
    n = len(starts)
    codes = rng.integers(0, 6, (n, 150)).astype(np.uint8)
    ln = np.full(n, 150, np.uint8)
    segs = engine.SegBatch(starts, ln, engine.pack_codes(codes), mm=None, pair=np.arange(n, dtype=np.uint32))
    exp = np.zeros((n_pos, 4), np.int64)
    for b in range(4):
        si, off = np.nonzero(codes == b)
        np.add.at(exp[:, b], starts[si].astype(np.int64) + off, 1)
    ref = rng.integers(0, 4, n_pos, dtype=np.uint8)
    # (random bases against a random reference: three of four columns are exceptions -- a segment travels as ~20 delta records)
    for window in (64, 192, 1024, 0):
        for layout in (0, NOPACK, SEG64):
            bt = engine.Batch(ctx, ref, [0, 150_000, n_pos], segs, n_mm_bins=1, min_cov=5, enable_linkage=True, window=window, layout=layout)
            bt.run()
            c = bt.fetch()["counts"]
            bt.close()
            assert (c.astype(np.int64) == exp).all(), (window, layout)


def _c2(scale=1.0, seed=2, skip_mm=True, with_n=False, p_keep=0.90):
    """with_n: positions that are not A/C/T/G in the reference (isolated ones and a run): a pipe slot then carries the bit plane
    that marks them beside its 2-bit reference plane"""
    from instrain_amd import synth
    w = synth.make_workload(genome_len=int(5_000_000 * scale), coverage=20, n_sites=int(5000 * scale), seed=seed, skip_mm=skip_mm, p_keep=p_keep)
    if with_n:
        rng = np.random.Generator(np.random.PCG64(seed + 99))
        ref = w["ref_codes"].copy()
        ref[rng.random(len(ref)) < 0.004] = 4
        ref[len(ref) // 3:len(ref) // 3 + 700] = 4
        w["ref_codes"] = ref
    return w


@pytest.mark.parametrize("skip_mm,linkage,layout,p_keep", [(True, False, 0, 0.9), (True, True, 0, 0.9), (True, True, NOPACK, 0.9), (True, True, SEG64, 0.9), (False, True, 0, 0.9),
                                                           (True, True, 0, 0.995), (True, False, 0, 1.0), (True, True, NOPACK, 0.995), (False, True, MMDELTA, 0.9), (False, False, MMDELTA, 0.995)])
def test_pipe_reads_equal_observation_batch(ctx, skip_mm, linkage, layout, p_keep):
    """a C2 slice through the read-level pipe == the same observations through a resident batch, every table
    (p_keep 0.9: every read has bases below the quality bar = full reference-delta records; 0.995: about half of them have none = halves
    of dual records, mixed with full ones; 1.0: dual records only)"""
    from instrain_amd import engine, synth
    w = _c2(0.1, seed=4, skip_mm=skip_mm, with_n=linkage, p_keep=p_keep)
    M = w["n_mm_bins"]
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    kw = dict(n_mm_bins=M, enable_linkage=linkage, min_snp=20, layout=layout)
    a = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"] if linkage else None, **kw)
    a.run()
    ra, sa = a.fetch(), a.sizes()
    a.close()
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=segs.n_seg, max_splits=len(w["split_bounds"]), depth=2,
                       host_threads=4, pin_threads=False, want_counts=True, **kw)
    tickets = [pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs) for _ in range(2)]
    for t in tickets:
        r = pipe.collect(t)
        assert r["stats"]["record_bytes"] == (32 if (M == 1 and layout != SEG64) or (layout & MMDELTA) else 64)
        assert r["sizes"] == sa
        if M == 1:
            assert (r["counts"] == ra["counts"]).all()
            assert r["clon"].tobytes() == ra["clon"].tobytes()
            assert (r["cov16"] == np.minimum(ra["counts"].sum(axis=1), 65535)).all()
        else:
            assert r["entries"].tobytes() == ra["entries"].tobytes()
        assert r["snv"].tobytes() == ra["snv"].tobytes()
        if linkage:
            assert r["ld"].tobytes() == ra["ld"].tobytes()
        pipe.release(t)
    # an observation batch is refused by a read-level pipe
    with pytest.raises(engine.IsxError, match="read-level"):
        pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], w["pair"] if linkage else None)
    pipe.close()


@pytest.mark.parametrize("linkage", [False, True])
def test_queued_submits_equal_blocking_submits(ctx, linkage):
    """isx_pipe_params.stage_async: submit_reads only queues; the pipe's stager encodes and enqueues in ticket order.  Same
    tables as the blocking submit, slot reuse over more batches than slots, slot exhaustion refused at submit, what the
    encoder rejects reported by collect (and the pipe goes on after it), a synchronous submit_bam-style drain at close."""
    from instrain_amd import engine, synth
    ws = [_c2(0.05, seed=11 + i, skip_mm=True) for i in range(3)]
    segs = [synth.segs_from_obs(w["obs"], w["pair"]) for w in ws]
    kw = dict(n_mm_bins=1, enable_linkage=linkage, min_snp=20)
    cap = dict(max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(s.n_seg for s in segs),
               max_splits=max(len(w["split_bounds"]) for w in ws), depth=2, host_threads=4, pin_threads=False)
    want = []
    plain = engine.Pipe(ctx, **cap, **kw)
    for w, sg in zip(ws, segs):
        t = plain.submit_reads(w["ref_codes"], w["split_bounds"], sg)
        r = plain.collect(t, want_ld=linkage)
        want.append((r["sizes"], r["snv"].copy(), r["cov16"].copy(), r["clon"].copy(), r["ld"].copy() if linkage else None))
        plain.release(t)
    plain.close()
    pipe = engine.Pipe(ctx, stage_async=True, **cap, **kw)
    order = [0, 1, 2, 1, 0, 2, 2]
    tickets, done = [], 0

    def take():
        nonlocal done
        r = pipe.collect(tickets[done], want_ld=linkage)
        sz, snv, cov, clon, ld = want[order[done]]
        assert r["sizes"] == sz and r["snv"].tobytes() == snv.tobytes()
        assert (r["cov16"] == cov).all() and r["clon"].tobytes() == clon.tobytes()
        if linkage:
            assert r["ld"].tobytes() == ld.tobytes()
        pipe.release(tickets[done])
        done += 1

    for k in order:
        if len(tickets) - done == 2:
            take()
        tickets.append(pipe.submit_reads(ws[k]["ref_codes"], ws[k]["split_bounds"], segs[k]))
    assert tickets == list(range(len(order)))
    with pytest.raises(engine.IsxError, match="every slot is in use"):
        pipe.submit_reads(ws[0]["ref_codes"], ws[0]["split_bounds"], segs[0])
    while done < len(tickets):
        take()
    # a batch the encoder rejects: the ticket is handed out, collect reports it, the next batch is fine
    bad = engine.SegBatch(segs[0].gpos, segs[0].len, segs[0].bases, np.full(segs[0].n_seg, 3, np.uint8), segs[0].pair)
    tb = pipe.submit_reads(ws[0]["ref_codes"], ws[0]["split_bounds"], bad)
    tg = pipe.submit_reads(ws[1]["ref_codes"], ws[1]["split_bounds"], segs[1])
    assert (tb, tg) == (len(order), len(order) + 1)
    with pytest.raises(engine.IsxError, match="mm >= n_mm_bins"):
        pipe.collect(tb)
    pipe.release(tb)
    r = pipe.collect(tg, want_ld=linkage)
    assert r["sizes"] == want[1][0] and r["snv"].tobytes() == want[1][1].tobytes()
    pipe.release(tg)
    # queued and never collected: close() stages and finishes it
    pipe.submit_reads(ws[2]["ref_codes"], ws[2]["split_bounds"], segs[2])
    pipe.close()


def test_full_c2_reads_equal_observations(ctx):
    """BASELINE configs[1] at full size: both hand-overs, all tables identical; the segments are 1 / 4.6 of the 2-byte records"""
    from instrain_amd import engine, synth
    w = _c2()
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    assert segs.n_seg == 2 * w["n_pairs"]
    a = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False)
    a.run()
    ra = a.fetch()
    a.close()
    for layout in (0, NOPACK, SEG64):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], segs, n_mm_bins=1, enable_linkage=False, layout=layout)
        b.run()
        rb = b.fetch()
        assert b.timings()["record_bytes"] == (64 if layout == SEG64 else 32)
        assert (b.timings()["pileup_window"] > 4096) == (layout == 0)          # 16-bit LDS counters: the wide window
        b.close()
        _tables_equal(ra, rb, "C2 layout %d" % layout)
        assert int(rb["counts"].sum()) == w["n_obs"]


@pytest.mark.parametrize("linkage", [False, True])
def test_empty_and_single_segment_batches(ctx, linkage):
    """no read at all (a scaffold nobody mapped to), then one 1-base and one 150-base segment: a resident batch and a pipe slot
    (first batch of a slot, then reused after a full one) hand back all-zero coverage / the one column, no SNV, no LD rows"""
    from instrain_amd import engine
    n_pos = 5000
    ref = np.zeros(n_pos, np.uint8)
    bounds = [0, 2500, n_pos]
    empty = engine.SegBatch(np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros((0, 15), np.uint32), np.zeros(0, np.uint8), np.zeros(0, np.uint32))
    b = engine.Batch(ctx, ref, bounds, empty, n_mm_bins=1, enable_linkage=linkage)
    b.run()
    r = b.fetch()
    assert r["counts"].sum() == 0 and len(r["snv"]) == 0 and len(r["ld"]) == 0 and np.isnan(r["clon"]).all()
    b.close()
    codes = np.full(150, 1, np.uint8)           # 150 x 'C' against a reference of 'A': below min_cov, so no SNV rows
    one = np.full(150, 4, np.uint8)             # unused slots hold code 4
    one[0] = 1
    two = engine.SegBatch(np.array([7, 2400], np.uint32), np.array([1, 150], np.uint8),
                          engine.pack_codes(np.stack([one, codes])), np.zeros(2, np.uint8), np.array([0, 1], np.uint32))
    pipe = engine.Pipe(ctx, max_pos=n_pos, max_obs=0, max_segs=4096, max_splits=4, depth=1, host_threads=2, pin_threads=False,
                       n_mm_bins=1, enable_linkage=linkage, want_counts=True)
    for segs, cols in ((empty, 0), (two, 151), (empty, 0)):
        t = pipe.submit_reads(ref, bounds, segs)
        r = pipe.collect(t)
        assert int(r["counts"].sum()) == cols and len(r["snv"]) == 0
        if cols:
            assert r["counts"][7, 1] == 1 and (r["counts"][2400:2550, 1] == 1).all() and r["counts"][:, [0, 2, 3]].sum() == 0
            assert (r["cov16"][2400:2550] == 1).all() and r["cov16"].sum() == 151
        pipe.release(t)
    pipe.close()


@pytest.mark.parametrize("skip_mm,linkage,layout", [(True, False, 0), (True, True, 0), (True, True, SEG64), (False, True, 0)])
def test_staged_batches_equal_submits(ctx, skip_mm, linkage, layout):
    """isx_pipe_stage_reads + isx_pipe_submit_wire (the zero-copy hand-over: all host work done once, into a pinned image) give the
    tables isx_pipe_submit_reads gives; a wire can be submitted again and again, interleaved with other wires and plain submits;
    a wire of another pipe is refused"""
    from instrain_amd import engine, synth
    ws = [_c2(0.05, seed=31 + i, skip_mm=skip_mm, with_n=(i == 1)) for i in range(2)]
    M = max(w["n_mm_bins"] for w in ws)
    segs = [synth.segs_from_obs(w["obs"], w["pair"]) for w in ws]
    kw = dict(n_mm_bins=M, enable_linkage=linkage, min_snp=20, layout=layout)
    cap = dict(max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(s.n_seg for s in segs),
               max_splits=max(len(w["split_bounds"]) for w in ws), depth=3, host_threads=4, pin_threads=False, want_counts=True)
    pipe = engine.Pipe(ctx, **cap, **kw)
    keys = ("counts", "clon", "snv", "ld") if M == 1 else ("entries", "snv", "ld")
    want = []
    for w, sg in zip(ws, segs):
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], sg)
        r = pipe.collect(t)
        want.append((r["sizes"], {k: r[k].copy() for k in keys}))
        pipe.release(t)
    wires = [pipe.stage_reads(w["ref_codes"], w["split_bounds"], sg) for w, sg in zip(ws, segs)]
    rb = 32 if (M == 1 and layout == 0) else 64
    for wr, sg, w in zip(wires, segs, ws):
        assert wr.bytes < sg.n_seg * rb * 1.1 + w["n_pos"] * 0.55 + (sg.n_seg * 4.4 if linkage else 0) + 200_000
    order = [0, 1, 1, 0, 0, 1]
    tickets, done = [], 0

    def take():
        nonlocal done
        r = pipe.collect(tickets[done])
        sz, tb = want[order[done]]
        assert r["sizes"] == sz and r["stats"]["record_bytes"] == rb and r["stats"]["encode_passes"] == 0
        for k in keys:
            assert r[k].tobytes() == tb[k].tobytes(), (k, done)
        pipe.release(tickets[done])
        done += 1

    for i, k in enumerate(order):
        if len(tickets) - done == 3:
            take()
        if i == 3:                                  # a plain submit in between
            tickets.append(pipe.submit_reads(ws[k]["ref_codes"], ws[k]["split_bounds"], segs[k]))
            continue
        tickets.append(pipe.submit_wire(wires[k]))
    while done < len(tickets):
        if done == 3:
            r = pipe.collect(tickets[done])
            assert r["sizes"] == want[order[done]][0]
            pipe.release(tickets[done])
            done += 1
            continue
        take()
    # the reference planes kept on the device (isx_wire_keep_reference): later submits bring in less and give the same tables -- also when
    # the slot has meanwhile held another batch's reference
    before = [wr.bytes for wr in wires]
    for wr in wires:
        wr.keep_reference()
    assert all(wr.bytes < b0 - w["n_pos"] // 4 + 64 for wr, b0, w in zip(wires, before, ws))
    for k in (1, 0, 1, 1, 0):
        t = pipe.submit_wire(wires[k])
        r = pipe.collect(t)
        assert r["sizes"] == want[k][0] and r["stats"]["h2d_bytes"] == wires[k].bytes
        for kk in keys:
            assert r[kk].tobytes() == want[k][1][kk].tobytes(), (kk, k)
        pipe.release(t)
    other = engine.Pipe(ctx, **cap, **kw)
    with pytest.raises(engine.IsxError, match="another pipe"):
        other.submit_wire(wires[0])
    other.close()
    pipe.close()


def test_lean_slot_output_equals_the_plain_slot(ctx):
    """isx_pipe_params.lean_output: a slot's kernel writes only what travels home (8- / 16-bit coverage + the sparse lists).  Same
    tables as a plain slot for a shallow batch (8-bit coverage), a deep one (16-bit) and one whose clonality list cannot be used
    (most positions carry a second base: the pass is repeated with the dense array); device summaries are refused on a lean slot"""
    from instrain_amd import engine, synth
    ws = [synth.make_workload(genome_len=400_000, coverage=3, n_sites=300, seed=59, skip_mm=True),      # 4-bit coverage plane, hardly a window beyond 15
          synth.make_workload(genome_len=400_000, coverage=5, n_sites=300, seed=60, skip_mm=True),      # ... with 16-bit rows for many windows
          synth.make_workload(genome_len=300_000, coverage=6, n_sites=300, seed=61, skip_mm=True),
          synth.make_workload(genome_len=200_000, coverage=60, n_sites=400, seed=62, skip_mm=True),
          synth.make_workload(genome_len=60_000, coverage=300, n_sites=100, seed=63, skip_mm=True, err=0.01)]
    segs = [synth.segs_from_obs(w["obs"], w["pair"]) for w in ws]
    cap = dict(max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(s.n_seg for s in segs),
               max_splits=max(len(w["split_bounds"]) for w in ws), depth=2, host_threads=4, pin_threads=False,
               n_mm_bins=1, enable_linkage=True, min_snp=20)
    out = {}
    n_rows = []
    for lean in (False, True):
        pipe = engine.Pipe(ctx, lean_output=lean, **cap)
        res = []
        for w, sg in zip(ws, segs):
            t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], sg)
            raw = pipe.collect(t, densify=False)
            assert ("cov4" in raw) == (lean and w["n_obs"] < 6 * w["n_pos"]), (lean, w["n_obs"] / w["n_pos"])
            if "cov4" in raw:
                n_rows.append(len(raw["cov_row_win"]))
                assert raw["cov_rows"].shape == (len(raw["cov_row_win"]), raw["cov_window"]) and len(set(raw["cov_row_win"].tolist())) == len(raw["cov_row_win"])
                exp_cov = np.bincount(w["obs"]["gpos"], minlength=w["n_pos"])
                W = raw["cov_window"]
                beyond = np.unique(np.flatnonzero(exp_cov > 15) // W)
                assert set(beyond.tolist()) == set(raw["cov_row_win"].tolist())          # exactly the windows beyond 15 have a row (the stripe path knows the coverage itself)
                assert (engine.dense_cov(raw, w["n_pos"]) == np.minimum(exp_cov, 65535)).all()
            r = pipe.collect(t)                         # densified: cov16 + clon arrays rebuilt on the host
            res.append({k: r[k].copy() for k in ("cov16", "clon", "snv", "ld")} | {"sizes": r["sizes"], "rare": r["rare"].copy()})
            if lean:
                with pytest.raises(engine.IsxError, match="lean"):
                    r["slot"].summarize(np.array([0, w["n_pos"]], np.int64))
            pipe.release(t)
        pipe.close()
        out[lean] = res
    for a, b in zip(out[False], out[True]):
        assert a["sizes"] == b["sizes"]
        for k in ("cov16", "snv", "ld", "rare"):
            assert a[k].tobytes() == b[k].tobytes(), k
        assert a["clon"].view(np.uint32).tobytes() == b["clon"].view(np.uint32).tobytes()
    assert len(n_rows) == 3 and n_rows[2] >= 3 and n_rows[0] <= n_rows[1] <= n_rows[2]         # (the depth-6 batch keeps 5.4 of 6 bases: 4-bit plane too)
    n2 = out[True][4]
    assert (~np.isnan(n2["clon"]) & (n2["clon"] != 1.0)).sum() * 2 > len(n2["clon"])        # the case that needs the dense array


@pytest.mark.parametrize("every,rarefied", [(800, 50), (700, 24)])
def test_lean_slot_with_a_pile_in_every_window_repeats_the_pass_without_the_4bit_plane(ctx, every, rarefied):
    """a shallow batch (mean depth 4.8) with 24 reads stacked every 800 positions (amplicon-like): every window holds positions beyond 15,
    the 16-bit rows would outgrow the slot's coverage block -> the pass is repeated with the 8-bit plane; tables equal the plain slot's.
    (700, 24): a lean slot takes a batch of mean depth below rarefied_coverage / 4 for shallow and writes the clonTR list alone; here one
    position in seven reaches the rarefied coverage, the list is no shorter than the array -> the pass is repeated with the dense array"""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=900_000, coverage=2, n_sites=400, seed=77, skip_mm=True)
    base = synth.segs_from_obs(w["obs"], w["pair"])
    starts = np.arange(300, w["n_pos"] - 200, every, dtype=np.uint32)
    sg = np.repeat(starts, 24)
    codes = np.full((len(sg), 150), 4, dtype=np.uint8)
    codes[:, :100] = w["ref_codes"][sg[:, None].astype(np.int64) + np.arange(100)[None, :]]
    spike_pair = (int(base.pair.max()) + 1 + np.arange(len(sg))).astype(np.uint32)
    g = np.concatenate([base.gpos, sg])
    order = np.argsort(g, kind="stable")
    segs = engine.SegBatch(g[order], np.concatenate([base.len, np.full(len(sg), 100, np.uint8)])[order],
                           np.concatenate([base.bases, engine.pack_codes(codes)])[order], None, np.concatenate([base.pair, spike_pair])[order])
    exp_cov = np.bincount(w["obs"]["gpos"], minlength=w["n_pos"])
    for s in starts:
        exp_cov[int(s):int(s) + 100] += 24
    assert exp_cov.sum() < 6 * w["n_pos"]
    cap = dict(max_pos=w["n_pos"], max_obs=0, max_segs=segs.n_seg, max_splits=len(w["split_bounds"]), depth=2, host_threads=4, pin_threads=False,
               n_mm_bins=1, enable_linkage=True, min_snp=20, rarefied_coverage=rarefied)
    got = {}
    for lean in (False, True):
        pipe = engine.Pipe(ctx, lean_output=lean, **cap)
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs)
        raw = pipe.collect(t, densify=False)
        assert "cov8" in raw and "cov4" not in raw, lean                # the lean slot fell back: rows for every window do not fit
        assert (engine.dense_cov(raw, w["n_pos"]) == exp_cov).all()
        r = pipe.collect(t)
        got[lean] = {k: r[k].copy() for k in ("cov16", "clon", "snv", "ld")} | {"sizes": r["sizes"], "rare": r["rare"].copy()}
        pipe.release(t)
        pipe.close()
    assert got[False]["sizes"] == got[True]["sizes"]
    n_reach = int((exp_cov >= rarefied).sum())
    assert len(got[True]["rare"]) == n_reach and (n_reach * 8 > w["n_pos"]) == (rarefied == 24)
    for k in ("cov16", "snv", "ld", "rare"):
        assert got[False][k].tobytes() == got[True][k].tobytes(), k
    assert got[False]["clon"].view(np.uint32).tobytes() == got[True]["clon"].view(np.uint32).tobytes()
