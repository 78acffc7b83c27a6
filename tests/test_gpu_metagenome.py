"""GPU: the metagenome configurations (BASELINE.json configs[3] / configs[4]; SURVEY.md 8(d) C4 / C5:
--database_mode, i.e. one mm bin, genomes below 1x dropped).
  * a slice (small genomes, same generator and shape) against the C oracle split by split, through both the
    one-shot path and the pipe;
  * the full per-GPU shard of C5 (1/8 of the kept genomes, LPT) streamed through the pipe, checked through
    size-independent properties: per-scaffold sums of counts == observations handed over, a position-weighted
    checksum of the counts against numpy, SNV rows consistent with the count table, idempotence."""
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


def test_metagenome_slice_vs_oracle(ctx):
    from instrain_amd import engine, synth
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    meta = synth.Metagenome(14, mean_coverage=5, seed=44, contigs=4, len_lo=30_000, len_hi=70_000, threads=4)
    kept = meta.kept_genomes()
    assert 3 <= len(kept) < 14                                # some genomes fall below 1x and are dropped
    w = meta.generate(kept)
    kw = dict(min_cov=5, min_freq=0.05, min_snp=10)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=True, **kw)
    b.run()
    res = b.fetch()
    got = prod.to_oracle_layout(res, lambda g: g.astype(np.int64))
    b.close()
    letters = np.array(list("ACTGN"))
    gpos = w["obs"]["gpos"].astype(np.int64)
    exp = {"entries": [], "snv": [], "ld": []}
    sb = w["split_bounds"]
    for s, e in zip(sb[:-1], sb[1:]):
        o = oracle.profile_split(gpos, w["obs"]["base"], w["obs"]["mm"].astype(np.int64), w["pair"].astype(np.int64),
                                 "".join(letters[w["ref_codes"][s:e]]), int(s), lut, fb, **kw)
        for k in exp:
            exp[k].append(o[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    assert len(exp["snv"]) > 100 and len(exp["ld"]) > 20
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=1e-6, what="metagenome slice")
    # the same batch through the pipe (the stream jumps at every uncovered stretch / contig end)
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"], max_splits=len(sb), depth=2, host_threads=4,
                       n_mm_bins=1, enable_linkage=True, want_counts=True, **kw)
    t = pipe.submit(w["ref_codes"], sb, w["obs"], w["pair"])
    r = pipe.collect(t)
    for k in ("counts", "clon", "snv", "ld"):
        a, e = r[k], res[k]
        assert a.shape == e.shape
        if a.dtype.names:
            for f in a.dtype.names:
                assert (a[f].view("u%d" % a[f].dtype.itemsize) == e[f].view("u%d" % e[f].dtype.itemsize)).all(), (k, f)
        else:
            assert (a.view(np.uint32) == e.view(np.uint32)).all(), k
    pipe.release(t)
    pipe.close()


def test_c5_per_gpu_shard_properties(ctx):
    """configs[4]: 1000 genomes, 10 Gbp of reads, --database_mode; rank 0's shard of 8, streamed in batches"""
    from instrain_amd import dist as idist
    from instrain_amd import engine, synth
    meta = synth.Metagenome(1000, total_read_bp=10e9, seed=5)
    kept = meta.kept_genomes()
    assert 600 < len(kept) < 760 and abs(meta.length.sum() / 1e9 - 4.0) < 0.3
    shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
    est_obs = (meta.pairs[shard] * 2 * meta.read_len * 0.92).astype(np.int64)
    batches = idist.pack_batches(meta.length[shard], est_obs, 40_000_000, 150_000_000)
    assert len(batches) >= 4
    ws = [meta.generate(shard[b]) for b in batches]
    total_bases = sum(w["profiled_bases"] for w in ws)
    assert 0.9e9 < total_bases < 1.4e9                        # ~ 10 Gbp x (kept share) / 8
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=3, n_mm_bins=1, enable_linkage=False,
                       jump_slack=0.5, want_counts=True)
    tickets = [None] * len(ws)

    def check(i, r, first):
        w = ws[i]
        counts = r["counts"]
        sb = w["scaffold_bounds"]
        o = w["obs"]
        assert counts.shape == (w["n_pos"], 4)
        # every observation carries an A,C,T,G base here: per-scaffold sums == observations handed over
        per_scaf = np.add.reduceat(counts.sum(axis=1, dtype=np.int64), sb[:-1])
        exp = np.bincount(np.searchsorted(sb, o["gpos"], side="right") - 1, minlength=len(sb) - 1)
        assert (per_scaf == exp).all()
        # position- and base-weighted checksum of the whole table against numpy on the raw records
        wgt = (np.arange(w["n_pos"], dtype=np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
        chk_dev = sum(int((counts[:, k].astype(np.uint64) * (wgt + np.uint64(k * 977))).sum()) for k in range(4))
        g = o["gpos"].astype(np.uint64)
        chk_host = int((((g * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) + o["base"].astype(np.uint64) * np.uint64(977)).sum())
        assert chk_dev == chk_host
        snv = r["snv"]
        assert len(snv) == r["sizes"]["n_snv"] > 0
        assert (snv["cnt"] == counts[snv["gpos"]]).all() and (np.diff(snv["gpos"].astype(np.int64)) > 0).all()
        cov = counts.sum(axis=1)
        assert np.isnan(r["clon"][cov < 5]).all() and not np.isnan(r["clon"][cov >= 5]).any()
        # the shrunk tables that travel by default: coverage16, and clonTR exactly where coverage >= 50
        assert (r["cov16"] == np.minimum(cov, 65535)).all()
        assert (r["rare"]["gpos"] == np.flatnonzero(cov >= 50)).all() and not np.isnan(r["rare"]["clon_rarefied"]).any()
        return (chk_dev, len(snv), int(snv["gpos"].astype(np.int64).sum()))

    sig = []
    for rnd in range(2):                                      # idempotence: the same stream twice, same tables
        out = []
        done = 0
        for i, w in enumerate(ws):
            if i - done == 3:
                out.append(check(done, pipe.collect(tickets[done]), rnd == 0))
                pipe.release(tickets[done])
                done += 1
            tickets[i] = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
        while done < len(ws):
            out.append(check(done, pipe.collect(tickets[done]), rnd == 0))
            pipe.release(tickets[done])
            done += 1
        sig.append(out)
    assert sig[0] == sig[1]
    pipe.close()


def test_c4_per_gpu_shard_properties(ctx):
    """configs[3]: 100 genomes U(2,6) Mbp x 50 contigs at 50x mean coverage (log-normal abundances), --database_mode; rank
    0's LPT shard of 8 (~2.5 Gbp of reads) streamed through the pipe with linkage on.  Size-independent properties per
    batch: the coverage table sums to the observations handed over, every SNV row's counts sum to the coverage at its
    position and exceed min_cov, LD rows stay inside one scaffold with counts that add up, deep genomes produce clonTR
    exactly where coverage >= 50."""
    from instrain_amd import dist as idist
    from instrain_amd import engine, synth
    meta = synth.Metagenome(100, mean_coverage=50, seed=4)
    kept = meta.kept_genomes()
    assert len(kept) == 100 and abs(float((meta.coverage * meta.length).sum()) / float(meta.length.sum()) - 50) < 1e-6
    shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
    est = (meta.pairs[shard] * 2 * meta.read_len * 0.92).astype(np.int64)
    batches = idist.pack_batches(meta.length[shard], est, 40_000_000, 400_000_000)
    ws_meta = [shard[b] for b in batches]
    pipe = None
    total = 0
    for sel in ws_meta:                                       # one batch in memory at a time (a batch is up to 5 GB of records)
        w = meta.generate(sel)
        if pipe is None or w["n_obs"] > cap[1] or w["n_pos"] > cap[0] or len(w["split_bounds"]) > cap[2]:
            if pipe is not None:
                pipe.close()
            cap = (max(w["n_pos"], 30_000_000), int(w["n_obs"] * 1.2), len(w["split_bounds"]) + 4096)
            pipe = engine.Pipe(ctx, max_pos=cap[0], max_obs=cap[1], max_splits=cap[2], depth=1, n_mm_bins=1,
                               enable_linkage=True, min_snp=20, jump_slack=0.3)
        t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], w["pair"])
        r = pipe.collect(t)
        cov = r["cov16"].astype(np.int64)
        assert r["n_saturated"] == 0 and int(cov.sum()) == w["n_obs"]
        snv, ld = r["snv"], r["ld"]
        assert len(snv) > 1000 and (snv["cnt"].sum(axis=1) == cov[snv["gpos"]]).all() and (snv["cnt"].sum(axis=1) >= 5).all()
        sb = w["scaffold_bounds"]
        assert len(ld) > 100
        assert (np.searchsorted(sb, ld["gpos_a"], side="right") == np.searchsorted(sb, ld["gpos_b"], side="right")).all()
        assert (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] == ld["total"]).all()
        assert (r["rare"]["gpos"] == np.flatnonzero(cov >= 50)).all()
        total += w["profiled_bases"]
        pipe.release(t)
        del w, r
    pipe.close()
    assert 1.5e9 < total < 4e9


# ---- the same configurations through the READ-LEVEL hand-over: the path bench.py times since round 3 ----

def _expected_coverage(w, per_base_at=None):
    """exact per-position coverage of a read-level workload, from the segments on the host (numpy, in chunks: segment starts
    +1 / ends -1, prefix sum, minus the bases a segment does not observe); per_base_at = sorted positions -> also the (A,C,T,G)
    counts at those positions"""
    from instrain_amd import engine
    sg = w["segs"]
    n_pos = w["n_pos"]
    diff = np.zeros(n_pos + 1, np.int64)
    np.add.at(diff, sg.gpos.astype(np.int64), 1)
    np.add.at(diff, sg.gpos.astype(np.int64) + sg.len, -1)
    cov = np.cumsum(diff[:-1])
    per_base = np.zeros((len(per_base_at), 4), np.int64) if per_base_at is not None else None
    j = np.arange(150, dtype=np.int64)[None, :]
    for c0 in range(0, sg.n_seg, 200_000):
        cd = engine.unpack_codes(sg.bases[c0:c0 + 200_000])
        inside = j < sg.len[c0:c0 + 200_000, None]
        g = sg.gpos[c0:c0 + 200_000, None].astype(np.int64) + j
        miss = inside & (cd >= 4)
        cov -= np.bincount(g[miss], minlength=n_pos)
        if per_base is not None:
            ok = inside & (cd < 4)
            gg, bb = g[ok], cd[ok]
            k = np.searchsorted(per_base_at, gg)
            hit = (k < len(per_base_at)) & (per_base_at[np.minimum(k, len(per_base_at) - 1)] == gg)
            np.add.at(per_base, (k[hit], bb[hit]), 1)
    return cov, per_base


def _slot_coverage(r):
    cov = (r["cov16"] if "cov16" in r else r["cov8"]).astype(np.int64)
    if "saturated" in r:
        cov[r["saturated"]["gpos"]] = r["saturated"]["coverage"]
    return cov


def _same_tables(a, b, keys, what):
    for k in keys:
        x, y = a[k], b[k]
        assert x.shape == y.shape, (what, k, x.shape, y.shape)
        assert x.tobytes() == y.tobytes(), (what, k)


def test_metagenome_slice_vs_oracle_as_segments(ctx):
    """the slice of test_metagenome_slice_vs_oracle, handed over as read segments (generate_segs: the reads whole, as the
    bench's C4 / C5 legs ship them) through the one-shot path and through Pipe.submit_reads, against the C oracle"""
    from instrain_amd import engine, synth
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    meta = synth.Metagenome(14, mean_coverage=5, seed=44, contigs=4, len_lo=30_000, len_hi=70_000, threads=4)
    kept = meta.kept_genomes()
    w = meta.generate_segs(kept)
    kw = dict(min_cov=5, min_freq=0.05, min_snp=10)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs"], n_mm_bins=1, enable_linkage=True, **kw)
    b.run()
    res = b.fetch()
    assert b.timings()["record_bytes"] == 32
    got = prod.to_oracle_layout(res, lambda g: g.astype(np.int64))
    b.close()
    gpos, base, mm, pair = util.segs_to_obs(w["segs"])
    assert len(gpos) == w["n_obs"]
    letters = np.array(list("ACTGN"))
    exp = {"entries": [], "snv": [], "ld": []}
    sb = w["split_bounds"]
    for s, e in zip(sb[:-1], sb[1:]):
        o = oracle.profile_split(gpos, base, mm.astype(np.int64), pair.astype(np.int64), "".join(letters[w["ref_codes"][s:e]]), int(s), lut, fb, **kw)
        for k in exp:
            exp[k].append(o[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    assert len(exp["snv"]) > 100 and len(exp["ld"]) > 20
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=1e-6, what="metagenome slice as segments")
    # the pipe: full tables (want_counts) == the one-shot batch; the default shrunk slot == what they shrink to
    for want_counts in (True, False):
        pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=w["segs"].n_seg, max_splits=len(sb), depth=2, host_threads=4,
                           n_mm_bins=1, enable_linkage=True, want_counts=want_counts, **kw)
        t = pipe.submit_reads(w["ref_codes"], sb, w["segs"])
        r = pipe.collect(t)
        if want_counts:
            _same_tables(r, res, ("counts", "clon", "snv", "ld"), "pipe/want_counts")
        else:
            cov = res["counts"].sum(axis=1)
            assert (_slot_coverage(r) == cov).all()
            assert (r["clon"].view(np.uint32) == res["clon"].view(np.uint32)).all()
            _same_tables(r, res, ("snv", "ld"), "pipe/shrunk")
        pipe.release(t)
        pipe.close()
    # a sub-slice again as segments REBUILT from the observation stream pair by pair (different cuts of the same reads)
    sub = meta.generate(kept[:2])
    segs2 = util.reassemble_segs(sub["obs"]["gpos"], sub["obs"]["base"], sub["obs"]["mm"], sub["pair"])
    outs = []
    for src, pr in ((sub["obs"], sub["pair"]), (segs2, None)):
        b = engine.Batch(ctx, sub["ref_codes"], sub["split_bounds"], src, pr, n_mm_bins=1, enable_linkage=True, **kw)
        b.run()
        outs.append(b.fetch())
        b.close()
    _same_tables(outs[0], outs[1], ("counts", "clon", "snv", "ld"), "reassembled")


def _stream_reads(pipe, ws, depth, check):
    tickets, done = [None] * len(ws), 0
    out = []
    for i, w in enumerate(ws):
        if i - done == depth:
            out.append(check(done, pipe.collect(tickets[done], densify=False)))
            pipe.release(tickets[done])
            done += 1
        tickets[i] = pipe.submit_reads(w["ref_codes"], w["split_bounds"], w["segs"])
    while done < len(ws):
        out.append(check(done, pipe.collect(tickets[done], densify=False)))
        pipe.release(tickets[done])
        done += 1
    return out


def test_c5_per_gpu_shard_read_segments(ctx):
    """configs[4] exactly as bench.py's headline streams it: rank 0's LPT shard of 8, ~4 150 contigs a batch, read segments through a
    read-level pipe with linkage on and the default (shrunk) slot output.  Per batch: the coverage table equals the exact
    per-position coverage computed on the host from the segments; SNV rows are strictly ordered, their counts sum to the
    coverage and follow the segments base by base; LD rows stay inside a scaffold and add up; the stream is idempotent.
    One whole batch is also handed over as observation records: byte-identical tables."""
    from instrain_amd import dist as idist
    from instrain_amd import engine, synth
    meta = synth.Metagenome(1000, total_read_bp=10e9, seed=5)
    kept = meta.kept_genomes()
    shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
    est = (meta.pairs[shard] * 2).astype(np.int64)
    batches = idist.pack_batches(meta.length[shard], est, 40_000_000, 1_000_000)         # (a finer cut than the bench's: 8+ batches)
    assert len(batches) >= 8
    ws = [meta.generate_segs(shard[b]) for b in batches]
    assert max(len(w["scaffold_bounds"]) - 1 for w in ws) >= 300                          # hundreds of contigs in one flat space
    total_bases = sum(w["profiled_bases"] for w in ws)
    assert 0.9e9 < total_bases < 1.4e9
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(w["segs"].n_seg for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=4, n_mm_bins=1, enable_linkage=True, min_snp=20)
    exp_cov = {}

    def check(i, r):
        w = ws[i]
        if i not in exp_cov:
            exp_cov[i] = _expected_coverage(w)[0]
        cov = _slot_coverage(r)
        assert int(cov.sum()) == w["n_obs"]
        assert (cov == exp_cov[i]).all()
        snv, ld = r["snv"], r["ld"]
        g = snv["gpos"].astype(np.int64)
        assert len(snv) == r["sizes"]["n_snv"] > 0 and (np.diff(g) > 0).all()
        assert (snv["cnt"].sum(axis=1) == cov[g]).all() and (cov[g] >= 5).all()
        assert (snv["ref_base"] == w["ref_codes"][g]).all()
        sb = w["scaffold_bounds"]
        if len(ld):
            assert (np.searchsorted(sb, ld["gpos_a"], side="right") == np.searchsorted(sb, ld["gpos_b"], side="right")).all()
            assert (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] == ld["total"]).all()
        cl = engine.dense_clon(cov, r["clon_sparse"], 5) if "clon_sparse" in r else r["clon"]
        assert np.isnan(cl[cov < 5]).all() and not np.isnan(cl[cov >= 5]).any()
        return (int((cov * (np.arange(len(cov)) % 65521)).sum()), len(snv), int(g.sum()), len(ld), snv["cnt"].tobytes())

    sig = [_stream_reads(pipe, ws, 4, check) for _ in range(2)]
    assert sig[0] == sig[1]
    # SNV row counts follow the segments base by base (one batch: per-base counts at the SNV positions from the host decode)
    i = int(np.argmax([w["n_obs"] for w in ws]))
    w = ws[i]
    t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], w["segs"])
    r = pipe.collect(t, densify=False)
    _, per_base = _expected_coverage(w, per_base_at=r["snv"]["gpos"].astype(np.int64))
    assert (per_base == r["snv"]["cnt"]).all()
    keep = {k: r[k].copy() for k in ("snv", "ld")}
    cov_segs = _slot_coverage(r)
    clon_segs = engine.dense_clon(cov_segs, r["clon_sparse"], 5) if "clon_sparse" in r else r["clon"].copy()
    pipe.release(t)
    pipe.close()
    # the same batch as observation records (isx_pipe_submit): byte-identical tables
    wo = meta.generate(w["genomes"])
    assert wo["n_obs"] == w["n_obs"]
    po = engine.Pipe(ctx, max_pos=wo["n_pos"], max_obs=wo["n_obs"], max_splits=len(wo["split_bounds"]), depth=1, n_mm_bins=1,
                     enable_linkage=True, min_snp=20, jump_slack=0.5)
    t = po.submit(wo["ref_codes"], wo["split_bounds"], wo["obs"], wo["pair"])
    ro = po.collect(t, densify=False)
    assert (_slot_coverage(ro) == cov_segs).all()
    clon_obs = engine.dense_clon(cov_segs, ro["clon_sparse"], 5) if "clon_sparse" in ro else ro["clon"]
    assert (clon_obs.view(np.uint32) == clon_segs.view(np.uint32)).all()
    _same_tables(ro, keep, ("snv", "ld"), "C5 batch: observations vs segments")
    po.release(t)
    po.close()


def test_c4_per_gpu_shard_read_segments(ctx):
    """configs[3] as read segments: rank 0's LPT shard of 8 (100 genomes x 50 contigs, 50x mean, log-normal abundances), linkage
    on, depth-2 pipe.  Per batch the properties of test_c4_per_gpu_shard_properties + the exact coverage from the host; the
    deepest batch also goes through the observation hand-over: byte-identical SNV and LD tables."""
    from instrain_amd import dist as idist
    from instrain_amd import engine, synth
    meta = synth.Metagenome(100, mean_coverage=50, seed=4)
    kept = meta.kept_genomes()
    shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
    est = (meta.pairs[shard] * 2).astype(np.int64)
    batches = idist.pack_batches(meta.length[shard], est, 40_000_000, 3_000_000)
    pipe, cap, total, deepest = None, None, 0, None
    for b in batches:
        w = meta.generate_segs(shard[b])
        if pipe is None or w["segs"].n_seg > cap[1] or w["n_pos"] > cap[0] or len(w["split_bounds"]) > cap[2]:
            if pipe is not None:
                pipe.close()
            cap = (max(w["n_pos"], 30_000_000), int(w["segs"].n_seg * 1.2), len(w["split_bounds"]) + 4096)
            pipe = engine.Pipe(ctx, max_pos=cap[0], max_obs=0, max_segs=cap[1], max_splits=cap[2], depth=2, n_mm_bins=1,
                               enable_linkage=True, min_snp=20)
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], w["segs"])
        r = pipe.collect(t, densify=False)
        cov = _slot_coverage(r)
        exp, _ = _expected_coverage(w)
        assert (cov == exp).all() and int(cov.sum()) == w["n_obs"]
        snv, ld = r["snv"], r["ld"]
        g = snv["gpos"].astype(np.int64)
        assert len(snv) > 1000 and (np.diff(g) > 0).all() and (snv["cnt"].sum(axis=1) == cov[g]).all() and (cov[g] >= 5).all()
        sb = w["scaffold_bounds"]
        assert len(ld) > 100
        assert (np.searchsorted(sb, ld["gpos_a"], side="right") == np.searchsorted(sb, ld["gpos_b"], side="right")).all()
        assert (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] == ld["total"]).all()
        rare = r["rare"]["gpos"] if "rare" in r else np.flatnonzero(~np.isnan(r["clon_r"]))
        assert (rare == np.flatnonzero(cov >= 50)).all()
        total += w["profiled_bases"]
        if deepest is None or w["n_obs"] / w["n_pos"] > deepest[0]:
            deepest = (w["n_obs"] / w["n_pos"], w["genomes"].copy(), {k: r[k].copy() for k in ("snv", "ld")}, cov)
        pipe.release(t)
        del w, r
    pipe.close()
    assert 1.5e9 < total < 4e9
    wo = meta.generate(deepest[1])
    po = engine.Pipe(ctx, max_pos=wo["n_pos"], max_obs=wo["n_obs"], max_splits=len(wo["split_bounds"]), depth=1, n_mm_bins=1,
                     enable_linkage=True, min_snp=20, jump_slack=0.3)
    t = po.submit(wo["ref_codes"], wo["split_bounds"], wo["obs"], wo["pair"])
    ro = po.collect(t, densify=False)
    assert (_slot_coverage(ro) == deepest[3]).all()
    _same_tables(ro, deepest[2], ("snv", "ld"), "C4 batch: observations vs segments")
    po.release(t)
    po.close()


def test_c5_headline_configuration_exactly(monkeypatch):
    """bench.py's headline, as it is timed: rank 0's C5 shard cut by bench.C5_BATCH_POS / C5_BATCH_SEGS (nothing restated), lean slots,
    a context with bench.C5_RESERVE_CUS compute units reserved, pipe depth 8, every batch handed over by isx_pipe_submit_planes (bit
    planes from the caller's arrays, staged inside the submit) -- and the same batches as pre-staged wires (the round-4 way, an extra
    of the line).  Every batch: exact per-position coverage from the host, SNV rows ordered / consistent / on the reference, LD rows
    inside a scaffold and adding up, clonality exactly where coverage reaches min_cov; the largest batch: SNV counts base by base;
    one batch byte-equal to a plain pipe (plain slots, isx_segs hand-over, no reserve, depth 1)."""
    import bench
    from instrain_amd import engine, synth
    monkeypatch.setenv("ISX_DIR_CHECK", "1")       # every window directory made on the stager's threads is compared with the plain one (aborts)
    lut, fb = util.load_lut()
    ctx5 = engine.Context(0, reserve_cus=bench.C5_RESERVE_CUS)
    ctx5.set_null_model(lut, fb)
    run = bench.C5Run(ctx5, 0, 8, 8, depth=8)
    assert run.pipe.read_level and bench.LEAN_SLOTS and bench.C5_LINKAGE
    assert max(w["n_pos"] for w in run.ws) <= bench.C5_BATCH_POS and max(w["n_seg"] for w in run.ws) <= bench.C5_BATCH_SEGS + 2 * 10 ** 5
    assert 2 <= len(run.ws) <= 8 and 0.9e9 < run.bases < 1.5e9
    exp = {}
    big = int(np.argmax([w["n_pos"] for w in run.ws]))
    seen = {False: [], True: []}

    def make_check(staged):
        def check(i, r):
            w = run.ws[i]
            if i not in exp:
                exp[i] = bench.planes_coverage(w)[0]
            assert ("cov4" in r) == (w["n_obs"] < 6 * w["n_pos"])                # lean slot: the 4-bit plane for a shallow batch
            cov = engine.dense_cov(r, w["n_pos"]).astype(np.int64)
            if "saturated" in r:
                cov[r["saturated"]["gpos"]] = r["saturated"]["coverage"]
            assert (cov == exp[i]).all() and int(cov.sum()) == w["n_obs"]
            snv, ld = r["snv"], r["ld"]
            g = snv["gpos"].astype(np.int64)
            assert len(snv) == r["sizes"]["n_snv"] > 0 and (np.diff(g) > 0).all()
            assert (snv["cnt"].sum(axis=1) == cov[g]).all() and (cov[g] >= 5).all()
            ref2 = w["ref_planes"].plane2
            assert (snv["ref_base"] == ((ref2[g >> 2] >> (2 * (g & 3)).astype(np.uint8)) & 3)).all()
            sb = w["scaffold_bounds"]
            if len(ld):
                assert (np.searchsorted(sb, ld["gpos_a"], side="right") == np.searchsorted(sb, ld["gpos_b"], side="right")).all()
                assert (ld["countAB"].astype(np.int64) + ld["countAb"] + ld["countaB"] + ld["countab"] == ld["total"]).all()
            cl = engine.dense_clon(cov, r["clon_sparse"], 5) if "clon_sparse" in r else r["clon"]
            assert np.isnan(cl[cov < 5]).all() and not np.isnan(cl[cov >= 5]).any()
            assert r["stats"]["record_bytes"] == 32 and r["stats"]["encode_passes"] == (0 if staged else r["stats"]["encode_passes"])
            if i == big:
                _, per_base = bench.planes_coverage(w, per_base_at=g)
                assert (per_base == snv["cnt"]).all()
            seen[staged].append((i, snv.tobytes(), ld.tobytes(), cl.view(np.uint32).tobytes(), r["sizes"]))
        return check

    sig = run.verify_pass()                                # bench.py's own untimed pass (incl. its exact check of the largest batch)
    assert len(sig) == len(run.ws) and run.exact_checked["batch"] == big
    run.checked_pass(make_check(False), staged=False)
    run.checked_pass(make_check(True), staged=True)
    assert len(seen[False]) == len(run.ws) and sorted(x[0] for x in seen[True]) == list(range(len(run.ws)))
    by_i = {x[0]: x for x in seen[False]}
    for x in seen[True]:                                    # staged wires == planes handed over inside the step
        assert x[1:] == by_i[x[0]][1:]
    # the timed configuration's passes are deterministic
    stats = []
    run.run(2, stats)
    run.check_timed(stats)
    # one batch through a PLAIN pipe: plain slots, segments (3-bit words) rebuilt from the planes, no reserve, depth 1
    k = int(np.argmin([w["n_pos"] for w in run.ws]))
    w = run.ws[k]
    pb = w["planes"]
    two = np.unpackbits(np.ascontiguousarray(pb.planes[:, 0:5]).view(np.uint8), axis=1, bitorder="little")[:, :300].reshape(pb.n_seg, 150, 2)
    code = (two[:, :, 0] | (two[:, :, 1] << 1)).astype(np.uint8)
    skip = np.unpackbits(np.ascontiguousarray(pb.planes[:, 5:8]).view(np.uint8), axis=1, bitorder="little")[:, :150].astype(bool)
    code[skip | (np.arange(150)[None, :] >= pb.len[:, None])] = 4
    segs = engine.SegBatch(pb.gpos, pb.len, engine.pack_codes(code), None, pb.pair)
    del two, skip, code
    rp = w["ref_planes"]
    refc = ((rp.plane2[np.arange(w["n_pos"]) >> 2] >> (2 * (np.arange(w["n_pos"]) & 3)).astype(np.uint8)) & 3).astype(np.uint8)
    if rp.nplane is not None:
        refc[np.unpackbits(rp.nplane, bitorder="little")[:w["n_pos"]].astype(bool)] = 4
    run.close()
    ctx5.close()
    ctx = engine.Context(0)
    ctx.set_null_model(lut, fb)
    plain = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=segs.n_seg, max_splits=len(w["split_bounds"]), depth=1, host_threads=8,
                        pin_threads=False, n_mm_bins=1, enable_linkage=True, min_snp=20)
    t = plain.submit_reads(refc, w["split_bounds"], segs)
    r = plain.collect(t)
    cl = r["clon"]
    assert (r["cov16"].astype(np.int64) == exp[k]).all()
    assert r["snv"].tobytes() == by_i[k][1] and r["ld"].tobytes() == by_i[k][2] and cl.view(np.uint32).tobytes() == by_i[k][3]
    assert r["sizes"] == by_i[k][4]
    plain.release(t)
    plain.close()
    ctx.close()


def test_c5_headline_whole_database_at_n1():
    """The headline as the driver runs it at N = 1: ALL of the kept database through one GPU -- bench.C5Run(rank 0 of 1): its 25 batches, the
    reference planes registered (isx_host_register), lean slots, reserve, depth 8.  bench.py's own untimed verify pass (every batch: coverage
    sums to the observations handed over, SNV rows ordered and consistent, LD counts add up; the largest batch exact), then exact per-position
    coverage from the host for five batches spread over the pass, then two passes exactly as they are timed, every batch compared with the
    verified pass by row counts and the checksum of its SNV + LD bytes (the shard test above is one rank of eight)."""
    import bench
    from instrain_amd import engine
    lut, fb = util.load_lut()
    ctx5 = engine.Context(0, reserve_cus=bench.C5_RESERVE_CUS)
    ctx5.set_null_model(lut, fb)
    run = bench.C5Run(ctx5, 0, 1, 16, depth=8)
    try:
        assert 20 <= len(run.ws) <= 30 and 8.5e9 < run.bases < 10.5e9 and run.ref_registered == bench.REGISTER_REF
        sig = run.verify_pass()
        assert len(sig) == len(run.ws) and all(s[0] > 0 for s in sig)
        pick = set(np.linspace(0, len(run.ws) - 1, 5).astype(int).tolist())
        seen = []

        def check(i, r):
            if i not in pick:
                return
            w = run.ws[i]
            exp = bench.planes_coverage(w)[0]
            cov = engine.dense_cov(r, w["n_pos"]).astype(np.int64)
            if "saturated" in r:
                cov[r["saturated"]["gpos"]] = r["saturated"]["coverage"]
            assert (cov == exp).all() and int(cov.sum()) == w["n_obs"]
            g = r["snv"]["gpos"].astype(np.int64)
            assert (np.diff(g) > 0).all() and (r["snv"]["cnt"].sum(axis=1) == cov[g]).all()
            seen.append(i)

        run.checked_pass(check, staged=False)
        assert sorted(seen) == sorted(pick)
        stats = []
        run.run(2, stats)
        assert len(stats) == 2 * len(run.ws)
        run.check_timed(stats)
    finally:
        run.close()
        ctx5.close()
