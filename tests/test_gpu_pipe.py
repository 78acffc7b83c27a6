"""GPU: the streaming hand-over (isx_pipe_*).  Every batch of a stream must come back with exactly the
tables the one-shot path (isx_batch_create / run / fetch, itself pinned against the oracle and the golden
vectors in test_gpu_parity.py) gives for the same input -- and directly against the oracle for a sample."""
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


def small_workload(seed, genome_len, coverage, skip_mm, n_sites=None):
    from instrain_amd import synth
    w = synth.make_workload(genome_len=genome_len, coverage=coverage, n_sites=n_sites or genome_len // 200, seed=seed,
                            skip_mm=skip_mm, af_lo=0.2, af_hi=0.5)
    if seed % 2:            # every other workload: a reference with positions that are not A/C/T/G (a slot's 2-bit plane + its bit plane)
        rng = np.random.Generator(np.random.PCG64(seed + 500))
        ref = w["ref_codes"].copy()
        ref[rng.random(len(ref)) < 0.003] = 4
        ref[len(ref) // 2:len(ref) // 2 + 300] = 4
        w["ref_codes"] = ref
    return w


def one_shot(ctx, w, **kw):
    from instrain_amd import engine
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=w["n_mm_bins"], **kw)
    b.run()
    res, sizes = b.fetch(), b.sizes()
    b.close()
    return res, sizes


def check_shrunk(got, exp, what):
    """the shrunk hand-back (coverage16 + clonality + sparse rarefied clonality) against the full tables"""
    cov = exp["counts"].sum(axis=1, dtype=np.int64)
    assert (got["cov16"] == np.minimum(cov, 65535)).all(), (what, "cov16")
    assert got["n_saturated"] == int((cov >= 65535).sum())
    assert (got["clon"].view(np.uint32) == exp["clon"].view(np.uint32)).all(), (what, "clon")
    k = np.flatnonzero(~np.isnan(exp["clon_r"]))
    assert len(got["rare"]) == len(k) and (got["rare"]["gpos"] == k).all(), (what, "rare positions")
    assert (got["rare"]["clon_rarefied"].view(np.uint32) == exp["clon_r"][k].view(np.uint32)).all(), (what, "rare values")


def same_tables(got, exp, what):
    if "counts" in exp and "cov16" in got:
        check_shrunk(got, exp, what)
    for k in ("counts", "clon", "clon_r", "entries", "snv", "ld"):
        if k not in exp or k not in got:
            continue
        a, e = got[k], exp[k]
        assert a.shape == e.shape, (what, k, a.shape, e.shape)
        if a.dtype.names:
            for f in a.dtype.names:
                x, y = a[f], e[f]
                if x.dtype.kind == "f":
                    assert (x.view("u%d" % x.dtype.itemsize) == y.view("u%d" % y.dtype.itemsize)).all(), (what, k, f)
                else:
                    assert (x == y).all(), (what, k, f)
        elif a.dtype.kind == "f":
            assert (a.view(np.uint32) == e.view(np.uint32)).all(), (what, k)
        else:
            assert (a == e).all(), (what, k)


@pytest.mark.parametrize("linkage,rarefied", [(False, 20), (True, 20), (False, 36)])
def test_stream_of_distinct_batches_dense(ctx, linkage, rarefied):
    """one mm bin (2-byte records): 7 distinct batches of different sizes through 3 slots, collected in
    order while later ones are in flight; bit-identical to the one-shot path.  rarefied 20: most positions
    have a clonTR value (it comes back as the dense array); 36: few do (the sparse list)"""
    from instrain_amd import engine
    ws = [small_workload(100 + i, 60_000 + 17_000 * (i % 3), 25 + 5 * (i % 2), True) for i in range(7)]
    kw = dict(enable_linkage=linkage, min_snp=5, seed=11, rarefied_coverage=rarefied)
    exp = [one_shot(ctx, w, **kw) for w in ws]
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=3, host_threads=4, n_mm_bins=1,
                       want_counts=linkage, **kw)           # once with the full tables, once with the shrunk ones only
    tickets = []
    done = 0
    for i, w in enumerate(ws):
        if len(tickets) - done == 3:                      # every slot busy: collect the oldest first
            r = pipe.collect(tickets[done])
            same_tables(r, exp[done][0], "batch %d" % done)
            assert r["sizes"] == exp[done][1]
            assert r["stats"]["record_bytes"] == 2 and r["stats"]["encode_passes"] == 1
            pipe.release(tickets[done])
            done += 1
        tickets.append(pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], w["pair"] if linkage else None))
    while done < len(ws):
        r = pipe.collect(tickets[done])
        same_tables(r, exp[done][0], "batch %d" % done)
        assert r["sizes"] == exp[done][1]
        if linkage:
            assert r["sizes"]["n_ld"] > 0
        pipe.release(tickets[done])
        done += 1
    assert tickets == list(range(7))
    pipe.close()


def test_stream_mm_profiling_with_linkage(ctx):
    """mm profiling on (4-byte records, entry table) + linkage through the pipe == one-shot path"""
    from instrain_amd import engine
    ws = [small_workload(200 + i, 50_000 + 9_000 * i, 30, False) for i in range(4)]
    M = max(w["n_mm_bins"] for w in ws)
    for w in ws:
        w["n_mm_bins"] = M
    kw = dict(enable_linkage=True, min_snp=5, seed=5, rarefied_coverage=20)
    exp = [one_shot(ctx, w, **kw) for w in ws]
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=2, host_threads=3, n_mm_bins=M, **kw)
    for rnd in range(2):                                  # slots are reused: second round through the same slots
        t0 = pipe.submit(ws[2 * rnd]["ref_codes"], ws[2 * rnd]["split_bounds"], ws[2 * rnd]["obs"], ws[2 * rnd]["pair"])
        t1 = pipe.submit(ws[2 * rnd + 1]["ref_codes"], ws[2 * rnd + 1]["split_bounds"], ws[2 * rnd + 1]["obs"], ws[2 * rnd + 1]["pair"])
        for t, i in ((t0, 2 * rnd), (t1, 2 * rnd + 1)):
            r = pipe.collect(t)
            assert r["stats"]["record_bytes"] == 4
            same_tables(r, exp[i][0], "mm batch %d" % i)
            assert r["sizes"] == exp[i][1] and r["sizes"]["n_entries"] > 0 and r["sizes"]["n_ld"] > 0
            pipe.release(t)
    pipe.close()


@pytest.mark.parametrize("reads", [False, True])
def test_shrunk_entry_hand_back_equals_the_entries(ctx, reads):
    """isx_pipe_fetch_entries_shrunk (four 4-byte columns: position | mm << 24 | coverage of the level | clonality | rarefied
    clonality) == the 32-byte entries of the same collected batch, column by column and bit by bit; big enough for the staging
    detour of the copy (> 1 MiB pieces) and small enough for the direct one"""
    from instrain_amd import engine, synth
    for genome, cov in ((60_000, 30), (1_500_000, 40)):
        w = small_workload(300 + genome % 7, genome, cov, False)
        M = w["n_mm_bins"]
        kw = dict(enable_linkage=False, rarefied_coverage=20, seed=9, n_mm_bins=M)
        if reads:
            segs = synth.segs_from_obs(w["obs"], w["pair"])
            pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=segs.n_seg, max_splits=len(w["split_bounds"]), depth=1, host_threads=3, **kw)
            t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs)
        else:
            pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"], max_splits=len(w["split_bounds"]), depth=1, host_threads=3, **kw)
            t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], None)
        full = pipe.collect(t)["entries"].copy()
        g, mc, cl, cr = pipe.collect(t, shrunk_entries=True)["entries_soa"]
        assert len(full) == len(g) > 50_000
        assert (g == full["gpos"]).all() and ((mc >> 24) == full["mm"]).all()
        assert ((mc & 0xFFFFFF) == full["cnt"].sum(axis=1)).all()
        assert cl.tobytes() == full["clon"].tobytes() and cr.tobytes() == full["clon_rarefied"].tobytes()
        assert np.isfinite(cr).sum() > 100
        pipe.release(t)
        pipe.close()


@pytest.mark.parametrize("skip_mm,ring_kib", [(True, 64), (False, 64), (False, 4096)])
def test_staging_ring(ctx, skip_mm, ring_kib):
    """records staged through a small pinned ring (the mode a pipe picks by itself beyond 512 MiB of records): waves of
    half a ring leave for the device while the next is encoded; tables identical to the one-shot path, slots reused,
    and a batch with more pair-id runs than the slot was created for"""
    from instrain_amd import engine
    ws = [small_workload(400 + i, 40_000 + 11_000 * i, 30, skip_mm) for i in range(4)]
    M = max(w["n_mm_bins"] for w in ws)
    for w in ws:
        w["n_mm_bins"] = M
    # the last batch: every record its own pair (far more runs than cap_rec / 64)
    w = ws[3]
    w["pair"] = np.arange(len(w["obs"]), dtype=np.uint32)
    kw = dict(enable_linkage=True, min_snp=5, seed=5, rarefied_coverage=20)
    exp = [one_shot(ctx, w, **kw) for w in ws]
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws),
                       max_splits=max(len(w["split_bounds"]) for w in ws), depth=2, host_threads=3, n_mm_bins=M,
                       ring_kib=ring_kib, **kw)          # 4096: the entry table also comes back through the ring's halves
    for rnd in range(2):
        ts = [pipe.submit(ws[i]["ref_codes"], ws[i]["split_bounds"], ws[i]["obs"], ws[i]["pair"]) for i in (2 * rnd, 2 * rnd + 1)]
        for t, i in zip(ts, (2 * rnd, 2 * rnd + 1)):
            r = pipe.collect(t)
            same_tables(r, exp[i][0], "ring batch %d" % i)
            assert r["sizes"] == exp[i][1]
            pipe.release(t)
    pipe.close()


def test_pipe_against_oracle(ctx):
    """a pipe batch directly against the C oracle, split by split"""
    from instrain_amd import engine
    from oracle import oracle
    from tests import prod
    lut, fb = util.load_lut()
    w = small_workload(300, 45_000, 40, False)
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"], max_splits=len(w["split_bounds"]), depth=2,
                       host_threads=2, n_mm_bins=w["n_mm_bins"], enable_linkage=True, min_snp=5)
    t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], w["pair"])
    r = pipe.collect(t)
    got = prod.to_oracle_layout(r, lambda g: g.astype(np.int64))
    pipe.release(t)
    pipe.close()
    letters = np.array(list("ACTGN"))
    exp = {"entries": [], "snv": [], "ld": []}
    b = w["split_bounds"]
    for s, e in zip(b[:-1], b[1:]):
        o = oracle.profile_split(w["obs"]["gpos"].astype(np.int64), w["obs"]["base"], w["obs"]["mm"].astype(np.int64),
                                 w["pair"].astype(np.int64), "".join(letters[w["ref_codes"][s:e]]), int(s), lut, fb, min_snp=5)
        for k in exp:
            exp[k].append(o[k])
    exp = {k: np.concatenate(v) for k, v in exp.items()}
    assert len(exp["ld"]) > 20 and len(exp["snv"]) > 50
    util.assert_same(util.canon_from_struct(got), util.canon_from_struct(exp), float_tol=1e-6, what="pipe vs oracle")


def test_jumping_stream_learns_its_slack(ctx):
    """a database-like batch (islands of reads 100 kbp apart): the first submit needs the second, exact
    layout; the pipe then sets slack aside and the same stream goes through in one pass"""
    from instrain_amd import engine
    w = small_workload(400, 30_000, 30, True)
    K, step = 6, 100_000
    n_pairs = int(w["pair"].max()) + 1
    obs = np.concatenate([w["obs"]] * K)
    obs["gpos"] = np.concatenate([w["obs"]["gpos"] + k * step for k in range(K)])
    pair = np.concatenate([w["pair"] + k * n_pairs for k in range(K)]).astype(np.uint32)
    n_pos = (K - 1) * step + w["n_pos"]
    ref = np.zeros(n_pos, np.uint8)
    bounds = [0]
    for k in range(K):
        ref[k * step:k * step + w["n_pos"]] = w["ref_codes"]
        bounds += [k * step + int(x) for x in w["split_bounds"][1:]]
        if k + 1 < K:
            bounds.append((k + 1) * step)
    bounds = np.unique(bounds)
    big = {"ref_codes": ref, "split_bounds": bounds, "obs": obs, "pair": pair, "n_mm_bins": 1}
    exp, sizes = one_shot(ctx, big, enable_linkage=True, min_snp=5)
    pipe = engine.Pipe(ctx, max_pos=n_pos, max_obs=len(obs), max_splits=len(bounds), depth=2, host_threads=4,
                       n_mm_bins=1, enable_linkage=True, min_snp=5)
    passes = []
    for _ in range(3):
        t = pipe.submit(ref, bounds, obs, pair)
        r = pipe.collect(t)
        same_tables(r, exp, "jumping")
        assert r["sizes"] == sizes
        passes.append(r["stats"]["encode_passes"])
        pipe.release(t)
    assert passes[0] == 2 and passes[-1] == 1, passes
    pipe.close()


def test_tables_grow_inside_a_slot(ctx):
    """SNS rows at every position outgrow the slot's SNV table: collect grows it, repeats the pass and still
    returns the right tables; the next batch in the same slot is unaffected"""
    from instrain_amd import engine
    n_pos, depth = 1_200_000, 6
    pos = np.repeat(np.arange(n_pos, dtype=np.uint32), depth)
    obs = engine.pack_obs(pos, np.ones(len(pos), np.uint8), np.zeros(len(pos), int))
    ref = np.zeros(n_pos, np.uint8)
    pipe = engine.Pipe(ctx, max_pos=n_pos, max_obs=len(obs), max_splits=16, depth=2, host_threads=4, n_mm_bins=1,
                       enable_linkage=False)
    t = pipe.submit(ref, [0, n_pos], obs)
    r = pipe.collect(t)
    assert r["sizes"]["n_snv"] == n_pos and (r["snv"]["cls"] == 2).all() and (r["snv"]["gpos"] == np.arange(n_pos)).all()
    assert (r["snv"]["cnt"][:, 1] == depth).all() and (r["snv"]["cnt"][:, [0, 2, 3]] == 0).all()
    assert (r["cov16"] == depth).all() and (r["clon"] == 1.0).all() and len(r["rare"]) == 0
    pipe.release(t)
    w = small_workload(500, 40_000, 20, True)
    exp, sizes = one_shot(ctx, w, enable_linkage=False)
    t2 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    t3 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    for t in (t2, t3):
        r = pipe.collect(t)
        same_tables(r, exp, "after growth")
        pipe.release(t)
    pipe.close()


def test_pipe_errors(ctx):
    from instrain_amd import engine
    from instrain_amd._lib import IsxError
    w = small_workload(600, 20_000, 10, True)
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"], max_splits=len(w["split_bounds"]), depth=2, host_threads=2)
    t0 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    t1 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    with pytest.raises(IsxError) as e:                    # both slots busy
        pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    assert e.value.code == -6
    with pytest.raises(IsxError):
        pipe.collect(5)
    pipe.release(t0)                                      # release without collect is allowed
    r = pipe.collect(t1)
    assert r["sizes"]["n_snv"] >= 0
    pipe.release(t1)
    with pytest.raises(IsxError):
        pipe.release(t1)
    big = small_workload(601, 40_000, 10, True)
    with pytest.raises(IsxError) as e:
        pipe.submit(big["ref_codes"], big["split_bounds"], big["obs"])
    assert e.value.code == -3
    # an empty batch is legal
    t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"][:0])
    r = pipe.collect(t)
    assert r["sizes"]["n_snv"] == 0 and (r["cov16"] == 0).all() and np.isnan(r["clon"]).all() and len(r["rare"]) == 0
    pipe.release(t)
    pipe.close()


def test_single_slot_pipe_with_a_large_result_block(ctx):
    """depth 1 and more than 64 MB of dense results: the block is plain (not pinned) memory; same tables"""
    from instrain_amd import engine
    w = small_workload(800, 50_000, 25, True)
    exp, sizes = one_shot(ctx, w, enable_linkage=False, rarefied_coverage=20)
    pipe = engine.Pipe(ctx, max_pos=12_000_000, max_obs=w["n_obs"], max_splits=len(w["split_bounds"]), depth=1, host_threads=2,
                       n_mm_bins=1, enable_linkage=False, rarefied_coverage=20)
    for _ in range(2):
        t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
        r = pipe.collect(t)
        same_tables(r, exp, "pageable result block")
        assert r["sizes"] == sizes
        pipe.release(t)
    pipe.close()


def test_error_found_while_finishing_reaches_collect(ctx):
    """an mm level the pipe has no bin for is only seen by the kernel; the pipe's finishing thread finds the flag, collect()
    of THAT batch raises with its message, the batches around it are unaffected"""
    from instrain_amd import engine
    from instrain_amd._lib import IsxError
    w = small_workload(700, 30_000, 15, False)
    M = int(w["n_mm_bins"])
    exp, sizes = one_shot(ctx, w, enable_linkage=False)
    bad = w["obs"].copy()
    bad["mm"][len(bad) // 2] = M + 3
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"], max_splits=len(w["split_bounds"]), depth=3, host_threads=2,
                       n_mm_bins=M, enable_linkage=False)
    t0 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    t1 = pipe.submit(w["ref_codes"], w["split_bounds"], bad)
    t2 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    same_tables(pipe.collect(t0), exp, "before the bad batch")
    with pytest.raises(IsxError) as e:
        pipe.collect(t1)
    assert "mm" in str(e.value)
    same_tables(pipe.collect(t2), exp, "after the bad batch")
    for t in (t0, t1, t2):
        pipe.release(t)
    t3 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])          # the slot that held the bad batch
    t4 = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"])
    same_tables(pipe.collect(t4), exp, "reused slot")
    pipe.release(t3); pipe.release(t4)
    pipe.close()


def test_submit_bam_equals_array_submit(ctx, tmp_path):
    """the fused path (front end expanding straight into the slot's staging, pair ids as runs) gives exactly the tables of
    expand_refs + array submit and of the one-shot path, for a messy multi-scaffold BAM, with and without mm profiling"""
    from instrain_amd import engine
    from tests import bamwriter
    refs = [("scafA", 4000), ("scafB", 900), ("scafC", 12500), ("empty", 700)]
    rng = np.random.Generator(np.random.PCG64(5))
    seqs = ["".join(rng.choice(list("ACGT"), ln)) for _, ln in refs]
    path = str(tmp_path / "m.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(21, refs[:3], 9000))
    bam = engine.BamFile(path, threads=4)
    bam.scan()
    bam.filter(min_read_ani=0.9)
    for skip_mm in (False, True):
        n_mm = 1 if skip_mm else bam.info["max_mm"] + 1
        kw = dict(min_cov=5, min_freq=0.05, min_snp=8, n_mm_bins=n_mm, enable_linkage=True, rarefied_coverage=10, seed=2)
        for sel in ([0, 1, 2, 3], [2], [0, 3]):
            ekw = dict(min_read_ani=0.9, skip_mm=skip_mm, window_length=1000)
            obs, pair, bounds, sref = bam.expand_refs(sel, **ekw)
            ref = np.concatenate([engine.encode_seq(seqs[t]) for t in sel])
            b = engine.Batch(ctx, ref, bounds, obs, pair, **kw)
            b.run()
            exp, sizes = b.fetch(), b.sizes()
            b.close()
            pipe = engine.Pipe(ctx, max_pos=len(ref), max_obs=max(len(obs), 1), max_splits=len(bounds), depth=2, host_threads=3,
                               jump_slack=1.0, want_counts=True, **kw)
            t1 = pipe.submit(ref, bounds, obs, pair)
            t2 = pipe.submit_bam(bam, sel, ref, None, **ekw)
            assert bam.info["n_obs"] == len(obs) and bam.info["n_splits"] == len(bounds) - 1
            for t in (t1, t2):
                r = pipe.collect(t)
                same_tables(r, exp, "bam %s %s" % (sel, skip_mm))
                assert r["sizes"] == sizes
                pipe.release(t)
            pipe.close()
            if sel == [2]:
                assert sizes["n_ld"] > 0 and sizes["n_snv"] > 100
    bam.close()


def test_summaries_on_a_reused_slot_with_growing_batches(ctx):
    """a slot's summary buffers follow the batch: a small batch first, then a larger one through the same slot (the
    position-sized scratch of isx_batch_summarize used to keep the first batch's size)"""
    from instrain_amd import engine, synth
    ws = [synth.make_workload(genome_len=g, coverage=12, n_sites=40, seed=31 + i) for i, g in enumerate((30_000, 400_000, 90_000))]
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws), max_splits=64, depth=1,
                       host_threads=2, pin_threads=False, n_mm_bins=1, enable_linkage=False, want_counts=True)
    for w in ws:
        t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], None)
        r = pipe.collect(t)
        lv, _ = r["slot"].summarize([0, w["n_pos"] // 3, w["n_pos"]])
        cov = r["counts"].sum(axis=1)
        for j, (a, e) in enumerate(((0, w["n_pos"] // 3), (w["n_pos"] // 3, w["n_pos"]))):
            assert int(lv[j, 0]["sum_cov"]) == int(cov[a:e].sum()) and int(lv[j, 0]["nonzero"]) == int((cov[a:e] > 0).sum())
            assert float(lv[j, 0]["median_cov"]) == float(np.median(cov[a:e]))
        pipe.release(t)
    pipe.close()


@pytest.mark.parametrize("reads", [False, True])
@pytest.mark.parametrize("linkage", [False, True])
def test_shallow_batch_comes_back_sparse(ctx, reads, linkage):
    """a slot that keeps no count table hands back 1-byte coverage for a shallow batch (+ exact coverage of positions at 255
    or beyond) and the clonalities other than 1.0 as a sorted list; densified, every table equals the resident batch's"""
    from instrain_amd import engine, synth
    w = synth.make_workload(genome_len=600_000, coverage=3, n_sites=900, seed=41)
    o = w["obs"].copy()
    # a pile of 400 reads over one spot: coverage far beyond 255 inside a shallow batch
    extra = np.zeros(400 * 50, dtype=o.dtype)
    extra["gpos"] = np.tile(np.arange(300_000, 300_050, dtype=np.uint32), 400)
    extra["base"] = np.repeat(np.arange(400) % 4, 50)
    obs = np.concatenate([o, extra])
    pair = np.concatenate([w["pair"], np.repeat(np.arange(400, dtype=np.uint32) + w["n_pairs"], 50)])
    kw = dict(n_mm_bins=1, enable_linkage=linkage, min_snp=5, min_cov=5, rarefied_coverage=4)
    a = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], obs, pair if linkage else None, **kw)
    a.run()
    exp, sizes = a.fetch(), a.sizes()
    a.close()
    segs = synth.segs_from_obs(obs, pair)
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=len(obs) if not reads else 0, max_segs=segs.n_seg if reads else 0,
                       max_splits=len(w["split_bounds"]), depth=2, host_threads=2, pin_threads=False, **kw)
    for rep in range(2):
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs) if reads else pipe.submit(w["ref_codes"], w["split_bounds"], obs, pair if linkage else None)
        raw = pipe.collect(t, densify=False)
        assert "cov8" in raw and "clon_sparse" in raw and "cov16" not in raw and "clon" not in raw
        cov = exp["counts"].sum(axis=1, dtype=np.int64)
        assert (raw["cov8"] == np.minimum(cov, 255)).all()
        sat = raw["saturated"]
        k = np.flatnonzero(cov >= 255)
        assert raw["n_saturated"] == len(k) == 50 and (np.sort(sat["gpos"]) == k).all()
        assert (sat["coverage"][np.argsort(sat["gpos"])] == cov[k]).all()
        cs = raw["clon_sparse"]
        has = np.flatnonzero(~np.isnan(exp["clon"]))
        other = np.flatnonzero(~np.isnan(exp["clon"]) & (exp["clon"] != 1.0))
        assert (cs["gpos"] == other).all() and cs["clon"].tobytes() == exp["clon"][other].tobytes() and 0 < len(other) < len(has)
        assert (np.flatnonzero(cov >= 5) == has).all()          # ... and every other position with coverage >= min_cov has 1.0
        assert engine.dense_clon(raw["cov8"], cs, 5).tobytes() == exp["clon"].tobytes()
        assert raw["sizes"] == sizes
        assert raw["snv"].tobytes() == exp["snv"].tobytes()
        if linkage:
            assert raw["ld"].tobytes() == exp["ld"].tobytes()
        # the rarefied clonality list and the device summaries work without the count table
        r = exp["clon_r"]
        hr = np.flatnonzero(~np.isnan(r))
        assert (raw["rare"]["gpos"] == hr).all()
        lv, _ = raw["slot"].summarize([0, w["n_pos"]])
        assert int(lv[0, 0]["sum_cov"]) == int(cov.sum()) and int(lv[0, 0]["nonzero"]) == int((cov > 0).sum())
        assert int(lv[0, 0]["counted"]) == len(has)
        with pytest.raises(engine.IsxError, match="count table"):
            raw["slot"].fetch()
        pipe.release(t)
        # the default collect() rebuilds the arrays every other batch has
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs) if reads else pipe.submit(w["ref_codes"], w["split_bounds"], obs, pair if linkage else None)
        d = pipe.collect(t)
        assert (d["cov16"] == np.minimum(cov, 65535)).all() and d["clon"].tobytes() == exp["clon"].tobytes()
        pipe.release(t)
    pipe.close()


def test_deep_batch_after_shallow_on_one_slot(ctx):
    """the hand-back follows the batch: shallow, deep, shallow through the same slot"""
    from instrain_amd import engine, synth
    ws = [synth.make_workload(genome_len=200_000, coverage=c, n_sites=200, seed=51 + i) for i, c in enumerate((2, 30, 3))]
    pipe = engine.Pipe(ctx, max_pos=200_000, max_obs=max(w["n_obs"] for w in ws), max_splits=64, depth=1, host_threads=2, pin_threads=False,
                       n_mm_bins=1, enable_linkage=False)
    for w, shallow in zip(ws, (True, False, True)):
        a = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False)
        a.run()
        exp = a.fetch()
        a.close()
        t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], None)
        raw = pipe.collect(t, densify=False)
        assert "clon_sparse" in raw and ("cov8" in raw) == shallow and ("cov16" in raw) != shallow
        pipe.release(t)
        t = pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], None)
        d = pipe.collect(t)
        assert (d["cov16"] == exp["counts"].sum(axis=1)).all() and d["clon"].tobytes() == exp["clon"].tobytes()
        assert d["snv"].tobytes() == exp["snv"].tobytes()
        pipe.release(t)
    pipe.close()


def test_context_with_a_cu_reserve_gives_the_same_tables(ctx):
    """isx_ctx_reserve_cus: pass queues masked off 4 CUs of every XCD, side queues onto them, the persistent grid sized for 224 CUs --
    a stream of read-level batches with linkage (lean slots, staged wires, large tables home by DMA) comes back byte for byte as from a
    context that keeps the whole device; the call is refused once the context has made a batch"""
    from instrain_amd import engine, synth
    from instrain_amd._lib import IsxError, check
    ws = [synth.make_workload(genome_len=600_000, coverage=c, n_sites=800, seed=s, skip_mm=True) for c, s in ((4, 91), (12, 92), (40, 93))]
    segs = [synth.segs_from_obs(w["obs"], w["pair"]) for w in ws]
    cap = dict(max_pos=max(w["n_pos"] for w in ws), max_obs=0, max_segs=max(s.n_seg for s in segs),
               max_splits=max(len(w["split_bounds"]) for w in ws), depth=4, host_threads=4, pin_threads=False,
               n_mm_bins=1, enable_linkage=True, min_snp=20, lean_output=True)
    lut, fb = util.load_lut()
    out = []
    for reserve in (None, 4):
        c = ctx if reserve is None else engine.Context(0, reserve_cus=reserve)
        if reserve is not None:
            c.set_null_model(lut, fb)
        pipe = engine.Pipe(c, **cap)
        wires = [pipe.stage_reads(w["ref_codes"], w["split_bounds"], sg) for w, sg in zip(ws, segs)]
        tickets = [pipe.submit_wire(x) for x in wires] + [pipe.submit_wire(wires[0])]
        res = []
        for t in tickets:
            r = pipe.collect(t)
            res.append({k: r[k].copy() for k in ("cov16", "clon", "snv", "ld", "rare")} | {"sizes": r["sizes"]})
            pipe.release(t)
        if reserve is not None:
            with pytest.raises(IsxError, match="before the context's first"):
                check(c.lib.isx_ctx_reserve_cus(c.h, 2))
        for x in wires:
            x.close()
        pipe.close()
        if reserve is not None:
            c.close()
        out.append(res)
    for a, b in zip(*out):
        assert a["sizes"] == b["sizes"]
        for k in ("cov16", "snv", "ld", "rare"):
            assert a[k].tobytes() == b[k].tobytes(), k
        assert a["clon"].view(np.uint32).tobytes() == b["clon"].view(np.uint32).tobytes()
    cov = np.bincount(ws[0]["obs"]["gpos"], minlength=ws[0]["n_pos"])
    assert (out[1][0]["cov16"] == cov).all() and out[1][3]["snv"].tobytes() == out[1][0]["snv"].tobytes()
