"""CPU-only, world_size 2 over gloo: the multi-GPU layer (sharding + the single final gather)."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist
    from instrain_amd._lib import SNV_DT, LD_DT
    rank, local, world = idist.init_from_env(backend="gloo")
    assert world == 2
    # each rank owns the scaffolds LPT gives it and "profiles" them (fake rows tagged by scaffold id)
    costs = [5, 1, 4, 2, 3]
    mine = idist.lpt_shards(costs, world)[rank]
    snv = np.zeros(sum(costs[i] for i in mine), dtype=SNV_DT)
    k = 0
    for i in mine:
        snv["gpos"][k:k + costs[i]] = 1000 * i + np.arange(costs[i])
        k += costs[i]
    ld = np.zeros(rank, dtype=LD_DT)          # ragged: rank 0 contributes an EMPTY table
    ld["gpos_a"] = 7
    out = idist.gather_tables({"snv": snv, "ld": ld}, dst=0)
    if rank == 0:
        got = np.sort(out["snv"]["gpos"])
        exp = np.sort(np.concatenate([1000 * i + np.arange(c) for i, c in enumerate(costs)]))
        assert (got == exp).all(), (got, exp)
        assert len(out["ld"]) == 1 and out["ld"]["gpos_a"][0] == 7
        print("GATHER_OK")
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_gather_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout


SHARD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist, engine
    rank, local, world = idist.init_from_env(backend="gloo")
    path = sys.argv[1]
    bam = engine.BamFile(path, threads=2)
    bam.scan(); info = bam.filter(min_read_ani=0.9)
    refs = bam.refs()
    _, pairs = bam.ref_counts()
    shards = idist.shard_scaffolds(pairs, [r[1] for r in refs], world)
    mine = shards[rank]
    # this rank's part of the observation stream (host front end only: no GPU here), keyed by (tid, position)
    obs, pair, bounds, sref = bam.expand_refs(sorted(mine), min_read_ani=0.9)
    offs = np.r_[0, np.cumsum([refs[t][1] for t in sorted(mine)])]
    which = np.searchsorted(offs, obs["gpos"], side="right") - 1
    dt = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("base", "u1"), ("mm", "<u2")])
    t = np.zeros(len(obs), dtype=dt)
    t["tid"] = np.array(sorted(mine), dtype=np.int32)[which] if len(obs) else 0
    t["pos"] = obs["gpos"] - offs[which]; t["base"] = obs["base"]; t["mm"] = obs["mm"]
    med = np.array([info["median_insert"]])
    out = idist.gather_tables({"obs": t, "median": med.view([("m", "<f8")])}, dst=0)
    if rank == 0:
        whole = engine.BamFile(path, threads=2)
        o, p, b, s = whole.expand(min_read_ani=0.9)
        woffs = np.r_[0, np.cumsum([r[1] for r in refs])]
        ww = np.searchsorted(woffs, o["gpos"], side="right") - 1
        exp = np.zeros(len(o), dtype=dt)
        exp["tid"] = ww; exp["pos"] = o["gpos"] - woffs[ww]; exp["base"] = o["base"]; exp["mm"] = o["mm"]
        got = np.sort(out["obs"], order=["tid", "pos", "base", "mm"])
        exp = np.sort(exp, order=["tid", "pos", "base", "mm"])
        assert len(got) == len(exp) and (got == exp).all()
        assert (out["median"]["m"] == whole.info["median_insert"]).all() and len(out["median"]) == world
        assert sorted(x for sh in shards for x in sh) == [i for i in range(len(refs)) if refs[i][1] > 0]
        print("SHARD_OK", [len(sh) for sh in shards], len(got))
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_one_bam_sharded_over_two_ranks(tmp_path):
    """both ranks scan the same BAM (same whole-file median insert without a collective), LPT gives each its
    scaffolds, each expands only those; the gathered union equals the single-process expansion"""
    sys.path.insert(0, REPO)
    from tests import bamwriter
    refs = [("s%d" % i, ln) for i, ln in enumerate([4000, 900, 12500, 700, 2600, 5100])]
    path = str(tmp_path / "shard.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(41, refs[:5], 5000))
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29534", str(script), path],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "SHARD_OK" in r.stdout


SCAN_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist, engine
    rank, local, world = idist.init_from_env(backend="gloo")
    path = sys.argv[1]
    bam = engine.BamFile(path, threads=2)
    refs = bam.refs()
    bam.scan(part=(rank, world))                       # this rank's share of the file only
    ins = idist.all_gather_concat(bam.insert_sizes())  # the exchange: the filter's median insert is a whole-file property
    median = float(np.median(ins))
    info = bam.filter(median_insert=median, min_read_ani=0.9)
    reads, pairs = bam.ref_counts()
    mine = [int(t) for t in np.flatnonzero(reads)]
    owned = idist.all_gather_concat((reads > 0).astype(np.uint8)).reshape(world, -1)
    obs, pair, bounds, sref = bam.expand_refs(mine, min_read_ani=0.9) if mine else (np.zeros(0, engine.OBS_DT), None, None, None)
    offs = np.r_[0, np.cumsum([refs[t][1] for t in mine])]
    which = np.searchsorted(offs, obs["gpos"], side="right") - 1
    dt = np.dtype([("tid", "<i4"), ("pos", "<i4"), ("base", "u1"), ("mm", "<u2")])
    t = np.zeros(len(obs), dtype=dt)
    if len(obs):
        t["tid"] = np.array(mine, dtype=np.int32)[which]
        t["pos"] = obs["gpos"] - offs[which]; t["base"] = obs["base"]; t["mm"] = obs["mm"]
    out = idist.gather_tables({"obs": t}, dst=0)
    if rank == 0:
        whole = engine.BamFile(path, threads=2)
        o, p, b, s = whole.expand(min_read_ani=0.9)
        assert whole.info["median_insert"] == median and len(ins) > 1000
        w_reads, _ = whole.ref_counts()
        assert ((owned.sum(axis=0) == 1) == (w_reads > 0)).all() and owned.sum(axis=0).max() == 1
        assert owned.sum(axis=1).min() >= 1            # every rank got something to do
        woffs = np.r_[0, np.cumsum([r[1] for r in refs])]
        ww = np.searchsorted(woffs, o["gpos"], side="right") - 1
        exp = np.zeros(len(o), dtype=dt)
        exp["tid"] = ww; exp["pos"] = o["gpos"] - woffs[ww]; exp["base"] = o["base"]; exp["mm"] = o["mm"]
        got = np.sort(out["obs"], order=["tid", "pos", "base", "mm"])
        exp = np.sort(exp, order=["tid", "pos", "base", "mm"])
        assert len(got) == len(exp) and (got == exp).all()
        print("SCAN_OK", owned.sum(axis=1).tolist(), len(got))
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_sharded_scan_two_ranks(tmp_path):
    """each rank scans HALF of the BAM (isx_bam_scan_part), the insert sizes are all-gathered for the whole-file median,
    each rank filters and expands the scaffolds it owns; the union equals the single-process result"""
    sys.path.insert(0, REPO)
    from tests import bamwriter
    refs = [("s%d" % i, 6000 + 500 * i) for i in range(12)]
    path = str(tmp_path / "scan.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(43, refs[:11], 9000))
    script = tmp_path / "scan_worker.py"
    script.write_text(SCAN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29535", str(script), path],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "SCAN_OK" in r.stdout


def _run_workers(tmp_path, script_text, nproc, port, *args, env_extra=None):
    script = tmp_path / ("worker_%d.py" % port)
    script.write_text(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + list(args),
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


SCAN4_WORKER = SCAN_WORKER.replace("assert owned.sum(axis=1).min() >= 1            # every rank got something to do",
                                   "assert (owned.sum(axis=1).min() == 0) == (os.environ.get('EXPECT_EMPTY') == '1'), owned.sum(axis=1)")


def test_sharded_scan_four_ranks_uneven_shares(tmp_path):
    """world 4, scaffolds of very different sizes: the shares own 1 .. many scaffolds each, the union is still the file"""
    sys.path.insert(0, REPO)
    from tests import bamwriter
    refs = [("s%d" % i, ln) for i, ln in enumerate([30000, 800, 900, 15000, 700, 650, 9000, 12000, 600, 5000])]
    path = str(tmp_path / "scan4.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(47, refs, 14000))
    out = _run_workers(tmp_path, SCAN4_WORKER, 4, 29541, path, env_extra={"EXPECT_EMPTY": "0"})
    assert "SCAN_OK" in out


def test_sharded_scan_four_ranks_one_owns_nothing(tmp_path):
    """world 4 over a BAM dominated by one long scaffold: a share that lies inside it owns no scaffold at all (its rank
    contributes empty tables to the gather), the others still cover the file"""
    sys.path.insert(0, REPO)
    from tests import bamwriter
    refs = [("big", 60000), ("small1", 2000), ("small2", 2500)]
    path = str(tmp_path / "scan4e.bam")
    reads = bamwriter.random_reads(48, refs[:1], 12000) + bamwriter.random_reads(49, refs[1:], 900)
    for r in reads[12000 * 2:]:
        pass
    # reads of the second call carry tids relative to refs[1:]: shift them
    n_big = sum(1 for _ in bamwriter.random_reads(48, refs[:1], 12000))
    for r in reads[n_big:]:
        r["tid"] += 1
        if "mtid" in r:
            r["mtid"] += 1
    reads.sort(key=lambda r: (r["tid"], r["pos"]))
    bamwriter.write_bam(path, refs, reads)
    out = _run_workers(tmp_path, SCAN4_WORKER, 4, 29542, path, env_extra={"EXPECT_EMPTY": "1"})
    assert "SCAN_OK" in out


FALLBACK_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist, engine
    rank, local, world = idist.init_from_env(backend="gloo")
    path = sys.argv[1]
    bam = engine.BamFile(path, threads=2)
    if rank == 1:                                       # this rank's share scan fails (what a share without a record start does)
        real = bam.scan
        def failing(part=None):
            if part is not None:
                raise engine.IsxError(-5, "no record starts in the two segments before this share")
            return real(part)
        bam.scan = failing
    sharded = idist.scan_share(bam, rank, world)
    assert sharded is False                            # EVERY rank learns of it, nobody hangs in the next collective
    bam.close()
    bam = engine.BamFile(path, threads=2)              # a handle scans one share only: a fresh one for the whole file
    bam.scan()
    info = bam.filter(min_read_ani=0.9)
    med = idist.all_gather_concat(np.asarray([info["median_insert"]]))
    assert len(med) == world and (med == med[0]).all()
    if rank == 0:
        print("FALLBACK_OK")
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_failed_share_scan_falls_back_on_every_rank(tmp_path):
    sys.path.insert(0, REPO)
    from tests import bamwriter
    refs = [("s%d" % i, 5000) for i in range(6)]
    path = str(tmp_path / "fb.bam")
    bamwriter.write_bam(path, refs, bamwriter.random_reads(50, refs, 3000))
    assert "FALLBACK_OK" in _run_workers(tmp_path, FALLBACK_WORKER, 2, 29543, path)


CROSS_WORKER = textwrap.dedent("""
    import os, sys, zlib
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from instrain_amd import dist as idist, engine
    rank, local, world = idist.init_from_env(backend="gloo")
    path, mode, expect = sys.argv[1], sys.argv[2], sys.argv[3]
    prio = ["p7", "p13", "p21", "p40", "p55"] if len(sys.argv) > 4 else []
    fkw = dict(pairing_filter=mode, min_read_ani=0.9)
    bam = engine.BamFile(path, threads=2)
    refs = bam.refs()
    bam.scan(part=(rank, world))
    n_cross = idist.resolve_cross_names(bam, rank, world)
    if prio:
        bam.set_priority_reads(prio)
    ok, why = 1, ""
    try:
        bam.filter(median_insert=0.0, **fkw)
    except engine.IsxError as e:
        ok, why = 0, str(e)
    oks = idist.all_gather_concat(np.asarray([ok], dtype=np.int32))
    if expect == "keyerror":
        assert oks.min() == 0 and (ok == 1 or "three scaffolds" in why)
        if rank == 0:
            print("CROSS_OK keyerror", oks.tolist())
        dist.barrier(); dist.destroy_process_group(); sys.exit(0)
    assert oks.min() == 1, why
    ins = idist.all_gather_concat(bam.filter_insert_sizes())
    median = float(np.median(ins))
    info = bam.filter(median_insert=median, **fkw)
    reads, pairs = bam.ref_counts()
    dt = np.dtype([("tid", "<i4"), ("name", "<u4"), ("len", "<i4"), ("mm", "<i4")])
    rows = []
    for t in np.flatnonzero(reads):
        for name, mm in bam.r2m(int(t)).items():
            rows.append((int(t), zlib.crc32(name.encode()), len(name), int(mm)))
    mine = np.array(rows, dtype=dt) if rows else np.zeros(0, dtype=dt)
    tal = np.asarray([info[k] for k in ("unfiltered_reads", "unfiltered_pairs", "unfiltered_singletons", "filtered_pairs", "filtered_singletons", "filtered_bases")], dtype=np.int64)
    tals = idist.all_gather_concat(tal).reshape(world, -1)
    crosses = idist.all_gather_concat(np.asarray([n_cross], dtype=np.int64))
    out = idist.gather_tables({"r2m": mine}, dst=0)
    if rank == 0:
        whole = engine.BamFile(path, threads=2)
        whole.scan()
        if prio:
            whole.set_priority_reads(prio)
        winfo = whole.filter(**fkw)
        assert winfo["median_insert"] == median, (winfo["median_insert"], median)
        exp = []
        for t in range(len(refs)):
            for name, mm in whole.r2m(t).items():
                exp.append((t, zlib.crc32(name.encode()), len(name), int(mm)))
        exp = np.sort(np.array(exp, dtype=dt), order=["tid", "name", "len"])
        got = np.sort(out["r2m"], order=["tid", "name", "len"])
        assert len(got) == len(exp) > 1000 and (got == exp).all(), (len(got), len(exp))
        for i, k in enumerate(("unfiltered_reads", "unfiltered_pairs", "unfiltered_singletons", "filtered_pairs", "filtered_singletons", "filtered_bases")):
            assert tals[:, i].sum() == winfo[k], (k, tals[:, i].tolist(), winfo[k])
        assert crosses.sum() > 100 and (crosses > 0).sum() >= 2      # the names really straddle shares
        print("CROSS_OK", mode, crosses.tolist(), len(got))
    dist.barrier()
    dist.destroy_process_group()
""") % REPO


def test_non_discordant_and_all_reads_over_share_scans(tmp_path):
    """the cross-scaffold name look-ups of non_discordant / all_reads with every rank scanning only its share: name hashes
    all-gathered, repeated names resolved, isx_bam_set_cross_names -- R2M, tallies and median equal the whole-file filter's"""
    sys.path.insert(0, REPO)
    from tests import bamwriter
    p2 = str(tmp_path / "cross2.bam")
    bamwriter.cross_scaffold_bam(p2, 71, triple=False)
    out = _run_workers(tmp_path, CROSS_WORKER, 3, 29551, p2, "non_discordant", "ok")
    assert "CROSS_OK non_discordant" in out
    out = _run_workers(tmp_path, CROSS_WORKER, 3, 29552, p2, "non_discordant", "ok", "prio")
    assert "CROSS_OK non_discordant" in out
    p3 = str(tmp_path / "cross3.bam")
    bamwriter.cross_scaffold_bam(p3, 73, triple=True)
    out = _run_workers(tmp_path, CROSS_WORKER, 4, 29553, p3, "all_reads", "ok")
    assert "CROSS_OK all_reads" in out
    out = _run_workers(tmp_path, CROSS_WORKER, 2, 29554, p3, "non_discordant", "keyerror")
    assert "CROSS_OK keyerror" in out
