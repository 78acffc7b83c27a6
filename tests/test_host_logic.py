"""CPU-only host logic: split geometry, synthetic generator determinism, LPT sharding, packing."""
import os

import numpy as np

from instrain_amd import dist, engine, synth
from tests import util


def test_iterate_splits_matches_reference_sweep():
    rows = np.load(os.path.join(util.GOLD, "iterate_splits.npy"))     # from the reference's fasta.iterate_splits
    for L in np.unique(rows[:, 0]):
        for W in (1000, 10000):
            exp = rows[(rows[:, 0] == L) & (rows[:, 1] == W)][:, 3:5]
            assert (np.array(synth.iterate_splits(int(L), W)) == exp).all(), (L, W)


def test_split_bounds_flat():
    b = synth.split_bounds_for([29879, 126], 10000)
    assert list(b) == [0, 9959, 19918, 29879, 30005]


def test_synth_is_deterministic_and_clustered():
    a = synth.make_workload(genome_len=50_000, coverage=10, n_sites=50, seed=7, skip_mm=False)
    b = synth.make_workload(genome_len=50_000, coverage=10, n_sites=50, seed=7, skip_mm=False)
    assert a["obs"].tobytes() == b["obs"].tobytes() and a["pair"].tobytes() == b["pair"].tobytes()
    g = a["obs"]["gpos"].astype(np.int64)
    assert (g < 50_000).all() and a["n_mm_bins"] == int(a["obs"]["mm"].max()) + 1
    # BAM order: per-1024-record chunk minima are non-decreasing up to a read length
    m = g[: len(g) // 1024 * 1024].reshape(-1, 1024).min(axis=1)
    assert (np.diff(m) > -400).all()


def test_pack_obs_layout():
    o = engine.pack_obs(np.array([5, 7]), np.array([1, 4]), np.array([0, 300]))
    raw = np.frombuffer(o.tobytes(), dtype="<u4").reshape(2, 2)
    assert list(raw[:, 0]) == [5, 7]
    assert list(raw[:, 1]) == [0 | (1 << 16), 300 | (4 << 16)]       # mm | base << 16  (what the kernels decode)


def test_lpt_shards():
    costs = [10, 1, 1, 9, 2, 8, 3, 7]
    sh = dist.lpt_shards(costs, 3)
    assert sorted(i for s in sh for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in sh]
    assert max(loads) - min(loads) <= 3


def test_dense_cov_rebuilds_coverage_from_every_shrunk_form():
    """engine.dense_cov (host, numpy only): 16-bit and 8-bit planes with their saturated lists, and the lean slots' 4-bit plane + 16-bit
    rows of the windows beyond 15 (isx_pipe_result.coverage4), odd lengths and a last partial window included"""
    import numpy as np
    from instrain_amd import engine
    from instrain_amd._lib import SAT_DT
    rng = np.random.Generator(np.random.PCG64(3))
    n, W = 10_001, 256
    cov = rng.poisson(4, n).astype(np.int64)
    cov[3000:3300] += 40                      # windows 11, 12 go beyond 15
    cov[9990:] += 70                          # ... and the last, partial window
    cov[5000] = 70_000                        # beyond 16 bits: a saturated-list entry
    sat = np.zeros(1, dtype=SAT_DT)
    sat["gpos"], sat["coverage"] = 5000, 70_000
    exp = np.minimum(cov, 65535).astype(np.uint16)
    # 16-bit plane
    assert (engine.dense_cov({"cov16": exp}) == exp).all()
    # 8-bit plane + exact values of what saturated
    big = np.flatnonzero(cov >= 255)
    s8 = np.zeros(len(big), dtype=SAT_DT)
    s8["gpos"], s8["coverage"] = big, cov[big]
    assert (engine.dense_cov({"cov8": np.minimum(cov, 255).astype(np.uint8), "saturated": s8}) == exp).all()
    # 4-bit plane + rows
    nib = np.minimum(cov, 15).astype(np.uint8)
    pad = np.concatenate([nib, np.zeros(n % 2, np.uint8)])
    cov4 = (pad[0::2] | (pad[1::2] << 4)).astype(np.uint8)
    wins = np.unique(np.flatnonzero(cov > 15) // W)[::-1].copy()              # rows in no particular order
    rows = np.zeros((len(wins), W), dtype=np.uint16)
    for k, w in enumerate(wins):
        seg = np.minimum(cov[w * W:(w + 1) * W], 65535)
        rows[k, :len(seg)] = seg
    res = {"cov4": cov4, "cov_rows": rows, "cov_row_win": wins.astype(np.uint32), "cov_window": W, "saturated": sat}
    got = engine.dense_cov(res, n)
    assert got.dtype == np.uint16 and len(got) == n and (got == exp).all()
    assert len(wins) >= 3 and (n - 1) // W in wins
