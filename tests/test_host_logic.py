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
