"""GPU parity for the coverage-overlap step of `inStrain compare`: isx_compare_coverage
vs golden vectors produced by the reference's own calc_mm2overlap (readComparer.py:145-191)."""
import os

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


def _batch(ctx, seq_codes, bounds, pos, base, mm, pair, n_mm=None):
    from instrain_amd import engine
    n_mm = int(mm.max()) + 1 if n_mm is None else n_mm
    b = engine.Batch(ctx, seq_codes, bounds, engine.pack_obs(pos.astype(np.uint32), base, mm), pair.astype(np.uint32),
                     n_mm_bins=n_mm, enable_linkage=False)
    b.run()
    return b


@pytest.mark.parametrize("name", ["compare_a", "compare_b"])
def test_overlap_vs_reference_vectors(ctx, name):
    from instrain_amd import compare, engine
    g = util.load_case(name)
    codes = engine.encode_seq(str(g["seq"]))
    b1 = _batch(ctx, codes, [0, len(codes)], g["a_pos"], g["a_base"], g["a_mm"], g["a_pair"])
    b2 = _batch(ctx, codes, [0, len(codes)], g["b_pos"], g["b_base"], g["b_mm"], g["b_pair"])
    mm2overlap, mm2coverage, ms = compare.calc_mm2overlap(b1, b2, [0, len(codes)], min_cov=5)
    b1.close(); b2.close()
    assert sorted(mm2overlap[0]) == list(g["mm"])
    assert [mm2overlap[0][m] for m in g["mm"]] == list(g["both"])
    assert np.max(np.abs(np.array([mm2coverage[0][m] for m in g["mm"]]) - g["coverage"])) < 1e-15
    assert ms > 0


def test_overlap_dense_vs_mm_and_multi_scaffold(ctx):
    """a dense (skip-mm) sample against an mm sample over three scaffolds, vs plain numpy"""
    from instrain_amd import compare, engine
    from tests.test_gpu_parity import _random_split
    seq, p1, b1_, m1, r1 = _random_split(501, 7000, 14, 1, 50)
    _, p2, b2_, m2, r2 = _random_split(502, 7000, 9, 4, 50)
    codes = engine.encode_seq(seq)
    sb = np.array([0, 1000, 1001, 7000])
    A = _batch(ctx, codes, sb, p1, b1_, m1, r1, n_mm=1)
    B = _batch(ctx, codes, sb, p2, b2_, m2, r2, n_mm=4)
    mm2overlap, mm2coverage, _ = compare.calc_mm2overlap(A, B, sb, min_cov=5)
    A.close(); B.close()
    cov1 = np.bincount(p1[b1_ < 4], minlength=7000)
    for i, (s, e) in enumerate(zip(sb[:-1], sb[1:])):
        c2 = np.zeros(7000, dtype=np.int64)
        keysA = {0} if cov1[s:e].any() else set()
        for m in range(4):
            k = (m2 == m) & (b2_ < 4)
            lvl = np.bincount(p2[k], minlength=7000)
            c2 += lvl
            present = bool(lvl[s:e].any()) or (m in keysA)
            assert (m in mm2overlap[i]) == present, (i, m)
            if not present:
                continue
            t1, t2 = cov1[s:e] >= 5, c2[s:e] >= 5
            assert mm2overlap[i][m] == int((t1 & t2).sum())
            either = int((t1 | t2).sum())
            assert abs(mm2coverage[i][m] - ((t1 & t2).sum() / either if either else 0)) < 1e-15
