"""CPU: isx_levels_expand (csrc/lev_expand.cpp) -- the level-sparse hand-back of mm profiling (isx_pipe_result.lev_*) expanded to the four
columns shrink_basewise's inputs are cut from (profile_utilities.py:337-350), against a plain Python restatement of the format's
definition (include/instrain_amd.h): windows in any order, saturated elements, the lists, every mask width, refusals."""
import ctypes as C

import numpy as np
import pytest

from instrain_amd import _lib


def make_tables(rng, n_pos, W, M, cov_bytes, min_cov, p_level=0.4, deep=0.01):
    """random per-(position, level) coverages -> (the tables as the device would write them, the expected columns)"""
    present = rng.random((n_pos, M)) < p_level
    cov = rng.integers(0, 6, (n_pos, M)).astype(np.int64)            # a present level may have coverage 0 (a non-ACGT base alone)
    big = rng.random((n_pos, M)) < deep
    cov[big] = rng.integers(250, 300 if cov_bytes == 1 else 70000, int(big.sum()))
    cov[~present] = 0
    mask = (present * (1 << np.arange(M, dtype=np.int64))).sum(axis=1)
    mdt = np.uint8 if M <= 8 else np.uint16 if M <= 16 else np.uint32
    n_win = (n_pos + W - 1) // W
    per_win = [int(present[w * W:(w + 1) * W].sum()) for w in range(n_win)]
    order = rng.permutation(n_win)                                    # the windows get to the cursor in any order
    win_off = np.zeros(n_win, np.uint32)
    at = 0
    for w in order:
        win_off[w] = at
        at += per_win[w]
    n_lev = at
    sat_thr = 255 if cov_bytes == 1 else 65535
    stream = np.zeros(n_lev, np.uint8 if cov_bytes == 1 else np.uint16)
    sat, exp = [], []
    dev_of = {}
    for w in range(n_win):
        d = int(win_off[w])
        for p in range(w * W, min(n_pos, (w + 1) * W)):
            cum = 0
            for m in range(M):
                if not present[p, m]:
                    continue
                c = int(cov[p, m])
                stream[d] = min(c, sat_thr)
                if c >= sat_thr:
                    sat.append((d, c))
                cum += c
                exp.append([p, m, c, 1.0 if cum >= min_cov else np.nan, np.nan, d])
                dev_of[(p, m)] = d
                d += 1
    exp.sort(key=lambda e: (e[0], e[1]))
    clon_list, rare_list = [], []
    for e in exp:
        if not np.isnan(e[3]) and rng.random() < 0.1:
            e[3] = float(np.float32(rng.random()))
            clon_list.append((e[5], e[3]))
        if rng.random() < 0.2:
            e[4] = float(np.float32(rng.random()))
            rare_list.append((e[5], e[4]))
    rng.shuffle(clon_list), rng.shuffle(rare_list), rng.shuffle(sat)
    t = {"mask": mask.astype(mdt), "cov": stream, "win_off": win_off,
         "clon": np.array(clon_list, dtype=_lib.RARE_DT) if clon_list else np.empty(0, _lib.RARE_DT),
         "rare": np.array(rare_list, dtype=_lib.RARE_DT) if rare_list else np.empty(0, _lib.RARE_DT),
         "sat": np.array(sat, dtype=_lib.SAT_DT) if sat else np.empty(0, _lib.SAT_DT)}
    cols = (np.array([e[0] for e in exp], np.uint32), np.array([(e[1] << 24) | e[2] for e in exp], np.uint32),
            np.array([e[3] for e in exp], np.float32), np.array([e[4] for e in exp], np.float32))
    return t, cols


def as_result(t, n_pos, W, min_cov):
    r = _lib.PipeResult()
    r.n_pos = n_pos
    for k, f in (("mask", "lev_mask"), ("cov", "lev_cov"), ("win_off", "lev_win_off"), ("clon", "lev_clon"), ("rare", "lev_rare"), ("sat", "lev_sat")):
        setattr(r, f, t[k].ctypes.data if len(t[k]) else None)
    if not len(t["cov"]):
        r.lev_cov = t["win_off"].ctypes.data        # (never read)
    r.n_lev, r.n_lev_clon, r.n_lev_rare, r.n_lev_sat = len(t["cov"]), len(t["clon"]), len(t["rare"]), len(t["sat"])
    r.lev_mask_bytes, r.lev_cov_bytes = t["mask"].dtype.itemsize, t["cov"].dtype.itemsize
    r.lev_window, r.n_lev_windows, r.lev_min_cov = W, (n_pos + W - 1) // W, min_cov
    return r


def expand(r, threads):
    lib = _lib.load()
    n = max(1, int(r.n_lev))
    cols = (np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.float32), np.empty(n, np.float32))
    rc = lib.isx_levels_expand(C.byref(r), threads, *(c.ctypes.data for c in cols))
    return rc, tuple(c[:int(r.n_lev)] for c in cols)


@pytest.mark.parametrize("M,cov_bytes,W,n_pos,threads", [(6, 1, 64, 1000, 1), (8, 1, 128, 5000, 4), (13, 2, 192, 3000, 3), (26, 1, 64, 2049, 16), (32, 2, 1024, 4096, 2), (2, 1, 64, 63, 8)])
def test_expand_equals_the_definition(M, cov_bytes, W, n_pos, threads):
    rng = np.random.Generator(np.random.PCG64(M * 1000 + n_pos))
    t, want = make_tables(rng, n_pos, W, M, cov_bytes, min_cov=5)
    rc, got = expand(as_result(t, n_pos, W, 5), threads)
    assert rc == 0
    assert len(want[0]) > 0 and (cov_bytes == 2 or n_pos < 100 or len(t["sat"]) > 0)
    for k in range(4):
        assert got[k].tobytes() == want[k].tobytes(), k


def test_empty_and_refusals():
    rng = np.random.Generator(np.random.PCG64(1))
    t, want = make_tables(rng, 500, 64, 5, 1, 5, p_level=0.0)
    rc, got = expand(as_result(t, 500, 64, 5), 2)
    assert rc == 0 and len(got[0]) == 0
    t, want = make_tables(rng, 500, 64, 5, 1, 5)
    r = as_result(t, 500, 64, 5)
    r.n_lev -= 1                                    # masks and n_lev disagree
    assert expand(r, 2)[0] == -6
    r = as_result(t, 500, 64, 5)
    bad = t["clon"].copy()
    bad["gpos"][0] = 10 ** 7                        # a list entry outside every window's range
    r.lev_clon = bad.ctypes.data
    assert expand(r, 2)[0] == -6
    r = as_result(t, 500, 64, 5)
    r.lev_mask = None
    assert expand(r, 2)[0] == -6
    r = as_result(t, 500, 64, 5)
    r.n_lev_windows += 1
    assert expand(r, 2)[0] == -6
