#!/usr/bin/env python3
"""Experiment: LDS bank conflicts in the stream loop.  With 2-byte records a lane holds 8 CONSECUTIVE records
(= 8 consecutive positions of one read), so the 64 lanes of a wave hit LDS words 8 apart: 4 banks.  Feeding the
records of every 512-group transposed (lane i gets records i, 64+i, ...) makes a wave-wide step touch 64
consecutive positions.  Linkage off: the tables do not depend on the record order."""
import os
import sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from instrain_amd import engine  # noqa: E402
from tests import util  # noqa: E402

ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
w = bench.c2_workload(seed=2, with_mm=True)


def run(obs, M, tag, per_lane, group):
    res = {}
    for name in ("arrival order", "transposed groups"):
        o = obs
        if name != "arrival order":
            n = len(o) // group * group
            o = o.copy()
            o[:n] = o[:n].reshape(-1, per_lane, 64).transpose(0, 2, 1).reshape(-1)
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], o, None, n_mm_bins=M, enable_linkage=False)
        for _ in range(3):
            b.run()
        ks = []
        for _ in range(10):
            b.run()
            ks.append(b.pileup_ms())
        f = b.fetch()
        res[name] = f
        print("%s | %s: kernel %.4f ms (min %.4f)" % (tag, name, float(np.mean(ks)), float(np.min(ks))), flush=True)
        b.close()
    a, c = res["arrival order"], res["transposed groups"]
    for k in a:
        if k in ("counts", "snv"):
            assert (a[k] == c[k]).all() if not a[k].dtype.names else all((a[k][f] == c[k][f]).all() for f in a[k].dtype.names), k
    if "entries" in a:
        assert all((a["entries"][f] == c["entries"][f]).all() for f in ("gpos", "mm", "cnt"))


run(w["obs"], 1, "dense, 2-byte records", 8, 512)
run(w["obs_mm"], w["n_mm_bins_mm"], "mm on, 4-byte records", 4, 256)
ctx.close()
