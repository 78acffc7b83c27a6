#!/usr/bin/env python3
"""Sweep (window, block) of k_pileup_call on the C2 workload; prints avg kernel ms (HIP events).
Tuning aid only -- run on the GPU box: python tools/tune_pileup.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import engine
from tests import util

ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
scale = float(os.environ.get("SCALE", "1.0"))
w = bench.c2_workload(2, scale=scale)
abytes = bench.pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, True)
combos = [(0, 1024)] + [(W, 1024) for W in (2560, 3008, 3264, 3328, 3904)]
for W, B in combos:
    os.environ["ISX_BLOCK"] = str(B)
    try:
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False, window=W)
    except Exception as e:
        print(W, B, "ERR", e)
        continue
    for _ in range(3):
        b.run()
    ts = []
    for _ in range(20):
        b.run()
        ts.append(b.timings()["pileup_ms"])
    t = b.timings()
    ts = np.array(ts)
    print("W=%5d block=%4d blocks=%5d lds=%6d  avg %.4f ms  min %.4f  -> %.0f GB/s (min: %.0f)" % (
        t["pileup_window"], B, t["pileup_blocks"], t["pileup_lds_bytes"], ts.mean(), ts.min(),
        abytes / ts.mean() / 1e6, abytes / ts.min() / 1e6), flush=True)
    b.close()
