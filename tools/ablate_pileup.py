#!/usr/bin/env python3
"""Ablation of k_pileup_call phases on the C2 workload (ISX_DEBUG_MODE bits: 1 no LDS atomics,
2 no epilogue, 4 no streaming). Tuning aid only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import engine
from tests import util
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
w = bench.c2_workload(2, scale=float(os.environ.get("SCALE", "1.0")))
for W, B in ((2048, 1024),):
    os.environ["ISX_BLOCK"] = str(B)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False, window=W)
    for mode, name in ((0, "full"), (1, "no-atomics"), (2, "no-epilogue"), (3, "stream only"), (4, "no-stream"), (6, "zero only"), (8, "no stores"), (16, "stores only"), (32, "no mask store"), (24, "epi loop only")):
        os.environ["ISX_DEBUG_MODE"] = str(mode)
        ts = []
        for i in range(15):
            b.run()
            if i >= 3:
                ts.append(b.timings()["pileup_ms"])
        print("W=%d block=%d %-12s avg %.4f ms min %.4f" % (W, B, name, np.mean(ts), np.min(ts)), flush=True)
    b.close()
os.environ["ISX_DEBUG_MODE"] = "0"
