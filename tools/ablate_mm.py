#!/usr/bin/env python3
"""Ablation of k_pileup_mm on C2 with mm on. ISX_DEBUG_MODE bits: 2 stream only (no level pass),
64 skip the entry stores. Tuning aid.
(Tried and dropped: bounding the level loop by the window's own highest mm -- +3 % at 16 bins, -2 % at 6.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import engine
from tests import util
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
w = bench.c2_workload(2, scale=float(os.environ.get("SCALE", "1.0")), with_mm=True)
for env in ({"ISX_BLOCK": "512"}, {"ISX_BLOCK": "1024"}):
    for k in ("ISX_NO_PACKED", "ISX_BLOCK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs_mm"], None, n_mm_bins=w["n_mm_bins_mm"], enable_linkage=False)
    for rep in range(2):
        for mode, name in ((0, "full"), (64, "no entry stores"), (2, "stream only")):
            os.environ["ISX_DEBUG_MODE"] = str(mode)
            ts = []
            for i in range(12):
                b.run()
                if i >= 2:
                    ts.append(b.timings()["pileup_ms"])
            t = b.timings()
            print(env, "blocks=%d W=%d lds=%d %-20s avg %.4f ms" % (t["pileup_blocks"], t["pileup_window"], t["pileup_lds_bytes"], name, np.mean(ts)), flush=True)
    b.close()
os.environ["ISX_DEBUG_MODE"] = "0"
