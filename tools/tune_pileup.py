#!/usr/bin/env python3
"""Sweep (window, grid) of k_pileup_dense on the C2 workload + the epilogue-off ablation (ISX_DEBUG_MODE=2);
prints avg kernel ms (HIP events).  Tuning aid only -- run on the GPU box: python tools/tune_pileup.py"""
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
combos = [(0, 0, 0, {})] + [(W, 0, 0, {}) for W in (2048, 2560, 3264, 3776)] + [(0, 256, 0, {}), (0, 768, 0, {}), (0, 0, 2, {}),
          (0, 0, 0, {"ISX_NO_SHORT_RECORDS": "1"}), (0, 0, 2, {"ISX_NO_SHORT_RECORDS": "1"}), (0, 0, 0, {"ISX_WIDE_RECORDS": "1"})]
for W, G, dbg, env in combos:
    for k in ("ISX_GRID", "ISX_NO_SHORT_RECORDS", "ISX_WIDE_RECORDS"):
        os.environ.pop(k, None)
    if G:
        os.environ["ISX_GRID"] = str(G)
    os.environ.update(env)
    os.environ["ISX_DEBUG_MODE"] = str(dbg)
    try:
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False, window=W)
    except Exception as e:
        print(W, G, "ERR", e)
        continue
    for _ in range(3):
        b.run()
    ts = []
    for _ in range(30):
        b.run()
        ts.append(b.timings()["pileup_ms"])
    t = b.timings()
    ts = np.array(ts)
    abytes = bench.pileup_algorithmic_bytes(w["n_obs"], w["n_pos"], 0, True, t["record_bytes"])
    print("W=%5d grid=%5d dbg=%d rec=%dB lds=%6d  avg %.4f ms  min %.4f  -> %.0f GB/s (min: %.0f)" % (
        t["pileup_window"], t["pileup_blocks"], dbg, t["record_bytes"], t["pileup_lds_bytes"], ts.mean(), ts.min(),
        abytes / ts.mean() / 1e6, abytes / ts.min() / 1e6), flush=True)
    b.close()
os.environ["ISX_DEBUG_MODE"] = "0"
