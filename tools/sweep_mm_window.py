#!/usr/bin/env python3
"""k_pileup_mm over the resident C2 batch with mm profiling on: kernel time by record layout and window size (one-shot batches)."""
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
w = bench.c2_workload(2, with_mm=True)
M = w["n_mm_bins_mm"]
for layout in (0, 32):
    for window in [int(x) for x in os.environ.get("WINDOWS", "0,256,384,448,512,640,704,1024,1408,1728,2048").split(",")]:
        try:
            b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs_mm"], None, n_mm_bins=M, enable_linkage=False, layout=layout, window=window)
            for _ in range(3):
                b.run()
            ks = []
            for _ in range(8):
                b.run()
                ks.append(b.pileup_ms())
            t = b.timings()
            b.close()
            print("layout %d window %4d -> W=%d block=%d grid=%d lds=%d: %.4f ms (min %.4f)" % (layout, window, t["pileup_window"], t["pileup_threads"], t["pileup_blocks"], t["pileup_lds_bytes"], np.mean(ks), np.min(ks)), flush=True)
        except Exception as e:
            print("layout %d window %d: %r" % (layout, window, e), flush=True)
