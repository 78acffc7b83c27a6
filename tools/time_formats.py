#!/usr/bin/env python3
"""Resident C2 batch: kernel time of the read-level stream formats (32-byte reference-delta records vs 64-byte segment records),
linkage off / on.  usage: python tools/time_formats.py [scale]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from instrain_amd import engine
from tests import util

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
w = bench.c2_workload(seed=2, scale=scale)
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
for link in (False, True):
    for layout, name in ((0, "delta32"), (8, "seg64")):
        segs = w["segs"]
        if link:
            segs = engine.SegBatch(segs.gpos, segs.len, segs.bases, segs.mm, w["segs"].pair)
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], segs, None, n_mm_bins=1, enable_linkage=link, layout=layout)
        for _ in range(4):
            b.run()
        ks = []
        for _ in range(10):
            b.run()
            ks.append(b.pileup_ms())
        t = b.timings()
        print("%-8s linkage=%d  kernel %.4f ms (min %.4f)  W=%d lds=%d blocks=%d  n_snv=%d" % (name, link, np.mean(ks), np.min(ks), t["pileup_window"], t["pileup_lds_bytes"], t["pileup_blocks"], b.sizes()["n_snv"]), flush=True)
        b.close()
ctx.close()
