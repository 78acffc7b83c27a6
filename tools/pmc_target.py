#!/usr/bin/env python3
"""The launches the hardware counters are collected on (run under rocprofv3 --pmc ...): ONLY resident C2 batches, so every
dispatch of a kernel name is the same work -- k_pileup_dense<false, 64> on the read segments and <false, 2, true> on the 2-byte
observation records (one mm bin), k_pileup_mm<..., SEGS> / <...> with mm profiling on; 8 launches each.  bench.py's roofline
objects are priced on exactly these launches (its resident / mm legs).
usage: python tools/pmc_target.py [--no-mm]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import engine
from tests import util

with_mm = "--no-mm" not in sys.argv
w = bench.c2_workload(seed=2, with_mm=with_mm)
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
jobs = [(w["segs"], 1), (w["obs"], 1)]
if with_mm:
    jobs += [(w["segs_mm"], w["n_mm_bins_mm"]), (w["obs_mm"], w["n_mm_bins_mm"])]
for src, M in jobs:
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=M, enable_linkage=False)
    for _ in range(8):
        b.run()
    print(type(src).__name__, M, b.timings()["pileup_ms"], flush=True)
    b.close()
ctx.close()
