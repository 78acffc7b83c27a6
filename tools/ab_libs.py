#!/usr/bin/env python3
"""Same-box A/B of two in-tree builds of the library: interleaved runs of the C2 dense and mm kernels.
usage: python tools/ab_libs.py libA.so libB.so   (each timing runs in its own python process)"""
import os, subprocess, sys
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, ".")
import bench
from instrain_amd import engine
from tests import util
ctx = engine.Context(0)
lut, fb = util.load_lut(); ctx.set_null_model(lut, fb)
w = bench.c2_workload(2, scale=1.0, with_mm=True)
out = []
for name, obs, M in (("dense", w["obs"], 1), ("mm", w["obs_mm"], w["n_mm_bins_mm"]), ("mm16", w["obs_mm"], 16)):
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], obs, None, n_mm_bins=M, enable_linkage=False)
    ts = []
    for i in range(40):
        b.run()
        if i >= 5: ts.append(b.timings()["pileup_ms"])
    out.append("%s %.4f ms (min %.4f) W=%d" % (name, np.mean(ts), np.min(ts), b.timings()["pileup_window"]))
    b.close()
print(os.environ["ISX_LIB"], " | ".join(out), flush=True)
'''
for rep in range(3):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, ISX_LIB=lib), check=True)
