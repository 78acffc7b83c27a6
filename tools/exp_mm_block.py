#!/usr/bin/env python3
"""mm kernel at 6 / 16 bins: two 512-lane workgroups per CU (78 KB of LDS each) vs one of 1024 lanes (156 KB, window twice as
wide).  Needs the tuning build (ISX_BLOCK).  usage: ISX_LIB=instrain_amd/libinstrain_amd_tuning.so python tools/exp_mm_block.py"""
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
for M in (w["n_mm_bins_mm"], 8, 12, 16, 32):
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs_mm"], None, n_mm_bins=M, enable_linkage=False)
    ts = []
    for i in range(30):
        b.run()
        if i >= 5: ts.append(b.timings()["pileup_ms"])
    t = b.timings()
    out.append("M=%d %.4f ms W=%d lds=%d" % (M, np.mean(ts), t["pileup_window"], t["pileup_lds_bytes"]))
    b.close()
print("ISX_BLOCK=%s" % os.environ.get("ISX_BLOCK"), " | ".join(out), flush=True)
'''
for rep in range(2):
    for blk in ("512", "1024", None):           # None: the library's own choice (batch_pick_block)
        env = dict(os.environ)
        env.pop("ISX_BLOCK", None)
        if blk:
            env["ISX_BLOCK"] = blk
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
