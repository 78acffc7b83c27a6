#!/usr/bin/env python3
"""What bounds the stream loop of k_pileup_dense (2-byte records)?  Needs the tuning build:
    make -C instrain_amd/csrc tuning && ISX_LIB=instrain_amd/libinstrain_amd_tuning.so python tools/ablate_dense.py
ISX_DEBUG_MODE bits: 2 no epilogue, 8 decode but no LDS atomic, 16 every atomic on a lane-private word (no conflicts, no
address math dependence), 32 one record of eight."""
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
w = bench.c2_workload(2)
for dbg, what in ((0, "full kernel"), (2, "stream loop only (no epilogue)"), (2 | 8, "stream: loads + decode, no LDS atomics"),
                  (2 | 16, "stream: atomics on lane-private words"), (2 | 32, "stream: 1 atomic per 8 records"), (2 | 64, "stream: loads only"), (8, "epilogue + loads + decode"), (64, "epilogue + loads only")):
    os.environ["ISX_DEBUG_MODE"] = str(dbg)
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], None, n_mm_bins=1, enable_linkage=False)
    for _ in range(3):
        b.run()
    ts = []
    for _ in range(20):
        b.run()
        ts.append(b.timings()["pileup_ms"])
    print("dbg %2d  %-45s avg %.4f ms  min %.4f ms" % (dbg, what, float(np.mean(ts)), float(np.min(ts))), flush=True)
    b.close()
os.environ["ISX_DEBUG_MODE"] = "0"
