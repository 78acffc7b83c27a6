#!/usr/bin/env python3
"""Sweep (window, workgroup size) of k_pileup_dense on the read-level C2 batch + ablations of its stream loop
(ISX_DEBUG_MODE: 2 no epilogue, 64 loads only, 8 decode without LDS, 16 conflict-free atomics); prints the kernel's own ms.
Tuning aid only -- needs the tuning build:
    make -C instrain_amd/csrc tuning && ISX_LIB=instrain_amd/libinstrain_amd_tuning.so python tools/tune_reads.py"""
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
w = bench.c2_workload(2, scale=float(os.environ.get("SCALE", "1.0")))
LAYOUT = int(os.environ.get("LAYOUT", "0"))        # 0 = reference-delta records, 8 = 64-byte segment records
if "--sweep" in sys.argv:
    combos = [(0, 0, 0)] + [(W, B, 0) for B in (1024, 512, 256) for W in (768, 1024, 1536, 2048, 2560, 3200) if W <= 4 * B]
elif LAYOUT == 0:       # delta records: 8 = no skip-plane walk, 16 = no materialise phase
    combos = [(0, 0, d) for d in (0, 2, 64, 8, 16, 8 | 16, 2 | 64, 2 | 8, 128, 256, 512, 1024, 2048, 128 | 256 | 512)]
else:
    combos = [(0, 0, d) for d in (0, 2, 64, 8, 16, 2 | 64, 128, 256, 512, 1024, 2048, 1024 | 2048, 128 | 256, 128 | 256 | 512, 128 | 512, 64 | 128 | 256 | 512)]
if "COMBOS" in os.environ:       # COMBOS=W:block[:dbg],...
    combos = [tuple((list(map(int, c.split(":"))) + [0])[:3]) for c in os.environ["COMBOS"].split(",")]
for W, B, dbg in combos:
    for k in ("ISX_GRID", "ISX_BLOCK"):
        os.environ.pop(k, None)
    if B:
        os.environ["ISX_BLOCK"] = str(B)
    os.environ["ISX_DEBUG_MODE"] = str(dbg)
    try:
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs"], None, n_mm_bins=1, enable_linkage=False, window=W, layout=LAYOUT)
        for _ in range(3):
            b.run()
        ts = []
        for _ in range(20):
            b.run()
            ts.append(b.timings()["pileup_ms"])
        t = b.timings()
        print("W=%5d block=%5d grid=%5d dbg=%3d lds=%6d  avg %.4f ms  min %.4f" % (t["pileup_window"], t["pileup_threads"], t["pileup_blocks"], dbg,
                                                                                t["pileup_lds_bytes"], np.mean(ts), np.min(ts)), flush=True)
        b.close()
    except Exception as e:
        print(W, B, dbg, "ERR", str(e)[:100], flush=True)
os.environ["ISX_DEBUG_MODE"] = "0"
