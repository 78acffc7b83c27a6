#!/usr/bin/env python3
"""One rank's share of an 8-GPU C5 job on one GPU (rank 0 of 8: one shard, 1/8 of the database): ms per pass for different batch budgets --
the per-rank figure the 8-GPU scaling run is made of.   usage: python tools/c5_rank_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from instrain_amd import engine
from tests import util

lut, fb = util.load_lut()
for mb in (0, 8, 16):
    ctx = engine.Context(0, reserve_cus=bench.C5_RESERVE_CUS)
    ctx.set_null_model(lut, fb)
    run = bench.C5Run(ctx, 0, 8, 16, depth=8, min_batches=mb)
    run.verify_pass()
    run.stage_all()
    for staged in (False, True):                # hand-over inside the step (the headline's way) / pre-staged images replayed
        run.run(2, staged=staged)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.run(10, [], staged=staged)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("min_batches %2d, %s: %2d batches of <= %.0f Mbp, %.2f ms per pass = %.1f Gbp/s for this rank (x8 = %.0f)" %
              (mb, "staged replay" if staged else "submit_planes", len(run.ws), max(w["n_pos"] for w in run.ws) / 1e6, dt * 1e3, run.bases / dt / 1e9, 8 * run.bases / dt / 1e9), flush=True)
    run.close()
    ctx.close()
