#!/usr/bin/env python3
"""The linkage stages' two chains side by side on one box: the bucket chain (default) and the sorted chain (ISX_LINK_CHAIN=sorted).
(a) C3 (configs[2]) as a resident read-level batch: ms per step and the chain's device time;
(b) one rank's share of an 8-GPU C5 job, pre-staged images replayed (nothing but the device work and the finishers' chains).
usage: python tools/link_chain_ab.py [--c3-bp N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from instrain_amd import engine
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--c3-bp", type=int, default=5_000_000)
ap.add_argument("--no-c5", action="store_true")
ap.add_argument("--reverse", action="store_true", help="the sorted chain first")
ap.add_argument("--c5-only", default=None, help="bucket | sorted: that chain's C5 shard alone (for a profiler)")
args = ap.parse_args()
os.environ["ISX_BENCH_C3_BP"] = str(args.c3_bp)
lut, fb = util.load_lut()
out = {}
for chain in ((args.c5_only,) if args.c5_only else (("sorted", "bucket") if args.reverse else ("bucket", "sorted"))):
    if chain == "sorted":
        os.environ["ISX_LINK_CHAIN"] = "sorted"
    else:
        os.environ.pop("ISX_LINK_CHAIN", None)
    out[chain] = {}
    if args.c5_only is None:
      ctx = engine.Context(0)
      ctx.set_null_model(lut, fb)
      leg = bench.linkage_leg(ctx)
      r = leg["reads"]
      k = r["kernel_ms"]
      chain_ms = k["sites_ms"] + k["allele_ms"] + k["group_ms"] + k["incr_ms"] + k["ld_ms"]
      out[chain] = {"c3_ms_per_step": r["ms_per_step"], "c3_chain_device_ms": chain_ms, "c3_kernel_ms": k, "c3_edges": r["edges"], "c3_ld_rows": r["ld_rows"],
                  "c3_pair_increments": r["pair_increments"], "c3_snv_pairs_linked_per_s": r["snv_pairs_linked_per_s"],
                  "c3_obs_sparse_ms_per_step": leg["sparse"]["ms_per_step"]}
      print(chain, json.dumps(out[chain]), flush=True)
      ctx.close()
    if args.no_c5:
        continue
    ctx = engine.Context(0, reserve_cus=bench.C5_RESERVE_CUS)
    ctx.set_null_model(lut, fb)
    run = bench.C5Run(ctx, 0, 8, 16, depth=8)
    run.verify_pass()
    run.stage_all()
    for staged in (True, False):
        run.run(2, staged=staged)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.run(10, [], staged=staged)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        out[chain]["c5_shard_%s_ms_per_pass" % ("staged_replay" if staged else "submit_planes")] = dt * 1e3
        print(chain, "C5 shard 0 of 8, %s: %d batches, %.2f ms per pass = %.1f Gbp/s for this rank" %
              ("staged replay" if staged else "submit_planes", len(run.ws), dt * 1e3, run.bases / dt / 1e9), flush=True)
    run.close()
    ctx.close()
print(json.dumps(out))
