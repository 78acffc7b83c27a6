#!/usr/bin/env python3
"""The launches the hardware counters are collected on (run under rocprofv3 --pmc ... or --kernel-trace --stats): every dispatch of a
kernel name is the same work.
  default   resident C2 batches (count table out, linkage off), 8 launches each:
              k_pileup_dense<false, 32, true>   reference-delta records, 16-bit LDS rows (the production stream of a read-level batch)
              k_pileup_dense<false, 32, false>  the same records, 32-bit LDS rows (very deep batches)
              k_pileup_dense<false, 64, false>  64-byte segment records (round 3)
              k_pileup_dense<false, 2, true>    2-byte observation records (round 2)
            + with mm profiling on: k_pileup_mm<..., SEGS> (segment records) / <..., DREC> (reference-delta records, round 6) / <...> (observations)
  --c5      ONE batch of the C5 headline (the one of this shard closest to the N = 1 average launch, ~108 Mbp; linkage on) through a pipe slot -- the shrunk slot
            output, what bench.py's roofline object is priced on -- submitted 8 times from its staged wire:
              k_pileup_dense<true, 32, true>
usage: python tools/pmc_target.py [--no-mm] [--c5]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from instrain_amd import engine
from tests import util

ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
if "--c5" in sys.argv:
    from instrain_amd import dist as idist
    run = bench.C5Run(ctx, 0, 8, 4, depth=1, min_batches=0)        # rank 0 of 8: one shard is enough to find a batch (of the N = 1 size)
    run.stage_all()
    sizes = [w["n_pos"] for w in run.ws]
    # the batch closest to the AVERAGE launch of the N = 1 headline (all 8 shards' positions over all their batches): what bench.py's
    # roofline.traffic is quoted against (the round-5 pass picked a 117 Mbp batch against a 108 Mbp average)
    meta, kept = run.meta, run.kept
    n_b = sum(len(idist.pack_batches(meta.length[kept[sh]], (meta.pairs[kept[sh]] * 2).astype(np.int64), bench.C5_BATCH_POS, bench.C5_BATCH_SEGS)) for sh in run.shards)
    avg = float(meta.length[kept].sum()) / max(n_b, 1)
    k = int(np.argmin([abs(x - avg) for x in sizes]))
    print("N = 1 headline: %d batches, average %.1f Mbp a launch; this shard's batches: %s Mbp" % (n_b, avg / 1e6, [round(x / 1e6, 1) for x in sizes]), flush=True)
    w, wire = run.ws[k], run.wires[k]
    for _ in range(8):
        t = run.pipe.submit_wire(wire)
        r = run.pipe.collect(t, densify=False)
        st = r["stats"]
        run.pipe.release(t)
    print("c5 batch %d: %d positions, %d segments, %d kept observations, kernel %.4f ms, h2d %d bytes, record_bytes %d"
          % (k, w["n_pos"], w["n_seg"], w["n_obs"], st["kernel_ms"], st["h2d_bytes"], st["record_bytes"]), flush=True)
    run.close()
else:
    with_mm = "--no-mm" not in sys.argv
    w = bench.c2_workload(seed=2, with_mm=with_mm)
    jobs = [(w["segs"], 1, 0), (w["segs"], 1, 4), (w["segs"], 1, 8), (w["obs"], 1, 0)]
    if with_mm:         # (layout 32: ISX_LAYOUT_MM_DELTA_RECORDS, the round-6 reference-delta records with the mm level in the header)
        jobs += [(w["segs_mm"], w["n_mm_bins_mm"], 0), (w["segs_mm"], w["n_mm_bins_mm"], 32), (w["obs_mm"], w["n_mm_bins_mm"], 0)]
    for src, M, layout in jobs:
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, None, n_mm_bins=M, enable_linkage=False, layout=layout)
        for _ in range(8):
            b.run()
        t = b.timings()
        print(type(src).__name__, M, layout, t["record_bytes"], t["pileup_window"], t["pileup_ms"], flush=True)
        b.close()
ctx.close()
