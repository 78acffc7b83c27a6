#!/usr/bin/env python3
"""The two BAM-on-disk legs of bench.py alone (profile_bam on the 0.9 Gbp probe, the C5 shard as a BAM), for several host thread counts
(ISX_BAM_TIMING=1 / ISX_PIPE_TIMING=1 print the front end's and the pipe's stage times).   usage: python tools/bench_bam_legs.py [threads ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from instrain_amd import engine
from tests import util

torch.cuda.set_device(0)
bench.bind_to_gpu_numa_node(torch, 0)
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
for thr in [int(a) for a in sys.argv[1:]] or [16]:
    for name, fn in (("profile_bam", lambda: bench.profile_bam_leg(ctx, thr)), ("c5_bam", lambda: bench.c5_bam_leg(ctx, thr))):
        r = fn()
        print("threads %d %s: %.3f s = %.2f Gbp/s  stages %s" % (thr, name, r.get("seconds", 0), r.get("gbp_per_s", 0), json.dumps(r.get("stages_ms"))), flush=True)
        if "mm_on" in r:
            print("threads %d %s mm_on: %.3f s = %.2f Gbp/s  stages %s" % (thr, name, r["mm_on"]["seconds"], r["mm_on"]["gbp_per_s"], json.dumps(r["mm_on"]["stages_ms"])), flush=True)
ctx.close()
