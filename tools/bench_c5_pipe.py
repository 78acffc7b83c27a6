#!/usr/bin/env python3
"""C5 per-GPU shard through the pipe with linkage off / on, per-batch stage times (ISX_PIPE_TIMING=1 for the submit split)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
import bench
from instrain_amd import dist as idist, engine, synth
from tests import util
bench.bind_to_gpu_numa_node(torch, 0)
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
meta = synth.Metagenome(1000, total_read_bp=10e9, seed=5)
kept = meta.kept_genomes()
mine = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
est = (meta.pairs[mine] * 2 * meta.read_len * 0.92).astype(np.int64)
ws = [meta.generate(mine[b]) for b in idist.pack_batches(meta.length[mine], est, 40_000_000, 150_000_000)]
for link in (False, True, True):
    pipe = engine.Pipe(ctx, max_pos=max(w["n_pos"] for w in ws), max_obs=max(w["n_obs"] for w in ws), max_splits=max(len(w["split_bounds"]) for w in ws),
                       depth=3, host_threads=int(os.environ.get("THREADS", 24)), n_mm_bins=1, enable_linkage=link, min_snp=20, jump_slack=0.5)
    bench.stream(pipe, ws, 2, 3)
    stats = []
    t0 = time.perf_counter()
    bench.stream(pipe, ws, len(ws), 3, stats)
    dt = time.perf_counter() - t0
    print("linkage", link, "wall %.1f ms" % (dt * 1e3), "encode", [round(s[0]["encode_ms"], 1) for s in stats], "collect_wait", [round(s[0]["collect_wait_ms"], 1) for s in stats], flush=True)
    pipe.close()
