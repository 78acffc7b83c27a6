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
    t_sub, t_col, t_rel = [], [], []
    t0 = time.perf_counter()
    tickets = []
    done = 0
    def take():
        global done
        a = time.perf_counter()
        r = pipe.collect(tickets[done], want_ld=link)
        b = time.perf_counter()
        stats.append((r["stats"], r["sizes"]))
        pipe.release(tickets[done])
        t_col.append((b - a) * 1e3); t_rel.append((time.perf_counter() - b) * 1e3)
        done += 1
    for i, v in enumerate(ws):
        if len(tickets) - done == 3:
            take()
        a = time.perf_counter()
        tickets.append(pipe.submit(v["ref_codes"], v["split_bounds"], v["obs"], v["pair"] if link else None))
        t_sub.append((time.perf_counter() - a) * 1e3)
    while done < len(tickets):
        take()
    dt = time.perf_counter() - t0
    print("caller: submit %.1f ms" % sum(t_sub), [round(x, 1) for x in t_sub], "collect %.1f ms" % sum(t_col), [round(x, 1) for x in t_col], "release %.1f" % sum(t_rel), flush=True)
    print("linkage", link, "wall %.1f ms" % (dt * 1e3), "encode", [round(s[0]["encode_ms"], 1) for s in stats], "collect_wait", [round(s[0]["collect_wait_ms"], 1) for s in stats], "passes", [s[0]["encode_passes"] for s in stats], flush=True)
    pipe.close()
