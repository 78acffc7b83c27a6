#!/usr/bin/env python3
"""Sweep (window, workgroup size) of k_pileup_dense on ONE C5 batch (shallow metagenome: ~40 Mbp of positions, depth ~3, linkage
on), resident.  Tuning build: tools/build_tuning.sh && ISX_LIB=instrain_amd/libinstrain_amd_tuning.so python tools/tune_c5.py
LAYOUT=8 for the 64-byte segment records; DBG=<bits,...> for ablations instead of the sweep."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instrain_amd import dist as idist
from instrain_amd import engine, synth
from tests import util

LAYOUT = int(os.environ.get("LAYOUT", "0"))
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
meta = synth.Metagenome(1000, total_read_bp=10e9, seed=5)
kept = meta.kept_genomes()
shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
b0 = idist.pack_batches(meta.length[shard], (meta.pairs[shard] * 2).astype(np.int64), 40_000_000, 1_000_000)[int(os.environ.get("BATCH", "3"))]
w = meta.generate_segs(shard[b0])
print("batch: %d positions, %d segments, depth %.2f" % (w["n_pos"], w["segs"].n_seg, w["n_obs"] / w["n_pos"]), flush=True)
link = os.environ.get("LINK", "1") == "1"
if "DBG" in os.environ:
    combos = [(0, 0, int(d)) for d in os.environ["DBG"].split(",")]
else:
    combos = [(0, 0, 0)] + [(W, B, 0) for B in (1024, 512, 256) for W in (512, 768, 1024, 1536, 2048, 2688) if W <= 4 * B]
if "COMBOS" in os.environ:       # COMBOS=W:block[:dbg],...
    combos = [tuple((list(map(int, c.split(":"))) + [0])[:3]) for c in os.environ["COMBOS"].split(",")]
for W, B, dbg in combos:
    for k in ("ISX_GRID", "ISX_BLOCK"):
        os.environ.pop(k, None)
    if B:
        os.environ["ISX_BLOCK"] = str(B)
    os.environ["ISX_DEBUG_MODE"] = str(dbg)
    try:
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs"], None, n_mm_bins=1, enable_linkage=link, window=W, layout=LAYOUT, min_snp=20)
        for _ in range(2):
            b.run()
        ts = []
        for _ in range(6):
            b.run()
            ts.append(b.timings()["pileup_ms"])
        t = b.timings()
        print("W=%5d block=%5d grid=%5d dbg=%4d lds=%6d  avg %.4f ms  min %.4f   (%.1f Gpos/s)" % (t["pileup_window"], t["pileup_threads"], t["pileup_blocks"], dbg,
              t["pileup_lds_bytes"], np.mean(ts), np.min(ts), w["n_pos"] / np.min(ts) / 1e6), flush=True)
        b.close()
    except Exception as e:
        print(W, B, dbg, "ERR", str(e)[:100], flush=True)
