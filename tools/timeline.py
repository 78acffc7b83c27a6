#!/usr/bin/env python3
"""Per-phase timeline of ONE workgroup of k_pileup_dense (tuning build, debug bit 4096: wall_clock64 stamps of workgroup 0 after each
barrier of its first 32 windows).  WORK=c2 (resident C2 batch, linkage off) or c5 (one C5 batch, linkage on):
    tools/build_tuning.sh && ISX_LIB=instrain_amd/libinstrain_amd_tuning.so WORK=c5 python tools/timeline.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import _lib
from instrain_amd import dist as idist
from instrain_amd import engine, synth
from tests import util

NAMES = ["zero rows", "stream", "materialise a", "materialise b", "epilogue pass 1", "cursors (+clon barrier)", "deferred clonality (+barrier)",
         "rare + rows", "allele pass", "prefetch + end barrier"]
work = os.environ.get("WORK", "c5")
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
if work == "c2":
    w = bench.c2_workload(2)
    link = os.environ.get("LINK", "0") == "1"
    kw = {}
else:
    meta = synth.Metagenome(1000, total_read_bp=10e9, seed=5)
    kept = meta.kept_genomes()
    shard = kept[idist.lpt_shards(meta.pairs[kept], 8)[0]]
    b0 = idist.pack_batches(meta.length[shard], (meta.pairs[shard] * 2).astype(np.int64), 40_000_000, 1_000_000)[3]
    w = meta.generate_segs(shard[b0])
    link = os.environ.get("LINK", "1") == "1"
    kw = {"min_snp": 20}
os.environ["ISX_DEBUG_MODE"] = str(4096 | int(os.environ.get("DBG", "0")))
b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs"], None, n_mm_bins=1, enable_linkage=link, window=int(os.environ.get("WINDOW", "0")), layout=int(os.environ.get("LAYOUT", "0")), **kw)
for _ in range(4):
    b.run()
t = b.timings()
ts = np.zeros(32 * 16, dtype=np.uint64)
lib = _lib.load()
lib.isx_debug_read_ts.argtypes = [ctypes.c_void_p]
rc = lib.isx_debug_read_ts(ts.ctypes.data)
assert rc == 0, rc
ts = ts.reshape(32, 16).astype(np.int64)
n_w = -(-(-(-w["n_pos"] // t["pileup_window"])) // t["pileup_blocks"])
print("%s: W=%d block=%d grid=%d kernel %.4f ms; workgroup 0 runs ~%d windows" % (work, t["pileup_window"], t["pileup_threads"], t["pileup_blocks"], t["pileup_ms"], n_w))
rows = []
for i in range(min(32, n_w)):
    s = ts[i]
    if s[0] == 0:
        break
    d = []
    prev = s[0]
    for k in range(1, 11):
        if s[k] >= prev and s[k] != 0 and (i == 0 or s[k] >= ts[i][0]):
            d.append((s[k] - prev) / 100.0)      # 100 MHz -> us
            prev = s[k]
        else:
            d.append(0.0)                         # phase not stamped in this window (uniform skip)
    rows.append(d + [(s[10] - s[0]) / 100.0])
    print("win %2d: " % i + " ".join("%6.2f" % x for x in rows[-1]))
rows = np.array(rows)
print("phase means (us):")
for k, nm in enumerate(NAMES):
    print("  %-32s %6.2f" % (nm, rows[:, k].mean()))
print("  %-32s %6.2f" % ("window", rows[:, 10].mean()))
b.close()
