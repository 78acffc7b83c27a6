#!/usr/bin/env python3
"""Per-phase timeline of ONE workgroup of k_pileup_mm on the C2 batch with mm profiling on (tuning build, debug bit 4096: wall_clock64
stamps of workgroup 0 after each barrier of its first 32 windows).
    tools/build_tuning.sh && ISX_LIB=instrain_amd/libinstrain_amd_tuning.so python tools/timeline_mm.py
LAYOUT=32: reference-delta records with the mm level in the header (ISX_LAYOUT_MM_DELTA_RECORDS) instead of the 64-byte segment records; SPARSE=1: through a pipe slot (level-sparse tables), LEAN=0: a plain slot (entries kept)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from instrain_amd import _lib
from instrain_amd import engine
from tests import util

NAMES = ["zero", "stream", "materialise", "level masks (sparse)", "level loop", "lists + deferred clonality", "rows", "allele pass", "end barrier"]
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
w = bench.c2_workload(2, with_mm=True)
M = w["n_mm_bins_mm"]
layout = int(os.environ.get("LAYOUT", "0"))
os.environ["ISX_DEBUG_MODE"] = str(4096 | int(os.environ.get("DBG", "0")))
if os.environ.get("SPARSE"):
    pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=0, max_segs=int(w["segs_mm"].n_seg), max_splits=len(w["split_bounds"]), depth=1, host_threads=8,
                       n_mm_bins=M, enable_linkage=False, lean_output=bool(int(os.environ.get("LEAN", "1"))), layout=layout)
    for _ in range(3):
        t = pipe.submit_reads(w["ref_codes"], w["split_bounds"], w["segs_mm"])
        r = pipe.collect(t, shrunk_entries=True, densify=False)
        tm = r["slot"].timings()
        pipe.release(t)
else:
    b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["segs_mm"], None, n_mm_bins=M, enable_linkage=False, layout=layout)
    for _ in range(4):
        b.run()
    tm = b.timings()
ts = np.zeros(32 * 16, dtype=np.uint64)
lib = _lib.load()
lib.isx_debug_read_ts.argtypes = [ctypes.c_void_p]
assert lib.isx_debug_read_ts(ts.ctypes.data) == 0
ts = ts.reshape(32, 16).astype(np.int64)
n_w = -(-(-(-w["n_pos"] // tm["pileup_window"])) // tm["pileup_blocks"])
print("C2 mm on (%d bins), record bytes %d: W=%d block=%d grid=%d lds=%d kernel %.4f ms; workgroup 0 runs ~%d windows"
      % (M, tm["record_bytes"], tm["pileup_window"], tm["pileup_threads"], tm["pileup_blocks"], tm["pileup_lds_bytes"], tm["pileup_ms"], n_w))
rows = []
for i in range(min(32, n_w)):
    s = ts[i]
    if s[0] == 0:
        break
    d, prev = [], s[0]
    for k in range(1, 10):
        d.append((s[k] - prev) / 100.0 if s[k] >= prev and s[k] != 0 else 0.0)      # 100 MHz -> us
        if s[k] >= prev and s[k] != 0:
            prev = s[k]
    rows.append(d + [(s[9] - s[0]) / 100.0])
    print("win %2d: " % i + " ".join("%6.2f" % x for x in rows[-1]))
rows = np.array(rows)
print("phase means (us):")
for k, nm in enumerate(NAMES):
    print("  %-32s %6.2f" % (nm, rows[:, k].mean()))
print("  %-32s %6.2f" % ("window", rows[:, 9].mean()))
