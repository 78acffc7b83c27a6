#!/usr/bin/env python3
"""Where does the dense int8-MFMA co-occurrence path (linkage_mode 2: X^T X per split on the matrix cores) overtake the sparse
pair-increment chain (linkage_mode 1)?  Sweep of the SNV density at fixed coverage: the sparse chain's work grows with the
square of the SNV sites a read pair spans, the dense path's with the square of the sites of a split.
usage: python tools/mfma_crossover.py [genome_len] [coverage]   (run on a GPU box; prints a markdown table)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instrain_amd import engine, synth
from tests import util

G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
cov = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = engine.Context(0)
lut, fb = util.load_lut()
ctx.set_null_model(lut, fb)
print("| SNV spacing (bp) | sites | pairs linked | pair increments | sparse total ms (incr ms) | dense total ms (MFMA ms, tiles, GMAC) | dense / sparse |")
print("|---:|---:|---:|---:|---:|---:|---:|")
for spacing in (400, 200, 100, 50, 25, 12, 6):
    w = synth.make_workload(genome_len=G, coverage=cov, n_sites=G // spacing, seed=5, skip_mm=True, af_lo=0.2, af_hi=0.5)
    res = {}
    for mode in (1, 2):
        try:
            b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=True, linkage_mode=mode, min_snp=20)
            for _ in range(2):
                b.run()
            ts = []
            for _ in range(3):
                b.run()
                ts.append(b.timings())
            t = {k: float(np.median([x[k] for x in ts])) for k in ts[0]}
            res[mode] = (t, b.sizes())
            b.close()
        except engine.IsxError as e:
            res[mode] = (None, str(e))
    (ts_, ss), (td, sd) = res[1], res[2]
    if ts_ is None or td is None:
        print("| %d | - | - | - | %s | %s | - |" % (spacing, "ok" if ts_ else ss[:60], "ok" if td else sd[:60]))
        continue
    assert ss["n_ld"] == sd["n_ld"] and ss["n_edges"] == sd["n_edges"]
    print("| %d | %d | %d | %d | %.2f (%.2f) | %.2f (%.2f, %d, %.1f) | %.2f |" %
          (spacing, ss["n_sites"], ss["n_edges"], ss["n_increments"], ts_["total_ms"], ts_["incr_ms"], td["total_ms"], td["mfma_ms"],
           int(td["dense_tiles"]), td["dense_macs"] / 1e9, td["total_ms"] / ts_["total_ms"]), flush=True)
ctx.close()
