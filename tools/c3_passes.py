"""Tuning aid: time of the dense pileup kernel at 200x coverage with / without the allele pass
(second stream over the window's records), for several explicit windows."""
import sys
import numpy as np
sys.path.insert(0, ".")
from instrain_amd import engine, synth
from tests import util
ctx = engine.Context(0)
lut, fb = util.load_lut(); ctx.set_null_model(lut, fb)
w = synth.make_workload(genome_len=2_000_000, coverage=200, n_sites=20000, seed=3, skip_mm=True, af_lo=0.2, af_hi=0.5)
for window in [0] + [int(x) for x in sys.argv[1:]]:
    for link in (False, True):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=link, window=window)
        for _ in range(3): b.run()
        t = b.timings()
        print("window", window, "linkage", link, "pileup_ms", round(t["pileup_ms"], 4), "W", t["pileup_window"], "blocks", t["pileup_blocks"],
              "lds", t.get("pileup_lds_bytes", t.get("lds_bytes")), "GB/s", round(len(w["obs"]) * 8 / t["pileup_ms"] / 1e6, 1), flush=True)
        b.close()
