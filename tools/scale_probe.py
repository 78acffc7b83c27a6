"""One-off scale probe: the per-GPU shard of BASELINE configs[4] (1000-genome database, 10 Gbp of reads
over 8 GPUs = 125 genomes / 1.25 Gbp of reads per GPU) as ONE resident batch.
usage: python tools/scale_probe.py [genome_len] [coverage]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from instrain_amd import engine, synth
from tests import util
G = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
t0 = time.time()
w = synth.make_workload(genome_len=G, coverage=cov, n_sites=G // 1000, seed=5, skip_mm=True, n_scaffolds=125)
print("generated", w["n_obs"], "obs in", round(time.time() - t0, 1), "s", flush=True)
ctx = engine.Context(0)
lut, fb = util.load_lut(); ctx.set_null_model(lut, fb)
t0 = time.time()
b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], w["obs"], w["pair"], n_mm_bins=1, enable_linkage=False)
print("batch_create", round(time.time() - t0, 2), "s", flush=True)
for _ in range(3):
    t0 = time.time(); b.run(); dt = time.time() - t0
t = b.timings(); sz = b.sizes()
print("run", round(dt * 1e3, 3), "ms  pileup_ms", round(t["pileup_ms"], 3), "W", t["pileup_window"], "GB/s", round(w["n_obs"] * 8 / t["pileup_ms"] / 1e6, 1),
      "Gbp/s", round(w["profiled_bases"] / dt / 1e9, 1), "n_snv", sz["n_snv"], flush=True)
# property: total of the counts == number of ACGT observations
d = b.fetch()
tot = int(d["counts"].astype(np.int64).sum())
print("counts total", tot, "obs with base<4", int((w["obs"]["base"] < 4).sum()))
assert tot == int((w["obs"]["base"] < 4).sum())
b.close()
