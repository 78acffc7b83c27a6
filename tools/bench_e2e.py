#!/usr/bin/env python3
"""End-to-end timing of instrain_amd.profile.profile_bam on a synthetic BAM: front end, device batch,
table fetch, SplitObject assembly.  usage: python tools/bench_e2e.py [n_pairs] [genome_len]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_front import write_simple_bam

n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
path = "/tmp/bench_front_%d.bam" % n_pairs
if not os.path.exists(path):
    write_simple_bam(path, G, n_pairs)
rng = np.random.Generator(np.random.PCG64(1))                # the same reference write_simple_bam draws first
ref = rng.integers(0, 4, G, dtype=np.uint8)
seq = "".join(np.array(list("ACTG"))[ref])
import instrain_amd.profile as amd
from instrain_amd import engine
from tests import util
lut, fb = util.load_lut()
nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
nm[-1] = fb
ctx = engine.Context(0)
for rep in range(2):
    for skip in (False, True):
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        out = amd.profile_bam(path, None, None, None, s2s={"scaf": seq}, null_model=nm, ctx=ctx, skip_mm_profiling=skip)
        pr.disable()
        dt = time.perf_counter() - t0
        print("skip_mm=%s: %d SplitObjects in %.2f s -> %.3f Gbp/s end to end" % (skip, len(out), dt, n_pairs * 300 / 1e9 / dt), flush=True)
        if rep == 1:
            pstats.Stats(pr).sort_stats("tottime").print_stats(22)
