#!/usr/bin/env python3
"""End-to-end timing of instrain_amd.profile.profile_bam on a synthetic sorted BAM written by the fast generator
(libisx_synth.so): front end scan / filter / read segments, device batches, tables, SplitObjects, with the stage times
profile_bam reports.  usage: python tools/bench_e2e.py [n_pairs] [genome_len] [n_contigs] [--mm] [--prof]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_pairs = int(float(args[0])) if len(args) > 0 else 3_000_000
G = int(float(args[1])) if len(args) > 1 else 24_000_000
contigs = int(args[2]) if len(args) > 2 else 1
import instrain_amd.profile as amd
from instrain_amd import engine, synth
from tests import util

meta = synth.Metagenome(1, total_read_bp=n_pairs * 300.0, seed=21, contigs=contigs, len_lo=G, len_hi=G, abundance_sigma=0.0,
                        min_genome_coverage=0.0, site_frac=0.001)
path = "/tmp/isx_e2e_%d_%d_%d.bam" % (n_pairs, G, contigs)
t0 = time.perf_counter()
info = meta.write_bam([0], path)
print("BAM: %d reads, %.2f Gbp, %.1f MB, written in %.1f s" % (info["n_reads"], info["profiled_bases"] / 1e9, os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
letters = np.array(list("ACTG"))
sb = info["scaffold_bounds"]
s2s = {n: "".join(letters[info["ref_codes"][sb[i]:sb[i + 1]]]) for i, n in enumerate(info["names"])}
lut, fb = util.load_lut()
nm = {i: int(v) for i, v in enumerate(lut) if v >= 0}
nm[-1] = fb
ctx = engine.Context(0)
modes = (False, True) if "--mm" in sys.argv else (True,)
for rep in range(3):
    for skip in modes:
        st = {}
        pr = cProfile.Profile()
        ctx.wait_closers()
        t0 = time.perf_counter()
        if "--prof" in sys.argv and rep == 2:
            pr.enable()
        out = amd.profile_bam(path, None, None, None, s2s=s2s, null_model=nm, ctx=ctx, skip_mm_profiling=skip, stats=st)
        pr.disable()
        dt = time.perf_counter() - t0
        print("skip_mm=%s: %d SplitObjects in %.3f s -> %.2f Gbp/s end to end; stages ms: %s" %
              (skip, len(out), dt, info["profiled_bases"] / 1e9 / dt, {k: round(v, 1) for k, v in st.items()}), flush=True)
        if "--prof" in sys.argv and rep == 2:
            pstats.Stats(pr).sort_stats("tottime").print_stats(18)
ctx.close()
