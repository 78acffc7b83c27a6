#!/usr/bin/env python3
"""The reference's DEFAULT mode -- mm profiling on, linkage on -- as a stream of batches through one pipe, for a profiler: how many launches
a batch costs (run under `rocprofv3 --kernel-trace --stats`; tools/link_prof.sh does and divides by the batches).
usage: python tools/mm_launch_count.py [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from instrain_amd import _lib, engine, synth
from tests import util

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lut, fb = util.load_lut()
ctx = engine.Context(0)
ctx.set_null_model(lut, fb)
w = synth.make_workload(genome_len=2_000_000, coverage=20, n_sites=20000, seed=4, skip_mm=False, af_lo=0.1, af_hi=0.5)
M = int(w["obs"]["mm"].max()) + 1
segs = synth.segs_from_obs(w["obs"], w["pair"])
planes = engine.PlaneBatch.from_segs(segs, threads=8)
ref = engine.RefPlanes.from_codes(w["ref_codes"], threads=8)
pipe = engine.Pipe(ctx, max_pos=len(w["ref_codes"]), max_obs=0, max_segs=int(segs.n_seg), max_splits=len(w["split_bounds"]), depth=4, host_threads=8,
                   n_mm_bins=M, enable_linkage=True, lean_output=True, layout=_lib.LAYOUT_MM_DELTA_RECORDS, jump_slack=1.0)
tickets = []
sizes = None
t0 = time.perf_counter()
for i in range(n_batches):
    if len(tickets) >= 4:
        t = tickets.pop(0)
        r = pipe.collect(t, densify=False, shrunk_entries=True)
        sizes = r["sizes"]
        pipe.release(t)
    tickets.append(pipe.submit_planes(ref, w["split_bounds"], planes))
for t in tickets:
    r = pipe.collect(t, densify=False, shrunk_entries=True)
    pipe.release(t)
dt = time.perf_counter() - t0
pipe.close()
ctx.close()
print("default mode (mm on, %d bins; linkage on): %d batches of %.1f Mbp / %d segments in %.1f ms; per batch: %d SNV rows, %d sites, %d allele observations, %d increments, %d LD rows"
      % (M, n_batches, len(w["ref_codes"]) / 1e6, segs.n_seg, dt * 1e3, sizes["n_snv"], sizes["n_sites"], sizes["n_allele_obs"], sizes["n_increments"], sizes["n_ld"]))
print("BATCHES %d" % n_batches)
