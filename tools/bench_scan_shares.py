#!/usr/bin/env python3
"""Host only: pass 1 over a whole BAM vs over one share of it (what a rank of an N-GPU run scans, isx_bam_scan_part).
usage: python tools/bench_scan_shares.py [n_pairs genome_len n_scaffolds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_front import write_simple_bam
from instrain_amd import engine
n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000
G = int(float(sys.argv[2])) if len(sys.argv) > 2 else 3_000_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
path = "/tmp/bench_shares_%d_%d.bam" % (n_pairs, R)
if not os.path.exists(path):
    write_simple_bam(path, G, n_pairs, n_refs=R)
threads = int(os.environ.get("THREADS", 0))
for world in (1, 2, 4, 8):
    ts, owned = [], []
    for rank in range(world):
        best = 1e9
        for rep in range(3):
            bf = engine.BamFile(path, threads=threads)
            t0 = time.perf_counter()
            info = bf.scan(part=(rank, world))
            best = min(best, time.perf_counter() - t0)
            reads, _ = bf.ref_counts()
            bf.close()
        ts.append(best * 1e3); owned.append(int((reads > 0).sum()))
    print("world %d: scan of a share %s ms (slowest %.1f), scaffolds owned %s" % (world, [round(t, 1) for t in ts], max(ts), owned), flush=True)
