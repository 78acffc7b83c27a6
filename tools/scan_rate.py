#!/usr/bin/env python3
"""Pass 1 of the BAM front end (isx_bam_open + isx_bam_scan) on the bench's probe BAM, host only: the table-driven block decoder
(csrc/fast_inflate.h) against zlib (ISX_BAM_ZLIB=1) -- run it once per setting; ISX_BAM_TIMING=1 prints the pass's stages.
usage: python tools/scan_rate.py [threads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instrain_amd import engine, synth

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
path = "/tmp/isx_inflate_probe.bam"
if not os.path.exists(path):
    n_pairs, G = 3_000_000, 24_000_000
    meta = synth.Metagenome(1, total_read_bp=n_pairs * 300.0, seed=21, contigs=1, len_lo=G, len_hi=G, abundance_sigma=0.0,
                            min_genome_coverage=0.0, site_frac=0.001, threads=threads)
    meta.write_bam([0], path)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter()
    b = engine.BamFile(path, threads=threads)
    b.scan()
    dt = time.perf_counter() - t0
    best = min(best, dt)
    b.close()
print("open + scan of %.0f MB on %d threads: %.0f ms (%s)" % (os.path.getsize(path) / 1e6, threads, best * 1e3, "zlib" if os.environ.get("ISX_BAM_ZLIB") else "fast_inflate.h"), flush=True)
