#!/usr/bin/env python3
"""How fast the device inflates a BAM's BGZF blocks (isx_bgzf_inflate_device) next to zlib on the host: the bench's probe BAM
(3 M read pairs on a 24 Mbp scaffold, ~340 MB) -> bytes equal, kernel ms, end-to-end ms of the call (H2D + kernel + D2H into pageable
memory), zlib ms on ONE host thread and on all of them.   usage: python tools/inflate_rate.py [pairs] [scaffold bp]"""
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instrain_amd import engine, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 24_000_000
threads = len(os.sched_getaffinity(0))
meta = synth.Metagenome(1, total_read_bp=n_pairs * 300.0, seed=21, contigs=1, len_lo=G, len_hi=G, abundance_sigma=0.0,
                        min_genome_coverage=0.0, site_frac=0.001, threads=max(2, threads))
path = "/tmp/isx_inflate_probe.bam"
if not os.path.exists(path):                    # (kept: the sweep over ISX_INFLATE_LPW runs this script once per value)
    meta.write_bam([0], path)
img = np.fromfile(path, dtype=np.uint8)
blocks, total = engine.bgzf_index(img)
print("BAM: %.1f MB in %d BGZF blocks -> %.1f MB inflated (%.2fx)" % (len(img) / 1e6, len(blocks), total / 1e6, total / len(img)), flush=True)


def host_range(lo, hi):
    out = []
    for b in blocks[lo:hi]:
        o, n = int(b["in_off"]), int(b["in_len"])
        out.append(zlib.decompress(img[o:o + n].tobytes(), -15))
    return b"".join(out)


t0 = time.perf_counter()
ref = host_range(0, len(blocks))
t_one = time.perf_counter() - t0
if os.environ.get("DEVICE_ONLY"):
    threads = 1
cuts = np.linspace(0, len(blocks), threads * 4 + 1).astype(int)
with ThreadPoolExecutor(threads) as ex:
    t0 = time.perf_counter()
    parts = list(ex.map(lambda k: host_range(int(cuts[k]), int(cuts[k + 1])), range(len(cuts) - 1)))
    t_all = time.perf_counter() - t0
assert b"".join(parts) == ref
print("zlib on the host: %.0f ms on one thread (%.2f GB/s out), %.0f ms on %d threads (%.2f GB/s)" % (t_one * 1e3, total / t_one / 1e9, t_all * 1e3, threads, total / t_all / 1e9), flush=True)
ctx = engine.Context(0)
best_k, best_w = 1e9, 1e9
for rep in range(int(os.environ.get("REPS", "4"))):
    t0 = time.perf_counter()
    out, ms = engine.bgzf_inflate(img, blocks, ctx=ctx)
    w = (time.perf_counter() - t0) * 1e3
    best_k, best_w = min(best_k, ms), min(best_w, w)
assert out.tobytes() == ref, "device output differs from zlib"
print("device: kernel %.2f ms (%.1f GB/s out, %.1f GB/s in), whole call %.0f ms (pageable buffers both ways); bytes equal zlib's"
      % (best_k, total / best_k / 1e6, len(img) / best_k / 1e6, best_w), flush=True)
ctx.close()
