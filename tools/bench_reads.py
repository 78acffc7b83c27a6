#!/usr/bin/env python3
"""Read-level hand-over vs observation hand-over on C2 (tuning aid): kernel time over resident batches and streamed
throughput through the pipe.  python tools/bench_reads.py [--scale S] [--window W]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--mm", action="store_true")
    ap.add_argument("--linkage", action="store_true")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    from instrain_amd import engine, synth
    from tests import util
    ctx = engine.Context(0)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    w = synth.make_workload(genome_len=int(5_000_000 * a.scale), coverage=20, n_sites=int(5000 * a.scale), seed=2, skip_mm=not a.mm)
    M = w["n_mm_bins"]
    t0 = time.perf_counter()
    segs = synth.segs_from_obs(w["obs"], w["pair"])
    print("segments: %d (%.1f MB payload) from %d observations in %.2f s" % (segs.n_seg, segs.n_seg * 64 / 1e6, w["n_obs"], time.perf_counter() - t0))
    kw = dict(n_mm_bins=M, enable_linkage=a.linkage, window=a.window)
    for name, src, pr in (("obs", w["obs"], w["pair"] if a.linkage else None), ("reads", segs, None)):
        b = engine.Batch(ctx, w["ref_codes"], w["split_bounds"], src, pr, **kw)
        for _ in range(3):
            b.run()
        ks = []
        for _ in range(10):
            b.run()
            ks.append(b.pileup_ms())
        t = b.timings()
        print("%-5s resident: kernel %.4f ms (min %.4f), window %d, %d x %d lanes, lds %d, record bytes %d" %
              (name, float(np.mean(ks)), float(np.min(ks)), t["pileup_window"], t["pileup_blocks"], t["pileup_threads"], t["pileup_lds_bytes"], t["record_bytes"]))
        b.close()
    for name in ("obs", "reads", "reads queued"):
        queued = name.endswith("queued")
        name = name.split()[0]
        pipe = engine.Pipe(ctx, max_pos=w["n_pos"], max_obs=w["n_obs"] if name == "obs" else 0, max_segs=segs.n_seg if name == "reads" else 0,
                           max_splits=len(w["split_bounds"]), depth=a.depth, host_threads=a.threads, pin_threads=False, stage_async=queued, **kw)
        sub = (lambda: pipe.submit(w["ref_codes"], w["split_bounds"], w["obs"], w["pair"] if a.linkage else None)) if name == "obs" else \
              (lambda: pipe.submit_reads(w["ref_codes"], w["split_bounds"], segs))
        for phase in range(2):
            tickets, done, stats = [], 0, []
            t0 = time.perf_counter()
            for i in range(a.steps):
                if len(tickets) - done == a.depth:
                    stats.append(pipe.collect(tickets[done], want_ld=a.linkage, densify=False)["stats"]); pipe.release(tickets[done]); done += 1
                tickets.append(sub())
            while done < len(tickets):
                stats.append(pipe.collect(tickets[done], want_ld=a.linkage, densify=False)["stats"]); pipe.release(tickets[done]); done += 1
            dt = time.perf_counter() - t0
        mean = lambda k: float(np.mean([s[k] for s in stats]))
        print("%-5s streamed%s: %.2f Gbp/s, step %.3f ms (encode %.3f, h2d %.3f [%.1f MB], kernel %.4f, d2h %.3f [%.1f MB])" %
              (name, " (queued submits)" if queued else "", w["profiled_bases"] * a.steps / dt / 1e9, dt / a.steps * 1e3, mean("encode_ms"), mean("h2d_ms"), mean("h2d_bytes") / 1e6,
               mean("kernel_ms"), mean("d2h_ms"), mean("d2h_bytes") / 1e6))
        pipe.close()
    ctx.close()


if __name__ == "__main__":
    main()
