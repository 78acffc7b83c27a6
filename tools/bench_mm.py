#!/usr/bin/env python3
"""The mm-on legs of bench.py alone (C2 with mm profiling on: the kernel over a resident batch, and the stream): a few seconds
instead of the whole line.  python tools/bench_mm.py [--steps N] [--scale S]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--variants", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--pin", action="store_true")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--skip-mm-too", action="store_true", help="also the M = 1 stream on the same box (the ratio the verdict asks about)")
    args = ap.parse_args()
    args.queued_submit = False
    import instrain_amd  # noqa: F401
    from instrain_amd import engine
    from tests import util
    ctx = engine.Context(0)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    ht = args.host_threads or max(2, min(48, bench.host_cpus()))
    w = bench.c2_workload(seed=2, scale=args.scale, with_mm=True)
    out = {}
    if not args.no_resident:
        out["mm_on"] = bench.mm_leg(ctx, w)
    out["c2_mm_stream"] = bench.c2_mm_stream_leg(ctx, w, args, ht, args.steps, args.warmup)
    if args.skip_mm_too:
        out["c2_stream"] = bench.c2_stream_leg(ctx, w, args, ht, args.steps, args.warmup)
        out["mm_over_skip_mm"] = out["c2_stream"]["gbp_per_s"] / out["c2_mm_stream"]["gbp_per_s"]
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
