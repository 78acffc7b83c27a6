#!/usr/bin/env python3
"""Sweep of the streaming hand-over on the C2 workload: encoder threads x thread pinning x where the
caller's buffers live (first touched anywhere / on the GPU's NUMA node).  One generation, many pipes.
  python tools/bench_pipe.py [--scale 1.0] [--variants 8] [--steps 24]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def gpu_numa_node(dev=0):
    import torch
    p = torch.cuda.get_device_properties(dev)
    bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
    try:
        return int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read()), bdf
    except Exception:
        return -1, bdf


def node_cpus(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--variants", type=int, default=8)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--threads", default="8,16,24,32")
    ap.add_argument("--depth", type=int, default=4)
    args = ap.parse_args()
    import torch                               # before the library: both bring a HIP runtime, torch's must initialise first
    torch.cuda.set_device(0)
    node, bdf = gpu_numa_node(0)
    from instrain_amd import engine
    from tests import util
    ctx = engine.Context(0)
    lut, fb = util.load_lut()
    ctx.set_null_model(lut, fb)
    print("gpu %s numa node %d; cgroup cpus %s; os.cpu_count %d" % (bdf, node, bench.cgroup_cpus(), os.cpu_count()), flush=True)
    w = bench.c2_workload(seed=2, scale=args.scale)
    variants = bench.make_variants(w, args.variants)

    def sweep(tag):
        for nt in [int(x) for x in args.threads.split(",")]:
            for pin in (True, False):
                pipe = engine.Pipe(ctx, max_pos=max(v["n_pos"] for v in variants), max_obs=int(w["n_obs"]),
                                   max_splits=max(len(v["split_bounds"]) for v in variants), depth=args.depth,
                                   host_threads=nt, pin_threads=pin, n_mm_bins=1, enable_linkage=False)
                bench.stream(pipe, variants, 4, args.depth)
                stats = []
                c0 = time.process_time()
                t0 = time.perf_counter()
                bench.stream(pipe, variants, args.steps, args.depth, stats)
                dt = time.perf_counter() - t0
                cpu = time.process_time() - c0
                pipe.close()
                m = lambda k: float(np.mean([s[0][k] for s in stats]))
                print(json.dumps({"buffers": tag, "threads": nt, "pin": pin, "ms_per_step": round(dt / args.steps * 1e3, 3),
                                  "gbp_per_s": round(w["profiled_bases"] * args.steps / dt / 1e9, 2),
                                  "encode_ms": round(m("encode_ms"), 3), "h2d_ms": round(m("h2d_ms"), 3), "d2h_ms": round(m("d2h_ms"), 3),
                                  "cpu_ms_per_step": round(cpu / args.steps * 1e3, 1)}), flush=True)

    sweep("first-touch anywhere")
    if node >= 0:
        os.sched_setaffinity(0, node_cpus(node))
        for v in variants:                       # re-touch the caller's buffers on the GPU's node
            v["obs"] = v["obs"].copy()
            v["ref_codes"] = v["ref_codes"].copy()
        sweep("gpu node")
    ctx.close()


if __name__ == "__main__":
    main()
