#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into a short table with
readable kernel names.  usage: prof_summary.py <dir-with-csvs> [more dirs] > profiles/xyz.md"""
import glob
import os
import re
import sys

import pandas as pd


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"rocprim::[A-Za-z0-9_]+::detail::(?:trampoline_kernel<)?.*?(radix_sort_[a-z_]+|merge_sort_block_merge|"
                  r"merge_sort_block_sort|radix_sort_block_sort|lookback_scan_kernel|init_lookback_scan_state_kernel|"
                  r"reduce_by_key[a-z_]*|run_length[a-z_]*|scan[a-z_]*|onesweep[a-z_]*|histogram[a-z_]*|transform[a-z_]*)", name)
    if "rocprim" in name:
        return "rocprim::" + (m.group(1) if m else "kernel")
    return re.sub(r"\(.*", "", name)[:60]


for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "*kernel_stats.csv"))):
        s = pd.read_csv(f)
        s["Kernel"] = s["Name"].map(short)
        g = s.groupby("Kernel", sort=False).agg(Calls=("Calls", "sum"), TotalNs=("TotalDurationNs", "sum"),
                                                MinNs=("MinNs", "min"), MaxNs=("MaxNs", "max")).reset_index()
        g["AvgNs"] = (g["TotalNs"] / g["Calls"]).round(1)
        g["Pct"] = (100 * g["TotalNs"] / g["TotalNs"].sum()).round(2)
        print("## kernel stats: %s\n" % f)
        print(g.sort_values("TotalNs", ascending=False).to_markdown(index=False))
        print()
    for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
        c = pd.read_csv(f)
        c["Kernel"] = c["Kernel_Name"].map(short)
        g = c.groupby(["Kernel", "Counter_Name"])["Counter_Value"].agg(["count", "mean", "min", "max"]).reset_index()
        g = g[~g["Kernel"].str.startswith("__amd")]
        print("## counters: %s\n" % f)
        print(g.to_markdown(index=False))
        print()
