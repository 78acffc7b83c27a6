#!/usr/bin/env python3
"""rocprofv3 kernel trace (csv) -> how busy the device was: union of kernel intervals, per-kernel totals, per-queue totals over
the window [t0, t1] of the trace that holds `frac` of the dispatches around its middle.  usage: trace_busy.py <kernel_trace.csv>"""
import sys

import pandas as pd

df = pd.read_csv(sys.argv[1])
df = df.sort_values("Start_Timestamp")
n = len(df)
lo, hi = int(n * 0.3), int(n * 0.9)
d = df.iloc[lo:hi]
t0, t1 = d["Start_Timestamp"].min(), d["End_Timestamp"].max()
iv = d[["Start_Timestamp", "End_Timestamp"]].to_numpy()
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.1f ms, device busy (union of kernels) %.1f ms = %.0f %%, sum of kernel durations %.1f ms" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), (d["End_Timestamp"] - d["Start_Timestamp"]).sum() / 1e6))
d = d.assign(dur=d["End_Timestamp"] - d["Start_Timestamp"], short=d["Kernel_Name"].str.slice(0, 60))
print(d.groupby("short")["dur"].agg(["count", "sum"]).sort_values("sum", ascending=False).head(14).assign(ms=lambda x: x["sum"] / 1e6).drop(columns="sum").to_string())
if "Queue_Id" in d.columns:
    print(d.groupby("Queue_Id")["dur"].agg(["count", "sum"]).assign(ms=lambda x: x["sum"] / 1e6).drop(columns="sum").to_string())
