#!/usr/bin/env python3
"""rocprofv3 kernel trace (csv) -> what ran WHILE the pileup kernels ran: per pileup dispatch its duration and the other kernels whose
intervals overlap it (summed overlap per kernel name), averaged over the last 60 % of the pileup dispatches.
usage: trace_overlap.py <kernel_trace.csv> [name-substring-of-the-kernel, default k_pileup_dense]"""
import sys

import numpy as np
import pandas as pd

df = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
key = sys.argv[2] if len(sys.argv) > 2 else "k_pileup_dense"
is_p = df["Kernel_Name"].str.contains(key)
P = df[is_p]
P = P.iloc[int(len(P) * 0.4):]
O = df[~is_p | True]
s_all, e_all, names = O["Start_Timestamp"].to_numpy(), O["End_Timestamp"].to_numpy(), O["Kernel_Name"].str.slice(0, 48).to_numpy()
acc, durs = {}, []
for idx, (s, e) in zip(P.index, P[["Start_Timestamp", "End_Timestamp"]].to_numpy()):
    durs.append(e - s)
    sel = np.flatnonzero((s_all < e) & (e_all > s))
    for j in sel:
        if O.index[j] == idx:
            continue
        ov = min(e, e_all[j]) - max(s, s_all[j])
        a = acc.setdefault(names[j], [0, 0])
        a[0] += 1
        a[1] += ov
durs = np.array(durs)
print("%d dispatches of %s: mean %.3f ms, median %.3f, min %.3f, max %.3f" % (len(durs), key, durs.mean() / 1e6, np.median(durs) / 1e6, durs.min() / 1e6, durs.max() / 1e6))
print("overlapping kernels per dispatch (count, overlap as a share of the dispatch's duration):")
for nm, (c, ov) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-50s %6.1f  %5.1f %%" % (nm, c / len(durs), 100.0 * ov / durs.sum()))
