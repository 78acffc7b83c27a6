#!/usr/bin/env python3
"""gpurun_out/<tag>/ (made by tools/make_profiles.sh on the GPU box) -> profiles/<tag>_*.md, <tag>_bench_n1.json,
pmc_traffic.json.  usage: python tools/write_profiles.py <tag>"""
import json, subprocess, sys
import pandas as pd
tag = sys.argv[1]
R = 'gpurun_out/' + tag
j = json.load(open(R + '/bench_n1.json'))
json.dump(j, open('profiles/%s_bench_n1.json' % tag, 'w'), indent=1)


def summ(*dirs):
    return subprocess.check_output([sys.executable, 'tools/prof_summary.py', *dirs]).decode()


ks = pd.read_csv(R + '/trace_c2/%s_kernel_stats.csv' % tag)
row = ks[ks['Name'].str.contains('k_pileup_dense')].iloc[0]
avg = row['TotalDurationNs'] / row['Calls'] / 1e3
ev = j["roofline"]["kernel_ms_avg"] * 1e3
evs = j["roofline"].get("kernel_ms_in_stream", 0) * 1e3
open('profiles/%s_c2_kernel_stats.md' % tag, 'w').write(f'''# Round {int(tag[1:])} — rocprofv3 --kernel-trace --stats, C2 workload (final kernels of the round)

Command (tools/make_profiles.sh): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-c5-leg`

k_pileup_dense<false, 2> (linkage off, 2-byte record stream): {int(row['Calls'])} calls, average {avg:.1f} us, min {row['MinNs']/1e3:.1f} us, max {row['MaxNs']/1e3:.1f} us under rocprof.
The calls are of two kinds: 34 launches inside the streamed pipe (4 warm-up + 30 timed batches; a launch is ~3 % of a PCIe-bound
step, the GPU idles in between, un-profiled bench: {evs:.1f} us each = roofline.kernel_ms_in_stream) and 44 back-to-back launches of
the resident leg (4 + 10 blocking + 30 pipelined; un-profiled bench: {ev:.1f} us = roofline.kernel_ms_avg, the figure the roofline is
priced on; MinNs above is the same kernel at full clocks).

''' + summ(R + '/trace_c2'))
cf = pd.read_csv(R + '/pmc_fetch/%s_counter_collection.csv' % tag)
cw = pd.read_csv(R + '/pmc_write/%s_counter_collection.csv' % tag)


def mean(c, k):
    return c[c['Kernel_Name'].str.contains(k)]['Counter_Value'].mean()


fd, wd = mean(cf, 'k_pileup_dense'), mean(cw, 'k_pileup_dense')
fm, wm = mean(cf, 'k_pileup_mm'), mean(cw, 'k_pileup_mm')
rd, wr = fd * 1024 * 2, wd * 1024
rm, wmm = fm * 1024 * 2, wm * 1024
ad = j['roofline']['algorithmic_bytes_per_launch']
am = j['mm_on']['roofline']['algorithmic_bytes_per_launch']
open('profiles/%s_c2_pmc.md' % tag, 'w').write(f'''# Round {int(tag[1:])} — HBM traffic of the pileup kernels on C2 (separate --pmc passes, bench.py --steps 5 --no-linkage-leg --no-c5-leg)

FETCH_SIZE / WRITE_SIZE are in KiB. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads, so read bytes = FETCH_SIZE x 1024 x 2.
Algorithmic bytes are priced on the resident layout: {j["roofline"]["record_bytes"]} bytes per observation for one mm bin, 4 with mm profiling on, 1 B/pos reference, 20 B/pos (dense) or 32 B/entry (mm) out
(pipe slots additionally write 2 B/pos of coverage16: the streamed launches of this run).

* k_pileup_dense (C2, skip-mm): read {rd/1e6:.1f} MB + written {wr/1e6:.1f} MB = **{(rd+wr)/1e6:.1f} MB per launch** vs {ad/1e6:.1f} MB algorithmic = {(rd+wr)/ad:.2f}x (window over-scan of the record stream).
* k_pileup_mm (C2, mm on, W = {j["mm_on"]["roofline"]["window"]}): read {rm/1e6:.1f} MB + written {wmm/1e6:.1f} MB = {(rm+wmm)/1e6:.1f} MB vs {am/1e6:.1f} MB algorithmic = {(rm+wmm)/am:.2f}x.

''' + summ(R + '/pmc_fetch', R + '/pmc_write'))
json.dump({"c2_pileup_bytes_per_launch": int(rd + wr), "fetch_size_kib": float(fd), "write_size_kib": float(wd),
           "c2_mm_pileup_bytes_per_launch": int(rm + wmm),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on bench.py C2 (profiles/%s_c2_pmc.md); FETCH_SIZE doubled per the gfx950 correction" % tag,
           "round": int(tag[1:])}, open('profiles/pmc_traffic.json', 'w'), indent=1)
l = j['linkage']
open('profiles/%s_linkage_kernel_stats.md' % tag, 'w').write(f'''# Round {int(tag[1:])} — rocprofv3 --kernel-trace --stats, bench.py with the linkage leg (BASELINE configs[2]: 5 Mbp, 200x, 50 000 SNV sites)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mm-leg --no-c5-leg`
(7 runs of the sparse path and 7 of the dense MFMA path over the same resident workload; k_pileup_dense<true, ...> is the linkage-on pileup
with the allele pass, k_pileup_dense<false, ...> the C2 steps of the same command.)

Un-profiled bench line of the same box: sparse {l["sparse"]["snv_pairs_linked_per_s"]/1e6:.1f} M SNV pairs/s ({l["sparse"]["ms_per_step"]:.2f} ms per step:
{l["sparse"]["kernel_ms"]}); dense MFMA pass {l["dense_mfma"]["mfma"]["pass_ms"]:.3f} ms = {l["dense_mfma"]["mfma"]["achieved_tops"]:.0f} int8 TOPS
= {l["dense_mfma"]["mfma"]["utilisation"]*100:.1f} % of the 5 POPS dense peak (useful tiles only).

''' + summ(R + '/trace_linkage'))
c5 = j.get('c5', {})
open('profiles/%s_c5_kernel_stats.md' % tag, 'w').write(f'''# Round {int(tag[1:])} — rocprofv3 --kernel-trace --stats, bench.py with the C5 leg (configs[4]: per-GPU shard of the 1000-genome database)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-resident-leg`
The k_pileup_dense rows mix the 5 C2 batches with the C5 shard's batches (2 warm-up + {c5.get("roofline", {}).get("launches", "?")} timed, 25-40 Mbp of positions and
~100 M records each: ~0.25 ms per launch).  Un-profiled bench line of the same box: {c5.get("gbp_per_s", 0):.1f} Gbp/s for the shard
({c5.get("seconds", 0)*1e3:.0f} ms; stages {c5.get("stages_ms_total")}), kernel total {c5.get("roofline", {}).get("kernel_ms_total", 0):.2f} ms = {c5.get("roofline", {}).get("frac", 0)*100:.0f} % of the HBM roof
on its algorithmic bytes.

''' + summ(R + '/trace_c5'))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["cpu_baseline"]["value"], c5.get("gbp_per_s"))
