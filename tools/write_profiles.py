#!/usr/bin/env python3
"""gpurun_out/<tag>/ (made by tools/make_profiles.sh on the GPU box) -> profiles/r01_*.md, r01_bench_n1.json,
pmc_traffic.json.  usage: python tools/write_profiles.py <tag>"""
import json, subprocess, sys
import pandas as pd
tag = sys.argv[1]
R = 'gpurun_out/' + tag
j = json.load(open(R + '/bench_n1.json'))
json.dump(j, open('profiles/r01_bench_n1.json', 'w'), indent=1)


def summ(*dirs):
    return subprocess.check_output([sys.executable, 'tools/prof_summary.py', *dirs]).decode()


ks = pd.read_csv(R + '/trace_c2/%s_kernel_stats.csv' % tag)
row = ks[ks['Name'].str.contains('k_pileup_dense')].iloc[0]
avg = row['TotalDurationNs'] / row['Calls'] / 1e3
ev = j["roofline"]["kernel_ms_avg"] * 1e3
open('profiles/r01_c2_kernel_stats.md', 'w').write(f'''# Round 1 — rocprofv3 --kernel-trace --stats, C2 workload (final kernels of the round)

Command (tools/make_profiles.sh): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-linkage-leg --no-mm-leg`

k_pileup_dense<false, 2> (linkage off, 2-byte record stream) average {avg:.1f} us (rocprof, all 44 calls) vs {ev:.1f} us (the dispatch's own time stamps on the 10
blocking steps of the un-profiled bench.py run of the same box, profiles/r01_bench_n1.json, roofline.kernel_ms_avg): agree
within {abs(avg - ev) / avg * 100:.1f} %.  (44 calls = 4 warm-up + 10 blocking + 30 timed steps; in the timed region consecutive passes
run in two queues and overlap at their tails -- MaxNs below, roofline.kernel_ms_avg_overlapped in the bench line -- and the
one-wave k_publish_state then waits for a free slot next to the other queue's pileup kernel.)

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
open('profiles/r01_c2_pmc.md', 'w').write(f'''# Round 1 — HBM traffic of the pileup kernels on C2 (separate --pmc passes, bench.py --steps 5 --no-linkage-leg)

FETCH_SIZE / WRITE_SIZE are in KiB. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads, so read bytes = FETCH_SIZE x 1024 x 2.
Algorithmic bytes are priced on the resident layout: {j["roofline"]["record_bytes"]} bytes per observation for one mm bin, 4 with mm profiling on, 1 B/pos reference, 20 B/pos (dense) or 32 B/entry (mm) out.

* k_pileup_dense (C2, skip-mm, W = {j["config"]["window"]}): read {rd/1e6:.1f} MB + written {wr/1e6:.1f} MB = **{(rd+wr)/1e6:.1f} MB per launch** vs {ad/1e6:.1f} MB algorithmic = {(rd+wr)/ad:.2f}x (window over-scan of the record stream).
* k_pileup_mm (C2, mm on, W = {j["mm_on"]["roofline"]["window"]}): read {rm/1e6:.1f} MB + written {wmm/1e6:.1f} MB = {(rm+wmm)/1e6:.1f} MB vs {am/1e6:.1f} MB algorithmic = {(rm+wmm)/am:.2f}x.

''' + summ(R + '/pmc_fetch', R + '/pmc_write'))
json.dump({"c2_pileup_bytes_per_launch": int(rd + wr), "fetch_size_kib": float(fd), "write_size_kib": float(wd),
           "c2_mm_pileup_bytes_per_launch": int(rm + wmm),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on bench.py C2 (profiles/r01_c2_pmc.md); FETCH_SIZE doubled per the gfx950 correction",
           "round": 1}, open('profiles/pmc_traffic.json', 'w'), indent=1)
l = j['linkage']
open('profiles/r01_linkage_kernel_stats.md', 'w').write(f'''# Round 1 — rocprofv3 --kernel-trace --stats, bench.py with the linkage leg (BASELINE configs[2]: 5 Mbp, 200x, 50 000 SNV sites)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mm-leg`
(7 runs of the sparse path and 7 of the dense MFMA path over the same resident workload; k_pileup_dense<true, ...> is the linkage-on pileup
with the allele pass, k_pileup_dense<false, ...> the C2 headline steps of the same command.)

Un-profiled bench line of the same box: sparse {l["sparse"]["snv_pairs_linked_per_s"]/1e6:.1f} M SNV pairs/s ({l["sparse"]["ms_per_step"]:.2f} ms per step:
{l["sparse"]["kernel_ms"]}); dense MFMA pass {l["dense_mfma"]["mfma"]["pass_ms"]:.3f} ms = {l["dense_mfma"]["mfma"]["achieved_tops"]:.0f} int8 TOPS
= {l["dense_mfma"]["mfma"]["utilisation"]*100:.1f} % of the 5 POPS dense peak (useful tiles only).

''' + summ(R + '/trace_linkage'))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["upload"], j["cpu_baseline"]["value"])
