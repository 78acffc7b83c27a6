#!/usr/bin/env python3
"""gpurun_out/<tag>/ (made by tools/make_profiles.sh on the GPU box) -> profiles/<tag>_*.md, <tag>_bench_n1.json,
pmc_traffic.json.  usage: python tools/write_profiles.py <tag>"""
import json
import subprocess
import sys

import pandas as pd

tag = sys.argv[1]
R = 'gpurun_out/' + tag
rnd = int(tag[1:])
j = json.load(open(R + '/bench_n1.json'))
json.dump(j, open('profiles/%s_bench_n1.json' % tag, 'w'), indent=1)
READS, OBS = r'k_pileup_dense<false, 64', r'k_pileup_dense<false, 2'
MM_READS, MM_OBS = r'k_pileup_mm<\w+, \w+, \w+, true>', r'k_pileup_mm<\w+, \w+, \w+, false>'


def summ(*dirs):
    return subprocess.check_output([sys.executable, 'tools/prof_summary.py', *dirs]).decode()


def stat_row(ks, pat):
    r = ks[ks['Name'].str.contains(pat, regex=True)]
    if not len(r):
        return None
    r = r.iloc[0]
    return int(r['Calls']), r['TotalDurationNs'] / r['Calls'] / 1e3, r['MinNs'] / 1e3, r['MaxNs'] / 1e3


ks = pd.read_csv(R + '/trace_c2/%s_kernel_stats.csv' % tag)
rd, ob = stat_row(ks, READS), stat_row(ks, OBS)
ro = j["roofline"]
ev, evs = ro["kernel_ms_avg"] * 1e3, ro.get("kernel_ms_in_stream", 0) * 1e3
obs_ev = j["roofline_observation_kernel"]["kernel_ms_avg"] * 1e3
open('profiles/%s_c2_kernel_stats.md' % tag, 'w').write(f'''# Round {rnd} — rocprofv3 --kernel-trace --stats, C2 workload (final kernels of the round)

Command (tools/make_profiles.sh): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-c5-leg --no-bam-leg`

**k_pileup_dense<false, 64, false>** (read-segment records, linkage off -- the kernel of the timed step): {rd[0]} calls, average {rd[1]:.1f} us,
min {rd[2]:.1f} us, max {rd[3]:.1f} us under rocprof.  The calls are of two kinds: 34 launches inside the streamed pipe (4 warm-up + 30 timed
batches; a launch is a few % of a PCIe-bound step, the GPU idles in between and its clocks sag; un-profiled bench: {evs:.1f} us each =
roofline.kernel_ms_in_stream) and 4 + 10 + 1 + 30 launches of the resident leg (un-profiled bench: {ev:.1f} us = roofline.kernel_ms_avg, the figure
the roofline is priced on; MinNs above is the same kernel at full clocks).
**k_pileup_dense<false, 2, true>** (2-byte observation records, the round-2 kernel, resident leg only): {ob[0]} calls, average {ob[1]:.1f} us
(bench: {obs_ev:.1f} us).

''' + summ(R + '/trace_c2'))

cf = pd.read_csv(R + '/pmc_fetch/%s_counter_collection.csv' % tag)
cw = pd.read_csv(R + '/pmc_write/%s_counter_collection.csv' % tag)


def mean(c, pat, name=None):
    r = c[c['Kernel_Name'].str.contains(pat, regex=True)]
    if name is not None:
        r = r[r['Counter_Name'] == name]
    return float(r['Counter_Value'].mean())


def traffic(pat):
    f, w = mean(cf, pat), mean(cw, pat)
    return f * 1024 * 2, w * 1024, f, w          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B


tr = {k: traffic(p) for k, p in (("reads", READS), ("obs", OBS), ("mm_reads", MM_READS), ("mm_obs", MM_OBS))}
alg = {"reads": ro["algorithmic_bytes_per_launch"], "obs": j["roofline_observation_kernel"]["algorithmic_bytes_per_launch"],
       "mm_reads": j["mm_on"]["reads"]["roofline"]["algorithmic_bytes_per_launch"],
       "mm_obs": j["mm_on"]["observations"]["roofline"]["algorithmic_bytes_per_launch"]}
label = {"reads": "k_pileup_dense<false, 64> (C2 as read segments, skip-mm; resident batch: counts + clonality out)",
         "obs": "k_pileup_dense<false, 2, true> (C2 as 2-byte observation records)",
         "mm_reads": "k_pileup_mm<..., SEGS> (C2 as read segments, mm profiling on, W = %d)" % j["mm_on"]["reads"]["roofline"]["window"],
         "mm_obs": "k_pileup_mm (C2 as 4-byte observation records, mm on)"}
lines = []
for k in ("reads", "obs", "mm_reads", "mm_obs"):
    r_, w_, _, _ = tr[k]
    lines.append(f"* {label[k]}: read {r_/1e6:.1f} MB + written {w_/1e6:.1f} MB = **{(r_+w_)/1e6:.1f} MB per launch** vs {alg[k]/1e6:.1f} MB algorithmic = {(r_+w_)/alg[k]:.2f}x.")
open('profiles/%s_c2_pmc.md' % tag, 'w').write(f'''# Round {rnd} — HBM traffic of the pileup kernels on C2 (separate --pmc passes over tools/pmc_target.py)

`rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_target.py` and the same with WRITE_SIZE: only resident C2 batches, 8 launches per kernel, every
launch of a kernel name the same work (the launches bench.py's roofline objects are priced on).  FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950
correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads, so
read bytes = FETCH_SIZE x 1024 x 2.  Algorithmic bytes as in DESIGN.md section 3 (64 B per read-segment record + 4 B per 16 of them, or
2 / 4 B per observation; 1 B/pos reference; 20 B/pos out, or 32 B per (position, mm) entry).

''' + "\n".join(lines) + "\n\n" + summ(R + '/pmc_fetch', R + '/pmc_write'))
json.dump({"c2_reads_bytes_per_launch": int(tr["reads"][0] + tr["reads"][1]), "c2_pileup_bytes_per_launch": int(tr["obs"][0] + tr["obs"][1]),
           "c2_mm_reads_bytes_per_launch": int(tr["mm_reads"][0] + tr["mm_reads"][1]),
           "c2_mm_pileup_bytes_per_launch": int(tr["mm_obs"][0] + tr["mm_obs"][1]),
           "fetch_size_kib": {k: v[2] for k, v in tr.items()}, "write_size_kib": {k: v[3] for k, v in tr.items()},
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/pmc_target.py (profiles/%s_c2_pmc.md); FETCH_SIZE doubled per the gfx950 correction" % tag,
           "round": rnd}, open('profiles/pmc_traffic.json', 'w'), indent=1)

# ---- SQ counters of the two one-bin kernels ----
sq = pd.concat([pd.read_csv(R + '/sq1/%s_counter_collection.csv' % tag), pd.read_csv(R + '/sq2/%s_counter_collection.csv' % tag)])
names = sorted(sq['Counter_Name'].unique())
rows = []
for n in names:
    a, b = mean(sq, READS, n), mean(sq, OBS, n)
    rows.append((n, a, b))
wc = {k: dict((n, v) for n, *_ in [] ) for k in ()}
wr = dict((n, a) for n, a, _ in rows)
wo = dict((n, b) for n, _, b in rows)
tab = "| counter | reads kernel | % of wave cycles | observation kernel | % of wave cycles |\n|:--|--:|--:|--:|--:|\n"
for n, a, b in rows:
    tab += "| %s | %.4g | %.1f | %.4g | %.1f |\n" % (n, a, 100 * a / wr["SQ_WAVE_CYCLES"], b, 100 * b / wo["SQ_WAVE_CYCLES"])
n_obs = j["config"]["kept_observations"]
open('profiles/%s_sq_counters.md' % tag, 'w').write(f'''# Round {rnd} — SQ counters of k_pileup_dense on C2: read-segment records vs 2-byte observation records

Two `rocprofv3 --pmc` passes of 8 counters over `tools/pmc_target.py --no-mm` (resident C2 batch, 8 launches per kernel; mean per launch; the
cycle counters are in quad-cycles summed over all waves, MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES).

{tab}
Reading (reads kernel, {ev:.1f} us alone; {n_obs/1e6:.0f} M kept bases = {n_obs/1e6:.0f} M LDS read-modify-writes per launch):
* LDS instructions: {wr["SQ_INSTS_LDS"]/1e6:.2f} M wave-level = {wr["SQ_INSTS_LDS"]*64/n_obs:.2f} lane slots per kept base (the skip slots of a record's last word and the
  lanes of short records are masked off); `SQ_LDS_BANK_CONFLICT` = {wr["SQ_LDS_BANK_CONFLICT"]/1e6:.1f} M cycles against `SQ_LDS_IDX_ACTIVE` = {wr["SQ_LDS_IDX_ACTIVE"]/1e6:.1f} M:
  **{100*wr["SQ_LDS_BANK_CONFLICT"]/wr["SQ_LDS_IDX_ACTIVE"]:.0f} % of the LDS pipe's busy cycles are bank-conflict replays** (observation kernel: {100*wo["SQ_LDS_BANK_CONFLICT"]/wo["SQ_LDS_IDX_ACTIVE"]:.0f} %).
  A wave's 64 atomics go to 16 records x 4 quarters at unrelated columns: the expected deepest bank is 3-4 of 64 lanes.
* VALU: {wr["SQ_INSTS_VALU"]/1e6:.1f} M wave-level instructions ({wr["SQ_INSTS_VALU"]*64/n_obs:.1f} lane-instructions per kept base) vs {wo["SQ_INSTS_VALU"]/1e6:.1f} M for the observation
  records; waves parked (`SQ_WAIT_ANY`) {100*wr["SQ_WAIT_ANY"]/wr["SQ_WAVE_CYCLES"]:.0f} % of their life vs {100*wo["SQ_WAIT_ANY"]/wo["SQ_WAVE_CYCLES"]:.0f} %: with 1 / 4.6 of the bytes to stream the loads no longer
  set the pace; issue stalls (`SQ_WAIT_INST_ANY`, {100*wr["SQ_WAIT_INST_ANY"]/wr["SQ_WAVE_CYCLES"]:.0f} %) and the LDS queue do.
* Consequence for the roofline: the kernel moves {ro["algorithmic_bytes_per_launch"]/1e6:.0f} MB in {ev:.1f} us = {ro["frac"]*100:.0f} % of the HBM roof -- not because it wastes traffic
  (measured traffic = {(tr["reads"][0]+tr["reads"][1])/ro["algorithmic_bytes_per_launch"]:.2f}x algorithmic) but because its bound moved from the stream to the LDS atomics:
  {j["roofline_lds"]["frac"]*100:.0f} % of the conflict-free ds_add rate (bench.py roofline_lds), ~3.3x conflict replays on top.
''')

l = j['linkage']
open('profiles/%s_linkage_kernel_stats.md' % tag, 'w').write(f'''# Round {rnd} — rocprofv3 --kernel-trace --stats, bench.py with the linkage leg (BASELINE configs[2]: 5 Mbp, 200x, 50 000 SNV sites)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-mm-leg --no-c5-leg --no-bam-leg`
(7 runs each of: the read-level batch through the sparse chain, the observation batch through the sparse chain, the observation batch through
the dense MFMA path; k_pileup_dense<true, ...> is the linkage-on pileup with the allele pass, <false, ...> the C2 steps of the same command.)

Un-profiled bench line of the same box: read-level {l["reads"]["snv_pairs_linked_per_s"]/1e6:.1f} M SNV pairs/s ({l["reads"]["ms_per_step"]:.2f} ms per step: {l["reads"]["kernel_ms"]}),
observations {l["sparse"]["snv_pairs_linked_per_s"]/1e6:.1f} M SNV pairs/s ({l["sparse"]["ms_per_step"]:.2f} ms); dense MFMA pass {l["dense_mfma"]["mfma"]["pass_ms"]:.3f} ms =
{l["dense_mfma"]["mfma"]["achieved_tops"]:.0f} int8 TOPS = {l["dense_mfma"]["mfma"]["utilisation"]*100:.1f} % of the 5 POPS dense peak (useful tiles only); see r03_mfma_crossover.md.

''' + summ(R + '/trace_linkage'))
c5 = j.get('c5', {})
open('profiles/%s_c5_kernel_stats.md' % tag, 'w').write(f'''# Round {rnd} — rocprofv3 --kernel-trace --stats, bench.py with the C5 leg (configs[4]: the whole 1000-genome database through one GPU)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-resident-leg --no-bam-leg`
The k_pileup_dense<true, 64> rows are the C5 batches (8 warm-up + {c5.get("roofline", {}).get("launches", "?")} timed; 25-40 Mbp of positions, ~0.75 M read segments each).
Un-profiled bench line of the same box: {c5.get("gbp_per_s", 0):.1f} Gbp/s for all 8 shards on one GPU ({c5.get("seconds", 0)*1e3:.0f} ms; stage totals {c5.get("stages_ms_total")}),
kernel total {c5.get("roofline", {}).get("kernel_ms_total", 0):.2f} ms = {c5.get("roofline", {}).get("frac", 0)*100:.0f} % of the HBM roof on {c5.get("roofline", {}).get("bytes_per_position", 0):.1f} algorithmic bytes per
position; the leg is bound by the copy-in queue ({c5.get("roofline_pcie", {}).get("host_to_device_bytes", 0)/1e9:.2f} GB over PCIe).

''' + summ(R + '/trace_c5'))
print(j["value"], j["ms_per_step"], ro["frac"], j["cpu_baseline"]["value"], c5.get("gbp_per_s"))
