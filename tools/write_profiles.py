#!/usr/bin/env python3
"""gpurun_out/<tag>/ (made by tools/make_profiles.sh on the GPU box) -> profiles/<tag>_*.md, <tag>_bench_n1.json, pmc_traffic.json.
usage: python tools/write_profiles.py <tag>"""
import hashlib
import json
import os
import subprocess
import sys

import pandas as pd

tag = sys.argv[1]
R = 'gpurun_out/' + tag
rnd = int(tag[1:])
line = json.loads([l for l in open(R + '/bench.json') if l.startswith('{')][-1])
detail = json.load(open(R + '/bench_detail.json'))
json.dump({"line": line, "detail": detail}, open('profiles/%s_bench_n1.json' % tag, 'w'), indent=1)
commit = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD']).decode().strip()
sha = hashlib.sha1(open('instrain_amd/csrc/isx_pileup.hip', 'rb').read()).hexdigest()[:16]

K = {"delta": r'k_pileup_dense<false, 32, true>', "delta_u32": r'k_pileup_dense<false, 32, false>', "seg64": r'k_pileup_dense<false, 64, false>',
     "obs": r'k_pileup_dense<false, 2, true>', "mm_reads": r'k_pileup_mm<\w+, \w+, \w+, true, \w+, false>', "mm_delta": r'k_pileup_mm<\w+, \w+, \w+, true, \w+, true>',
     "mm_obs": r'k_pileup_mm<\w+, \w+, \w+, false',
     "c5": r'k_pileup_dense<true, 32, true>'}
LABEL = {"delta": "k_pileup_dense<false, 32, true> -- C2 as 32-byte reference-delta records, 16-bit LDS rows (production)",
         "delta_u32": "k_pileup_dense<false, 32, false> -- the same records, 32-bit LDS rows (very deep batches)",
         "seg64": "k_pileup_dense<false, 64, false> -- C2 as 64-byte segment records (round 3)",
         "obs": "k_pileup_dense<false, 2, true> -- C2 as 2-byte observation records (round 2)",
         "mm_reads": "k_pileup_mm<..., SEGS> -- C2 as segment records, mm profiling on", "mm_obs": "k_pileup_mm -- C2 as 4-byte observation records, mm on",
         "mm_delta": "k_pileup_mm<..., SEGS, ., DREC> -- C2 as reference-delta records with the mm level in the header (round 6), mm profiling on",
         "c5": "k_pileup_dense<true, 32, true> -- one C5 batch of the headline's average size (linkage on) in a pipe slot (shrunk output)"}


def summ(*dirs):
    return subprocess.check_output([sys.executable, 'tools/prof_summary.py', *dirs]).decode()


def stat_row(ks, pat):
    r = ks[ks['Name'].str.contains(pat, regex=True)]
    if not len(r):
        return None
    r = r.iloc[0]
    return int(r['Calls']), r['TotalDurationNs'] / r['Calls'] / 1e3, r['MinNs'] / 1e3, r['MaxNs'] / 1e3


def counter_mean(c, pat, name=None):
    r = c[c['Kernel_Name'].str.contains(pat, regex=True)]
    if name is not None:
        r = r[r['Counter_Name'] == name]
    return float(r['Counter_Value'].mean()) if len(r) else float('nan')


def traffic(cf, cw, pat):
    f, w = counter_mean(cf, pat), counter_mean(cw, pat)
    return f * 1024 * 2, w * 1024, f, w          # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section)


# ---- kernel stats ----
ks2 = pd.read_csv(R + '/trace_c2/%s_kernel_stats.csv' % tag)
ks5 = pd.read_csv(R + '/trace_c5b/%s_kernel_stats.csv' % tag)
rows = {k: stat_row(ks2, p) for k, p in K.items() if k != "c5"}
rows["c5"] = stat_row(ks5, K["c5"])
res = detail.get("resident", {})
rc2 = detail.get("roofline_c2_resident", {})
body = "\n".join("* **%s**: %d calls, average %.1f us, min %.1f us, max %.1f us" % ((LABEL[k],) + rows[k]) for k in K if rows.get(k))
open('profiles/%s_c2_kernel_stats.md' % tag, 'w').write(f'''# Round {rnd} -- rocprofv3 --kernel-trace --stats over the resident launches (tools/pmc_target.py)

Commands (tools/make_profiles.sh): `cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python tools/pmc_target.py` and
the same with `--c5`: only resident / re-submitted batches, 8 launches per kernel, every launch of a kernel name the same work -- the launches bench.py's roofline
objects are priced on (its `resident` leg times the same kernels with the dispatches' own time stamps: un-profiled {rc2.get("kernel_ms_avg", 0) * 1e3:.1f} us for the
reference-delta kernel).

{body}

''' + summ(R + '/trace_c2') + "\n" + summ(R + '/trace_c5b'))

c5 = detail["c5"]
open('profiles/%s_c5_kernel_stats.md' % tag, 'w').write(f'''# Round {rnd} -- rocprofv3 --kernel-trace --stats, bench.py headline (configs[4]: the whole 1000-genome database through one GPU)

Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python bench.py --only-c5 --steps 3 --warmup 1`
(verification pass + 1 warm-up + 3 timed passes of {c5["batches"]} batches each; the k_pileup_dense<true, 32, true> rows are those batches).
Un-profiled bench line of the same box: **{line["value"]:.1f} Gbp/s** ({line["ms_per_step"]:.1f} ms per pass; per pass: {c5["stages_ms_per_pass"]});
pileup kernel {c5["roofline"]["kernel_ms_per_pass"]:.1f} ms per pass on {c5["roofline"]["bytes_per_position"]:.2f} algorithmic bytes per position;
{c5["roofline_pcie"]["bytes_per_pass"] / 1e9:.2f} GB over PCIe per pass = {c5["roofline_pcie"]["bytes_per_profiled_base"]:.3f} B per profiled base.

''' + summ(R + '/trace_c5'))

# ---- PMC traffic ----
cf, cw = pd.read_csv(R + '/pmc_fetch/%s_counter_collection.csv' % tag), pd.read_csv(R + '/pmc_write/%s_counter_collection.csv' % tag)
cf5, cw5 = pd.read_csv(R + '/pmc_fetch_c5/%s_counter_collection.csv' % tag), pd.read_csv(R + '/pmc_write_c5/%s_counter_collection.csv' % tag)
tr = {k: traffic(cf, cw, p) for k, p in K.items() if k != "c5"}
tr["c5"] = traffic(cf5, cw5, K["c5"])
w = None
alg = {"delta": rc2.get("algorithmic_bytes_per_launch"), "obs": detail.get("roofline_observation_kernel", {}).get("algorithmic_bytes_per_launch"),
       "mm_reads": detail.get("mm_on", {}).get("reads", {}).get("roofline", {}).get("algorithmic_bytes_per_launch"),
       "mm_obs": detail.get("mm_on", {}).get("observations", {}).get("roofline", {}).get("algorithmic_bytes_per_launch"),
       "mm_delta": detail.get("mm_on", {}).get("reads_delta_records", {}).get("roofline", {}).get("algorithmic_bytes_per_launch")}
c5log = open(R + '/pmc_fetch_c5.log').read()
lines = []
for k in K:
    r_, w_, _, _ = tr[k]
    if r_ != r_:
        continue
    a = alg.get(k)
    lines.append(f"* {LABEL[k]}: read {r_/1e6:.1f} MB + written {w_/1e6:.1f} MB = **{(r_+w_)/1e6:.1f} MB per launch**" +
                 (f" vs {a/1e6:.1f} MB algorithmic = {(r_+w_)/a:.2f}x." if a else "."))
open('profiles/%s_c2_pmc.md' % tag, 'w').write(f'''# Round {rnd} -- HBM traffic of the pileup kernels (separate --pmc passes over tools/pmc_target.py)

`rocprofv3 --pmc FETCH_SIZE -- python tools/pmc_target.py [--c5]` and the same with WRITE_SIZE (separate passes, no tracing): 8 launches per kernel, every
launch of a kernel name the same work.  FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B
requests at 64 B for wide coalesced streaming reads, so read bytes = FETCH_SIZE x 1024 x 2.  Algorithmic bytes as in DESIGN.md section 3.
The C5 batch of the `--c5` pass: {[l for l in c5log.splitlines() if l.startswith("c5 batch")][-1] if "c5 batch" in c5log else "?"}

''' + "\n".join(lines) + "\n\n" + summ(R + '/pmc_fetch', R + '/pmc_write') + "\n" + summ(R + '/pmc_fetch_c5', R + '/pmc_write_c5'))
tot = lambda k: int(tr[k][0] + tr[k][1]) if tr[k][0] == tr[k][0] else None
json.dump({"c2_reads_bytes_per_launch": tot("delta"), "c2_delta_u32_bytes_per_launch": tot("delta_u32"), "c2_seg64_bytes_per_launch": tot("seg64"),
           "c2_pileup_bytes_per_launch": tot("obs"), "c2_mm_reads_bytes_per_launch": tot("mm_reads"), "c2_mm_pileup_bytes_per_launch": tot("mm_obs"), "c2_mm_delta_bytes_per_launch": tot("mm_delta"),
           "c5_dense_linkage_bytes_per_launch": tot("c5"),
           "fetch_size_kib": {k: v[2] for k, v in tr.items()}, "write_size_kib": {k: v[3] for k, v in tr.items()},
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/pmc_target.py [--c5] (profiles/%s_c2_pmc.md); FETCH_SIZE doubled per the "
                   "gfx950 correction.  bench.py withholds these figures when instrain_amd/csrc/isx_pileup.hip no longer hashes to kernel_source_sha" % tag,
           "round": rnd, "tag": tag, "commit": commit, "kernel_source_sha": sha}, open('profiles/pmc_traffic.json', 'w'), indent=1)

# ---- SQ counters: the reference-delta kernel next to the round-3 segment kernel ----
sq = pd.concat([pd.read_csv(R + '/sq1/%s_counter_collection.csv' % tag), pd.read_csv(R + '/sq2/%s_counter_collection.csv' % tag)])
names = sorted(sq['Counter_Name'].unique())
cols = ["delta", "seg64", "obs"]
val = {k: {n: counter_mean(sq, K[k], n) for n in names} for k in cols}
tab = "| counter | " + " | ".join("%s | %% of wave cycles" % k for k in cols) + " |\n|:--|" + "--:|--:|" * len(cols) + "\n"
for n in names:
    tab += "| %s | " % n + " | ".join("%.4g | %.1f" % (val[k][n], 100 * val[k][n] / val[k]["SQ_WAVE_CYCLES"]) for k in cols) + " |\n"
d, s6 = val["delta"], val["seg64"]
n_obs = detail.get("c2_stream", {}).get("kept_observations", 0)
open('profiles/%s_sq_counters.md' % tag, 'w').write(f'''# Round {rnd} -- SQ counters of k_pileup_dense on C2: reference-delta records (16-bit LDS rows) vs segment records vs observation records

Two `rocprofv3 --pmc` passes of 8 counters over `tools/pmc_target.py --no-mm` (resident C2 batch, 8 launches per kernel; mean per launch; the cycle counters
are in quad-cycles summed over all waves, MI355X_MICROARCH.md).  delta = k_pileup_dense<false, 32, true>, seg64 = <false, 64, false>, obs = <false, 2, true>.

{tab}
Reading:
* LDS instructions per launch: {d["SQ_INSTS_LDS"]/1e6:.2f} M (delta) vs {s6["SQ_INSTS_LDS"]/1e6:.2f} M (segments): the difference array replaces one `ds_add_u32` per kept base
  ({n_obs/1e6:.0f} M per launch) by +1 / -1 at a record's ends, one per skipped column and one per exception.
* `SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` = {100*d["SQ_LDS_BANK_CONFLICT"]/d["SQ_LDS_IDX_ACTIVE"]:.0f} % (delta) vs {100*s6["SQ_LDS_BANK_CONFLICT"]/s6["SQ_LDS_IDX_ACTIVE"]:.0f} % (segments);
  LDS pipe busy cycles {d["SQ_LDS_IDX_ACTIVE"]/1e6:.1f} M vs {s6["SQ_LDS_IDX_ACTIVE"]/1e6:.1f} M.
* waves parked (`SQ_WAIT_ANY`) {100*d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]:.0f} % of their life (segments: {100*s6["SQ_WAIT_ANY"]/s6["SQ_WAVE_CYCLES"]:.0f} %); VALU instructions {d["SQ_INSTS_VALU"]/1e6:.1f} M vs {s6["SQ_INSTS_VALU"]/1e6:.1f} M.
''')
# ---- SQ counters on the C5 batch: the stripe path against the per-position epilogue ----
import re
c5sq = {}
for L in (0, 64):
    try:
        c = pd.concat([pd.read_csv(R + '/sq1_c5_%d/%s_counter_collection.csv' % (L, tag)), pd.read_csv(R + '/sq2_c5_%d/%s_counter_collection.csv' % (L, tag))])
    except FileNotFoundError:
        continue
    c5sq[L] = {n: counter_mean(c, K["c5"], n) for n in sorted(c['Counter_Name'].unique())}
    lg = open(R + '/c5_batch_layout%d.log' % L).read()
    m = re.search(r"c5 batch \d+: (\d+) positions, (\d+) segments, (\d+) kept observations, kernel ([0-9.]+) ms", lg)
    if m:
        c5sq[L]["positions"], c5sq[L]["kernel_ms"] = int(m.group(1)), float(m.group(4))
if 0 in c5sq and 64 in c5sq:
    a_, b_ = c5sq[0], c5sq[64]
    names5 = [n for n in sorted(a_) if n.startswith("SQ_")]
    tab5 = "| counter | stripe path | per position | per-position epilogue (ISX_LAYOUT_NO_STRIPES) | per position |\n|:--|--:|--:|--:|--:|\n"
    for n in names5:
        tab5 += "| %s | %.4g | %.3g | %.4g | %.3g |\n" % (n, a_[n], a_[n] / a_["positions"], b_[n], b_[n] / b_["positions"])
    open('profiles/%s_sq_counters.md' % tag, 'a').write(f"""
## The headline's average-size C5 batch in a lean pipe slot (`tools/pmc_target.py --c5`): k_pileup_dense<true, 32, true>

{a_["positions"] / 1e6:.1f} M positions, linkage on; un-profiled kernel time of the same submit: **{a_["kernel_ms"]:.3f} ms with the stripe path**
(round 6: coverage straight from the difference row, a thread owns eight positions, only positions at min_cov go on) against {b_["kernel_ms"]:.3f} ms with the
per-position epilogue (`ISX_BENCH_LAYOUT=64` = `ISX_LAYOUT_NO_STRIPES`), same build.  Instruction counters are WAVE instructions per launch
(x 64 lanes for lane-instructions); "per position" divides by the batch's positions.

{tab5}
* VALU: **{a_["SQ_INSTS_VALU"] / a_["positions"]:.2f} wave instructions = {64 * a_["SQ_INSTS_VALU"] / a_["positions"]:.0f} lane-instructions per position** (per-position epilogue:
  {b_["SQ_INSTS_VALU"] / b_["positions"]:.2f} = {64 * b_["SQ_INSTS_VALU"] / b_["positions"]:.0f}); scalar {a_["SQ_INSTS_SALU"] / a_["positions"]:.2f} vs {b_["SQ_INSTS_SALU"] / b_["positions"]:.2f}; LDS {a_["SQ_INSTS_LDS"] / a_["positions"]:.3f} vs {b_["SQ_INSTS_LDS"] / b_["positions"]:.3f}.
* `SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES` x 8 waves a SIMD = {800 * a_["SQ_ACTIVE_INST_VALU"] / a_["SQ_WAVE_CYCLES"]:.0f} % of the SIMDs' VALU issue time (was {800 * b_["SQ_ACTIVE_INST_VALU"] / b_["SQ_WAVE_CYCLES"]:.0f} %);
  waves parked (`SQ_WAIT_ANY`) {100 * a_["SQ_WAIT_ANY"] / a_["SQ_WAVE_CYCLES"]:.0f} % of their life.
""")
# the VALU-issue yardstick of the resident C2 launch (bench.py roofline_c2_resident.valu_issue): quad-cycles in which a wave issued a VALU
# instruction, and the instruction count, per launch
pj = json.load(open('profiles/pmc_traffic.json'))
pj["c2_delta_sq"] = {n: d[n] for n in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY") if n in d}
json.dump(pj, open('profiles/pmc_traffic.json', 'w'), indent=1)
print(line["value"], line["ms_per_step"], line["roofline"], tot("delta"), tot("c5"))
