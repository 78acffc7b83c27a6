#!/usr/bin/env python3
"""gpurun_out/link/ (made by tools/link_prof.sh on the GPU box) -> profiles/<tag>_link_chain.md: the linkage stages' bucket chain next to the
sorted chain -- per-kernel times on C3 (resident batch) and inside the stream of one rank's C5 shard, launches per batch, the A/B lines.
usage: python tools/write_link_profile.py <tag>"""
import csv
import glob
import json
import sys

tag = sys.argv[1]
R = 'gpurun_out/link'


def stats(d):
    f = glob.glob('%s/%s/**/link_kernel_stats.csv' % (R, d), recursive=True)[0]
    return list(csv.DictReader(open(f)))


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if 'rocprim' in n:
        i = n.find('detail::')
        n = 'rocprim::' + n[i + 8:]
    return n.split('(')[0][:70]


def table(rows, keep=None, top=40):
    out = ['| kernel | calls | avg us | total ms |', '|:--|--:|--:|--:|']
    for r in rows[:top]:
        nm = short(r['Name'])
        if keep and not keep(nm):
            continue
        out.append('| %s | %s | %.1f | %.2f |' % (nm, r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
    return '\n'.join(out)


c3, c5b, c5s = stats('trace'), stats('trace_c5'), stats('trace_c5s')
ab = [l.rstrip() for l in open(R + '/ab.txt') if not l.startswith('{')]
res = json.loads([l for l in open(R + '/ab.txt') if l.startswith('{')][-1])
nb, ns = sum(int(r['Calls']) for r in c5b), sum(int(r['Calls']) for r in c5s)
pile = [r for r in c5b if 'k_pileup_dense<true, 32, true>' in r['Name']]
n_batches = int(pile[0]['Calls']) if pile else 100
link = ('k_link_prep', 'k_ao_chain', 'k_pair_walk', 'k_scan_u32', 'k_site_edges', 'k_edge_rows', 'k_ao_rank', 'k_pair_incr', 'k_ld_rows', 'k_site_split',
        'rocprim', 'k_copy_small', 'fillBuffer')
with open('profiles/%s_link_chain.md' % tag, 'w') as o:
    o.write('# Round %s -- the linkage stages: bucket chain (default) vs sorted chain (`ISX_LINK_CHAIN=sorted`)\n\n' % tag[1:])
    o.write('`tools/link_prof.sh` on one box: `rocprofv3 --kernel-trace --stats` over `tools/link_chain_ab.py` (C3 = BASELINE configs[2] as a resident read-level\n'
            'batch, both chains in one process; then one rank\'s C5 shard -- 4 batches a pass, pre-staged images replayed and handed over inside the step -- with\n'
            'each chain in its own process), then the un-profiled A/B.\n\n')
    o.write('## A/B (no profiler)\n\n```\n' + '\n'.join(ab) + '\n```\n\n')
    b, s = res['bucket'], res['sorted']
    o.write('C3: chain device time **%.2f ms (bucket) vs %.2f ms (sorted)**, step %.2f vs %.2f ms, **%.1f vs %.1f M SNV pairs linked/s**; '
            'same %d edges / %d LD rows / %d increments (`tests/test_gpu_link_chain.py`: byte-identical rows).\n\n'
            % (b['c3_chain_device_ms'], s['c3_chain_device_ms'], b['c3_ms_per_step'], s['c3_ms_per_step'], b['c3_snv_pairs_linked_per_s'] / 1e6,
               s['c3_snv_pairs_linked_per_s'] / 1e6, b['c3_edges'], b['c3_ld_rows'], b['c3_pair_increments']))
    o.write('## Launches per batch inside the C5 stream (everything the process launched / pileup launches)\n\n')
    o.write('| chain | launches | batches | per batch |\n|:--|--:|--:|--:|\n| bucket | %d | %d | **%.1f** |\n| sorted | %d | %d | %.1f |\n\n'
            % (nb, n_batches, nb / n_batches, ns, n_batches, ns / n_batches))
    o.write('### bucket chain, C5 shard (all kernels)\n\n' + table(c5b) + '\n\n')
    o.write('### sorted chain, C5 shard (top 30)\n\n' + table(c5s, top=30) + '\n\n')
    o.write('## C3 resident batch: the chains\' kernels (one process ran both chains; 14 launches each)\n\n' + table(c3, keep=lambda n: any(k in n for k in link), top=60) + '\n')
# the reference's default mode (mm on + linkage on) as a stream: tools/mm_launch_prof.sh
mm = glob.glob('gpurun_out/mmcount/trace/**/mm_kernel_stats.csv', recursive=True)
if mm:
    rows = list(csv.DictReader(open(mm[0])))
    nb = 20
    for l in open('gpurun_out/mmcount/run.log'):
        if l.startswith('BATCHES'):
            nb = int(l.split()[1])
    head = [l.rstrip() for l in open('gpurun_out/mmcount/run.log') if l.startswith('default mode')]
    tot = sum(int(r['Calls']) for r in rows)
    with open('profiles/%s_link_chain.md' % tag, 'a') as o:
        o.write('\n## The default mode (mm profiling on, linkage on) as a stream: launches per batch\n\n`tools/mm_launch_prof.sh` (`rocprofv3 --kernel-trace --stats -- python tools/mm_launch_count.py %d`).\n' % nb)
        o.write('%s\n\n**%d launches / %d batches = %.1f a batch** (29.4 while the site table of `k_pileup_mm` went through a device-wide sort in front of the chain; `k_site_order` puts it in position order window by window now -- 22.4 -- and one copy launch takes the level tables of a small batch home).\n\n' % ('\n'.join(head), tot, nb, tot / nb))
        o.write(table(rows, top=30) + '\n')
print('profiles/%s_link_chain.md' % tag)
