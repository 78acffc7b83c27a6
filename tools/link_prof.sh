#!/bin/bash
# per-kernel times of the linkage chains (rocprofv3 --kernel-trace --stats) on C3 and on one rank's C5 shard, then the A/B of tools/link_chain_ab.py
REPO=$(pwd); OUT=$REPO/gpurun_out/link; mkdir -p $OUT
python -m pytest tests/test_gpu_link_chain.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o link -- python $REPO/tools/link_chain_ab.py --no-c5 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -o link -- python $REPO/tools/link_chain_ab.py --c5-only bucket > $OUT/trace_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5s -o link -- python $REPO/tools/link_chain_ab.py --c5-only sorted > $OUT/trace_c5s.log 2>&1
cd $REPO
for t in trace trace_c5 trace_c5s; do
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/link/$t/**/link_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("== $t")
tot=sum(int(r['Calls']) for r in rows)
print("launches", tot)
for r in rows[:24]:
    print("%-60s calls %6s avg %10.1f us  total %8.2f ms" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
python tools/link_chain_ab.py > $OUT/ab.txt 2> $OUT/ab.err; grep -v "^{" $OUT/ab.txt
