REPO=$(pwd); OUT=$REPO/gpurun_out/mmcount; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o mm -- python $REPO/tools/mm_launch_count.py 20 > $OUT/run.log 2>&1
cd $REPO; tail -3 $OUT/run.log
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/mmcount/trace/**/mm_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(int(r['Calls']) for r in rows)
print("launches", tot, "per batch", tot/20.0)
for r in rows[:40]:
    print("%-70s calls %6s avg %9.1f us" % (r['Name'].replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
