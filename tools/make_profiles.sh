#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh <tag>      -> gpurun_out/<tag>/{bench_n1.json, trace_c2, trace_linkage, pmc_fetch, pmc_write}
# Summaries are made afterwards with tools/prof_summary.py and committed under profiles/.
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -o $TAG -- $B --steps 30 --warmup 4 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-c5-leg --no-bam-leg > $OUT/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_linkage -o $TAG -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-mm-leg --no-c5-leg --no-bam-leg > $OUT/trace_linkage.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-linkage-leg --no-c5-leg --no-bam-leg > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-linkage-leg --no-c5-leg --no-bam-leg > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -o $TAG -- $B --steps 4 --warmup 1 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-resident-leg --no-bam-leg > $OUT/trace_c5.log 2>&1
cd $REPO
# keep only the small CSVs (kernel stats / counter collection); traces are large
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
