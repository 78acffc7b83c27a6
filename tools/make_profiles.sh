#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh <tag>      -> gpurun_out/<tag>/{bench_n1.json, trace_*, pmc_*, sq*}
# Summaries are made afterwards with tools/write_profiles.py <tag> and committed under profiles/.
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
T="python $REPO/tools/pmc_target.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -o $TAG -- $B --steps 30 --warmup 4 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-c5-leg --no-bam-leg > $OUT/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_linkage -o $TAG -- $B --steps 3 --warmup 2 --no-cpu-baseline --no-mm-leg --no-c5-leg --no-bam-leg > $OUT/trace_linkage.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -o $TAG -- $B --steps 4 --warmup 1 --no-cpu-baseline --no-linkage-leg --no-mm-leg --no-resident-leg --no-bam-leg > $OUT/trace_c5.log 2>&1
# hardware counters: their own passes, no tracing, on launches that are all the same work (tools/pmc_target.py)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $T > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $T > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -o $TAG -- $T --no-mm > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/sq2 -o $TAG -- $T --no-mm > $OUT/sq2.log 2>&1
cd $REPO
# keep only the small CSVs (kernel stats / counter collection); traces are large
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
