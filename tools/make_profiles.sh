#!/bin/bash
# Regenerates the raw material of profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/make_profiles.sh <tag>      -> gpurun_out/<tag>/{bench.json, bench_detail.json, trace_*, pmc_*, sq*}
# Summaries are made afterwards with tools/write_profiles.py <tag> and committed under profiles/.
set -u
TAG=${1:-r06}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
python bench.py --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py"
T="python $REPO/tools/pmc_target.py"
# kernel time: the headline leg, and the resident launches the counters are collected on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -o $TAG -- $B --only-c5 --steps 3 --warmup 1 --detail /tmp/d.json > $OUT/trace_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -o $TAG -- $T > $OUT/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5b -o $TAG -- $T --c5 > $OUT/trace_c5b.log 2>&1
# hardware counters: their own passes, no tracing
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $T > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $T > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c5 -o $TAG -- $T --c5 > $OUT/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c5 -o $TAG -- $T --c5 > $OUT/pmc_write_c5.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq1 -o $TAG -- $T --no-mm > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/sq2 -o $TAG -- $T --no-mm > $OUT/sq2.log 2>&1
# ... and on the headline's average-size C5 batch in a lean pipe slot: the stripe path (round 6) and the per-position epilogue (ISX_LAYOUT_NO_STRIPES)
for L in 0 64; do
  ISX_BENCH_LAYOUT=$L rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq1_c5_$L -o $TAG -- $T --c5 > $OUT/sq1_c5_$L.log 2>&1
  ISX_BENCH_LAYOUT=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $OUT/sq2_c5_$L -o $TAG -- $T --c5 > $OUT/sq2_c5_$L.log 2>&1
  ISX_BENCH_LAYOUT=$L $T --c5 > $OUT/c5_batch_layout$L.log 2>&1
done
# the device inflate prototype next to zlib (tools/inflate_rate.py), and its lanes-per-wave sweep
python $REPO/tools/inflate_rate.py > $OUT/inflate_rate.log 2>&1
for l in 64 16 8; do ISX_INFLATE_LPW=$l DEVICE_ONLY=1 python $REPO/tools/inflate_rate.py 2>&1 | tail -1 | sed "s/^/LPW $l: /" >> $OUT/inflate_rate.log; done
rm -f /tmp/isx_inflate_probe.bam
cd $REPO
# keep only the small CSVs (kernel stats / counter collection); traces are large
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
