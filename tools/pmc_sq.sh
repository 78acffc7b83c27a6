set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $OUT/sq_counters.txt
B="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-linkage-leg --no-c5-leg --no-bam-leg"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $OUT/p2 -o p2 -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_WAVE_CYCLES --output-format csv -d $OUT/p3 -o p3 -- $B > $OUT/p3.log 2>&1
cd $REPO
find $OUT -name "*.db" -delete
ls $OUT/*
