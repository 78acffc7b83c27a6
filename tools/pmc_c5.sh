# SQ counters of k_pileup_dense<true, 32, true> on ONE C5 batch of the N = 1 average size through a lean pipe slot (tools/pmc_target.py --c5),
# stripe path (default) and the per-position epilogue (ISX_BENCH_LAYOUT=64): gpurun_out/pmc_c5/{stripe,nostripe}/p{1,2}
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in 0 64; do
  D=$OUT/layout$L; mkdir -p $D
  export ISX_BENCH_LAYOUT=$L
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $D/p1 -o p1 -- python $REPO/tools/pmc_target.py --c5 > $D/p1.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT --output-format csv -d $D/p2 -o p2 -- python $REPO/tools/pmc_target.py --c5 > $D/p2.log 2>&1
done
cd $REPO
find $OUT -name "*.db" -delete
python - <<'PY'
import csv, glob, collections
for L in (0, 64):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("gpurun_out/pmc_c5/layout%d/p*/**/*counter_collection.csv" % L, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pileup_dense" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("layout", L)
    for k in sorted(tot): print("  %-24s %.4g per launch (%d launches)" % (k, tot[k] / max(n[k], 1), n[k]))
PY
