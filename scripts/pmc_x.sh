#!/bin/bash
# PMC probe of the contraction tiles on one shape (separate passes per counter group).  Usage: scripts/pmc_x.sh <tag> "<--only substr>" <cfgs>
TAG=${1:-pmcx}
ONLY=${2:-"linear ff2 L1280"}
CF=${3:-15,36,40,41}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
      python $GRAFT_REPO_ROOT/scripts/bench_kernels.py --only "$ONLY" --reps 3 --cfg-sweep --cfgs $CF > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/summary.log
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
run sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY
run tcc1 TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python scripts/pmc_report.py $OUT > $OUT/report.txt 2>&1
cat $OUT/summary.log; grep -A40 "conv_gemm" $OUT/report.txt | head -150
find $OUT -name "*.csv" -delete
