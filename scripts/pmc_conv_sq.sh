#!/bin/bash
# SQ counters of the contraction kernel on the 3x3 conv 320->320 @64x64 (AA_FORCE_CFG picks the tile).
TAG=${1:-pmcs}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
      python $GRAFT_REPO_ROOT/scripts/bench_kernels.py --only "conv3x3 L320" --reps 2 > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/summary.log
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE
python $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT > $OUT/report.txt 2>&1
cat $OUT/summary.log; grep -A18 "Li256ELi320" $OUT/report.txt | head -60
