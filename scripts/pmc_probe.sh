#!/bin/bash
# PMC probe of selected kernels (separate passes per counter group, kernel-trace only: see MI355X_MICROARCH.md).
# Usage: scripts/pmc_probe.sh <tag> "<bench_kernels --only substr>"
TAG=${1:-pmc}
ONLY=${2:-"conv3x3 L320"}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
      python $GRAFT_REPO_ROOT/scripts/bench_kernels.py --only "$ONLY" --reps 3 --cfg-sweep > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/summary.log
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
run tcc1 TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
find $OUT -name "*.csv" | head -20 >> $OUT/summary.log
cat $OUT/summary.log
