#!/bin/bash
# L2 behaviour of the contraction kernel on the 3x3 conv 320->320 @64x64 (tile choice = library heuristic / AA_FORCE_CFG).
TAG=${1:-pmcc}
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
run tcc1 TCC_HIT_sum TCC_MISS_sum
run tcc2 FETCH_SIZE
run tcc3 WRITE_SIZE
python $GRAFT_REPO_ROOT/scripts/pmc_report.py $OUT > $OUT/report.txt 2>&1
cat $OUT/summary.log; cat $OUT/report.txt | head -40
