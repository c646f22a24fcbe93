#!/bin/bash
# Matrix-pipe / VALU / LDS counters of one eager denoising step per kernel family (separate rocprofv3 --pmc passes, kernel-trace only:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Writes gpurun_out/<tag>/{sq1,sq2}/...csv and <tag>/step_sq.json
TAG=${1:-stepsq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $OUT/tile_cache.json > $OUT/prep.log 2>&1
cd /tmp
run() {
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $OUT/tile_cache.json > $OUT/$name.log 2>&1
  echo "$name rc=$?" >> $OUT/summary.log
}
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES
python $GRAFT_REPO_ROOT/scripts/pmc_step_sq_report.py $OUT > $OUT/step_sq.json 2>$OUT/report.err
cat $OUT/summary.log; cat $OUT/step_sq.json
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.csv" -size +8M -delete
