#!/bin/bash
# HBM traffic of one eager denoising step per kernel (FETCH_SIZE / WRITE_SIZE in separate passes, see
# MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots").  Writes gpurun_out/<tag>/traffic_{fetch,write}/...csv
TAG=${1:-traffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# tile choices first (no profiler), so the profiled passes contain no autotuning launches
python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $OUT/tile_cache.json --launch-list $OUT/launch_list.json > $OUT/prep.log 2>&1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do  # (the profiled command makes the same three eager steps as the prep run: warm-up, timed, launch list)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-other-form --no-vae --tile-cache $OUT/tile_cache.json --launch-list $OUT/launch_list_$c.json > $OUT/$c.log 2>&1
  echo "$c rc=$?" >> $OUT/summary.log
done
python $GRAFT_REPO_ROOT/scripts/pmc_traffic_report.py $OUT > $OUT/traffic.json 2>$OUT/report.err
python $GRAFT_REPO_ROOT/scripts/pmc_traffic_report.py $OUT --by-shape $OUT/launch_list.json > $OUT/traffic_by_shape.json 2>>$OUT/report.err
cat $OUT/summary.log; cat $OUT/traffic.json
find $OUT -name "*kernel_trace.csv" -size +1M -delete; find $OUT -name "*.csv" -size +8M -delete
