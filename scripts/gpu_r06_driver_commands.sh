#!/bin/bash
# round 6, last call: the shipped tree with the driver's own commands - smoke, `python bench.py`, serial `pytest -m gpu`; + the SVD / RGBA bench lines with the final bench.py
OUT=gpurun_out/r06final; mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/summary.log
timeout 900 python bench.py > $OUT/bench.json 2>$OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --workload svd > $OUT/bench_svd.json 2>$OUT/bench_svd.err; echo "bench svd rc=$?" >> $OUT/summary.log
timeout 900 python bench.py --workload rgba > $OUT/bench_rgba.json 2>$OUT/bench_rgba.err; echo "bench rgba rc=$?" >> $OUT/summary.log
timeout 3000 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> $OUT/summary.log
cat $OUT/summary.log; tail -1 $OUT/smoke.log; tail -2 $OUT/gpu_tests.log; cut -c1-1500 $OUT/bench.json; cut -c1-600 $OUT/bench_svd.json
