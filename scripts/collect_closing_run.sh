#!/bin/bash
# Copy the judged summaries of a closing run (scripts/gpu_r06_final.sh <run>) from gpurun_out/<run>/ into profiles/<tag>_*  (usage: collect_closing_run.sh r06zz [tag])
RUN=$1; TAG=${2:-$1}; S=gpurun_out/$RUN; P=profiles
for n in bench bench_bf16 bench_rgba bench_svd; do cp $S/$n.json $P/${TAG}_$n.json; done
cp $S/gemm_breakdown.txt $P/${TAG}_gemm_breakdown.txt
cp $S/svd_gemm_breakdown.txt $P/${TAG}_svd_gemm_breakdown.txt
cp $S/graph_step_kernels.txt $P/${TAG}_graph_step_kernels.txt
cp $S/kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $S/parity_3steps.txt $P/${TAG}_parity_3steps_vs_oracle.txt
cp $S/parity_8steps.txt $P/${TAG}_parity_8steps_vs_oracle.txt
cp $S/sq/step_sq.json $P/${TAG}_pmc_step_sq.json
cp $S/traffic/traffic.json $P/${TAG}_traffic_pmc.json
cp $S/traffic/traffic_by_shape.json $P/${TAG}_traffic_by_shape.json
cp $S/smi.txt $P/${TAG}_clock_power_after_run.txt
{ echo "round 6 closing run (scripts/gpu_r06_final.sh $RUN), python -m pytest tests -m gpu -x -q -n 3:"; tail -2 $S/gpu_tests.log; echo; tail -1 $S/smoke.log; echo; cat $S/summary.log; } > $P/${TAG}_gpu_tests_summary.txt
{ echo "round 6 closing run ($RUN), ONE box, bench.py (hipGraph, 10 timed steps), each knob alternating with the default library state (ms per step, autotuned signatures):"
  for f in $S/ab_*.json; do python -c "
import json,sys,os; d=json.load(open('$f')); print('  %-90s %7.3f  %d' % (os.path.basename('$f')[3:-5], d['ms_per_step'], d['autotuned_signatures']))"; done; } > $P/${TAG}_step_ab_same_box.txt
cp $S/tile_cache.json animate_anything_amd/tile_cache_gfx950.json
ls $P/${TAG}_* | wc -l
