#!/bin/bash
# r04 call C: lean K walk + Upsample2D as four parity convolutions: kernel parity, full-size UNet parity, bench + breakdown, VAE
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_kernels.py -m gpu -q -x -n 4 > $OUT/test_kernels.log 2>&1; echo "test_kernels rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unet.py tests/test_gpu_pipeline.py -m gpu -q -x > $OUT/test_full.log 2>&1; echo "test_full rc=$?" >> $OUT/summary.log
tail -3 $OUT/test_full.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-form --gemm-breakdown $OUT/gemm_breakdown.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/summary.log
head -c 600 $OUT/bench.json; echo
timeout 300 python scripts/bench_vae.py > $OUT/vae.log 2>&1; tail -4 $OUT/vae.log
cat $OUT/summary.log
