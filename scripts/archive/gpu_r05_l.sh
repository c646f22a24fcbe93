#!/bin/bash
OUT=gpurun_out/r05l; mkdir -p $OUT
python scripts/attn_tail_probe.py 2>&1 | tee $OUT/attn_tail_probe.txt
