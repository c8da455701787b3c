#!/bin/bash
# 256-row tile of the pre-split kernel (one block per CU) against the 128-row tile: per-layer table, then the step.
set -e
for t in 1 0 1 0; do
  echo "DPA_X6P_TALL=$t"
  DPA_X6P_TALL=$t timeout -k 10 200 python3 scripts/bench_conv_layers.py --dtype fp32 --batch 1024 > gpurun_out/tall_${t}_layers.txt 2>&1
  tail -1 gpurun_out/tall_${t}_layers.txt
done
for t in 1 0 1 0; do
  echo "DPA_X6P_TALL=$t"
  DPA_X6P_TALL=$t timeout -k 10 300 python3 bench.py --no-secondary --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
