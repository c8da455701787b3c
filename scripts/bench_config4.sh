#!/bin/bash
# BASELINE.json config 4 (ResNet-18, CIFAR-shaped 32x32 input) and the headline config, eager vs hipGraph replay.
# Usage (on a GPU box): bash scripts/bench_config4.sh > gpurun_out/config4.jsonl
set -e
export DPA_BENCH_NO_FALLBACK=1
for dt in fp32 bf16; do
  for g in "" "--graph"; do
    timeout -k 10 200 python bench.py --worker --comm native --model resnet --shape cifar --batch 256 --dtype $dt --steps 30 --warmup 5 --no-secondary $g | grep '"metric"'
    timeout -k 10 200 python bench.py --worker --comm native --model resnet --shape cifar --batch 2048 --dtype $dt --steps 20 --warmup 5 --no-secondary $g | grep '"metric"'
  done
done
timeout -k 10 300 python bench.py --worker --comm native --steps 10 --warmup 3 --no-secondary --graph | grep '"metric"'
