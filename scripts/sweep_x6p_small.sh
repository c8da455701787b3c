#!/bin/bash
# Sweep of the pre-split kernel's eligibility knobs on the batch-64 zoo (reference default batch): one process per setting.
set -e
for cfg in "256 0" "256 512" "128 512" "96 512" "64 512" "128 0" "64 0"; do
  set -- $cfg
  echo "MIN_TILES=$1 NARROW_BELOW=$2"
  DPA_X6P_MIN_TILES=$1 DPA_X6P_NARROW_BELOW=$2 timeout -k 10 150 python3 scripts/bench_zoo.py --fp32-only --no-torch --steps 20 --models ${MODELS:-resnet,alexnet,vgg,squeezenet,densenet,inception}
done
