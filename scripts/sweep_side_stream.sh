#!/bin/bash
# Where does the side-stream wgrad start to pay?  DPA_BWD_SIDE_MIN_GFLOP sweep over small- and large-kernel workloads.
set -e
for t in 0 20 40 80 1e9; do
  echo "SIDE_MIN_GFLOP=$t"
  DPA_BWD_SIDE_MIN_GFLOP=$t timeout -k 10 200 python3 scripts/bench_zoo.py --fp32-only --no-torch --steps 20 --models resnet,alexnet,vgg,squeezenet,densenet,inception
  DPA_BWD_SIDE_MIN_GFLOP=$t timeout -k 10 100 python3 scripts/bench_zoo.py --fp32-only --no-torch --steps 10 --batch 128 --models resnet50
  DPA_BWD_SIDE_MIN_GFLOP=$t timeout -k 10 100 python3 scripts/bench_zoo.py --fp32-only --no-torch --steps 10 --batch 512 --models resnet50
done
