#!/bin/bash
# Runtime switches on the batch-64 zoo (CPU-side cost of the side stream / arena / fusions on launch-bound models).
set -e
M=${MODELS:-densenet,alexnet}
for cfg in "" "DPA_BWD_STREAMS=1" "DPA_ACT_ARENA=0" "DPA_BWD_STREAMS=1 DPA_ACT_ARENA=0" "DPA_FUSE_BN_BWD=0" "DPA_BWD_STREAMS=1 DPA_ACT_ARENA=0 DPA_FUSE_BN_BWD=0"; do
  echo "cfg: $cfg"
  env $cfg timeout -k 10 150 python3 scripts/bench_zoo.py --fp32-only --no-torch --steps 20 --models $M
done
