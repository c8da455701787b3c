#!/usr/bin/env bash
# 1/2/4/8-GPU weak-scaling sweep of the headline benchmark on one 8 x MI355X node (BASELINE.json configs 3-5).
# Usage: scripts/scaling.sh [extra bench.py flags]     e.g.  scripts/scaling.sh --bucket-mb 32
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps 30 --warmup 5 "$@"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus "$n" --steps 30 --warmup 5 "$@"
  fi
done
