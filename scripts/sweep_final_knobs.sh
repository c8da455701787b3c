#!/bin/bash
# Cheap environment switches re-measured on the final round-3 tree (same box, two rounds, stops at the first failure).
for round in 1 2; do
for cfg in "" "DPA_BWD_STREAMS=2" "DPA_BN_EW_KB=32" "DPA_BN_EW_KB=8" "DPA_X6S_BLOCKS=768" "DPA_WGRAD_BLOCKS=768" "DPA_BN_BLOCKS=2048" "DPA_RESNET_DOWN_LAST=0"; do
  r=$(env $cfg timeout -k 10 200 python3 bench.py --no-secondary --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])") || exit 1
  echo "round $round [$cfg] $r"
done
done
