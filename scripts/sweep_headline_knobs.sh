#!/bin/bash
# Re-measure the cheap environment switches on the headline step (same box, two rounds): settings that lost earlier in the
# round can win after later changes (the bf16 BatchNorm fusion did).
for round in 1 2; do
for cfg in "" "DPA_BWD_LAG=2" "DPA_BWD_STREAMS=1" "DPA_BN_EW_KB=8" "DPA_BN_EW_KB=32" "DPA_WGRAD_BLOCKS=768" "DPA_WGRAD_BLOCKS=384" "DPA_BN_BLOCKS=2048" "DPA_BWD_SIDE_MIN_GFLOP=150"; do
  r=$(env $cfg timeout -k 10 200 python3 bench.py --no-secondary --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])") || exit 1
  echo "round $round [$cfg] $r"
done
done
