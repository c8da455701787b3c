#!/usr/bin/env bash
# 1/2/4/8-GPU weak-scaling sweep of the headline benchmark on one 8 x MI355X node (BASELINE.json configs 3-5).
# Prints bench.py's JSON line for every N and, at the end, ONE table (images/s, ms/step, efficiency against N x the
# 1-GPU value, the communication facts bench.py recorded: transport ranks, preflight, in-sync check, bucket sizes) as
# JSON on stdout and in gpurun_out/scaling_table.json.  No scaling curve has been measured by the builder (one-GPU boxes).
# Usage: scripts/scaling.sh [extra bench.py flags]     e.g.  scripts/scaling.sh --bucket-mb 32
set -uo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
lines=gpurun_out/scaling_lines.jsonl
: > "$lines"
for n in ${DPA_SCALING_NS:-1 2 4 8}; do
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --steps 30 --warmup 5 --no-secondary "$@" | grep '^{' | tee -a "$lines"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus "$n" --steps 30 --warmup 5 "$@" | grep '^{' | tee -a "$lines"
  fi
done
python - "$lines" <<'PY' | tee gpurun_out/scaling_table.json
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = next((r["value"] for r in rows if r["n_gpus"] == 1), None)
table = []
for r in rows:
    c = r["config"]
    table.append({"n_gpus": r["n_gpus"], "images_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                  "efficiency": round(r["value"] / (base * r["n_gpus"]), 4) if base else None,
                  "comm": c.get("comm"), "comm_ranks": c.get("comm_ranks"), "preflight_ok": c.get("preflight_ok"),
                  "in_sync_after": c.get("in_sync_after"), "buckets_mb": c.get("buckets_mb")})
print(json.dumps({"scaling_table": table}))
PY
