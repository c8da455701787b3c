#!/usr/bin/env bash
# First contact with a multi-GPU node: ONE command that produces every communication table the bucket-size decision
# needs (VERDICT r2 missing #4) - for N = 2, 4, 8: all-reduce bus bandwidth of the native RCCL transport, of the IPC mesh
# transport and of torch.distributed's RCCL group, plus the ResNet-50 gradient cut into 4..128 MiB buckets.
# One JSON line per (N, transport) is appended to profiles/r3_allreduce_<host>.jsonl.  Every run starts with a
# known-answer preflight and is bounded by the collective watchdog (DPA_COLLECTIVE_TIMEOUT_S, default 120 s here).
# Usage: scripts/comm_first_contact.sh [N ...]        (default: 2 4 8)
set -uo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
export DPA_COLLECTIVE_TIMEOUT_S="${DPA_COLLECTIVE_TIMEOUT_S:-120}"
out="profiles/r3_allreduce_$(hostname -s 2>/dev/null || echo node).jsonl"
ns=("$@"); [ ${#ns[@]} -eq 0 ] && ns=(2 4 8)
port=29700
for n in "${ns[@]}"; do
  for tr in rccl mesh; do
    port=$((port + 20))
    extra=""; [ "$tr" = rccl ] && extra="--also-torch"
    echo "[first contact] N=$n transport=$tr" >&2
    timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
      scripts/allreduce_bench.py --transport "$tr" --sweep $extra | grep '^{' | tee -a "$out" \
      || echo "{\"world\": $n, \"transport\": \"$tr\", \"error\": \"run failed or timed out (see stderr)\"}" | tee -a "$out"
  done
done
echo "[first contact] tables in $out" >&2
