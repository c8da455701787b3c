#!/usr/bin/env bash
# Where the waves of the streaming 1x1 kernel spend their cycles (SQ counters), against conv_x6p.hip's kernel:
#   scripts/diag/pmc_x6s_sq.sh   (on the GPU box; writes gpurun_out/pmc_x6s_sq/summary.txt)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_x6s_sq; mkdir -p $out
run() {   # tag, env assignment
  tag=$1; envv=$2
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
              "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    export $envv
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$tag/p$i -- python3 scripts/diag/one_x6s.py 256 fprop > $out/$tag.p$i.log 2>&1 || echo "pass $tag $i failed"
  done
}
run x6s DPA_X6S=1
run x6p DPA_X6S=0
python3 - "$out" <<'PY' > $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("x6s", "x6p"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{tag}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            if "conv_x6" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(tag, k)
        for c, v in sorted(d.items()):
            print(f"   {c:40s} {v / cnt[(k, c)]:16.0f}")
PY
cat $out/summary.txt
