#!/usr/bin/env bash
# TCC write / read request mix of the streaming 1x1 kernel against conv_x6p.hip's kernel and a plain fill:
#   scripts/diag/pmc_x6s.sh   (on the GPU box; writes gpurun_out/pmc_x6s/summary.txt)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_x6s; mkdir -p $out
run() {   # tag, env assignment, args...
  tag=$1; envv=$2; shift 2
  i=0
  for ctrs in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
              "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_WRITE_sum TCC_WRITEBACK_sum GRBM_GUI_ACTIVE" \
              "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
    i=$((i+1))
    export $envv
    timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$tag/p$i -- python3 scripts/diag/one_x6s.py "$@" > $out/$tag.p$i.log 2>&1 || echo "pass $tag $i failed"
  done
}
run x6s DPA_X6S=1 256 fprop
run x6p DPA_X6S=0 256 fprop
run fill DPA_X6S=1 256 copy
python3 - "$out" <<'PY' > $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for tag in ("x6s", "x6p", "fill"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{out}/{tag}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            if not ("conv_x6" in k or "Fill" in k): continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(tag, k)
        for c, v in sorted(d.items()):
            print(f"   {c:40s} {v / cnt[(k, c)]:16.0f}")
PY
cat $out/summary.txt
