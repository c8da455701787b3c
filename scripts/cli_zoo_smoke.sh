#!/usr/bin/env bash
# End-to-end CLI smoke of every zoo model on one GPU: 1 debug epoch (200 samples) + test from the best checkpoint.
# Usage: scripts/cli_zoo_smoke.sh [fp32|bf16]
set -uo pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
DT=${1:-fp32}
W=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from distributedpytorch_amd.data import mnist
mnist.generate_synthetic_mnist("$W/data", n_train=2000, n_test=300, seed=5)
PY
rc=0
for m in resnet alexnet vgg squeezenet densenet inception; do
  mkdir -p $W/$m && cd $W/$m
  if DPA_SYNTHETIC_DATA=1 DPA_GPUS=0 DPA_MODEL_NAME=$m DPA_COMPUTE_DTYPE=$DT DPA_MASTER_PORT=29641 PYTHONPATH=$ROOT timeout -k 10 300 python $ROOT/main.py train -d $W/data -e 1 -b 32 --debug > train.log 2>&1 \
     && DPA_SYNTHETIC_DATA=1 DPA_GPUS=0 DPA_COMPUTE_DTYPE=$DT DPA_MASTER_PORT=29641 PYTHONPATH=$ROOT timeout -k 10 300 python $ROOT/main.py test -d $W/data -b 50 -f rsl/bestmodel-mnist-$m.pt.tar > test.log 2>&1; then
    echo "$m $DT: $(grep -h 'Train  ' train.log | tail -1 | sed 's/ *| */ /g') | $(grep -h 'Acc:' test.log | tail -1)"
  else
    echo "$m $DT: FAILED"; tail -5 train.log test.log 2>/dev/null; rm -rf $W; exit 1      # (nothing more on the GPU after a failed step)
  fi
done
rm -rf $W
exit $rc
