#!/usr/bin/env bash
# The reference's default workflow end to end on one GPU, synthetic digits (opt-in): train 2 epochs, resume for a third
# from the last checkpoint, test the best-validation model.  Stops at the first failing step.
# Usage: scripts/cli_train_resume_test.sh [fp32|bf16]
set -euo pipefail
cd "$(dirname "$0")/.."
ROOT=$PWD
DT=${1:-fp32}
W=$(mktemp -d)
cd $W
export DPA_SYNTHETIC_DATA=1 DPA_GPUS=0 DPA_COMPUTE_DTYPE=$DT DPA_MASTER_PORT=29651 PYTHONPATH=$ROOT
echo "# ---- train ----"
timeout -k 10 400 python $ROOT/main.py train -d $W/data -e 2 -b 64
echo "# ---- resume ----"
timeout -k 10 300 python $ROOT/main.py train -d $W/data -e 3 -b 64 -f rsl/checkpoint-mnist-resnet-001.pt.tar
echo "# ---- test ----"
timeout -k 10 200 python $ROOT/main.py test -d $W/data -b 64 -f rsl/bestmodel-mnist-resnet.pt.tar
rm -rf $W
