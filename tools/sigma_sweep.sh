#!/bin/bash
# GPU box: logits of the generated ResNet-20 on synthetic weight files of growing sigma (tools/model_weights.py) -- picks SIGMA
set -e
mkdir -p gpurun_out
for s in "$@"; do
  f=$(python3 tools/model_weights.py resnet20 $s | python3 -c "import sys,ast; print(ast.literal_eval(sys.stdin.read())[0])")
  echo "sigma $s"
  ACEHIP_SEED=20261004 MODEL_ENC_SEED=1000 MODEL_DATA_FILE=$f ACEHIP_RT_DATA_FILE=$f workloads/_gen/examples/model_resnet20_cifar10_pre 1 2>&1 | grep -E "logits9|abort|error|Assert" || true
done
