#!/bin/bash
# GPU box: logits of a generated ResNet on synthetic weight files of growing sigma (tools/model_weights.py) -- picks SIGMA
#   tools/sigma_sweep.sh <resnet20|resnet32|resnet32c100|resnet44|resnet56|resnet110> sigma...
set -e
mkdir -p gpurun_out
KEY=$1; shift
EXE=model_$(python3 -c "import sys; sys.path.insert(0, 'tools'); import model_weights; print(model_weights.PROGRAM['$KEY'])")
for s in "$@"; do
  f=$(python3 tools/model_weights.py $KEY $s | python3 -c "import sys,ast; print(ast.literal_eval(sys.stdin.read())[0])")
  echo "$KEY sigma $s"
  ACEHIP_SEED=20261004 MODEL_ENC_SEED=1000 MODEL_DATA_FILE=$f ACEHIP_RT_DATA_FILE=$f workloads/_gen/examples/$EXE 1 2>&1 | grep -E "logits9|abort|error|Assert" || true
  rm -f $f
done
