#!/bin/bash
# GPU box: one image of a generated ResNet on the weight file of a given generator / sigma -> logits (is the network in range?)
#   bash tools/try_weights.sh <key> <sigma> <gen>
set -u
KEY=$1; SIGMA=$2; GEN=$3
EXE=model_$(python3 -c "import sys; sys.path.insert(0, 'tools'); import model_weights; print(model_weights.PROGRAM['$KEY'])")
f=$(python3 tools/model_weights.py $KEY $SIGMA $GEN | python3 -c "import sys,ast; print(ast.literal_eval(sys.stdin.read())[0])")
echo "== $KEY sigma $SIGMA gen $GEN: $f"
ACEHIP_SEED=20261004 MODEL_ENC_SEED=1000 ACEHIP_RT_DATA_FILE=$f MODEL_DATA_FILE=$f timeout -k 10 300 workloads/_gen/examples/$EXE 1 2>&1 | grep "logits9\|MODEL" | head -4
