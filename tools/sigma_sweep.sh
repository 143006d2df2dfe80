#!/bin/bash
# GPU box: logits of a generated ResNet on synthetic weight files of growing sigma (tools/model_weights.py) -- picks SIGMA
#   bash tools/sigma_sweep.sh <key> <gen> <sigma>...      (gen: ih12 | numpy)
set -u
KEY=$1; GEN=$2; shift 2
for s in "$@"; do bash tools/try_weights.sh $KEY $s $GEN; done
