#!/bin/bash
# Per-kernel summary of the generated ResNet-20 program on ONE stream:
#   gpurun -- 'bash tools/prof_model.sh <tag> <images> <batch>'  -> gpurun_out/<tag>_model_kernel_stats.csv
set -u
TAG=${1:-prof}; IMAGES=${2:-8}; BATCH=${3:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_model
export ACEHIP_RT_DATA_SYNTH=1 MODEL_BATCH=$BATCH
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_model -- "$ROOT/workloads/_gen/examples/model_resnet20_cifar10_pre" $IMAGES \
  > "$ROOT/gpurun_out/${TAG}_model.log" 2>&1
f=$(find /tmp/prof_model -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_model_kernel_stats.csv"
grep "MODEL" "$ROOT/gpurun_out/${TAG}_model.log" | tail -4
head -40 "$f"
