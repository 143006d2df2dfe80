#!/bin/bash
# Throughput of the headline (three streams) with the launches of whole kernel families skipped (-DACEHIP_ABLATION build made on the box;
# $ACEHIP_ABLATE = bit mask of kernels.hpp AblateFamily; results are wrong by construction, only the time of what is left counts).
# 255 = no kernel at all: what the HOST side alone can issue.   usage (under gpurun): bash tools/ablate_bench.sh <tag> 0 1 255 ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
export ACEHIP_BENCH_NO_VERIFY=1
OUT=gpurun_out/${TAG}_ablate.txt
: > $OUT
exp_build "-DACEHIP_ABLATION" || { echo "build failed"; exit 1; }
for m in "$@"; do
  ACEHIP_ABLATE=$m timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-shard-leg --steps 2 --warmup 1 ${AB_ARGS:-} > gpurun_out/${TAG}_abl$m.json 2> gpurun_out/${TAG}_abl$m.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_abl$m.json'))
print('ABLATE $m:', 'images/s', d['value'], 'ms/step', d['ms_per_step'])" >> $OUT 2>&1
  echo "mask $m done"
done
exp_restore
cat $OUT
