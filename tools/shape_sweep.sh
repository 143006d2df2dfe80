#!/bin/bash
# Headline under streams x images-per-launch shapes (same box).  usage (under gpurun): bash tools/shape_sweep.sh <tag> "3 12" "4 12" "3 15" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_shape_sweep.txt
: > $OUT
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --steps ${AB_STEPS:-2} --warmup 1 --streams $1 --batch $2 > gpurun_out/${TAG}_shape.json 2> gpurun_out/${TAG}_shape.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_shape.json'))
print('streams $1 x batch $2:', 'images/s', d['value'], 'ms/step', d['ms_per_step'], 'verified', d.get('verified'))" >> $OUT 2>&1
  echo "shape $cfg done"
done
cat $OUT
