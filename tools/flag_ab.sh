#!/bin/bash
# Same-box A/B of the headline between builds of the kernel library with different compile-time flags (rebuilt on the GPU box;
# the product build is restored at the end).   usage (under gpurun): bash tools/flag_ab.sh <tag> "" "-DACEHIP_HW_LANES=4" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_flag_ab.txt
: > $OUT
i=0
for flags in "$@"; do
  i=$((i + 1))
  exp_build "$flags" || { echo "[$flags] build failed" >> $OUT; continue; }
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${TAG}_flag$i.json 2> gpurun_out/${TAG}_flag$i.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_flag$i.json'))
print('[$flags]', 'images/s', d['value'], 'ms/step', d['ms_per_step'])" >> $OUT 2>&1
  echo "variant $i done"
done
exp_restore
cat $OUT
