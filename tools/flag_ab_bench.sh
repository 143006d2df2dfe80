#!/bin/bash
# Same-box A/B of the HEADLINE (three streams) between builds of the kernel library with different compile-time flags (timing experiments
# such as -DNTT_EXP=1: results are wrong, so the runs are not verified).  usage (under gpurun): bash tools/flag_ab_bench.sh <tag> "" "-DNTT_EXP=1" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_flag_ab_bench.txt
: > $OUT
i=0
for flags in "$@"; do
  i=$((i + 1))
  exp_build "$flags" || { echo "[$flags] build failed" >> $OUT; continue; }
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-shard-leg --steps ${AB_STEPS:-2} --warmup 1 > gpurun_out/${TAG}_fab$i.json 2> gpurun_out/${TAG}_fab$i.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_fab$i.json'))
print('[$flags]', 'images/s', d['value'], 'ms/step', d['ms_per_step'])" >> $OUT 2>&1
  echo "variant $i done"
done
exp_restore
cat $OUT
