#!/bin/bash
# Same-box A/B of the headline: runs bench.py once per environment setting given as an argument ("" = defaults) and prints
# images/s, ms per step, NTT batch ms, key-switch ms for each.   usage (under gpurun): bash tools/ab_bench.sh <tag> "" "VAR=1" "VAR=2 OTHER=3"
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_ab.txt
: > $OUT
i=0
for cfg in "$@"; do
  i=$((i + 1))
  env $cfg timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps ${AB_STEPS:-2} --warmup 1 ${AB_ARGS:-} > gpurun_out/${TAG}_ab$i.json 2> gpurun_out/${TAG}_ab$i.err
  python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_ab$i.json'))
print('[$cfg]', 'images/s', d['value'], 'ms/step', d['ms_per_step'], 'ntt_ms', d['roofline']['launch_ms'], 'ks_ms', d.get('key_switch',{}).get('ms'))" >> $OUT 2>&1
  echo "config $i done"
done
cat $OUT
