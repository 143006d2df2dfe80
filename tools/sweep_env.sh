#!/bin/bash
# headline A/B over one environment variable: tools/sweep_env.sh <tag> <VAR> v1 v2 ...   (bench.py --no-cpu-baseline, SWEEP_STEPS steps)
# one JSON line per value under gpurun_out/<tag>_sweep.jsonl: images/s, ms per step, verified (digest of the first timed batch = the reference's)
tag=$1; var=$2; shift 2
out=gpurun_out/${tag}_sweep.jsonl
: > $out
for v in "$@"; do
  env $var=$v timeout -k 10 500 python bench.py --no-cpu-baseline --steps ${SWEEP_STEPS:-3} --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'$var': '$v', 'images_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'verified': d.get('verified'), 'roofline_frac': d['roofline']['frac']}))" >> $out || echo "{\"$var\": \"$v\", \"failed\": true}" >> $out
  tail -1 $out
done
