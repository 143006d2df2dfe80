#!/bin/bash
# headline sweep: image streams x images per batch (bench.py --streams S --batch B); one JSON line per point
# usage: sweep_batch.sh <tag> "S B" "S B" ...   (extra environment is inherited)
tag=$1; shift
out=gpurun_out/${tag}_sweep.jsonl
: > $out
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu-baseline --streams $1 --batch $2 --steps ${SWEEP_STEPS:-2} --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read())
print(json.dumps({'streams': d['config']['streams_per_gpu'], 'batch': d['config']['images_per_batch'], 'images_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'env': {k: v for k, v in os.environ.items() if k.startswith('ACEHIP_')}}))" >> $out || echo "{\"cfg\": \"$cfg\", \"failed\": true}" >> $out
  tail -1 $out
done
