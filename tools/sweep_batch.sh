#!/bin/bash
# headline sweep: image streams x images per batch (bench.py --streams S --batch B); one JSON line per point
out=gpurun_out/$1_sweep.jsonl
: > $out
for cfg in "1 4" "1 8" "2 4" "2 8" "4 2" "4 4" "1 16" "2 16" "3 8"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu-baseline --streams $1 --batch $2 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({'streams': d['config']['streams_per_gpu'], 'batch': d['config']['images_per_batch'], 'images_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $out || echo "{\"cfg\": \"$cfg\", \"failed\": true}" >> $out
  tail -1 $out
done
