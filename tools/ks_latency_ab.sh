#!/bin/bash
# The single C3 key-switch (BASELINE configs[2]) and the 12-ciphertext launch set under ENVIRONMENT settings (e.g. the row count up to which
# N = 2^16 transforms run as narrow passes).  usage (under gpurun): bash tools/ks_latency_ab.sh <tag> "" "ACEHIP_NTT_NARROW=64" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_ks_latency_ab.txt
: > $OUT
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg python3 bench.py --workload keyswitch --no-cpu-baseline --steps 20 --warmup 3 2> gpurun_out/${TAG}_ks.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['key_switch']; b=k.get('batched') or {}
print('[$cfg]', 'single ms', k['ms'], 'batched ms per key-switch', b.get('ms_per_key_switch'), 'per_s', b.get('per_s'), 'equal', b.get('outputs_equal_single_operation'))" >> $OUT 2>&1
  done
done
cat $OUT
