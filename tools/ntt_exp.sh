#!/bin/bash
# Timing experiments on the forward NTT of the roofline batch with kernels that leave one ingredient out (-DNTT_EXP=..., results
# wrong by construction; ntt_fast.hip).  Rebuilds the library on the GPU box per variant, restores the product build at the end.
# usage (under gpurun): bash tools/ntt_exp.sh <tag> <exp> [<exp> ...]      -> gpurun_out/<tag>_ntt_exp.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_ntt_exp.txt
: > $OUT
for e in 0 "$@"; do
  exp_build "-DNTT_EXP=$e" || { echo "build failed for $e" >> $OUT; continue; }
  ACEHIP_BENCH_NO_VERIFY=1 python3 bench.py --roofline-only --no-cpu-baseline 2> gpurun_out/${TAG}_exp$e.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('NTT_EXP=$e', 'fwd_ms', r.get('launch_ms'), 'inv_ms', r.get('inverse_launch_ms'), 'achieved', r.get('achieved'), 'frac', r.get('frac'))" >> $OUT 2>&1
  echo "variant $e done"
done
exp_restore
cat $OUT
