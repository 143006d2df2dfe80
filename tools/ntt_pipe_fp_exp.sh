#!/bin/bash
# Experiment: the pipelined NTT passes compiled for the FP class alone at 4 workgroups per CU (-DNTT_PIPE_FP_ONLY -DNTT_PIPE_WG=4; the integer
# classes do nothing: results wrong by construction), timed on the scaling-primes-only roofline batch.  Rebuilds on the GPU box, restores the
# product build at the end.  usage (under gpurun): bash tools/ntt_pipe_fp_exp.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_ntt_pipe_fp_exp.txt
: > $OUT
run() {  # label, env...
  local label=$1; shift
  for rep in 1 2; do
    env "$@" ACEHIP_BENCH_NO_VERIFY=1 python3 bench.py --roofline-only --roofline-batch scaling --no-cpu-baseline 2> gpurun_out/${TAG}_fp.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['other_batches']['scaling_primes_only']
print('$label', 'fwd_ms', r.get('launch_ms'), 'inv_ms', r.get('inverse_launch_ms'))" >> $OUT 2>&1
  done
}
run "product PIPE=0" ACEHIP_NTT_PIPE=0
for wg in 4 3; do
  exp_build "-DNTT_PIPE_T4=1 -DNTT_PIPE_FP_ONLY -DNTT_PIPE_WG=$wg" || echo "build failed" >> $OUT
  run "fp-only WG=$wg PIPE=2" ACEHIP_NTT_PIPE=2
  run "fp-only WG=$wg PIPE=4" ACEHIP_NTT_PIPE=4
done
exp_restore
cat $OUT
