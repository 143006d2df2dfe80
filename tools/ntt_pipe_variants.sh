#!/bin/bash
# Round 6: the tile walk of the NTT passes under compile-time variants (workgroups per CU the walk kernels are compiled for, with / without the
# prefetch register set), strided pass only or both passes; roofline batches, ms per launch pair.  Rebuilds on the GPU box, restores the product build.
# usage (under gpurun): bash tools/ntt_pipe_variants.sh <tag> "<flags>" ["<flags>" ...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_ntt_pipe_variants.txt
: > $OUT
run() {
  local label=$1; shift
  env "$@" python3 bench.py --roofline-only --no-cpu-baseline 2> gpurun_out/${TAG}_v.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; o=r['other_batches']
print('$label', '| mix fwd %.4f inv %.4f | fp fwd %.4f inv %.4f | c3 fwd %.4f inv %.4f' % (r['launch_ms'], r['inverse_launch_ms'], o['scaling_primes_only']['launch_ms'], o['scaling_primes_only']['inverse_launch_ms'], o['c3_parameter_set']['launch_ms'], o['c3_parameter_set']['inverse_launch_ms']))" >> $OUT 2>&1
}
for flags in "$@"; do
  exp_build "$flags" || { echo "[$flags] build failed" >> $OUT; continue; }
  run "[$flags] one tile per workgroup      " ACEHIP_NTT_PIPE=0
  run "[$flags] walk, strided pass only     " ACEHIP_NTT_PIPE=2 ACEHIP_NTT_PIPE_CONTIG=0
  [ -n "${WITH_CONTIG:-}" ] && run "[$flags] walk, both passes           " ACEHIP_NTT_PIPE=2 ACEHIP_NTT_PIPE_CONTIG=1
  run "[$flags] walk, strided pass only (2) " ACEHIP_NTT_PIPE=2 ACEHIP_NTT_PIPE_CONTIG=0
  run "[$flags] one tile per workgroup (2)  " ACEHIP_NTT_PIPE=0
  echo "variant done"
done
exp_restore
cat $OUT
