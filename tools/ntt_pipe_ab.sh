#!/bin/bash
# A/B of the pipelined NTT passes (ACEHIP_NTT_PIPE = tiles per workgroup; csrc/ntt_fast.hip) on the roofline batches, plus the N = 2^16
# parity tests with each form forced.  usage (under gpurun): bash tools/ntt_pipe_ab.sh <tag> [pipe values]   -> gpurun_out/<tag>_ntt_pipe_ab.txt
set -u
TAG=$1; shift
VALS=${*:-"0 2 4"}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_ntt_pipe_ab.txt
: > $OUT
for p in $VALS; do
  for rep in 1 2; do
    ACEHIP_NTT_PIPE=$p python3 bench.py --roofline-only --no-cpu-baseline 2> gpurun_out/${TAG}_pipe$p.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('PIPE=$p', 'mix fwd_ms', r.get('launch_ms'), 'inv_ms', r.get('inverse_launch_ms'), 'frac', r.get('frac'), '| other:', json.dumps(r.get('other_batches', r.get('batches', {})))[:600])" >> $OUT 2>&1
  done
  echo "variant $p timed"
done
for p in $VALS; do
  [ "$p" = "0" ] && continue
  ACEHIP_NTT_PIPE=$p ACEHIP_NTT_NARROW=0 timeout -k 10 600 python3 -m pytest -q -x -m gpu tests/test_gpu_parity.py tests/test_gpu_encode.py -k "(test_against_reference_golden and n65536) or test_fused_ntt_paths_n65536 or test_fp_class_ntt_n65536 or (test_encode_matches_reference and n65536) or test_full_size_properties" > gpurun_out/${TAG}_pipe${p}_parity.log 2>&1
  echo "PIPE=$p parity exit=$? $(tail -1 gpurun_out/${TAG}_pipe${p}_parity.log)" >> $OUT
done
cat $OUT
