#!/bin/bash
# Runs a small example program N times and reports every abnormal exit with its stderr (round 6: one silent SIGSEGV of eg_rotate in a full GPU
# suite; the shim now writes a backtrace for such a death).  usage (under gpurun): bash tools/flake_loop.sh <tag> <program> <count>
set -u
TAG=$1; EXE=$2; N=${3:-100}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_flake_loop.txt
: > $OUT
bad=0
for i in $(seq 1 $N); do
  ACEHIP_SEED=20261004 ACEHIP_DUMP_OUTPUT=/tmp/flake_$i timeout -k 5 120 $EXE > /tmp/flake_out.txt 2> /tmp/flake_err.txt
  rc=$?
  rm -f /tmp/flake_$i.*
  if [ $rc -ne 0 ]; then
    bad=$((bad + 1))
    echo "== run $i: exit $rc" >> $OUT
    tail -5 /tmp/flake_out.txt >> $OUT
    tail -40 /tmp/flake_err.txt >> $OUT
  fi
  [ $((i % 25)) -eq 0 ] && echo "$i runs, $bad abnormal"
done
echo "$N runs of $EXE: $bad abnormal exits" >> $OUT
cat $OUT | tail -60
