#!/bin/bash
# Per-kernel summary of the default bench command (profiles/*_bench_kernel_stats.csv, *_bench_under_rocprof.json):
#   gpurun -- 'bash tools/prof_bench.sh <tag>'   -> gpurun_out/<tag>_bench_kernel_stats.csv, gpurun_out/<tag>_bench_under_rocprof.json
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline \
  > "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.json" 2> "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.err"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv"
head -12 "$f"
tail -1 "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.json" | cut -c1-300
