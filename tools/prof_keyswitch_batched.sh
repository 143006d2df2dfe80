#!/bin/bash
# Per-kernel table of B C3 key-switches per launch set (tools/ks_batched.py) under rocprofv3 --kernel-trace --stats.
#   usage (under gpurun): bash tools/prof_keyswitch_batched.sh <tag> [B]  -> gpurun_out/<tag>_ks_batched_kernel_stats.csv / .json
set -u
TAG=${1:-ksb}; B=${2:-12}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ksb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ksb -- python3 "$ROOT/tools/ks_batched.py" $B 100 > "$ROOT/gpurun_out/${TAG}_ks_batched.json" 2> "$ROOT/gpurun_out/${TAG}_ks_batched.err"
f=$(find /tmp/prof_ksb -name "*kernel_stats.csv" | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_ks_batched_kernel_stats.csv"
cat "$ROOT/gpurun_out/${TAG}_ks_batched.json"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
