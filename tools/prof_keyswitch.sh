#!/bin/bash
# Per-kernel breakdown of acehip_key_switch at C3 (N=2^16, L=25, dnum=4): rocprofv3 --kernel-trace --stats over
# `bench.py --workload keyswitch` (no image workload: every launch in the trace belongs to the key-switch loop, the NTT
# batch of the roofline leg or context creation).  usage (under gpurun): tools/prof_keyswitch.sh <tag> -> gpurun_out/<tag>/
set -u
TAG=${1:-ks}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --workload keyswitch --no-cpu-baseline --steps 200 --warmup 5 > "$OUT/bench_ks.json" 2> "$OUT/bench_ks.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for path in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    with open(out + "/kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r["Name"].split("(")[0], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
            print("%-70s calls %6s avg %9.1f us  min %8.1f us  %5s%%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Percentage"]))
PY
