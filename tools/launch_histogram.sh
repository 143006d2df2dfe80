#!/bin/bash
# Where the time of one image goes by launch size: rocprofv3 --kernel-trace over the generated ResNet-20 program (one stream,
# IMAGES images), then per kernel family a histogram of workgroups per launch with launch counts and summed durations.
# usage (under gpurun): tools/launch_histogram.sh <tag> [images]  -> gpurun_out/<tag>/histogram.txt
set -u
TAG=${1:-hist}
IMAGES=${2:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
EXE=$ROOT/workloads/_gen/examples/model_resnet20_cifar10_pre
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp ACEHIP_RT_DATA_SYNTH=1
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- "$EXE" "$IMAGES" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
rows = []
for path in glob.glob("/tmp/ktrace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void acehip::", "").replace("acehip::", ""),
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))))
rows.sort()
# keep the last image only: kernels after the last "sample_uniform"/encrypt burst are hard to find; use the second half by count
half = rows[len(rows) // 2:]
def fam(n):
    if n.startswith("ntt8_contig") or n.startswith("ntt8_strided") or n.startswith("ntt16"): return "ntt"
    return n.split("<")[0]
buckets = [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 1 << 30]
hist = defaultdict(lambda: [[0, 0.0] for _ in buckets])
for s, e, n, wg in half:
    f = fam(n)
    for i, b in enumerate(buckets):
        if wg <= b:
            hist[f][i][0] += 1
            hist[f][i][1] += (e - s) / 1e3
            break
span = (half[-1][1] - half[0][0]) / 1e6
busy = sum((e - s) for s, e, _, _ in half) / 1e6
with open(out + "/histogram.txt", "w") as f:
    f.write("# second half of the trace (%d launches, %.1f ms wall, %.1f ms of kernel time): launches and summed kernel time by workgroups per launch\n" % (len(half), span, busy))
    f.write("# family | <=16 | <=32 | <=64 | <=128 | <=256 | <=512 | <=1024 | <=2048 | <=4096 | more   (count / ms)\n")
    for k, h in sorted(hist.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        f.write("%-24s %s | total %d / %.1f ms\n" % (k, " | ".join("%d / %.1f" % (c, t / 1e3) for c, t in h), sum(c for c, _ in h), sum(t for _, t in h) / 1e3))
print(open(out + "/histogram.txt").read())
PY
