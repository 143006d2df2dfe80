#!/bin/bash
# Whole-image HBM traffic from hardware counters: two separate rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE; MI355X_MICROARCH.md "rocprofv3 PMC slots") over the generated ResNet-20 program run for IMAGES images on
# one stream, aggregated per kernel on the GPU box (the raw csv is > 64 MiB).  usage (under gpurun): tools/pmc_image.sh <tag> [images]
set -u
TAG=${1:-pmcimg}
IMAGES=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
EXE=$ROOT/workloads/_gen/examples/model_resnet20_cifar10_pre
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp ACEHIP_RT_DATA_SYNTH=1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -- "$EXE" "$IMAGES" > "$OUT/run_$ctr.log" 2>&1
done
python3 - "$OUT" "$IMAGES" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, images = sys.argv[1], int(sys.argv[2])
res = defaultdict(lambda: defaultdict(float))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != ctr:
                continue
            k = row["Kernel_Name"].split("(")[0]
            res[k][ctr] += float(row["Counter_Value"])
            res[k]["dispatches_" + ctr] += 1
tot_f = sum(v["FETCH_SIZE"] for v in res.values()) * 1024 * 2   # KB -> bytes, gfx950 halves wide streaming reads
tot_w = sum(v["WRITE_SIZE"] for v in res.values()) * 1024
summary = {"note": "whole process: context creation (keys, bootstrap tables) + %d image(s); FETCH_SIZE/WRITE_SIZE are KB; FETCH doubled per the gfx950 "
                   "correction for wide streaming reads (MI355X_MICROARCH.md HBM section): an upper bound for narrow accesses" % images,
           "images": images, "hbm_read_bytes_corrected": tot_f, "hbm_write_bytes": tot_w,
           "kernels": {k: {"read_bytes_corrected": v["FETCH_SIZE"] * 2048, "write_bytes": v["WRITE_SIZE"] * 1024,
                           "dispatches": int(v["dispatches_FETCH_SIZE"])} for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] * 2 + kv[1]["WRITE_SIZE"]))}}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print("read %.1f GB  write %.1f GB" % (tot_f / 1e9, tot_w / 1e9))
for k, v in list(summary["kernels"].items())[:14]:
    print("%-60s R %8.1f GB  W %8.1f GB  n %d" % (k[:60], v["read_bytes_corrected"] / 1e9, v["write_bytes"] / 1e9, v["dispatches"]))
PY
