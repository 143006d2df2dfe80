#!/bin/bash
# Same-box A/B of per-family KERNEL time between builds of the kernel library with different compile-time flags: for every flag set the
# library is rebuilt on the GPU box and the generated ResNet-20 runs 24 images, 12 per launch, one stream, under rocprofv3 --kernel-trace
# --stats; prints seconds per family.  The product build is restored at the end.
#   usage (under gpurun): bash tools/kernel_ab.sh <tag> "" "-DACEHIP_CONV_MIN_WG=4" ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
. tools/exp_build.sh
OUT=gpurun_out/${TAG}_kernel_ab.txt
: > $OUT
i=0
for flags in "$@"; do
  i=$((i + 1))
  exp_build "$flags" || { echo "[$flags] build failed" >> $OUT; continue; }
  bash tools/prof_model.sh ${TAG}_v$i 24 12 > /dev/null 2>&1
  python3 - "$flags" gpurun_out/${TAG}_v${i}_model_kernel_stats.csv >> $OUT <<'PY'
import csv, sys
fam = {}
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"]
    f = next((k for k, keys in (("ntt", ("ntt8_", "ntt4_")), ("hw_batch_ew", ("hw_batch_ew",)), ("key_mac", ("key_mac",)), ("bsgs", ("bsgs_inner",)),
                                ("base_conv", ("base_conv",)), ("rotate", ("rotate",))) if any(x in n for x in keys)), "other")
    fam[f] = fam.get(f, 0.0) + float(r["TotalDurationNs"]) / 1e9
print("[%s]" % sys.argv[1], " ".join("%s %.3f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])), "sum %.3f" % sum(fam.values()))
PY
  echo "variant $i done"
done
exp_restore
cat $OUT
