#!/bin/bash
# Same-box A/B of the generated ResNet-20 between ENVIRONMENT settings (no rebuild): for every setting ("" = defaults) the program runs
# IMAGES images, BATCH per launch, on one stream -- once plainly (wall time of the images) and once under rocprofv3 --kernel-trace --stats
# (kernel seconds per family).   usage (under gpurun): bash tools/env_ab.sh <tag> "" "ACEHIP_HW_KEEP=0" ...
set -u
TAG=$1; shift
IMAGES=${AB_IMAGES:-24}; BATCH=${AB_BATCH:-12}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${TAG}_env_ab.txt
: > $OUT
i=0
for cfg in "$@"; do
  i=$((i + 1))
  wall=$(env $cfg ACEHIP_RT_DATA_SYNTH=1 MODEL_BATCH=$BATCH timeout -k 10 400 workloads/_gen/examples/model_resnet20_cifar10_pre $IMAGES 2>&1 | grep "MODEL\] total" | tail -1)
  env $cfg bash tools/prof_model.sh ${TAG}_e$i $IMAGES $BATCH > /dev/null 2>&1
  python3 - "$cfg" "$wall" gpurun_out/${TAG}_e${i}_model_kernel_stats.csv >> $OUT <<'PY'
import csv, sys
fam = {}
for r in csv.DictReader(open(sys.argv[3])):
    n = r["Name"]
    f = next((k for k, keys in (("ntt", ("ntt8_", "ntt4_")), ("hw_batch_ew", ("hw_batch_ew",)), ("key_mac", ("key_mac",)), ("bsgs", ("bsgs_inner",)),
                                ("base_conv", ("base_conv",)), ("rotate", ("rotate",))) if any(x in n for x in keys)), "other")
    fam[f] = fam.get(f, 0.0) + float(r["TotalDurationNs"]) / 1e9
print("[%s] %s |" % (sys.argv[1], sys.argv[2].strip()), " ".join("%s %.3f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])), "sum %.3f" % sum(fam.values()))
PY
  echo "setting $i done"
done
cat $OUT
