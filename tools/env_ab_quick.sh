#!/bin/bash
# Same-box A/B of the generated ResNet-20 between ENVIRONMENT settings, kernel seconds per family only (one profiled run per setting):
#   usage (under gpurun): bash tools/env_ab_quick.sh <tag> "" "ACEHIP_X=1" ...
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
  env $cfg bash tools/prof_model.sh ${TAG}_e$i $IMAGES $BATCH > /dev/null 2>&1
  python3 - "$cfg" gpurun_out/${TAG}_e${i}_model_kernel_stats.csv >> $OUT <<'PY'
import csv, sys
fam = {}
for r in csv.DictReader(open(sys.argv[2])):
    n = r["Name"]
    f = next((k for k, keys in (("ntt", ("ntt8_", "ntt4_")), ("hw_batch_ew", ("hw_batch_ew",)), ("key_mac", ("key_mac",)), ("bsgs", ("bsgs_inner",)),
                                ("base_conv", ("base_conv",)), ("rotate", ("rotate",))) if any(x in n for x in keys)), "other")
    fam[f] = fam.get(f, 0.0) + float(r["TotalDurationNs"]) / 1e9
print("[%s]" % sys.argv[1], " ".join("%s %.3f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])), "sum %.3f" % sum(fam.values()))
PY
  echo "setting $i done"
done
cat $OUT
