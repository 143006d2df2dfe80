#!/bin/bash
# Instruction-issue counters per kernel family over one ResNet-20 image (one stream): rocprofv3 --pmc SQ_* over the generated
# program, aggregated on the GPU box.  usage (under gpurun): tools/pmc_image_sq.sh <tag> [images] -> gpurun_out/<tag>/summary.json
set -u
TAG=${1:-pmcsq}
IMAGES=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
EXE=$ROOT/workloads/_gen/examples/model_resnet20_cifar10_pre
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp ACEHIP_RT_DATA_SYNTH=1
rm -rf /tmp/pmc_sq
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_sq -- "$EXE" "$IMAGES" > "$OUT/run.log" 2>&1
python3 - "$OUT" "$IMAGES" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out, images = sys.argv[1], int(sys.argv[2])
res = defaultdict(lambda: defaultdict(float))
for path in glob.glob("/tmp/pmc_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void acehip::", "").replace("acehip::", "")
        fam = "ntt" if k.startswith("ntt") else k.split("<")[0]
        res[fam][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_BUSY_CYCLES":
            res[fam]["dispatches"] += 1
summ = {}
for fam, d in res.items():
    busy = d.get("SQ_BUSY_CYCLES", 0)
    share = d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (busy / 32 * 1024) if busy else 0
    summ[fam] = {"dispatches": int(d["dispatches"]), "valu_instructions": d.get("SQ_INSTS_VALU", 0), "valu_issue_share": round(share, 3),
                 "busy_cycles_per_se": busy / 32}
json.dump({"note": "whole process (context creation + %d image(s), one stream); valu_issue_share = SQ_ACTIVE_INST_VALU*4 / (SQ_BUSY_CYCLES/32 * 1024 SIMDs): "
                   "fraction of SIMD-cycles in which a VALU instruction issues while the family's kernels run (every instruction counted as 4 cycles)" % images,
           "families": dict(sorted(summ.items(), key=lambda kv: -kv[1]["busy_cycles_per_se"]))}, open(out + "/summary.json", "w"), indent=1)
for fam, v in sorted(summ.items(), key=lambda kv: -kv[1]["busy_cycles_per_se"])[:12]:
    print("%-28s n %7d  VALU share %.2f  busy Mcycles/SE %8.1f" % (fam, v["dispatches"], v["valu_issue_share"], v["busy_cycles_per_se"] / 1e6))
PY
