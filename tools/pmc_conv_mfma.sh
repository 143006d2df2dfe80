#!/bin/bash
# Matrix-core counters of the base-conversion kernel inside one batch of the generated ResNet-20 program (one stream):
#   gpurun -- 'bash tools/pmc_conv_mfma.sh <tag> [batch]'  ->  gpurun_out/<tag>_conv_mfma_pmc.json
set -u
TAG=${1:-pmcmfma}; BATCH=${2:-12}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
EXE=$ROOT/workloads/_gen/examples/model_resnet20_cifar10_pre
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp ACEHIP_RT_DATA_SYNTH=1 MODEL_BATCH=$BATCH
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_mfma -- "$EXE" "$BATCH" > "$ROOT/gpurun_out/${TAG}_conv_mfma_run.log" 2>&1
python3 - "$ROOT/gpurun_out/${TAG}_conv_mfma_pmc.json" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tot, n = defaultdict(float), 0
for path in glob.glob("/tmp/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "base_conv_mfma" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        n += row["Counter_Name"] == "SQ_BUSY_CYCLES"
busy = tot.get("SQ_BUSY_CYCLES", 0.0)
simd_cycles = busy / 32 * 1024 if busy else 0.0
out = {"kernel": "base_conv_mfma_kernel", "dispatches": n, "counters": dict(tot),
       "valu_issue_share": (tot.get("SQ_ACTIVE_INST_VALU", 0) * 4 / simd_cycles) if simd_cycles else None,
       "mfma_busy_share": (tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cycles) if simd_cycles else None,
       "note": "sums over every dispatch of the kernel in one batch; SQ_BUSY_CYCLES is summed over 32 shader engines: SIMD-cycles = busy / 32 * 1024; "
               "shares are fractions of those SIMD-cycles"}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out)[:900])
PY
