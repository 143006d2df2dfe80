#!/bin/bash
# Everything profiles/traffic.json is made of, in one gpurun call (counters in their own rocprofv3 --pmc passes, MI355X_MICROARCH.md):
#   * the roofline kernel (forward NTT over the resident 1024-limb batch): FETCH_SIZE / WRITE_SIZE per launch (tools/pmc_roofline.sh)
#   * the generated ResNet-20 program with BATCH images per launch, one stream: FETCH_SIZE / WRITE_SIZE and launches of runs with
#     BATCH and 2*BATCH images (tools/pmc_image.sh): steady state per image = (second - first) / BATCH
#   * kernel time per family of the same program (rocprofv3 --kernel-trace --stats, tools/prof_model.sh)
# usage (under gpurun): bash tools/measure_traffic.sh <tag> [batch]   ->  gpurun_out/<tag>_traffic.json (copy to profiles/traffic.json)
set -u
TAG=${1:-traffic}; BATCH=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/pmc_roofline.sh ${TAG}_roofline > gpurun_out/${TAG}_roofline.log 2>&1
echo "roofline counters done"
MODEL_BATCH=$BATCH bash tools/pmc_image.sh ${TAG}_img1 $BATCH > gpurun_out/${TAG}_img1.log 2>&1
echo "image counters (one batch) done"
MODEL_BATCH=$BATCH bash tools/pmc_image.sh ${TAG}_img2 $((2 * BATCH)) > gpurun_out/${TAG}_img2.log 2>&1
echo "image counters (two batches) done"
bash tools/prof_model.sh ${TAG}_t1 $BATCH $BATCH > /dev/null 2>&1
bash tools/prof_model.sh ${TAG}_t2 $((2 * BATCH)) $BATCH > /dev/null 2>&1
echo "kernel times done"
python3 tools/update_traffic.py $TAG $BATCH
