#!/bin/bash
# Collects the roofline kernel's counters on the GPU box: three separate rocprofv3 --pmc passes (the TCC block cannot
# hold FETCH_SIZE and WRITE_SIZE together; MI355X_MICROARCH.md "rocprofv3 PMC slots") over `bench.py --roofline-only --roofline-batch mix`,
# plus a --kernel-trace --stats pass.  Only the roofline object's own batch is run (--roofline-batch mix:
# the level-21 limb mix of the generated ResNet-20's parameter set), so every ntt8_* launch of the process has its size.  usage (under gpurun): tools/pmc_roofline.sh <tag>   -> gpurun_out/<tag>/summary.json
set -u
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --roofline-only --roofline-batch mix > "$OUT/bench_trace.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" --roofline-only --roofline-batch mix > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" --roofline-only --roofline-batch mix > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" --roofline-only --roofline-batch mix > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- python3 "$ROOT/bench.py" --roofline-only --roofline-batch mix > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(dict)
for sub in ("fetch", "write", "sq", "tcc"):
    for path in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        tot, cnt = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(path)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
        for (kern, ctr), v in tot.items():
            res[kern][ctr] = v / cnt[(kern, ctr)]
            res[kern]["dispatches_" + sub] = cnt[(kern, ctr)]
for path in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        res[row["Name"].split("(")[0]]["avg_us"] = float(row["AverageNs"]) / 1e3
        res[row["Name"].split("(")[0]]["min_us"] = float(row["MinNs"]) / 1e3
        res[row["Name"].split("(")[0]]["calls"] = int(row["Calls"])
for kern, d in res.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2   # KB; gfx950 reports 1/2 of wide streaming reads
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
json.dump({"note": "averages per dispatch; FETCH_SIZE/WRITE_SIZE in KB, FETCH doubled per the gfx950 correction; SQ_* quad-cycles summed over waves",
           "kernels": res}, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for kern, d in sorted(res.items()):
    if "ntt8" in kern:
        print(kern[:60], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if not k.startswith("dispatches")})
PY
