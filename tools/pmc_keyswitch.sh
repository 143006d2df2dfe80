#!/bin/bash
# Counters of the kernels of acehip_key_switch at C3: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*) over
# `bench.py --workload keyswitch`, averaged per kernel over the dispatches of the key-switch loop.
# usage (under gpurun): tools/pmc_keyswitch.sh <tag>   -> gpurun_out/<tag>/summary.json
set -u
TAG=${1:-pmcks}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload keyswitch --no-cpu-baseline --steps 40 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(dict)
for sub in ("fetch", "write", "sq"):
    for path in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        tot, cnt = defaultdict(float), defaultdict(int)
        for row in csv.DictReader(open(path)):
            gx = int(row.get("Grid_Size_X", row.get("Grid_Size", "0")) or 0)
            k = (row["Kernel_Name"].split("(")[0] + " grid " + str(gx // 256), row["Counter_Name"])
            tot[k] += float(row["Counter_Value"])
            cnt[k] += 1
        for (kern, ctr), v in tot.items():
            res[kern][ctr] = v / cnt[(kern, ctr)]
            res[kern]["dispatches"] = cnt[(kern, ctr)]
for kern, d in res.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "SQ_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip; SQ_BUSY_CYCLES is summed over 32 shader engines:
        # VALU-issue share of the time the SIMDs exist = ACTIVE_INST_VALU * 4 / (BUSY_CYCLES / 32 * 1024 SIMDs)
        d["valu_issue_share"] = d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (d["SQ_BUSY_CYCLES"] / 32 * 1024)
json.dump({"note": "per kernel and grid size (workgroups of the x dimension), averages over dispatches; FETCH doubled (gfx950); valu_issue_share: "
                   "fraction of SIMD-cycles in which a VALU instruction issues, every instruction counted as 4 cycles (an upper estimate for VOP1/2)",
           "kernels": {k: v for k, v in sorted(res.items())}}, open(out + "/summary.json", "w"), indent=1)
for k, v in sorted(res.items()):
    if v.get("dispatches", 0) < 30:
        continue
    print("%-58s n %4d  R %7.1f MB  W %7.1f MB  VALU insts %9.0f  valu share %.2f" % (k.replace("void acehip::", "").replace("acehip::", "")[:58], v.get("dispatches", 0),
          v.get("hbm_read_bytes_corrected", 0) / 1e6, v.get("hbm_write_bytes", 0) / 1e6, v.get("SQ_INSTS_VALU", 0), v.get("valu_issue_share", 0)))
PY
