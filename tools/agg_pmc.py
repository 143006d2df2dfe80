#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV by kernel name: sum of the counter over all dispatches.
usage: agg_pmc.py <counter_collection.csv> <counter name> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

path, counter, out = sys.argv[1:4]
tot, cnt = defaultdict(float), defaultdict(int)
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
json.dump({"counter": counter, "kernels": {k: {"dispatches": cnt[k], "sum": tot[k]} for k in sorted(tot, key=lambda k: -tot[k])},
           "total": sum(tot.values())}, open(out, "w"), indent=1)
print(counter, "total", sum(tot.values()), "(raw counter units: FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 double FETCH_SIZE for wide streaming reads)")
