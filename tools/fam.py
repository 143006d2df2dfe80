#!/usr/bin/env python3
"""Kernel seconds per family from a rocprofv3 kernel_stats.csv:  python3 tools/fam.py <csv> [label]"""
import csv
import sys

fam, calls = {}, {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    f = next((k for k, keys in (("ntt", ("ntt8_", "ntt4_")), ("hw_batch_ew", ("hw_batch_ew",)), ("key_mac", ("key_mac",)), ("bsgs", ("bsgs_inner",)),
                                ("base_conv", ("base_conv",)), ("rotate", ("rotate",))) if any(x in n for x in keys)), "other")
    fam[f] = fam.get(f, 0.0) + float(r["TotalDurationNs"]) / 1e9
    calls[f] = calls.get(f, 0) + int(r["Calls"])
print("[%s]" % (sys.argv[2] if len(sys.argv) > 2 else ""), " ".join("%s %.3f (%d)" % (k, v, calls[k]) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])),
      "sum %.3f" % sum(fam.values()))
