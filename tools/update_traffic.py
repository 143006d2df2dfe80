#!/usr/bin/env python3
"""Assembles gpurun_out/<tag>_traffic.json from the outputs of tools/measure_traffic.sh (see there)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import csrc_fingerprint  # noqa: E402

FAMILIES = [("ntt", ("ntt8_", "ntt4_", "ntt_pass")), ("hw_batch_ew", ("hw_batch_ew",)), ("key_inner_product", ("key_mac",)),
            ("bsgs_inner", ("bsgs_inner",)), ("base_conv", ("base_conv",)), ("rotate", ("rotate",)), ("embed", ("embed_inv",))]


def family(kernel):
    for name, keys in FAMILIES:
        if any(k in kernel for k in keys):
            return name
    return "other"


def main():
    tag, batch = sys.argv[1], int(sys.argv[2])
    out = os.path.join(ROOT, "gpurun_out")
    roof = json.load(open(os.path.join(out, tag + "_roofline", "summary.json")))["kernels"]
    ntt_bytes = 0.0
    for k, d in roof.items():
        if k.startswith("void acehip::ntt8_strided_kernel<false") or k.startswith("void acehip::ntt8_contig_kernel<false"):
            ntt_bytes += d.get("hbm_read_bytes_corrected", 0.0) + d.get("hbm_write_bytes", 0.0)
    s1 = json.load(open(os.path.join(out, tag + "_img1", "summary.json")))
    s2 = json.load(open(os.path.join(out, tag + "_img2", "summary.json")))
    tot = lambda s: s["hbm_read_bytes_corrected"] + s["hbm_write_bytes"]  # noqa: E731
    fam = {}
    for k, v in s2["kernels"].items():
        a = s1["kernels"].get(k, {"read_bytes_corrected": 0, "write_bytes": 0, "dispatches": 0})
        f = fam.setdefault(family(k), {"read_GB": 0.0, "write_GB": 0.0, "launches": 0.0})
        f["read_GB"] += (v["read_bytes_corrected"] - a["read_bytes_corrected"]) / batch / 1e9
        f["write_GB"] += (v["write_bytes"] - a["write_bytes"]) / batch / 1e9
        f["launches"] += (v["dispatches"] - a["dispatches"]) / batch

    def times(t):
        res = {}
        for row in csv.DictReader(open(os.path.join(out, t + "_model_kernel_stats.csv"))):
            res[family(row["Name"])] = res.get(family(row["Name"]), 0.0) + float(row["TotalDurationNs"])
        return res

    t1, t2 = times(tag + "_t1"), times(tag + "_t2")
    ksec = {k: (t2[k] - t1.get(k, 0.0)) / batch / 1e9 for k in t2}
    commit = os.popen("git -C %s rev-parse --short HEAD 2>/dev/null" % ROOT).read().strip() or None
    res = {
        "csrc_fingerprint": csrc_fingerprint.fingerprint(), "git_commit": commit, "images_per_batch": batch,
        "ntt_forward_bytes_per_launch": int(ntt_bytes),
        "ntt_source": "tools/pmc_roofline.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --roofline-only; FETCH_SIZE x2 "
                      "(gfx950 correction); strided + contiguous forward pass of the 1024-limb batch of the roofline object (--roofline-batch mix)",
        "resnet20_bytes_per_image": int((tot(s2) - tot(s1)) / batch),
        "resnet20_launches_per_image": round(sum(f["launches"] for f in fam.values()), 1),
        "resnet20_kernel_seconds_per_image": {k: round(v, 5) for k, v in sorted(ksec.items(), key=lambda kv: -kv[1])},
        "resnet20_families_per_image": {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in sorted(fam.items(), key=lambda kv: -(kv[1]["read_GB"] + kv[1]["write_GB"]))},
        "resnet20_source": "tools/measure_traffic.sh: workloads/_gen/examples/model_resnet20_cifar10_pre with MODEL_BATCH=%d, one stream, runs of %d and %d "
                           "images under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, every dispatch summed, FETCH x2) and under "
                           "--kernel-trace --stats; steady state per image = (second run - first run) / %d.  L2-miss traffic: Infinity-Cache hits are "
                           "counted" % (batch, batch, 2 * batch, batch),
    }
    path = os.path.join(out, tag + "_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("ntt_forward_bytes_per_launch", "resnet20_bytes_per_image", "resnet20_launches_per_image", "resnet20_kernel_seconds_per_image")}, indent=1))


if __name__ == "__main__":
    main()
