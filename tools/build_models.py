#!/usr/bin/env python3
"""Compile the reference's checked-in ACE-generated model sources (rtlib/ant/dataset/*.onnx.inc), unchanged
and from where they lie under /root/reference, into workloads/_gen/models/libmodel_<name>.so against OUR headers
and OUR runtime (workloads/_gen/ is git-ignored; it holds generated callers only, no reference runtime code).
These libraries are the benchmark workload (BASELINE.json configs[3]/[4]); they only exist where
/root/reference exists (dev container) and travel to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/fhe-cmplr/rtlib/ant/dataset"
OUT = os.path.join(ROOT, "workloads", "_gen", "models")
MODELS = {"resnet20": "resnet20_cifar10_pre.onnx.inc", "resnet110": "resnet110_cifar10_train.onnx.inc"}


def build(verbose=False):
    if not os.path.isdir(REF):
        return []
    os.makedirs(OUT, exist_ok=True)
    built = []
    for name, inc in MODELS.items():
        src, out = os.path.join(REF, inc), os.path.join(OUT, "libmodel_%s.so" % name)
        newest = max(os.path.getmtime(os.path.join(ROOT, "tools", f)) for f in ("model_lib.c", "build_models.py"))
        if os.path.exists(out) and os.path.getmtime(out) >= newest:
            built.append(out)
            continue
        cmd = ["gcc", "-O1", "-w", "-fPIC", "-shared", os.path.join(ROOT, "tools", "model_lib.c"),
               "-DMODEL_INC=\"%s\"" % src, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "include", "rt_ant"),
               "-L", os.path.join(ROOT, "ace-compiler_amd", "lib"), "-lFHErt_ant", "-Wl,-rpath,$ORIGIN/../../../ace-compiler_amd/lib",
               "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(out)
    return built


if __name__ == "__main__":
    print(build(verbose=True))
