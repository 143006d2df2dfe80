#!/usr/bin/env python3
"""Fingerprint of the kernel / runtime sources: sha256 over the sorted files of ace-compiler_amd/csrc/** and include/** (ONE definition,
ace-compiler_amd/build.py source_fingerprint).  Both shared libraries embed it at build time (acehip_source_fingerprint(),
acehip_rt_source_fingerprint()) and binding.load_library() / Prepare_context compare; profiles/traffic.json records it at measurement
time and bench.py reports the counter-based figures as stale (null) when the sources have changed since -- a hand-maintained
"measured" number must not outlive the code it was measured on."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location("_acehip_build", os.path.join(ROOT, "ace-compiler_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def fingerprint(root=ROOT):
    assert os.path.samefile(root, ROOT), "the fingerprint is defined for this checkout"
    return _build_module().source_fingerprint()


if __name__ == "__main__":
    sys.stdout.write(fingerprint() + "\n")
