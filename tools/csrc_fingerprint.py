#!/usr/bin/env python3
"""Fingerprint of the kernel / runtime sources (sha256 over the sorted files of ace-compiler_amd/csrc and include/acehip.h).
profiles/traffic.json records it at measurement time; bench.py recomputes it and reports the counter-based figures as stale
(null) when the sources have changed since -- a hand-maintained "measured" number must not outlive the code it was measured on."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprint(root=ROOT):
    h = hashlib.sha256()
    files = [os.path.join(root, "include", "acehip.h")]
    for d, _, names in os.walk(os.path.join(root, "ace-compiler_amd", "csrc")):
        files += [os.path.join(d, n) for n in names if n.endswith((".hip", ".hpp", ".cpp", ".inc"))]
    for p in sorted(files):
        h.update(os.path.relpath(p, root).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    sys.stdout.write(fingerprint() + "\n")
