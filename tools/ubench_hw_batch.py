#!/usr/bin/env python3
"""Micro-benchmark of acehip_hw_batch on the shapes the generated ResNet code produces (N=2^16, L=34 dnum=3):
conv tap (2 MUL + 2 ADD per limb), plain add of two ciphertexts, key inner product loop.  Prints achieved GB/s
on the algorithmic bytes (24*N per ADD/MUL limb-op)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ace_compiler_amd as A
from ace_compiler_amd import binding as B

N, L = 65536, 34
rt = A.AceHip(N, L, 51, 50, 3, device=0)
T = rt.L + rt.K
arena = rt.buf(16 * T * N)
rt.check(rt.lib.acehip_memset(arena.ptr, 0, arena.nbytes, None))


def at(row, g):
    return arena.at((row * T + g) * N)


def run(name, prog, reps=50):
    arr = (B.HwOp * len(prog))(*[B.HwOp(o, g, r, a, b) for o, g, r, a, b in prog])
    fn = lambda: rt.check(rt.lib.acehip_hw_batch(rt.h, arr, len(prog), None))
    fn()
    ms = rt.time_ms(fn, reps)
    byts = sum(24 * N if o in (B.HW_ADD, B.HW_MUL, B.HW_SUB) else 32 * N if o == B.HW_MULADD else 16 * N for o, *_ in prog)
    print("%-46s %4d ops  %8.1f us  %7.1f GB/s (algorithmic)" % (name, len(prog), ms * 1e3, byts / ms / 1e6))


for level in (6, 15, 30):
    tap = []
    for g in range(level):
        tap += [(B.HW_MUL, g, at(4, g), at(0, g), at(2, g)), (B.HW_MUL, g, at(5, g), at(1, g), at(2, g)),
                (B.HW_ADD, g, at(6, g), at(6, g), at(4, g)), (B.HW_ADD, g, at(7, g), at(7, g), at(5, g))]
    run("conv tap level %d" % level, tap)
    add = []
    for g in range(level):
        add += [(B.HW_ADD, g, at(8, g), at(0, g), at(2, g)), (B.HW_ADD, g, at(9, g), at(1, g), at(3, g))]
    run("ct + ct level %d" % level, add)
    mac = []
    for d in range(3):
        for g in range(level + rt.K):
            gi = g if g < level else rt.L + g - level
            mac += [(B.HW_MUL, gi, at(4, g), at(d, g), at(10 + d, g)), (B.HW_ADD, gi, at(6, g), at(6, g), at(4, g)),
                    (B.HW_MUL, gi, at(5, g), at(d, g), at(13 + d, g)), (B.HW_ADD, gi, at(7, g), at(7, g), at(5, g))]
    run("key inner product (3 digits) level %d" % level, mac)
    one = [(B.HW_ADD, 0, at(8, 0), at(0, 0), at(2, 0))]
    run("single limb add", one)
rt.close()
