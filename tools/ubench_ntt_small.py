#!/usr/bin/env python3
"""NTT launch time vs number of limbs (N=2^16): where the fixed per-launch floor sits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ace_compiler_amd as A

N, L = 65536, 34
rt = A.AceHip(N, L, 51, 50, 3, device=0)
buf = rt.buf(64 * N)
rt.check(rt.lib.acehip_memset(buf.ptr, 0, buf.nbytes, None))
for inv in (0, 1):
    for n in (1, 2, 4, 8, 12, 16, 24, 34, 45):
        lvl = min(n, L)
        fn_ = rt.lib.acehip_ntt_inverse if inv else rt.lib.acehip_ntt_forward
        npos = min(n, L + rt.K)
        fn = lambda: rt.check(fn_(rt.h, buf.ptr, L, 0, npos if npos <= L else L, None))
        fn()
        ms = rt.time_ms(fn, 200)
        k = npos if npos <= L else L
        print("%s %2d limbs: %7.2f us  (%6.1f GB/s algorithmic)" % ("inv" if inv else "fwd", k, ms * 1e3, 16 * N * k / ms / 1e6))
rt.close()
