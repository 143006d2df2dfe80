#!/usr/bin/env python3
"""Does the intermediate of the two NTT passes stay in the Infinity Cache when a big batch is processed in chunks?
1024 limbs (512 MiB) as one launch pair vs k launch pairs of 1024/k limbs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ace_compiler_amd as A

N, L = 65536, 25
rt = A.AceHip(N, L, 60, 56, 4, device=0)
T = rt.L + rt.K
n_polys = 32
buf = rt.buf(n_polys * T * N)
rt.check(rt.lib.acehip_memset(buf.ptr, 0, buf.nbytes, None))
pw = T * N
for chunks in (1, 2, 4, 8, 16, 32):
    per = n_polys // chunks

    def fn():
        for k in range(chunks):
            rt.check(rt.lib.acehip_ntt_batch(rt.h, buf.at(k * per * pw), pw, per, L, 0, T, 0, None))

    fn()
    ms = rt.time_ms(fn, 10)
    print("%2d chunk(s) of %4d limbs (%5.0f MiB): %7.1f us  %7.1f GB/s algorithmic" % (chunks, per * T, per * T * 0.5, ms * 1e3, 16 * N * n_polys * T / ms / 1e6))
rt.close()
