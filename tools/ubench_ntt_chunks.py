#!/usr/bin/env python3
"""Does the hand-off between the two passes of a transform stay on the die when the launch is small enough?

The forward NTT at N = 2^16 is two kernels; the first leaves every limb in memory for the second (in place).  One launch over the roofline
batch (1024 limbs = 512 MiB) is far larger than the 256 MiB Infinity Cache, so the hand-off goes through HBM: 2.02 x the algorithmic bytes.
This experiment transforms the SAME batch (scaling primes of the generated ResNet-20's parameter set, FP64 butterflies) as 1, 2, 4, 8, 16 and 32
launches over groups of polynomials -- both passes of a group before the next group -- and times the whole batch with HIP events; run it under
`rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` for the bytes.   usage: python3 tools/ubench_ntt_chunks.py [groups ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace_compiler_amd as A  # noqa: E402

N = 65536


def main():
    groups = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]
    rt = A.AceHip(N, 34, 51, 50, 3, device=0)
    level, pos0, n_limbs, n_p = 33, 1, 32, 32  # 32 polynomials x limbs 1..32 (50-bit scaling primes)
    words = n_limbs * N
    rng = np.random.default_rng(5)
    src = np.empty((n_limbs, N), dtype=np.uint64)
    for i in range(n_limbs):
        src[i] = rng.integers(0, rt.primes[pos0 + i], size=N, dtype=np.uint64)
    buf = rt.buf(n_p * words)
    for p in range(n_p):
        rt.check(rt.lib.acehip_memcpy_h2d(buf.at(p * words), src.ctypes.data, words * 8, None))
    base = buf.ptr - pos0 * N * 8
    out = []
    for g in groups:
        per = n_p // g

        def run(inverse, per=per, g=g):
            for k in range(g):
                rt.check(rt.lib.acehip_ntt_batch(rt.h, base + k * per * words * 8, words, per, level, pos0, n_limbs, inverse, None))

        for _ in range(2):
            run(0)
            run(1)
        tf = ti = 0.0
        reps = 6
        for _ in range(reps):
            tf += rt.time_ms(lambda: run(0), 1)
            ti += rt.time_ms(lambda: run(1), 1)
        got = np.empty_like(src)
        rt.check(rt.lib.acehip_memcpy_d2h(got.ctypes.data, buf.at((n_p - 1) * words), words * 8, None))
        assert np.array_equal(got, src), "round trip is not the identity"
        row = {"launch_groups": g, "MiB_per_group": per * words * 8 >> 20, "forward_ms": round(tf / reps, 4), "inverse_ms": round(ti / reps, 4),
               "frac_of_hbm_peak": round(16 * N * n_limbs * n_p / (tf / reps * 1e-3) / 1e9 / 8000.0, 4)}
        out.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
