#!/usr/bin/env python3
"""B independent C3 key-switches per launch set (BASELINE configs[2]: N=2^16, L=25, dnum=4) through the replica mechanism of
include/acehip.h (acehip_ctx_set_arena / acehip_ctx_select) -- the loop alone, so that a kernel trace of this program is the per-kernel
table of the batched key-switch:
    rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/ks_batched.py [B] [reps]        (tools/prof_keyswitch_batched.sh)
Prints one JSON line (ms per launch set, key-switches/s)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ace_compiler_amd as A  # noqa: E402
from ace_compiler_amd.binding import ArenaCfg  # noqa: E402

N, L, Q0, SF, DNUM = 65536, 25, 60, 56, 4
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rt = A.AceHip(N, L, Q0, SF, DNUM, device=0)
lib, h = rt.lib, rt.h
T = L + rt.K
rng = np.random.default_rng(1234)
host = np.empty((T, N), dtype=np.uint64)
for l in range(T):
    host[l] = rng.integers(0, rt.primes[l], size=N, dtype=np.uint64)
key = rt.buf(DNUM * 2 * T * N)
for d in range(DNUM * 2):
    rt.check(lib.acehip_memcpy_h2d(key.at(d * T * N), host.ctypes.data, T * N * 8, None))
lib.acehip_workspace_words.restype = C.c_size_t
gran = lambda w: (w + 31) // 32 * 32  # noqa: E731
off_sc = gran(lib.acehip_workspace_words(h))
off_a = off_sc + 2 * N
off_o0 = off_a + gran(L * N)
off_o1 = off_o0 + gran(L * N)
rep_words = off_o1 + gran(L * N)
arena = rt.buf(rep_words * B)
cfg = ArenaCfg(arena.ptr, rep_words * 8, rep_words * 8, B, arena.at(0), arena.at(off_sc), 2)
rt.check(lib.acehip_ctx_set_arena(h, C.byref(cfg)))
rt.check(lib.acehip_ctx_select(h, 0, B))
rt.check(lib.acehip_upload(h, arena.at(off_a), host.ctypes.data, L * N * 8, None))


def ks():
    rt.check(lib.acehip_key_switch(h, arena.at(off_o0), arena.at(off_o1), arena.at(off_a), key.ptr, L, None))


for _ in range(3):
    ks()
ms = rt.time_ms(ks, REPS)
print(json.dumps({"B": B, "ms_per_launch_set": round(ms, 4), "ms_per_key_switch": round(ms / B, 4), "key_switches_per_s": round(B * 1e3 / ms, 1)}))
rt.sync()
