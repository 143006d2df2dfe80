#!/usr/bin/env python3
"""Latency of the limb-sharded key-switch (ace-compiler_amd/shard.py) on G GPUs of one node vs the single-GPU key-switch.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/shard_keyswitch_bench.py
Config C3 (N=2^16, L=25, dnum=4).  Every rank allocates only the key limbs it owns.  Rank 0 prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import torch.distributed as dist

rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
import ace_compiler_amd as A  # noqa: E402  (after torch: its HIP runtime initialises first)
from ace_compiler_amd import shard  # noqa: E402

N, L, Q0, SF, DNUM = 65536, 25, 60, 56, 4
rt = A.AceHip(N, L, Q0, SF, DNUM, device=local)
T, level = rt.L + rt.K, rt.L
sh = shard.LimbShard(rt.L, rt.K, world, rank)
rng = np.random.default_rng(7)
own_gi = sh.q_owned(level) + [rt.L + j for j in sh.p_owned()]
# owned key limbs only: [dnum][2][n_own][N]
key = rt.buf(DNUM * 2 * max(len(own_gi), 1) * N)
row = rng.integers(0, min(rt.primes), size=N, dtype=np.uint64)
for k in range(DNUM * 2 * max(len(own_gi), 1)):
    rt.check(rt.lib.acehip_memcpy_h2d(key.at(k * N), row.ctypes.data, N * 8, None))
idx = {gi: k for k, gi in enumerate(own_gi)}
x_own = rt.buf(max(len(sh.q_owned(level)), 1) * N)
for k in range(len(sh.q_owned(level))):
    rt.check(rt.lib.acehip_memcpy_h2d(x_own.at(k * N), row.ctypes.data, N * 8, None))
ks = shard.ShardedKeySwitch(rt, rank, world)
comm = shard.TorchComm(dist, torch.device("cuda", local))
key_limb = lambda d, comp, gi: key.at(((d * 2 + comp) * len(own_gi) + idx[gi]) * N)  # noqa: E731


def once():
    o0, o1 = shard.run_rank(ks, ks.run(level, x_own, key_limb), comm)
    o0.free()
    o1.free()


for _ in range(2):
    once()
dist.barrier()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    once()
rt.sync()
dist.barrier()
dt = (time.perf_counter() - t0) / reps
if rank == 0:
    print(json.dumps({"workload": "limb-sharded key-switch, C3 (N=2^16, L=25, dnum=4)", "n_gpus": world, "ms": round(dt * 1e3, 3),
                      "note": "reference implementation of the exchange schedule (per-limb launches, host-driven): measures the"
                              " two all-gathers + sharded compute; the single-GPU fused key-switch is 0.28 ms"}))
dist.destroy_process_group()
rt.close()
