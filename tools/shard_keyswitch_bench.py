#!/usr/bin/env python3
"""Latency of the limb-sharded key-switch + rescale (acehip_shard_* phases, ace-compiler_amd/shard.py RankRunner) on G GPUs
of one node, next to the fused single-GPU key-switch.  Also reachable as `bench.py --mode shard`.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P tools/shard_keyswitch_bench.py
Config C3 (N=2^16, L=25, dnum=4; BASELINE configs[2]).  Every rank holds only the limbs it owns (gi % G == rank) of the
input, the outputs and the switch key; the two all-gathers per key-switch and the broadcast per rescale go through RCCL on
torch's stream, the phases are launched on the same stream.  Rank 0 prints one JSON line.  With one GPU (G = 1) the
collectives degenerate to copies: that number is the overhead of the phase structure, not a scaling result."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    args, _ = ap.parse_known_args(argv)
    import numpy as np
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if "MASTER_ADDR" not in os.environ:
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import ace_compiler_amd as A  # after torch: its HIP runtime initialises first
    from ace_compiler_amd import shard

    N, L, Q0, SF, DNUM = 65536, 25, 60, 56, 4
    rt = A.AceHip(N, L, Q0, SF, DNUM, device=local)
    level = L
    comm = shard.TorchComm(dist, torch.device("cuda", local))
    rr = shard.RankRunner(rt, comm, rank)
    nq, n_own = rr.sh.num_q(level), rr.sh.num_q(L) + rt.lib.acehip_shard_num_p(rr.sh.h)
    rng = np.random.default_rng(7 + rank)
    gis = shard.owned_q(L, world, rank, L) + [L + j for j in shard.owned_p(L, rt.K, world, rank)]

    def rand_limbs(idx):
        return np.stack([rng.integers(0, rt.primes[g], size=N, dtype=np.uint64) for g in idx]) if idx else np.zeros((1, N), dtype=np.uint64)

    key = rt.to_device(np.stack([rand_limbs(gis) for _ in range(2 * DNUM)]))
    x, y = rt.to_device(rand_limbs(gis[:nq])), rt.to_device(rand_limbs(gis[:nq]))
    o0, o1 = rt.buf(max(nq, 1) * N), rt.buf(max(nq, 1) * N)
    keep = []

    def step():
        keep.append(rr.key_switch(x.ptr, key.ptr, o0.ptr, o1.ptr, level))
        rr.rescale(o0.ptr, o1.ptr, o0.ptr, o1.ptr, level)
        if len(keep) > 4:
            keep.pop(0)

    with rr.stream():  # launches and collectives on one explicit stream (not the default one: see RankRunner._stream)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) / args.steps * 1e3
    if rank == 0:
        exch = 8 * N * (rr.sh.pad_q(level) + 2 * rr.sh.pad_p + 2)
        print(json.dumps({
            "metric": "limb-sharded key-switch + rescale latency (N=2^16, L=25, dnum=4)", "value": round(ms, 4), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": False,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C3 key-switch + rescale of one ciphertext, RNS limbs sharded gi % world over the ranks; "
                                   "2 all-gathers + 1 broadcast per step through RCCL",
                       "owned_limbs_rank0": n_own, "exchange_bytes_sent_per_rank_per_step": exch,
                       "note": ("one rank: collectives are local copies; no multi-GPU hardware number exists for this path yet"
                                if world == 1 else "ranks on one node, RCCL over xGMI")}}))
    rr.close()
    rt.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
