"""N>1 path of bench.py on CPU: world_size-2 gloo run of the replica glue (barrier, max-reduce of the timed
region, whole-job throughput, unit sharding)."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import ace_compiler_amd
    from ace_compiler_amd.dist import Ranks
    r = Ranks(backend="gloo")
    assert r.world == 2
    r.barrier()
    elapsed = 0.5 if r.rank == 0 else 2.0          # rank 1 is the slow one
    assert r.max_over_ranks(elapsed) == 2.0
    assert r.sum_over_ranks(3) == 6.0
    tp = r.aggregate_throughput(4, elapsed)         # 8 units / 2.0 s
    assert abs(tp - 4.0) < 1e-12, tp
    shards = list(r.shard(7))
    assert shards == ([0, 1, 2, 3] if r.rank == 0 else [4, 5, 6]), shards
    r.barrier()
    r.close()
    import os
    os.write(1, ("rank " + str(r.rank) + " ok" + chr(10)).encode())   # one write: lines of the two ranks cannot interleave
""")


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_single_rank_defaults():
    sys.path.insert(0, ROOT)
    import ace_compiler_amd  # noqa: F401
    from ace_compiler_amd.dist import Ranks

    env = {k: os.environ.pop(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE") if k in os.environ}
    try:
        r = Ranks()
        assert (r.rank, r.world) == (0, 1)
        assert r.max_over_ranks(1.5) == 1.5 and list(r.shard(5)) == [0, 1, 2, 3, 4]
        assert r.aggregate_throughput(10, 2.0) == 5.0
    finally:
        os.environ.update(env)


SHARD_WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    import ctypes as C
    import torch
    import torch.distributed as dist
    import ace_compiler_amd as A
    from ace_compiler_amd import shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # the partition and the exchange layout come from the C library (a host-only context: no GPU needed)
    rt = A.AceHip(8, 7, 60, 51, 3, host_only=True)
    L, K, level, N = rt.L, rt.K, 5, 8
    sh = rt.lib.acehip_shard_create(rt.h, rank, world)
    assert sh
    q = (C.c_uint32 * L)()
    p = (C.c_uint32 * max(K, 1))()
    nq = rt.lib.acehip_shard_owned(sh, level, q, p)
    q_own, p_own = list(q[:nq]), list(p[:rt.lib.acehip_shard_num_p(sh)])
    assert q_own == shard.owned_q(L, world, rank, level) and p_own == shard.owned_p(L, K, world, rank)
    assert nq == rt.lib.acehip_shard_num_q(sh, level)
    pad_q, pad_p = rt.lib.acehip_shard_pad_q(sh, level), rt.lib.acehip_shard_pad_p(sh)
    comm = shard.TorchComm(dist, "cpu")
    # exchange 1: every rank sends its owned q-limbs (filled with 100 + i) in pad_q slots; phase 2 of the library expects
    # limb i in slot (i mod world) * pad_q + i // world of the rank-major gather
    local = torch.zeros((pad_q, N), dtype=torch.int64)
    for k, i in enumerate(q_own):
        local[k] = 100 + i
    got = comm.all_gather(local)
    assert got.shape[0] == world * pad_q
    assert [int(got[(i %% world) * pad_q + i // world][0]) for i in range(level)] == [100 + i for i in range(level)]
    # exchange 2: [2][pad_p] per rank; phase 3 expects p-limb j of accumulator z at ((r*2 + z) * pad_p + k), r = (L+j) %% world,
    # k = its index among the p-limbs rank r owns
    local = torch.zeros((2 * pad_p, N), dtype=torch.int64)
    for z in range(2):
        for k, j in enumerate(p_own):
            local[z * pad_p + k] = 1000 * (z + 1) + j
    got = comm.all_gather(local)
    for z in range(2):
        for j in range(K):
            r = (L + j) %% world
            k = shard.owned_p(L, K, world, r).index(j)
            assert int(got[(r * 2 + z) * pad_p + k][0]) == 1000 * (z + 1) + j
    # rescale: the owner of the last limb broadcasts
    t = torch.full((2, N), 7 if rank == shard.owner(level - 1, world) else 0, dtype=torch.int64)
    comm.broadcast(t, shard.owner(level - 1, world))
    assert int(t[0][0]) == 7
    # every limb has exactly one owner; phases fail cleanly without a GPU
    owners = [sum(1 for r in range(world) if gi %% world == r) for gi in range(L + K)]
    assert all(n == 1 for n in owners)
    assert rt.lib.acehip_shard_ks_phase1(sh, 8, 8, level, None) < 0
    rt.lib.acehip_shard_destroy(sh)
    rt.close()
    dist.barrier()
    dist.destroy_process_group()
    import os
    os.write(1, ("rank " + str(rank) + " shard ok" + chr(10)).encode())
""")


def test_limb_shard_exchange_layout_two_ranks_gloo(tmp_path):
    """the exchanges of limb-sharded execution on CPU tensors with two gloo ranks: ownership lists and pad sizes from the C
    library (acehip_shard_* on a host-only context), padded rank-major all-gathers and the slot arithmetic phases 2 and 3
    use to reassemble the limbs, broadcast from the owner of the last limb for the rescale"""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rank 0 shard ok" in r.stdout and "rank 1 shard ok" in r.stdout
