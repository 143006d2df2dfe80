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
    import torch
    import torch.distributed as dist
    import ace_compiler_amd
    from ace_compiler_amd import shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L, K, level, N = 7, 3, 5, 4
    sh = shard.LimbShard(L, K, world, rank)
    comm = shard.TorchComm(dist, "cpu")
    # exchange 1: every rank contributes its owned q-limbs (filled with 100 + i), padded to the largest share
    pad = sh.max_q(level)
    local = torch.zeros((pad, N), dtype=torch.int64)
    for k, i in enumerate(sh.q_owned(level)):
        local[k] = 100 + i
    got = comm.all_gather_tensor(local)
    slot = shard.gather_slots([sh.q_owned(level, r) for r in range(world)], pad)
    full = torch.stack([got[slot[i]] for i in range(level)])
    assert full[:, 0].tolist() == [100 + i for i in range(level)], full
    # exchange 2: both accumulators' p-limbs, [comp][pad_p] per rank
    pad_p = sh.max_p()
    local = torch.zeros((2 * pad_p, N), dtype=torch.int64)
    for comp in range(2):
        for k, j in enumerate(sh.p_owned()):
            local[comp * pad_p + k] = 1000 * (comp + 1) + j
    got = comm.all_gather_tensor(local)
    pslot = shard.gather_slots([[(comp, j) for comp in range(2) for j in sh.p_owned(r) + [None] * (pad_p - len(sh.p_owned(r)))]
                                for r in range(world)], 2 * pad_p)
    for comp in range(2):
        for j in range(K):
            assert int(got[pslot[(comp, j)]][0]) == 1000 * (comp + 1) + j
    # every limb has exactly one owner
    owners = [[r for r in range(world) if gi in (sh.q_owned(L, r) + [L + j for j in sh.p_owned(r)])] for gi in range(L + K)]
    assert all(len(o) == 1 for o in owners), owners
    dist.barrier()
    dist.destroy_process_group()
    import os
    os.write(1, ("rank " + str(rank) + " shard ok" + chr(10)).encode())
""")


def test_limb_shard_exchange_layout_two_ranks_gloo(tmp_path):
    """the two all-gathers of the limb-sharded key-switch (ace-compiler_amd/shard.py) on CPU tensors: padded rank-major
    gather + slot map must reassemble the limbs in position order on every rank"""
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
