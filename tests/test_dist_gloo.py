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
    print("rank", r.rank, "ok")
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
