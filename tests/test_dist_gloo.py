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


SCHEDULE_WORKER = textwrap.dedent("""
    import sys
    sys.path.insert(0, %r)
    sys.path.insert(0, %r)
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    import ace_compiler_amd as A
    import _oracle as O
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, L, q0, sf, dnum, level = 16, 7, 60, 51, 3, 6
    o = O.Oracle(N, L, q0, sf, dnum)
    K = o.K
    rt = A.AceHip(N, L, q0, sf, dnum, host_only=True)   # the schedule comes from the product library (no GPU needed)
    owns = lambda gi: gi %% world == rank
    POISON = np.uint64(0xDEADBEEFDEADBEEF)

    def schedule(op, lvl):
        cap = 256
        st, pos, root = (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
        n = rt.lib.acehip_shard_schedule(rt.h, world, op, lvl, st, pos, root, cap)
        assert 0 < n <= cap
        return [(st[i], pos[i], root[i]) for i in range(n)]

    def exchange(buf, items):          # in-place broadcast of every listed limb from its root, like the RCCL exchange
        for _, pos, root in items:
            t = torch.from_numpy(buf[pos].view(np.int64))
            dist.broadcast(t, root)

    def mine_only(x, gis):             # a rank's view: the limbs it does not own hold garbage
        y = x.copy()
        for l, gi in enumerate(gis):
            if not owns(gi):
                y[l] = POISON
        return y

    def check_owned(got, want, gis, what):
        for l, gi in enumerate(gis):
            if owns(gi):
                assert np.array_equal(got[l], want[l]), (what, "limb", l, "rank", rank)

    mulmod = lambda a, b, q: (int(a) * int(b)) %% int(q)
    q_gis = list(range(level))
    ext_gis = q_gis + [L + j for j in range(K)]
    a_full = o.uniform(level, level, 3)

    # ---- ModUp of every digit (Decompose_modup polynomial.c:1241-1335): exchange = the coefficient-domain q-limbs ----
    a = mine_only(a_full, q_gis)
    coef = a.copy()
    for i in range(level):
        if owns(i):
            coef[i] = o.ntt_inv(a[i:i + 1], [i])[0]
    items = schedule(0, level)
    assert sorted(p for _, p, _ in items) == q_gis and all(r == p %% world for _, p, r in items)
    exchange(coef, items)
    for d in range(o.num_decomp(level)):
        n2, hat_inv, compl, hat_mod = o.modup_tables(level, d)
        start = o.alpha * d
        out = np.full((level + K, N), POISON, dtype=np.uint64)
        for i in range(n2):
            if owns(start + i):
                out[start + i] = a[start + i]                      # the digit's own limbs pass through
        for j, gi in enumerate(compl):
            gi = int(gi)
            if not owns(gi):
                continue
            t = o.primes[gi]
            row = np.empty(N, dtype=np.uint64)
            for n in range(N):
                acc = 0
                for i in range(n2):
                    y = mulmod(coef[start + i][n], hat_inv[i], o.primes[start + i])
                    acc += y * int(hat_mod[i][j])                  # exact sum, one reduction: the uncorrected fast base conversion
                row[n] = acc %% t
            pos = gi if gi < L else level + (gi - L)
            out[pos] = o.ntt_fwd(row[None, :], [gi])[0]
        check_owned(out, o.decomp_modup(a_full, level, d), ext_gis, "modup digit %%d" %% d)

    # ---- ModDown (Reduce_rns_base polynomial.c:928-967): exchange = the coefficient-domain P-limbs ----
    e_full = o.uniform(level + K, level, 5)
    e = mine_only(e_full, ext_gis)
    pc = e.copy()
    for j in range(K):
        if owns(L + j):
            pc[level + j] = o.ntt_inv(e[level + j:level + j + 1], [L + j])[0]
    items = schedule(1, level)
    assert sorted(p for _, p, _ in items) == [level + j for j in range(K)] and all(r == (L + p - level) %% world for _, p, r in items)
    exchange(pc, items)
    phat_inv = o.table("phat_inv_modp", K)
    phat_modq = o.table("phat_modq", L * K).reshape(L, K)
    pinv = o.table("pinv_modq", L)
    md = np.full((level, N), POISON, dtype=np.uint64)
    for i in range(level):
        if not owns(i):
            continue
        q = o.primes[i]
        row = np.empty(N, dtype=np.uint64)
        for n in range(N):
            acc = 0
            for j in range(K):
                acc += mulmod(pc[level + j][n], phat_inv[j], o.primes[L + j]) * int(phat_modq[i][j])
            row[n] = acc %% q
        conv = o.ntt_fwd(row[None, :], [i])[0]
        md[i] = np.array([mulmod((int(e[i][n]) - int(conv[n])) %% q, pinv[i], q) for n in range(N)], dtype=np.uint64)
    check_owned(md, o.mod_down(e_full, level), q_gis, "mod_down")

    # ---- Rescale (Rescale_poly polynomial.c:1097-1163): exchange = the last limb, from its owner ----
    items = schedule(2, level)
    assert items == [(0, level - 1, (level - 1) %% world)]
    last = a.copy()
    if owns(level - 1):
        last[level - 1] = o.ntt_inv(a[level - 1:level], [level - 1])[0]
    exchange(last, items)
    ql = o.primes[level - 1]
    qlql = o.table("qlql", L * L).reshape(L, L)[level - 2]
    ql_inv = o.table("ql_inv_modqi", L * L).reshape(L, L)[level - 2]
    rs = np.full((level - 1, N), POISON, dtype=np.uint64)
    for i in range(level - 1):
        if not owns(i):
            continue
        q = o.primes[i]
        t = np.empty(N, dtype=np.uint64)
        for n in range(N):
            t[n] = mulmod(o.lib.orc_switch_modulus(int(last[level - 1][n]), ql, q), qlql[i], q)
        t = o.ntt_fwd(t[None, :], [i])[0]
        rs[i] = np.array([(mulmod(a[i][n], ql_inv[i], q) + int(t[n])) %% q for n in range(N)], dtype=np.uint64)
    check_owned(rs, o.rescale(a_full, level), list(range(level - 1)), "rescale")

    # ---- ModRaise (ckks_bootstrap_context.c:1527-1551): limb 0 from rank 0 ----
    assert schedule(3, level) == [(0, 0, 0)]
    # ownership is a partition, and launches on a GPU-less context still fail loudly
    assert sum(rt.lib.acehip_shard_owned_limbs(rt.h, r) for r in range(1)) == L + K     # not sharded: one rank owns all
    rt.close()
    o.close()
    dist.barrier()
    dist.destroy_process_group()
    import os
    os.write(1, ("rank " + str(rank) + " schedule ok" + chr(10)).encode())
""")


def test_sharded_mode_exchange_schedule_two_ranks_gloo(tmp_path):
    """Limb-sharded execution as a mode of the context (include/acehip.h acehip_ctx_shard_*): the exchange schedule the pipelines
    follow (acehip_shard_schedule: which limb positions meet at ModUp, ModDown, Rescale, ModRaise, and who sends them) is
    sufficient and correct -- two gloo ranks, each holding ONLY its own limbs (the others poisoned), rebuild ModUp of every digit,
    ModDown and Rescale from the oracle's per-limb pieces, exchange exactly the listed limbs by broadcast from the listed root,
    and find their owned limbs of the oracle's unsharded results bit for bit."""
    script = tmp_path / "schedule_worker.py"
    script.write_text(SCHEDULE_WORKER % (ROOT, os.path.join(ROOT, "tests")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rank 0 schedule ok" in r.stdout and "rank 1 schedule ok" in r.stdout


LEG_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import ace_compiler_amd
    from ace_compiler_amd.dist import Ranks
    import bench
    r = Ranks(backend="gloo")
    out = bench.limb_sharded_leg(r, timeout_s=240)     # no GPU here: every child fails; the leg must report that, not raise or hang
    if r.rank == 0:
        assert out is not None and out["ranks_succeeded"] is False and "error" in out, out
    else:
        assert out is None
    r.barrier()
    r.close()
    os.write(1, ("rank " + str(r.rank) + " leg ok" + chr(10)).encode())
""")


def test_limb_sharded_leg_of_the_bench_survives_failing_children_two_ranks_gloo(tmp_path):
    """bench.py's secondary limb-sharded leg (multi-GPU runs) starts one child per rank and then agrees on the outcome with three
    reductions: with children that cannot run (no GPU in this container) both ranks must still pass the same collectives and rank 0
    must get an object that says so -- the headline line of a node run never depends on this leg."""
    script = tmp_path / "leg_worker.py"
    script.write_text(LEG_WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rank 0 leg ok" in r.stdout and "rank 1 leg ok" in r.stdout


def test_exchange_volume_of_a_key_switch_matches_the_survey():
    """at the generated ResNets' parameter set (N = 2^16, L = 34, K = 11) a key-switch at the top level moves the whole
    input once for ModUp (L x 512 KiB = 17 MiB) and the K P-limbs of both accumulators for ModDown (2K x 512 KiB = 11 MiB); a rank of an
    8-GPU group receives the 7/8 of them it does not own.  Counted from the product library's own exchange schedule (host-only context,
    no GPU): the positions listed are exactly those limbs, each from its owner gi % world."""
    import ctypes as C

    import ace_compiler_amd as A

    N, L, q0, sf, dnum, world = 65536, 34, 51, 50, 3, 8
    rt = A.AceHip(N, L, q0, sf, dnum, host_only=True)
    try:
        K, level, limb = rt.K, L, N * 8

        def schedule(op):
            cap = 256
            st, pos, root = (C.c_uint32 * cap)(), (C.c_uint32 * cap)(), (C.c_uint32 * cap)()
            n = rt.lib.acehip_shard_schedule(rt.h, world, op, level, st, pos, root, cap)
            assert 0 < n <= cap
            return [(pos[i], root[i]) for i in range(n)]

        up, down = schedule(0), schedule(1)
        assert K == 11 and [p for p, _ in up] == list(range(level)) and [r for _, r in up] == [i % world for i in range(level)]
        assert [p for p, _ in down] == list(range(level, level + K)) and [r for _, r in down] == [(L + j) % world for j in range(K)]
        assert len(up) * limb == 17 * 2 ** 20 and 2 * len(down) * limb == 11 * 2 ** 20
        for rank in range(world):  # what one rank receives per key-switch: the limbs it does not own (c0 and c1 of the ModDown pair)
            recv = sum(limb for _, r in up if r != rank) + 2 * sum(limb for _, r in down if r != rank)
            assert 23.5 * 2 ** 20 <= recv <= 25 * 2 ** 20, (rank, recv)
    finally:
        rt.close()
