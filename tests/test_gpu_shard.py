"""Limb-sharded execution (BASELINE configs[4]) through the C-ABI phases acehip_shard_* (csrc/api_shard.cpp, shard.hip):
`world` simulated ranks on one GPU, limb gi on rank gi % world, exchanges replaced by device copies
(ace-compiler_amd/shard.py LocalWorld).  Assembling the ranks' owned output limbs must reproduce the unsharded oracle
results bit for bit -- key-switch (two all-gathers), rescale (broadcast of the last limb) and encode (broadcast message)
-- for every world size including ones that leave ranks without p-limbs or q-limbs."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

CFGS = [(64, 7, 60, 51, 3, [7, 4]), (4096, 6, 60, 50, 3, [6, 5, 2]), (65536, 5, 60, 56, 2, [5, 3])]


@pytest.mark.parametrize("cfg", CFGS, ids=["n64", "n4096", "n65536"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_key_switch_rescale_encode_match_oracle(cfg, world):
    import ace_compiler_amd as A
    from ace_compiler_amd import shard

    N, L, q0, sf, dnum, levels = cfg
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    lw = shard.LocalWorld(rt, world)
    try:
        key = np.ascontiguousarray(o.make_key(3000)).reshape(o.dnum, 2, L + o.K, N)
        key_own = [rt.to_device(shard.pack_key(key, L, o.K, world, r)) for r in range(world)]
        for level in levels:
            # key-switch
            x = o.uniform(level, level, 81 + level)
            e0, e1 = o.key_switch(x, key, level)
            x_own = lw.split(x, level)
            out0 = [rt.buf(max(s.num_q(level), 1) * N) for s in lw.shards]
            out1 = [rt.buf(max(s.num_q(level), 1) * N) for s in lw.shards]
            lw.key_switch(x_own, key_own, out0, out1, level)
            rt.sync()
            assert np.array_equal(lw.join(out0, level), e0) and np.array_equal(lw.join(out1, level), e1), (world, level)
            assert np.array_equal(lw.join(x_own, level), x)  # inputs intact
            # rescale of the pair (x, y)
            if level > 1:
                y = o.uniform(level, level, 181 + level)
                y_own = lw.split(y, level)
                lw.rescale(x_own, y_own, out0, out1, level)
                rt.sync()
                assert np.array_equal(lw.join(out0, level - 1), o.rescale(x, level)), (world, level)
                assert np.array_equal(lw.join(out1, level - 1), o.rescale(y, level)), (world, level)
                for b in y_own:
                    b.free()
            for b in x_own + out0 + out1:
                b.free()
        # encode: one integer message, every rank reduces and transforms its own limbs (scale degree 1 and 2)
        level = levels[0]
        slots = N // 2
        msg = (np.cos(0.37 * np.arange(slots)) * 0.5).astype(np.float64)
        d_vals, d_msg = rt.to_device(msg.view(np.uint64)), rt.buf(N)
        rt.check(rt.lib.acehip_encode_message(rt.h, d_msg.ptr, d_vals.ptr, 1, slots, slots, float(2 ** sf), None))
        for deg in (1, 2):
            exp, _ = o.encode(msg.astype(np.complex128), level, sf_degree=deg)
            outs = [rt.buf(max(s.num_q(level), 1) * N) for s in lw.shards]
            for s, ob in zip(lw.shards, outs):
                rt.check(rt.lib.acehip_shard_encode_limbs(s.h, ob.ptr, d_msg.ptr, float(2 ** sf), deg, level, None))
            rt.sync()
            assert np.array_equal(lw.join(outs, level), exp), (world, deg)
            for b in outs:
                b.free()
        for b in key_own + [d_vals, d_msg]:
            b.free()
    finally:
        lw.close()
        rt.close()
        o.close()


def test_sharded_key_switch_c3_is_batched():
    """BASELINE configs[2] size (N = 2^16, L = 25, dnum = 4) on ONE rank: the C++ path is a handful of batched launches per
    phase -- under 1 ms per key-switch (the Python-driven per-limb schedule of round 1 took 6.7 ms; the fused single-GPU
    acehip_key_switch 0.28 ms) -- and still bit-exact."""
    import ace_compiler_amd as A
    from ace_compiler_amd import shard

    N, L, q0, sf, dnum = 65536, 25, 60, 56, 4
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    lw = shard.LocalWorld(rt, 1)
    try:
        key = np.ascontiguousarray(o.make_key(5000)).reshape(o.dnum, 2, L + o.K, N)
        x = o.uniform(L, L, 7)
        e0, e1 = o.key_switch(x, key, L)
        key_own, x_own = [rt.to_device(key)], lw.split(x, L)
        out0, out1 = [rt.buf(L * N)], [rt.buf(L * N)]
        lw.key_switch(x_own, key_own, out0, out1, L)
        rt.sync()
        assert np.array_equal(out0[0].download((L, N)), e0) and np.array_equal(out1[0].download((L, N)), e1)
        ms = rt.time_ms(lambda: lw.key_switch(x_own, key_own, out0, out1, L), 20)
        print("sharded key-switch, 1 rank, C3: %.3f ms" % ms)
        assert ms < 1.0, ms
        for b in key_own + x_own + out0 + out1:
            b.free()
    finally:
        lw.close()
        rt.close()
        o.close()


NCCL_WORKER = """
import os, socket, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
import torch
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))   # torch's HIP runtime first, like bench.py does
import _oracle as O
import ace_compiler_amd as A
from ace_compiler_amd import shard
N, L, q0, sf, dnum, level = 4096, 6, 60, 50, 3, 5
o = O.Oracle(N, L, q0, sf, dnum)
rt = A.AceHip(N, L, q0, sf, dnum, device=0)
key = np.ascontiguousarray(o.make_key(4000)).reshape(o.dnum, 2, L + o.K, N)
x, y = o.uniform(level, level, 91), o.uniform(level, level, 92)
e0, e1 = o.key_switch(x, key, level)
d_key, d_x, d_y = rt.to_device(key), rt.to_device(x), rt.to_device(y)
o0, o1 = rt.buf(level * N), rt.buf(level * N)
comm = shard.TorchComm(dist, torch.device("cuda", 0))
rr = shard.RankRunner(rt, comm, 0)
try:
    rr.key_switch(d_x.ptr, d_key.ptr, o0.ptr, o1.ptr, level)   # on torch's default stream: refused (not the library's stream)
    raise SystemExit("the default stream was accepted")
except RuntimeError:
    pass
torch.cuda.synchronize()   # the uploads above went through the library's own stream
with rr.stream():
    keep = rr.key_switch(d_x.ptr, d_key.ptr, o0.ptr, o1.ptr, level)
    torch.cuda.synchronize()
    assert np.array_equal(o0.download((level, N)), e0) and np.array_equal(o1.download((level, N)), e1)
    rr.rescale(d_x.ptr, d_y.ptr, o0.ptr, o1.ptr, level)
    torch.cuda.synchronize()
assert np.array_equal(o0.download((level, N))[:level - 1], o.rescale(x, level))
assert np.array_equal(o1.download((level, N))[:level - 1], o.rescale(y, level))
rr.close()
rt.close()
dist.destroy_process_group()
print("nccl shard ok")
"""


def test_torch_communicator_single_rank_nccl(tmp_path):
    """the torch.distributed driver (shard.RankRunner + TorchComm, backend "nccl" = RCCL) with world size 1 on this box, in a
    process of its own (torch initialises its HIP runtime first, as in bench.py): exchange buffers are torch tensors whose
    device addresses go through the C ABI, launches and collectives share torch's stream; results = the oracle's.  World
    sizes > 1 need more GPUs than a test box has; the partition / gather layout for them is covered by the simulated ranks
    above and by the gloo test of tests/test_dist_gloo.py."""
    import subprocess
    import sys

    from conftest import ROOT

    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % (ROOT, ROOT))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl shard ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_base_conv_subset_against_full_conversions():
    """acehip_base_conv on chosen targets = the matching limbs of the full Decomp_modup / Mod_down (before their NTT/tail):
    checked through the identities ext[d][p] = NTT(conv) and (x - NTT(conv)) * P^-1, i.e. against the oracle's results."""
    import ace_compiler_amd as A

    N, L, q0, sf, dnum, level = 4096, 6, 60, 50, 3, 5
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    try:
        K = o.K
        a = o.uniform(level, level, 311)
        coef = o.ntt_inv(a, list(range(level)))
        d_coef = rt.to_device(coef)
        for d in range(o.num_decomp(level)):
            start, n2 = o.alpha * d, min(o.alpha, level - o.alpha * d)
            full = o.decomp_modup(a, level, d)
            targets = [p for p in range(level + K) if not (start <= p < start + n2)][::2]   # every other complement limb
            pos = np.asarray(targets, dtype=np.uint32)
            out = rt.buf(len(targets) * N)
            rt.check(rt.lib.acehip_base_conv(rt.h, out.ptr, d_coef.at(start * N), level, d, pos.ctypes.data, len(targets), None))
            got = out.download((len(targets), N))
            gis = [o.gidx(p, level) for p in targets]
            assert np.array_equal(o.ntt_fwd(got, gis), full[targets]), d
            out.free()
            # a limb of the digit itself is not a conversion target
            bad = np.asarray([start], dtype=np.uint32)
            assert rt.lib.acehip_base_conv(rt.h, d_coef.ptr, d_coef.ptr, level, d, bad.ctypes.data, 1, None) == -1
        assert rt.lib.acehip_base_conv(rt.h, d_coef.ptr, d_coef.ptr, level, 9, pos.ctypes.data, 1, None) == -1
        d_coef.free()
    finally:
        rt.close()
        o.close()
