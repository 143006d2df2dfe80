"""Limb-sharded key-switch (ace-compiler_amd/shard.py): G simulated ranks on one GPU, limb gi on rank gi % G,
two all-gathers per key-switch.  Assembling the ranks' owned output limbs must reproduce the unsharded key-switch of
the oracle bit for bit, for every world size including ones that leave some ranks without p-limbs or q-limbs."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(64, 7, 60, 51, 3, [7, 4]), (4096, 6, 60, 50, 3, [6, 5, 2]), (65536, 5, 60, 56, 2, [5, 3])],
                         ids=["n64", "n4096", "n65536"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_sharded_key_switch_matches_oracle(cfg, world):
    import ace_compiler_amd as A
    from ace_compiler_amd import shard

    N, L, q0, sf, dnum, levels = cfg
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    try:
        key = o.make_key(3000)
        for level in levels:
            x = o.uniform(level, level, 81 + level)
            e0, e1 = o.key_switch(x, key, level)
            g0, g1 = shard.run_local(rt, world, level, x, np.ascontiguousarray(key).reshape(o.dnum, 2, L + o.K, N))
            assert np.array_equal(g0, e0) and np.array_equal(g1, e1), (world, level)
    finally:
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
key = np.ascontiguousarray(o.make_key(4000))
x = o.uniform(level, level, 91)
e0, e1 = o.key_switch(x, key, level)
d_key, d_x = rt.to_device(key), rt.to_device(x)
T = L + o.K
ks = shard.ShardedKeySwitch(rt, 0, 1)
comm = shard.TorchComm(dist, torch.device("cuda", 0))
o0, o1 = shard.run_rank(ks, ks.run(level, d_x, lambda d, comp, gi: d_key.at(((d * 2 + comp) * T + gi) * N)), comm)
assert np.array_equal(o0.download((level, N)), e0) and np.array_equal(o1.download((level, N)), e1)
rt.close()
dist.destroy_process_group()
print("nccl shard ok")
"""


def test_torch_communicator_single_rank_nccl(tmp_path):
    """the torch.distributed driver (shard.run_rank + TorchComm, backend "nccl" = RCCL) with world size 1 on this box, in a
    process of its own (torch initialises its HIP runtime first, as in bench.py): exchange buffers are torch tensors whose
    device addresses go through the C ABI; result = the oracle's key-switch.  World sizes > 1 need more GPUs than a test
    box has; the partition / gather layout for them is covered by the simulated ranks above and by the gloo test of
    tests/test_dist_gloo.py."""
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
