"""Encode path on the GPU box: Encode_plain_from_float / Encode_val_at_level of the drop-in
shim against golden plaintexts produced by the reference itself (tests/golden/ref_encode_*.json, written by
oracle/ref_dump.c `encode`: Encode_at_level_with_sf ckks_encoder.c:395, Encode_val_at_level :464).
Bit-exact: the FP64 canonical embedding follows the reference's butterfly order with no FMA contraction, the
integer part runs in HIP kernels."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_encode_*.json")))


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    so = str(tmp_path_factory.mktemp("stub") / "libctxstub.so")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "c", "ctx_stub.c"), "-I", inc, "-I",
                           os.path.join(inc, "rt_ant"), "-L", bmod.LIBDIR, "-Wl,--no-as-needed", "-lFHErt_ant", "-Wl,-rpath," + bmod.LIBDIR, "-o", so])
    lib = C.CDLL(so, mode=C.RTLD_GLOBAL)
    lib.Stub_set_params.argtypes = [C.c_uint32] + [C.c_size_t] * 5
    lib.Stub_sizeof_plaintext.restype = C.c_size_t
    lib.Stub_plain_data.restype = C.c_void_p
    lib.Stub_plain_data.argtypes = [C.c_void_p]
    lib.Stub_plain_level.restype = C.c_size_t
    lib.Stub_plain_level.argtypes = [C.c_void_p]
    lib.Encode_plain_from_float.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    lib.Encode_plain_from_double.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    lib.Free_plain.argtypes = [C.c_void_p]
    lib.acehip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def _download(lib, pt, N):
    lib.Acehip_rt_sync()  # the shim batches per-limb work lazily; raw reads of Coeffs() memory sync first
    level = lib.Stub_plain_level(pt)
    out = np.empty((level, N), dtype=np.uint64)
    assert lib.acehip_memcpy_d2h(out.ctypes.data, lib.Stub_plain_data(pt), out.nbytes, None) == 0
    return out


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-5] for p in FILES])
def test_encode_matches_reference(stub, path):
    g = json.load(open(path))
    N, level = g["N"], g["level"]
    stub.Stub_set_params(N, g["L"] - 1, g["q0_bits"], g["sf_bits"], g["dnum_req"], 192)
    stub.Prepare_context()
    try:
        for case in g["cases"]:
            n = case["len"]
            msg = O.encode_message(n, g["seed"])
            pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
            stub.Encode_plain_from_float(pt, msg.ctypes.data, n, case["sf_degree"], level)
            got = _download(stub, pt, N)
            gold = case["poly"]
            assert got.size == gold["n"]
            if "data" in gold:
                assert got.reshape(-1).tolist() == gold["data"], (n, case["sf_degree"])
            assert O.sum64(got) == gold["sum64"] and O.xorw(got) == gold["xorw"], (n, case["sf_degree"])
            stub.Free_plain(pt)
        for k in g["consts"]:
            val = np.array([k["value"]], dtype=np.float64)
            pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
            stub.Encode_plain_from_double(pt, val.ctypes.data, 1, k["sf_degree"], level)
            got = _download(stub, pt, N)
            assert got[:, 0].tolist() == k["limb0"], k
            assert np.all(got == got[:, :1])  # constant polynomial in the NTT domain
            stub.Free_plain(pt)
    finally:
        stub.Finalize_context()


# ---- acehip_encode through the C ABI against the oracle (oracle/ckks_encode.c, itself pinned to the reference
# fixtures by tests/test_oracle_golden.py::test_encode_matches_reference) on inputs the fixtures do not cover:
# complex values, sparse slot counts, extended (q+p) plaintexts, ragged and empty messages, wide dynamic range.
ENC_CASES = [
    # N, L, q0, sf, dnum, level, slots, len, kind, sf_degree, n_p
    (16, 3, 60, 50, 2, 3, 0, 8, "f32", 1, 0),
    (16, 3, 60, 50, 2, 2, 2, 2, "c128", 1, 0),
    (16, 3, 60, 50, 2, 3, 1, 1, "f64", 2, 2),
    (64, 7, 60, 51, 3, 4, 0, 0, "f32", 1, 0),          # empty message: the zero plaintext
    (64, 7, 60, 51, 3, 7, 8, 5, "c128", 3, 3),
    (1024, 7, 60, 51, 3, 6, 0, 511, "f64", 1, 0),       # ragged
    (1024, 7, 60, 51, 3, 6, 256, 256, "c128", 1, 3),    # exactly the low-pass block size
    (1024, 7, 60, 51, 3, 6, 512, 300, "c128", 2, 0),    # one strided stage
    (4096, 6, 60, 50, 3, 6, 0, 2048, "f32", 1, 0),
    (65536, 4, 60, 56, 2, 4, 0, 32768, "f32", 1, 0),    # ResNet weight shape: full slots, float message
    (65536, 4, 60, 56, 2, 3, 4096, 4096, "c128", 1, 1), # bootstrap precompute shape: sparse slots, extended
    (131072, 3, 60, 50, 1, 2, 0, 65536, "f64", 1, 0),   # largest ring degree the compiler emits
]


def _message(kind, n, seed):
    rng = np.random.default_rng(seed)
    mag = np.ldexp(rng.standard_normal(n), rng.integers(-20, 4, size=n))   # wide dynamic range
    if kind == "f32":
        return mag.astype(np.float32)
    if kind == "f64":
        return mag.astype(np.float64)
    return (mag + 1j * rng.standard_normal(n)).astype(np.complex128)


@pytest.mark.parametrize("case", ENC_CASES, ids=["N%d_lv%d_s%d_n%d_%s_d%d_p%d" % (c[0], c[5], c[6], c[7], c[8], c[9], c[10]) for c in ENC_CASES])
def test_encode_abi_matches_oracle(case):
    import ace_compiler_amd as A

    N, L, q0, sf, dnum, level, slots, n, kind, deg, n_p = case
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    try:
        msg = _message(kind, n, 1000 + N + n)
        q, p = rt.encode(msg, level, slots=slots, sf_degree=deg, n_p=n_p)
        eq, ep = o.encode(msg.astype(np.complex128), level, slots=slots, sf_degree=deg, n_p=n_p)
        assert np.array_equal(q, eq)
        assert np.array_equal(p, ep)
    finally:
        rt.close()
        o.close()


def test_encode_overflow_is_reported():
    """|x * Delta| > 9.2e18 is the reference's "encode overflow" assert (ckks_encoder.c:255-258): the device path
    records it and acehip_encode_status fails; the oracle reports the same."""
    import ace_compiler_amd as A

    o = O.Oracle(64, 3, 60, 50, 1)
    rt = A.AceHip(64, 3, 60, 50, 1, device=0)
    try:
        msg = np.full(32, 1.0e5, dtype=np.float64)
        with pytest.raises(OverflowError):
            o.encode(msg.astype(np.complex128), 3)
        with pytest.raises(A.AceHipError, match="encode overflow"):
            rt.encode(msg, 3)
        q, _ = rt.encode(np.ones(32), 3)  # the flag is cleared once reported
        assert np.array_equal(q, o.encode(np.ones(32, dtype=np.complex128), 3)[0])
    finally:
        rt.close()
        o.close()


def test_two_host_threads_own_independent_contexts(stub):
    """All runtime state is per host thread (context, pool, queue, HIP stream): two threads that prepare a context each and
    encode concurrently must both reproduce the reference plaintexts (the bench runs 4 such image streams per GPU)."""
    import threading

    path = [p for p in FILES if "n1024" in p][0]
    g = json.load(open(path))
    N, level = g["N"], g["level"]
    stub.Stub_set_params(N, g["L"] - 1, g["q0_bits"], g["sf_bits"], g["dnum_req"], 192)
    errors = []
    gate = threading.Barrier(2)

    def run(tid):
        try:
            stub.Prepare_context()
            gate.wait()
            for _ in range(20):
                for case in g["cases"]:
                    msg = O.encode_message(case["len"], g["seed"])
                    pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
                    stub.Encode_plain_from_float(pt, msg.ctypes.data, case["len"], case["sf_degree"], level)
                    got = _download(stub, pt, N)
                    if O.sum64(got) != case["poly"]["sum64"] or O.xorw(got) != case["poly"]["xorw"]:
                        errors.append((tid, case["len"], case["sf_degree"]))
                    stub.Free_plain(pt)
            gate.wait()
            stub.Finalize_context()
        except BaseException as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
            gate.abort()

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


# ---- pre-encoded weight files (DE_PLAINTEXT): Pt_get / Pt_prefetch / Pt_free (pt_mgr.c:63-159) ----
def _ptfile_message(e, n, seed):
    """the message ref_dump.c `ptfile` encodes into entry e: ((splitmix64(seed + e, i) % 2001) - 1000) / 1024"""
    return ((O.splitmix64(seed + e, np.arange(n, dtype=np.uint64)) % np.uint64(2001)).astype(np.int64) - 1000).astype(np.float32) / np.float32(1024.0)


PT_FILES = [("ref_ptfile_n64_l5_lv3.bin", 64, 5, 60, 50, 2, 3, 3, 1, 77, None),
            (None, 65536, 5, 51, 50, 2, 4, 2, 1, 78, "65536 5 51 50 2 4 %s 2 1 78"),     # generated on the box when oracle/_ref is there
            (None, 4096, 6, 60, 50, 3, 6, 3, 2, 79, "4096 6 60 50 3 6 %s 3 2 79")]


@pytest.mark.parametrize("cfg", PT_FILES, ids=["n64_committed", "n65536", "n4096_deg2"])
def test_pt_get_serves_reference_plaintext_file(stub, cfg, tmp_path):
    """A DE_PLAINTEXT data file written with the reference's own Encode_plain_buffer (oracle/ref_dump.c `ptfile`): Pt_get must
    hand out exactly the stored residues (resident in HBM), which in turn equal what our device encode makes of the same
    message -- the pre-encoded and the encode-on-the-fly weight paths are interchangeable bit for bit."""
    fname, N, L, q0, sf, dnum, level, n_ent, deg, seed, gen = cfg
    if fname:
        path = os.path.join(GOLDEN, fname)
    else:
        ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
        if not os.path.exists(ref):
            pytest.fail("oracle/_ref/ref_dump not built (needs /root/reference) -- a build output of the dev container that must travel with the snapshot")
        path = str(tmp_path / "pt.bin")
        subprocess.check_call([ref, "ptfile"] + (gen % path).split())
    stub.Stub_set_data_file.argtypes = [C.c_char_p, C.c_int]
    stub.Pt_get.restype = C.c_void_p
    stub.Pt_get.argtypes = [C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32]
    stub.Pt_prefetch.argtypes = [C.c_uint32]
    stub.Pt_free.argtypes = [C.c_uint32]
    stub.Stub_set_params(N, L - 1, q0, sf, dnum, 192)
    stub.Stub_set_data_file(path.encode(), 2)
    stub.Prepare_context()
    try:
        raw = open(path, "rb").read()
        import struct

        lut_ofst = struct.unpack_from("<Q", raw, 24)[0]
        stub.Pt_prefetch(n_ent - 1)
        for e in range(n_ent):
            name, idx, size, ofst = struct.unpack_from("<16sIIQ", raw, lut_ofst + 32 * e)
            words = level * N
            stored = np.frombuffer(raw, dtype=np.uint64, count=words, offset=ofst + 16 + stub.Stub_sizeof_plaintext()).reshape(level, N)
            pt = stub.Pt_get(e, N // 2, deg, level)
            assert stub.Stub_plain_level(pt) == level
            got = _download(stub, pt, N)
            assert np.array_equal(got, stored), e
            msg = _ptfile_message(e, N // 2, seed)
            mine = C.create_string_buffer(stub.Stub_sizeof_plaintext())
            stub.Encode_plain_from_float(mine, msg.ctypes.data, N // 2, deg, level)
            assert np.array_equal(_download(stub, mine, N), stored), e
            stub.Free_plain(mine)
            assert stub.Pt_get(e, N // 2, deg, level) == pt   # resident: the same plaintext on every request
            stub.Pt_free(e)
    finally:
        stub.Finalize_context()
        stub.Stub_set_data_file(None, 0)


def test_plaintext_cache_equals_encoding(stub, tmp_path):
    """ACEHIP_PT_CACHE=1 (weight plaintexts kept in HBM after their first encode): what Pt_from_msg returns from the cache on
    the second and third request equals what it encodes without the cache, for two entries, levels and scale degrees."""
    N, L, q0, sf, dnum = 4096, 6, 60, 50, 3
    ent = tmp_path / "entries.txt"
    ent.write_text("0 2048\n1 1024\n2 1\n")
    wfile = str(tmp_path / "w.msg")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_weight_file.py"), "--entries", str(ent), "--out", wfile])
    stub.Stub_set_data_file.argtypes = [C.c_char_p, C.c_int]
    stub.Pt_from_msg.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32]
    stub.Stub_set_params(N, L - 1, q0, sf, dnum, 192)
    stub.Stub_set_data_file(wfile.encode(), 0)
    results = {}
    try:
        for cache in ("0", "1"):
            os.environ["ACEHIP_PT_CACHE"] = cache
            stub.Prepare_context()
            for rep in range(3 if cache == "1" else 1):
                for (idx, n, deg, level) in [(0, 2048, 1, 6), (1, 1024, 2, 4), (0, 2048, 1, 3), (2, 1, 1, 5)]:
                    pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
                    stub.Pt_from_msg(pt, idx, n, deg, level)
                    got = _download(stub, pt, N)
                    key = (idx, n, deg, level)
                    if cache == "0":
                        results[key] = got
                    else:
                        assert np.array_equal(got, results[key]), (key, rep)
                    stub.Free_plain(pt)
            stub.Finalize_context()
    finally:
        os.environ.pop("ACEHIP_PT_CACHE", None)
        stub.Stub_set_data_file(None, 0)


@pytest.mark.parametrize("N", [4096, 65536], ids=["n4096_sequential", "n65536_batched"])
def test_weight_prefetch_equals_encoding(stub, tmp_path, N):
    """Weight-plaintext prefetch (rt_io.cpp): the Pt_from_msg calls of the first input are recorded, later inputs get their
    plaintexts from batched encodes issued ahead of the calls (acehip_encode_batch at N = 2^16; one by one below).  What the
    calls return must equal the direct encodes of the first input bit for bit -- for runs of equal calls, a change of level /
    scale / length inside the sequence, and an input whose calls leave the recorded order (fallback)."""
    L, q0, sf, dnum = 4, 60, 50, 2
    n = N // 4
    ent = tmp_path / "entries.txt"
    ent.write_text("".join("%d %d\n" % (i, n if i != 5 else n // 2) for i in range(7)))
    wfile = str(tmp_path / "w.msg")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_weight_file.py"), "--entries", str(ent), "--out", wfile])
    stub.Stub_set_data_file.argtypes = [C.c_char_p, C.c_int]
    stub.Pt_from_msg.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32]
    stub.Acehip_rt_prefetched_count.restype = C.c_size_t
    stub.Stub_set_params(N, L - 1, q0, sf, dnum, 192)
    stub.Stub_set_data_file(wfile.encode(), 0)
    seq = [(0, n, 1, 4), (1, n, 1, 4), (2, n, 1, 4), (3, n, 1, 4), (4, n, 2, 3), (6, n, 2, 3), (5, n // 2, 1, 4), (0, n, 1, 4),
           (1, n, 1, 2)]
    os.environ["ACEHIP_PT_PREFETCH"] = "3"   # batches of 3: runs longer than a batch, and batches cut short by a change
    try:
        stub.Prepare_context()
        first = {}
        for image in range(4):
            stub.Acehip_rt_next_input()
            calls = seq if image < 3 else [seq[1], seq[0]] + seq[2:]   # the last input swaps two calls: prediction ends
            held = []
            for k, (idx, ln, deg, level) in enumerate(calls):
                pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
                stub.Pt_from_msg(pt, idx, ln, deg, level)
                held.append(pt)
                if len(held) > 2:   # plaintexts stay alive for a while, as in a convolution's tap loop
                    stub.Free_plain(held.pop(0))
                got = _download(stub, pt, N)
                key = (idx, ln, deg, level)
                if image == 0:
                    first.setdefault(key, got)
                assert np.array_equal(got, first[key]), (image, k, key)
            for pt in held:
                stub.Free_plain(pt)
            want = 0 if image == 0 else len(seq) * min(image, 2)
            assert stub.Acehip_rt_prefetched_count() == want, (image, stub.Acehip_rt_prefetched_count(), want)
        stub.Finalize_context()
    finally:
        os.environ.pop("ACEHIP_PT_PREFETCH", None)
        stub.Stub_set_data_file(None, 0)


def test_encode_batch_abi_matches_single_encodes():
    """acehip_encode_batch (embedding kernels over the batch, one NTT over separate output blocks) against acehip_encode, one
    message at a time: bit-identical q-limbs, scale degree 1 and 2."""
    import ace_compiler_amd as A

    N, L = 65536, 4
    rt = A.AceHip(N, L, 60, 50, 2)
    try:
        rng = np.random.default_rng(5)
        B, ln = 5, 12000
        msgs = [(rng.standard_normal(ln) * 0.1).astype(np.float32) for _ in range(B)]
        for deg, level in ((1, 4), (2, 3)):
            singles = [rt.encode(m, level, sf_degree=deg)[0] for m in msgs]
            dvals = [rt.to_device(m) for m in msgs]
            outs = [rt.buf(level * N) for _ in range(B)]
            hq = (C.c_void_p * B)(*[o.ptr for o in outs])
            hv = (C.c_void_p * B)(*[d.ptr for d in dvals])
            rt.check(rt.lib.acehip_encode_batch(rt.h, hq, hv, B, 0, ln, 0, float(2.0 ** rt.sf_bits), deg, level, None))
            rt.check(rt.lib.acehip_encode_status(rt.h))
            for b in range(B):
                assert np.array_equal(outs[b].download((level, N)), singles[b]), (deg, b)
            for d in dvals + outs:
                d.free()
    finally:
        rt.close()
