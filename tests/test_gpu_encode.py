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
            msg = np.array([np.float32(((i * 7 + g["seed"]) % 17) - 8) / np.float32(16.0) for i in range(n)], dtype=np.float32)
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
