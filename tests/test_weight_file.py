"""The synthetic weight files behind the end-to-end parity fixtures (tools/make_weight_file.py --gen ih12) must be the same bytes under
every numpy / libm / CPU: the generator is integer arithmetic (SplitMix64 lanes summed to an Irwin-Hall variate) plus one IEEE
multiplication and one IEEE rounding.  Pinned here by known answers and by the same arithmetic on Python integers."""
import hashlib
import os
import struct
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_weight_file as W  # noqa: E402


def test_splitmix64_known_answers():
    # SplitMix64 with state 0: the finaliser applied to k * golden (first outputs of the published generator seeded with 0)
    assert [W.mix_py(k * W.GOLDEN) for k in (1, 2, 3)] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


def test_vector_form_equals_integer_form():
    for seed, e, n, sigma in [(2, 0, 257, 0.2), (2, 6043, 64, 0.12), (7, 1, 1, 0.01), (2, 5, 0, 0.2)]:
        assert np.array_equal(W.values_python(seed, e, n, sigma), W.values_ih12(seed, e, n, sigma))


def test_known_values_and_moments():
    v = W.values_ih12(2, 5, 4, 0.2)
    assert v.dtype == np.float32
    assert [struct.pack("<f", x).hex() for x in v] == ['00ec6dbe', '0080643e', 'cdc4adbd', '6652a1be']  # little-endian float32 bit patterns
    big = W.values_ih12(2, 7, 1 << 20, 1.0).astype(np.float64)
    assert abs(big.mean()) < 3e-3 and abs(big.std() - 1.0) < 3e-3 and np.abs(big).max() < 6.0  # Irwin-Hall(12): |x| <= 6
    # entries are independent streams: generating entry 9 does not depend on what was generated before
    assert np.array_equal(W.values_ih12(2, 9, 100, 0.2), W.values_ih12(2, 9, 200, 0.2)[:100])


def test_file_bytes_are_pinned(tmp_path):
    ent = tmp_path / "entries.txt"
    ent.write_text("0 300\n1 17\n3 1024\n")
    out = tmp_path / "w.msg"
    cnt, size = W.write_file(str(ent), str(out), 2, 0.2, "ih12")
    data = out.read_bytes()
    assert cnt == 4 and data[:7] == b"!ANTFHE"
    # sha256 of the whole container (header page, aligned entries, lookup table), computed once with numpy 2.2 on x86-64 and equal
    # to the digest of the same file assembled from values_python (pure integers): any platform that disagrees fails here
    py = bytearray(data)
    ofs = 4096
    for e, n in ((0, 300), (1, 17), (2, 0), (3, 1024)):
        blob = W.values_python(2, e, n, 0.2).tobytes()
        py[ofs:ofs + len(blob)] = blob
        ofs = (ofs + len(blob) + 31) // 32 * 32
    assert bytes(py) == data
    assert hashlib.sha256(data).hexdigest() == PINNED


PINNED = "0faa13f7141b434613aef9ecca47bf13149f658cc67c47cecc675b45261153d9"
