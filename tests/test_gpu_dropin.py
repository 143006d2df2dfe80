"""Drop-in acceptance on the GPU (-m gpu): the reference's checked-in ACE-generated example programs
(rtlib/ant/example/eg_fhertlib_*.c + .inc; registered as ctest cases in rtlib/ant/CMakeLists.txt:79-92)
compiled UNCHANGED against our headers and libFHErt_ant.so by `make -C oracle examples` (dev container,
outputs in oracle/_ref/examples, which travels to the GPU box).  Each program embeds its expected
output and prints SUCESS! when |out - expected| < 1e-3 (eg_fhertlib_relin.c:16-17,59-60)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

EX_DIR = os.path.join(ROOT, "oracle", "_ref", "examples")
EXAMPLES = ["add", "add_const", "mul_const", "rotate", "rotate_02", "relin", "relin_02", "gemm", "gemm_02", "conv2d",
            "avg_pool", "relu", "bootstrap", "bootstrap_02"]


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_generated_example(name):
    exe = os.path.join(EX_DIR, "eg_" + name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/examples not built (needs /root/reference: make -C oracle examples)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SUCESS!" in r.stdout
    # stdout contract parsed by the reference's scripts/perf.py:233-276
    assert r.stdout.lstrip().startswith("ckks_param: _provider = 0")
    assert "Total memory size for keys: rot_key_cnt =" in r.stdout
