"""Bit-exact parity of the rt_ant shim under the reference's GENERATED (polynomial-level) programs, on the GPU (-m gpu).

The reference's acceptance tests are its generated programs (rtlib/ant/example/eg_fhertlib_*.c + .inc, rtlib/ant/CMakeLists.txt:79-92;
the ResNet sources under rtlib/ant/dataset).  They enter the runtime one limb at a time (Hw_modadd / Hw_modmul / Hw_rotate,
Decomp / Mod_up / Decomp_modup, Mod_down, Rescale, Init_ciph_*, Bootstrap: rtlib/ant/src/rtlib/rtlib.c:41-87, poly_eval.c), which on
the product side is the LAZY shim -- per-limb queue, dependency chains, register forwarding, dead-store elimination, lazy zero fills,
held-back Mod_down / Rescale pairs, the ModUp digit cache (csrc/rt/rt_poly.cpp, csrc/api_hw_batch.cpp).

Oracle: the reference rtlib itself, run in the dev container under the SAME unchanged programs with the key set and the encryption
randomness the product derives from ACEHIP_SEED injected into it (tests/c/gen_parity_ref.c, `make -C oracle refgen`); every
Set_output_data ciphertext was written in the product's dump layout and its sha256 committed (tests/golden/gen_parity.json, made by
tests/golden/gen_gen_parity.py).  Here the unchanged programs run against libFHErt_ant.so with that seed and every output ciphertext
must hash to the committed digest: byte-identical to the reference, with the lazy machinery on, with three images per launch
(image k = the k-th encryption of the stream; the reference ran once per k), under ACEHIP_POISON=1, and with the lazy machinery
switched off piece by piece.
"""
import glob
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

EX_DIR = os.path.join(ROOT, "workloads", "_gen", "examples")
FIX = json.load(open(os.path.join(GOLDEN, "gen_parity.json")))
EXAMPLES = sorted(FIX["examples"])


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _run(exe, args, env_extra, tmp, tag, timeout=900):
    if not os.path.exists(exe):
        pytest.skip("workloads/_gen/examples not built (needs /root/reference: make -C workloads)")
    prefix = os.path.join(str(tmp), tag)
    env = dict(os.environ, ACEHIP_SEED=str(FIX["seed"]), ACEHIP_DUMP_OUTPUT=prefix, **env_extra)
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout, {os.path.basename(p)[len(tag) + 1:]: _sha(p) for p in sorted(glob.glob(prefix + ".*"))}


def test_fixture_covers_the_reference_ctest_cases():
    assert len(EXAMPLES) == 14 and all(set(FIX["examples"][e]["batch3"]) == {"0", "1", "2"} for e in EXAMPLES)


@pytest.mark.parametrize("name", EXAMPLES)
def test_generated_example_is_byte_identical_to_the_reference(name, tmp_path):
    """unchanged eg_fhertlib_<name>: lazy shim on (the default) -- output ciphertext = the reference's, byte for byte"""
    out, got = _run(os.path.join(EX_DIR, "eg_" + name), [], {}, tmp_path, "lazy")
    assert "SUCESS!" in out
    assert got == FIX["examples"][name]["single"], "output ciphertext of eg_%s differs from the reference rtlib's" % name


@pytest.mark.parametrize("name", EXAMPLES)
def test_generated_example_three_images_per_launch(name, tmp_path):
    """ACEHIP_BATCH=3 under the unchanged program: one Main_graph, three images per launch, each image's output ciphertext equal
    to the reference's run on that image's encryption (Prepare_input of a program that knows nothing of batches encrypts its tensor
    three times in a row; the reference ran once per image with the other two encryptions' draws skipped)"""
    out, got = _run(os.path.join(EX_DIR, "eg_" + name), [], {"ACEHIP_BATCH": "3"}, tmp_path, "b3")
    assert "SUCESS!" in out
    want = {}
    for k in range(3):
        for key, dig in FIX["examples"][name]["batch3"][str(k)].items():
            call = key.split(".")[0]
            want["%s.%d" % (call, k)] = dig
    assert got == want, "eg_%s with three images per launch differs from the reference" % name


@pytest.mark.parametrize("name", EXAMPLES)
def test_generated_example_under_poison(name, tmp_path):
    """ACEHIP_POISON=1: deferred fills and parked blocks hold non-residues, declared operand lists are checked -- same bytes"""
    out, got = _run(os.path.join(EX_DIR, "eg_" + name), [], {"ACEHIP_POISON": "1"}, tmp_path, "poison")
    assert "SUCESS!" in out
    assert got == FIX["examples"][name]["single"]


@pytest.mark.parametrize("env", [{"ACEHIP_LAZY_ZERO": "0"}, {"ACEHIP_HW_DISCARD": "0"}, {"ACEHIP_MODUP_REUSE": "0"}, {"ACEHIP_CONV_MFMA": "0"},
                                 {"ACEHIP_PT_PREFETCH": "0"}], ids=lambda e: "_".join("%s=%s" % kv for kv in e.items()))
@pytest.mark.parametrize("name", ["rotate_02", "relin_02", "conv2d", "relu", "bootstrap_02"])
def test_generated_example_with_a_mechanism_switched_off(name, env, tmp_path):
    """the same digests with one piece of the lazy machinery (or the matrix-core conversion) off: every configuration a user can
    select computes the reference's bytes"""
    out, got = _run(os.path.join(EX_DIR, "eg_" + name), [], env, tmp_path, "off")
    assert "SUCESS!" in out
    assert got == FIX["examples"][name]["single"]
