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
    # a committed fixture + a missing workload binary is a FAILURE under -m gpu: the binaries are build outputs of the dev container
    # (__graft_entry__.build -> make -C workloads) that travel with the snapshot; a box without them proves nothing and must say so
    assert os.path.exists(exe), "%s is missing: the generated programs were not built (make -C workloads; needs /root/reference) " \
                                "or did not travel -- the parity evidence cannot be produced" % os.path.relpath(exe, ROOT)
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


# ------------------------------------------------------------------------------------------------------------------------------
# The benchmarked program itself: the ACE-generated ResNet-20 (rtlib/ant/dataset/resnet20_cifar10_pre.onnx.inc, unchanged; BASELINE
# configs[3]).  FIX["models"]["resnet20"] is the REFERENCE rtlib's CPU run of it (dev container, 0.5 h of one core): key set and
# encryption randomness of ACEHIP_SEED injected, synthetic weight file N(0, sigma) of tools/model_weights.py (sigma chosen so that the
# logits are of order 0.1-1: profiles/r04a_sigma_sweep.txt), image 0 of tools/model_main.c.
# ------------------------------------------------------------------------------------------------------------------------------
import re
import sys

sys.path.insert(0, os.path.join(ROOT, "tools"))
MODEL = FIX.get("models", {}).get("resnet20")
MODEL_EXE = os.path.join(EX_DIR, "model_resnet20_cifar10_pre")


def _weights_of(key, m):
    """the weight file the reference run of fixture entry m used, regenerated here; a different md5 is a FAILURE (the entry's
    generator -- "ih12": integer-only, tools/make_weight_file.py -- must write the same bytes everywhere; "numpy" entries of rounds 1-4
    depend on numpy's stream and say so when it moves)"""
    import model_weights

    wfile, meta = model_weights.ensure(key, m["weights"]["sigma"], m["weights"].get("gen", "numpy"))
    assert meta["md5"] == m["weights"]["md5"], "the synthetic weight file of %s (generator %s) has md5 %s here, the reference run used %s" % (
        key, meta["gen"], meta["md5"], m["weights"]["md5"])
    return wfile


def _model_env(extra=None):
    assert MODEL is not None, "tests/golden/gen_parity.json has no resnet20 entry (tests/golden/gen_gen_parity.py resnet20)"
    wfile = _weights_of("resnet20", MODEL)
    env = {"ACEHIP_RT_DATA_FILE": wfile, "MODEL_DATA_FILE": wfile, "MODEL_ENC_SEED": str(MODEL["enc_seed"])}
    env.update(extra or {})
    return env


def _logits9(stdout):
    return [m.split() for m in re.findall(r"logits9:((?: -?\d+\.\d+)+)", stdout)]


def test_resnet20_output_ciphertext_is_byte_identical_to_the_reference_cpu_run(tmp_path):
    """one image through the lazy shim -- 2.5 M queued limb-ops, 170 k deferred fills, 19 bootstraps, 6 044 weight plaintexts:
    the output ciphertext hashes to the reference's digest, the decrypted logits print the same nine decimals, and the reference's
    bookkeeping lines (scripts/ace_pre.log:28) are reproduced"""
    out, got = _run(MODEL_EXE, ["1"], _model_env(), tmp_path, "r20", timeout=1200)
    assert got == MODEL["outputs"], "ResNet-20 output ciphertext differs from the reference rtlib's CPU run"
    assert _logits9(out) == [["%.9f" % v for v in MODEL["logits9"]]]
    assert max(abs(v) for v in MODEL["logits9"]) > 0.05   # real digits: the old N(0,0.05) weights gave logits of 1e-3
    assert "rot_key_cnt = 227," in out and "Total memory size for weight plain: cnt = 6044," in out


def test_resnet20_batches_of_3_and_12_and_three_threads_match_the_reference(tmp_path):
    """what bench.py times is 3 image streams x 12 images per launch.  12 images one by one, in batches of 3, as one batch of 12: every
    image's output ciphertext is the same bytes in all three runs (image i is encrypted with the randomness of seed + i whatever
    carries it), and image 0 -- the reference's image -- hashes to the reference's digest.  Three OpenMP threads on one context
    (tools/model_main_omp.c, the reference's own main structure, resnet_cifar.main.inc:77-116), each running that image: three times
    the reference's digest."""
    env = _model_env()
    o1, d1 = _run(MODEL_EXE, ["12"], env, tmp_path, "b1", timeout=1500)
    o3, d3 = _run(MODEL_EXE, ["12"], dict(env, MODEL_BATCH="3"), tmp_path, "b3", timeout=1500)
    o12, d12 = _run(MODEL_EXE, ["12"], dict(env, MODEL_BATCH="12"), tmp_path, "b12", timeout=1500)
    # the reference's stdout contract in the benchmarked mode (context.c:103-116, parsed by scripts/perf.py:242-250): one counter per
    # process, one count per image and weight plaintext -- 12 images x 6 044 whatever carries them (one by one, batches, prefetch)
    for out in (o1, o3, o12):
        assert "rot_key_cnt = 227," in out and "Total memory size for weight plain: cnt = %d," % (12 * 6044) in out, out[-1500:]
    assert d1["0.0"] == MODEL["outputs"]["0.0"]
    assert len(set(d1.values())) == 12
    for i in range(12):
        assert d3["%d.%d" % (i // 3, i % 3)] == d1["%d.0" % i], "image %d differs in batches of 3" % i
        assert d12["0.%d" % i] == d1["%d.0" % i], "image %d differs in a batch of 12" % i
    omp = os.path.join(EX_DIR, "modelomp_resnet20_cifar10_pre")
    prefix = os.path.join(str(tmp_path), "omp")
    r = subprocess.run([omp, "3"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, ACEHIP_SEED=str(FIX["seed"]), OMP_NUM_THREADS="3", MODEL_DUMP_PREFIX=prefix, **env))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = sorted(glob.glob(prefix + ".img*"))
    assert len(files) == 3 and all(_sha(f) == MODEL["outputs"]["0.0"] for f in files)
    assert "Total memory size for weight plain: cnt = %d," % (3 * 6044) in r.stdout, r.stdout[-1500:]  # image THREADS: still one counter


def test_resnet20_batch_issued_in_replica_groups_matches_the_reference(tmp_path):
    """ACEHIP_REP_CHUNK (api_internal.hpp for_replica_chunks, an experiment that stays switched off): the transform pipelines of a
    12-image batch issued in groups of 5, 5 and 2 images -- the images are independent, so every output must keep its bytes"""
    env = _model_env()
    _, plain = _run(MODEL_EXE, ["12"], dict(env, MODEL_BATCH="12"), tmp_path, "g0", timeout=1500)
    _, grouped = _run(MODEL_EXE, ["12"], dict(env, MODEL_BATCH="12", ACEHIP_REP_CHUNK="5"), tmp_path, "g5", timeout=1500)
    assert plain["0.0"] == MODEL["outputs"]["0.0"]
    assert grouped == plain and len(set(plain.values())) == 12


def test_resnet20_logits_with_independent_keys_agree_to_ckks_precision(tmp_path, capsys):
    """the tolerance-level check, with digits: OUR fresh keys and encryption randomness (no seed: a new ChaCha20 master key from the OS
    every run), same weights and image -- the logits agree with the reference's to 2.5e-2 of the largest one.  Independent keys mean
    independent CKKS noise: every one of the 19 bootstraps adds about 1.2e-3 at this parameter set on either runtime
    (profiles/r01m_bootstrap_precision.md), i.e. 5e-3 absolute as a random walk and 2.3e-2 at worst on logits up to 0.44; measured on
    single draws 2.3e-3 (round 4) and 7.2e-3 (round 5) of the largest logit.  The bound has to hold for EVERY draw of the keys, so it sits
    at five times the random-walk figure -- and a single wrong rotation misses it by more than a factor of ten (asserted by the next test).  The bit-level statement
    is the test above.  Round 6 (review of round 5): the worst-case bound alone would let a bias or noise regression of the ChaCha20-keyed
    samplers pass until it reached 2.5e-2, so THREE independent draws are made and their MEDIAN error must also stay within 1.5e-2 -- about
    twice the largest single draw measured so far; the three figures are printed into the test log."""
    env = dict(os.environ, **_model_env())
    env.pop("MODEL_ENC_SEED")
    env.pop("ACEHIP_SEED", None)
    scale = max(abs(v) for v in MODEL["logits9"])
    errs = []
    for _ in range(3):
        r = subprocess.run([MODEL_EXE, "1"], capture_output=True, text=True, timeout=1200, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = [float(x) for x in _logits9(r.stdout)[0]]
        errs.append(max(abs(a - b) for a, b in zip(got, MODEL["logits9"])) / scale)
    with capsys.disabled():
        print("\n[independent keys] max |logit - reference| / largest logit over three draws: %s" % " ".join("%.2e" % e for e in errs))
    assert max(errs) <= 2.5e-2, errs
    assert sorted(errs)[1] <= 1.5e-2, errs


def test_a_single_changed_rotation_is_caught(tmp_path, capsys):
    """negative control: the same program with ONE rotation amount changed (workloads/Makefile model_resnet20_perturbed) on the same
    keys, weights and image -- the digest differs and the logits leave the tolerance of the previous test by orders of magnitude"""
    exe = os.path.join(EX_DIR, "model_resnet20_perturbed")
    out, got = _run(exe, ["1"], _model_env(), tmp_path, "pert", timeout=1200)
    assert got["0.0"] != MODEL["outputs"]["0.0"]
    bad = [float(x) for x in _logits9(out)[0]]
    scale = max(abs(v) for v in MODEL["logits9"])
    err = max(abs(a - b) for a, b in zip(bad, MODEL["logits9"]))
    with capsys.disabled():
        print("\n[negative control] one Rotate amount changed: max |logit - reference| = %.4f (tolerance of the parity test: %.6f)" % (err, 2.5e-2 * scale))
    assert err > 10 * 2.5e-2 * scale


# ------------------------------------------------------------------------------------------------------------------------------
# The reference's key-switch optimisation unit tests (rtlib/ant/unittest/ut_ksw_opt.cxx:115-660) as a bit-level contract:
# tests/c/ksw_variants.c, base forms through the rt_ant operator API.
# ------------------------------------------------------------------------------------------------------------------------------
KSW = FIX.get("ksw_variants", {})


def _build_ksw(tmp):
    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = os.path.join(str(tmp), "ksw_variants")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", os.path.join(ROOT, "tests", "c", "ksw_variants.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
                           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe])
    return exe


@pytest.mark.parametrize("name", sorted(KSW) or ["(no fixture)"])
def test_key_switch_base_forms_match_the_reference(name, tmp_path):
    """sum of rotations of one ciphertext, sum of rotations of three, multiply + relinearise + rescale, rotate-multiply-rescale-rotate:
    our results have the bytes of the reference's BASE forms.  The fixture also records which of the reference's OPT forms are NOT
    bit-equal to its own base forms (ModDown hoisted over a sum, ModDown merged with Rescale, both plus a hoisted ModUp): a runtime
    must not "optimise" into those -- only the hoisted ModUp keeps the bits, and that one the runtime does (rt_poly.cpp ModupCache)."""
    assert KSW, "tests/golden/gen_parity.json has no ksw_variants entry (tests/golden/gen_gen_parity.py ksw)"
    exe = _build_ksw(tmp_path)
    d = tmp_path / "out"
    d.mkdir()
    for env_extra in ({}, {"ACEHIP_MODUP_REUSE": "0"}):
        r = subprocess.run([exe, str(d)] + KSW[name]["args"].split(), capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, ACEHIP_SEED=str(FIX["seed"]), **env_extra))
        assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        got = {os.path.basename(p)[:-3]: _sha(p) for p in sorted(glob.glob(str(d) + "/*.ct"))}
        assert got == KSW[name]["base"]
    assert KSW[name]["opt_vs_base"] == {"modup_hoist": "EQUAL", "moddown_hoist": "DIFFERENT", "moddown_rescale": "DIFFERENT",
                                        "moddown_rescale_modup": "DIFFERENT"}


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]'s workload: the ACE-generated ResNet-110 (rtlib/ant/dataset/resnet110_cifar10_train.onnx.inc, unchanged), one image,
# against the reference rtlib's CPU run with the same injected key set (2.2 h of one core in the dev container).  With synthetic
# weights this 110-layer network leaves the range of the bootstrap on both runtimes, so the logits mean nothing -- the bytes do: every
# intermediate is exact integer arithmetic on both sides.
# ------------------------------------------------------------------------------------------------------------------------------
MODEL110 = FIX.get("models", {}).get("resnet110")


def _model110_env():
    assert MODEL110 is not None, "tests/golden/gen_parity.json has no resnet110 entry (tests/golden/gen_gen_parity.py resnet110)"
    wfile = _weights_of("resnet110", MODEL110)
    return {"ACEHIP_RT_DATA_FILE": wfile, "MODEL_DATA_FILE": wfile, "MODEL_ENC_SEED": str(MODEL110["enc_seed"])}


@pytest.mark.parametrize("mode", [{}, {"ACEHIP_SHARD_SIM": "2"}, {"ACEHIP_SHARD_SIM": "8"}], ids=["unsharded", "2_simulated_ranks", "8_simulated_ranks"])
def test_resnet110_output_ciphertext_is_byte_identical_to_the_reference_cpu_run(mode, tmp_path):
    """unsharded, and with its RNS limbs spread over 2 and 8 simulated ranks (the execution mode of configs[4]): the output ciphertext
    hashes to the digest of the reference's CPU run"""
    exe = os.path.join(EX_DIR, "model_resnet110_cifar10_train")
    out, got = _run(exe, ["1"], dict(_model110_env(), **mode), tmp_path, "r110", timeout=1500)
    assert got == MODEL110["outputs"], "ResNet-110 output ciphertext differs from the reference rtlib's CPU run"
    assert _logits9(out) == [["%.9f" % v for v in MODEL110["logits9"]]]


def test_resnet110_sharded_over_two_processes_matches_the_reference_cpu_run(tmp_path):
    """configs[4] with ranks that are PROCESSES: the limbs of one ResNet-110 image spread over two processes that exchange them through the
    RCCL entry points (tests/c/mock_rccl.c serves them on the one-GPU test box; on a node the same program dlopens librccl) -- both
    ranks end with the output ciphertext of the reference rtlib's CPU run, and limbs did travel"""
    import hashlib

    from test_gpu_batch_shard import _mock_rccl, _run_ranks

    exe = os.path.join(EX_DIR, "model_resnet110_cifar10_train")
    assert os.path.exists(exe), "workloads/_gen/examples/model_resnet110_cifar10_train is missing (make -C workloads)"
    env = dict(_model110_env(), ACEHIP_RCCL_LIB=_mock_rccl(tmp_path))
    assert str(FIX["seed"]) == "20261004"  # (_run_ranks seeds the ranks with it)
    for r, (out, dumps) in enumerate(_run_ranks(exe, ["1"], 2, env, tmp_path, "r110mp2", timeout=1500)):
        got = {k: hashlib.sha256(v).hexdigest() for k, v in dumps.items()}
        assert got == MODEL110["outputs"], "rank %d: ResNet-110 output differs from the reference rtlib's CPU run" % r
        line = [ln for ln in out.splitlines() if "limb exchanges:" in ln]
        assert line and int(line[0].split("limb exchanges:")[1].split()[0]) > 0 and "(simulated)" not in out, out[-1500:]


# ------------------------------------------------------------------------------------------------------------------------------
# The other generated programs of the reference's dataset directory (rtlib/ant/dataset/resnet{32,44,56}_cifar10_pre.onnx.inc,
# resnet32_cifar100_pre.onnx.inc, unchanged): one image each against the reference rtlib's CPU run with the injected key set
# (0.9 / 1.3 / 1.6 h of one core each in the dev container; sigma per depth so that the logits keep real digits,
# profiles/r04ae_sigma_sweep_more_models.txt).  A model without a fixture entry fails.
# ------------------------------------------------------------------------------------------------------------------------------
OTHER_MODELS = ["resnet32", "resnet32c100", "resnet44", "resnet56"]


@pytest.mark.parametrize("key", OTHER_MODELS)
def test_other_dataset_model_is_byte_identical_to_the_reference_cpu_run(key, tmp_path):
    import model_weights

    m = FIX.get("models", {}).get(key)
    assert m is not None, "tests/golden/gen_parity.json has no %s entry (tests/golden/gen_gen_parity.py %s)" % (key, key)
    exe = os.path.join(EX_DIR, "model_" + model_weights.PROGRAM[key])
    wfile = _weights_of(key, m)
    env = {"ACEHIP_RT_DATA_FILE": wfile, "MODEL_DATA_FILE": wfile, "MODEL_ENC_SEED": str(m["enc_seed"])}
    out, got = _run(exe, ["1"], env, tmp_path, key, timeout=1500)
    assert got == m["outputs"], "%s output ciphertext differs from the reference rtlib's CPU run" % key
    assert _logits9(out) == [["%.9f" % v for v in m["logits9"]]]
    assert max(abs(v) for v in m["logits9"]) > 0.02   # real digits


# ------------------------------------------------------------------------------------------------------------------------------
# The part of the reference's rt_ant surface that no checked-in generated program calls (tests/c/api_extras.c): Upscale_ciph /
# Downscale_ciph, the with-scale encoders, the message-level validation helpers, the diagnostics -- same bytes, same text.
# ------------------------------------------------------------------------------------------------------------------------------
EXTRAS = FIX.get("api_extras", {})


@pytest.mark.parametrize("name", sorted(EXTRAS) or ["(no fixture)"])
def test_api_extras_match_the_reference(name, tmp_path):
    assert EXTRAS, "tests/golden/gen_parity.json has no api_extras entry (tests/golden/gen_gen_parity.py extras)"
    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = os.path.join(str(tmp_path), "api_extras")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", os.path.join(ROOT, "tests", "c", "api_extras.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
                           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe])
    d = tmp_path / "out"
    d.mkdir()
    want = EXTRAS[name]
    r = subprocess.run([exe, str(d)] + want["args"].split(), capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ACEHIP_SEED=str(FIX["seed"])))
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    got = {os.path.basename(p)[:-3]: _sha(p) for p in sorted(glob.glob(str(d) + "/*.ct"))}
    assert got == want["files"]
    assert open(os.path.join(str(d), "text.txt")).read().splitlines() == want["text"]
    assert [ln for ln in r.stdout.splitlines() if "internal validation" in ln] == want["validate_stdout"]
    assert [ln for ln in r.stderr.splitlines() if ln.startswith(("ERROR: validation", "idx:", "res:", "std:"))] == want["validate_stderr"]
