"""Drop-in acceptance on the GPU (-m gpu): the reference's checked-in ACE-generated example programs
(rtlib/ant/example/eg_fhertlib_*.c + .inc; registered as ctest cases in rtlib/ant/CMakeLists.txt:79-92)
compiled UNCHANGED against our headers and libFHErt_ant.so by `make -C workloads examples` (dev container,
outputs in workloads/_gen/examples, which travels to the GPU box).  Each program embeds its expected
output and prints SUCESS! when |out - expected| < 1e-3 (eg_fhertlib_relin.c:16-17,59-60)."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

EX_DIR = os.path.join(ROOT, "workloads", "_gen", "examples")
EXAMPLES = ["add", "add_const", "mul_const", "rotate", "rotate_02", "relin", "relin_02", "gemm", "gemm_02", "conv2d",
            "avg_pool", "relu", "bootstrap", "bootstrap_02"]


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_generated_example(name):
    exe = os.path.join(EX_DIR, "eg_" + name)
    if not os.path.exists(exe):
        pytest.fail("workloads/_gen/examples not built (needs /root/reference: make -C workloads examples) -- build outputs of the dev container that must travel with the snapshot")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SUCESS!" in r.stdout
    # stdout contract parsed by the reference's scripts/perf.py:233-276
    assert "ckks_param: _provider = 0, _poly_degree = " in r.stdout
    assert "Total memory size for keys: rot_key_cnt =" in r.stdout


def test_own_program_config_c1(tmp_path):
    """BASELINE configs[0] through the drop-in API with a program of our own (tests/c/dropin_c1.c): HAdd spelled
    per limb, HMul + relinearise, Rescale, Rotate, plaintext multiply at N=2^14, 4 limbs; compiled here with gcc
    against include/ and libFHErt_ant.so exactly like a generated program would be."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "dropin_c1")
    inc = os.path.join(ROOT, "include")
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c", "dropin_c1.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_deferred_zero_fills_are_never_seen_late(tmp_path):
    """tests/c/lazy_fills.c: accumulators filled before a key-switch, a zero ciphertext as the input of a key-switch, a
    zero-filled ciphertext freed while its fill waits (block reuse), per-limb adds on a zero-filled polynomial -- correct
    against the clear computation, and bit-identical slot by slot with the deferral switched off (same ACEHIP_SEED)."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "lazy_fills")
    inc = os.path.join(ROOT, "include")
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c", "lazy_fills.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    outs = {}
    for lazy, discard in (("1", "1"), ("0", "1"), ("0", "0")):
        env = dict(os.environ, ACEHIP_SEED="12345", ACEHIP_LAZY_ZERO=lazy, ACEHIP_HW_DISCARD=discard)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        outs[(lazy, discard)] = [ln for ln in r.stdout.splitlines() if ln.startswith("slot ")]
        assert len(outs[(lazy, discard)]) == 61
    assert outs[("1", "1")] == outs[("0", "1")] == outs[("0", "0")]


def test_ops_kept_queued_across_declared_launches_are_never_misordered(tmp_path):
    """tests/c/keep_queue.c: generated-style code whose per-limb ops stay queued while later key-switches / rescales run (rt_poly.cpp
    "keeping ops queued"): accumulations across rotations, write-after-read / read-after-write / write-after-write against direct
    launches, readers of blocks freed in the meantime.  Correct against the clear computation; bit-identical slot by slot with the
    mechanism off (ACEHIP_HW_KEEP=0), with the library's stage order off (ACEHIP_HW_STAGES=0) and under ACEHIP_POISON=1; ops really
    do stay queued; and an operand left out of a declared list on purpose (ACEHIP_POISON_SELFTEST=1) aborts under poison."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "keep_queue")
    inc = os.path.join(ROOT, "include")
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c", "keep_queue.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    outs = {}
    for tag, extra in (("keep", {}), ("off", {"ACEHIP_HW_KEEP": "0"}), ("nostage", {"ACEHIP_HW_STAGES": "0"}),
                       ("all_off", {"ACEHIP_HW_KEEP": "0", "ACEHIP_HW_STAGES": "0", "ACEHIP_LAZY_ZERO": "0", "ACEHIP_HW_DISCARD": "0"}),
                       ("poison", {"ACEHIP_POISON": "1"}), ("batch3", {"ACEHIP_BATCH": "3"})):
        env = dict(os.environ, ACEHIP_SEED="99", ACEHIP_PROFILE="1", **extra)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "SUCESS!" in r.stdout, tag + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        outs[tag] = [ln for ln in r.stdout.splitlines() if ln.startswith("slot ")]
        assert len(outs[tag]) == 64 - 5 - 4
        kept = [ln for ln in r.stdout.splitlines() if "queue kept open" in ln]
        assert kept, r.stdout[-1500:]
        n_kept = int(kept[0].split("(")[1].split()[0])
        assert (n_kept > 0) == (extra.get("ACEHIP_HW_KEEP") != "0"), kept[0]
    assert outs["keep"] == outs["off"] == outs["nostage"] == outs["all_off"] == outs["poison"] == outs["batch3"]
    # the safety net: one input of the paired Mod_down missing from its declared list -> the ops that produce it stay queued, the
    # launch reads memory nobody wrote, and the poison check names the undeclared range and aborts
    env = dict(os.environ, ACEHIP_SEED="99", ACEHIP_POISON="1", ACEHIP_POISON_SELFTEST="1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not declare" in r.stderr, r.stdout[-1000:] + r.stderr[-2000:]


def test_raised_digits_are_never_reused_stale(tmp_path):
    """tests/c/modup_reuse.c: rotations spelled at the polynomial level like the generated Rotate(); the same ciphertext
    rotated three times (digits raised once), then changed in place by queued ops, rewritten at the same address by direct
    launches, freed and its memory reused -- correct against the clear computation and bit-identical, slot by slot, with
    ACEHIP_MODUP_REUSE=0 (digits raised for every rotation, as the reference does)."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "modup_reuse")
    inc = os.path.join(ROOT, "include")
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c", "modup_reuse.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    outs, raised = {}, {}
    for reuse in ("1", "0"):
        env = dict(os.environ, ACEHIP_SEED="4711", ACEHIP_MODUP_REUSE=reuse, ACEHIP_PROFILE="1")
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        outs[reuse] = [ln for ln in r.stdout.splitlines() if ln.startswith("slot ")]
        assert len(outs[reuse]) == 61
        stat = [ln for ln in r.stdout.splitlines() if "all-digit ModUp" in ln]
        assert stat, r.stdout[-1500:]
        raised[reuse] = [int(t) for t in stat[0].replace(",", " ").split() if t.isdigit()]
    assert outs["1"] == outs["0"]
    # 7 rotations: with reuse only the second and third tap of case 1 are served from digits raised before
    assert raised["0"] == [7, 0] and raised["1"] == [5, 2], raised


# (The end-to-end comparison of the generated ResNet-20 with the reference's CPU run -- byte-identical output ciphertext with injected
# keys, logits to CKKS precision with independent keys, a negative control -- lives in tests/test_gpu_gen_parity.py.)


@pytest.mark.parametrize("cfg", ["4096 33 51 50 3 192 2048 15", "4096 33 51 48 3 192 2048 15", "4096 33 51 48 3 192 512 17"],
                         ids=["delta50_full", "delta48_full", "delta48_sparse"])
def test_bootstrap_at_generated_model_parameters(tmp_path, cfg):
    """Bootstrap at the prime sizes the generated ResNets use (q0=51 with Delta=50: ResNet-20/32, Delta=48: ResNet-110),
    fully and sparsely packed, through tests/c/bootstrap_params.c (encrypt, burn levels down to 2 limbs, Bootstrap,
    decrypt, compare with the reference examples' tolerance 1e-3).  The reference's own bootstrap examples only cover
    q0=60/Delta=51 at N=16."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "bootstrap_params")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", os.path.join(ROOT, "tests", "c", "bootstrap_params.c"), "-I", inc, "-I",
                           os.path.join(inc, "rt_ant"), "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm",
                           "-Wl,-rpath," + bmod.LIBDIR, "-o", exe])
    r = subprocess.run([exe] + cfg.split(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_style_openmp_main_shares_one_context():
    """The reference's model main prepares the context once and runs Main_graph from an OpenMP loop, one image per thread
    (rtlib/ant/dataset/resnet_cifar.main.inc:77-116).  tools/model_main_omp.c has that structure around the UNCHANGED
    generated ResNet-20: worker threads attach to the prepared context (shared keys; own scratch, pool, queue and HIP stream).
    Every thread processes the same image, so every line of logits must be the same."""
    import re

    exe = os.path.join(EX_DIR, "modelomp_resnet20_cifar10_pre")
    if not os.path.exists(exe):
        pytest.fail("workloads/_gen/examples/modelomp_* not built (needs /root/reference: make -C workloads models) -- build outputs of the dev container that must travel with the snapshot")
    env = dict(os.environ, OMP_NUM_THREADS="3", GPU_MAX_HW_QUEUES="8", ACEHIP_RT_DATA_SYNTH="1")
    r = subprocess.run([exe, "6"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = re.findall(r"\[MODEL\] image \d+: logits:((?: -?\d+\.\d+)+)", r.stdout)
    assert len(rows) == 6
    vals = [[float(x) for x in row.split()] for row in rows]
    for v in vals[1:]:
        assert max(abs(a - b) for a, b in zip(v, vals[0])) <= 2e-4, vals
    assert "rot_key_cnt = 227," in r.stdout  # the keys exist once, not once per thread


def test_static_archive_program_and_timing_table(tmp_path):
    """The reference link line against libFHErt_ant.a / libFHErt_common.a (scripts/perf.py:202-207), run on the GPU with
    RTLIB_TIMING_OUTPUT=stdout (perf.py:164): the program must succeed and Finalize_context must print the per-function
    table in the layout perf.py:251-260 walks (header 'RTLib functions', item lines '<name>\t<count>\t<sec> sec', a
    'MAIN_GRAPH ... sub total' line closing the nested items)."""
    import re
    import sys

    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    inc = os.path.join(ROOT, "include")
    exe = str(tmp_path / "bootstrap_static")
    subprocess.check_call(["cc", "-O1", os.path.join(ROOT, "tests", "c", "bootstrap_params.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
                           bmod.RT_ARCHIVE, bmod.RT_COMMON_ARCHIVE, "-lm", "-o", exe])
    env = dict(os.environ, RTLIB_TIMING_OUTPUT="stdout")
    r = subprocess.run([exe] + "1024 33 51 50 3 192 512 15".split(), capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    lines = r.stdout.splitlines()
    start = next(i for i, l in enumerate(lines) if "RTLib functions" in l)
    assert re.match(r"RTLib functions\s+Count\s+Elapse", lines[start]) and lines[start + 1].startswith("-" * 20)
    items = {}
    for l in lines[start + 2:]:
        m = re.match(r"^( *)([A-Z_]+)\s+(\d+|sub total)\s+([0-9.]+) sec$", l)
        if not m:
            break
        items.setdefault(m.group(2), []).append((len(m.group(1)), m.group(3), float(m.group(4))))
    for name in ("PREPARE_CONTEXT", "FINALIZE_CONTEXT", "MAIN_GRAPH", "BOOTSTRAP", "BS_EVAL", "BS_COEFF_TO_SLOT", "BS_APPROX_MOD",
                 "BS_SLOT_TO_COEFF", "ENCODE_ARRAY"):
        assert name in items, (name, r.stdout[-3000:])
    assert items["BOOTSTRAP"][0][:2] == (1, "1") and items["BS_EVAL"][0][0] == 2 and items["BS_APPROX_MOD"][0][0] == 3
    assert ("sub total" in [e[1] for e in items["MAIN_GRAPH"]])          # what perf.py looks for to end the table
    assert items["BS_EVAL"][0][2] <= items["BOOTSTRAP"][0][2] * 1.05      # nested time is part of its parent's


def test_resnet110_workload_runs_on_one_gpu():
    """BASELINE configs[4]'s workload (the UNCHANGED ACE-generated ResNet-110 source, resnet110_cifar10_train.onnx.inc: N = 2^16,
    Delta = 2^48, 109 bootstraps) on ONE GPU with synthetic weights: the 8-GPU limb-sharded run cannot be made on a test box,
    but the program itself must execute and its bookkeeping must be the reference's -- 227 rotation keys, one weight plaintext
    per Pt_from_msg call site (36 464, tests/golden/resnet110_pt_entries.txt), ten finite logits.  (Synthetic N(0, 0.05)
    weights drive a 110-layer network out of its numeric range on the reference's CPU run too: the values are not compared.)"""
    import math
    import re

    exe = os.path.join(EX_DIR, "model_resnet110_cifar10_train")
    if not os.path.exists(exe):
        pytest.fail("workloads/_gen/examples/model_* not built (needs /root/reference: make -C workloads models) -- build outputs of the dev container that must travel with the snapshot")
    env = dict(os.environ, ACEHIP_RT_DATA_SYNTH="1")
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "_poly_degree = 65536" in r.stdout and "_scaling_mod_size = 48" in r.stdout and "_num_rot_idx = 197" in r.stdout
    assert "rot_key_cnt = 227," in r.stdout
    n_entries = sum(1 for _ in open(os.path.join(ROOT, "tests", "golden", "resnet110_pt_entries.txt")))
    assert "Total memory size for weight plain: cnt = %d," % n_entries in r.stdout
    m = re.search(r"logits:((?: -?\d+\.\d+)+)", r.stdout)
    assert m, r.stdout[-2000:]
    vals = [float(x) for x in m.group(1).split()]
    assert len(vals) == 10 and all(math.isfinite(v) for v in vals)


def test_bench_resnet110_workload_line():
    """`bench.py --workload resnet110` (the secondary measurement of the configs[4] network as replicas): one image on one stream
    ends in a JSON line that names the workload."""
    import json
    import sys

    lib = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet110.so")
    if not os.path.exists(lib):
        pytest.fail("workloads/_gen/models not built (needs /root/reference: tools/build_models.py) -- build outputs of the dev container that must travel with the snapshot")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "resnet110", "--no-cpu-baseline", "--streams", "1",
                        "--batch", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["unit"] == "images/s" and d["value"] > 0 and "ResNet-110" in d["metric"] and "REPLICAS" in d["config"]["workload"]
    # the timed image is the fixture's: its output ciphertext hashes to the digest of the reference rtlib's own 2.6-hour CPU run
    if d["verification"].get("note") is None:
        assert d["verified"] is True, d["verification"]


@pytest.mark.parametrize("name", ["add", "add_const", "mult_const", "conv2d_keep_shape"])
def test_ckks_level_provider_programs(name):
    """SURVEY 8f-2, the provider-level (ciphertext-granular) API: the reference's programs generated for the CKKS-level provider
    interface (rtlib/seal/example/eg_rtseal_*.cxx + .inc, `#include "rt_seal/rt_seal.h"`: Add_ciph / Add_plain / Mul_plain /
    Rotate_ciph / Copy_ciph / Encode_plain_from_float ... on whole ciphertexts) compiled UNCHANGED against
    include/rt_seal/rt_seal.h -> include/rt_acehip/rt_acehip.h and libFHErt_ant (make -C workloads provider).  Each embeds its
    expected output (tolerance 1e-2) and prints SUCCESS!."""
    exe = os.path.join(EX_DIR, "egseal_" + name)
    if not os.path.exists(exe):
        pytest.fail("workloads/_gen/examples/egseal_* not built (needs /root/reference: make -C workloads provider) -- build outputs of the dev container that must travel with the snapshot")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SUCCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_mod_down_pair_on_a_queued_key_inner_product(tmp_path):
    """tests/c/kmac_pair.c: generated-style rotations at N = 2^16 whose Mod_down pair runs on the raised digits and key parts themselves while the
    products and additions that would have filled the accumulators stay queued (rt_poly.cpp keymac_pair_from_queue; the accumulators are never
    stored): plain rotations, accumulators read afterwards, a digit rewritten behind the sums, fills already executed, a preloaded accumulator,
    an operand rewritten between a product and its addition.  Correct against the clear computation; bit-identical slot by slot with the
    shortcut off (ACEHIP_KMAC_SHIM=0), with the stored-accumulator kernels (ACEHIP_KMAC_FUSE=0), without kept ops, under ACEHIP_POISON=1 and
    with three images per launch; the shortcut is taken exactly where the queue proves it right."""
    import ace_compiler_amd  # noqa: F401
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path / "kmac_pair")
    inc = os.path.join(ROOT, "include")
    cmd = ["gcc", "-O1", os.path.join(ROOT, "tests", "c", "kmac_pair.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    outs, stats = {}, {}
    for tag, extra in (("shim", {"ACEHIP_KMAC_FUSE": "2"}), ("off", {"ACEHIP_KMAC_FUSE": "2", "ACEHIP_KMAC_SHIM": "0"}),
                       ("stored", {"ACEHIP_KMAC_FUSE": "0"}), ("keep_off", {"ACEHIP_KMAC_FUSE": "2", "ACEHIP_HW_KEEP": "0"}),
                       ("poison", {"ACEHIP_KMAC_FUSE": "2", "ACEHIP_POISON": "1"}), ("batch3", {"ACEHIP_BATCH": "3"})):
        # (ACEHIP_HW_KEEP_RUN: no periodic full hand-over, which would execute some accumulator fills early and make the counts below
        #  depend on where in the program it falls)
        env = dict(os.environ, ACEHIP_SEED="77", ACEHIP_PROFILE="1", ACEHIP_HW_KEEP_RUN="1000", **extra)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "SUCESS!" in r.stdout, tag + ": " + r.stdout[-2000:] + r.stderr[-2000:]
        outs[tag] = [ln for ln in r.stdout.splitlines() if ln.startswith("slot ")]
        assert len(outs[tag]) == 60
        line = [ln for ln in r.stdout.splitlines() if "Mod_down pairs behind a queued key inner product" in ln]
        assert line, r.stdout[-1500:]
        stats[tag] = [int(t) for t in line[0].replace(",", " ").replace(";", " ").replace(":", " ").split() if t.isdigit()]
    assert outs["shim"] == outs["off"] == outs["stored"] == outs["keep_off"] == outs["poison"] == outs["batch3"]
    # 8 pairs.  The three plain rotations and the one whose accumulators are read afterwards can take the shortcut; one finds its fills executed
    # (Acehip_rt_sync), three find a queue that does not prove the sums (digit rewritten, accumulator preloaded, operand rewritten before
    # the addition).  A full hand-over forced by the pool's bound on pinned blocks may execute further fills early: such a pair counts as
    # "fill already executed" whatever else is wrong with it, so only the totals are fixed.
    for tag in ("shim", "poison", "batch3"):
        tried, fused, no_zero, shape, other = stats[tag]
        # (observed: 8 / 4 / 2 / 0 / 2; the bounds leave room for a hand-over falling elsewhere -- the bit-identical slots above are the check)
        assert tried == 8 and fused + no_zero + shape + other == 8 and 1 <= fused <= 4 and no_zero >= 1 and shape == 0, stats
    assert stats["off"][1] == 0 and stats["stored"][1] == 0 and stats["keep_off"][1] == 0, stats
