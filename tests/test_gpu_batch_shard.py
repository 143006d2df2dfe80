"""Image batches and limb-sharded execution behind the rt_ant API, on the GPU (-m gpu).

Both are execution modes of the runtime under UNCHANGED generated programs (the reference's checked-in example programs and
ResNet sources, compiled by `make -C workloads`), so the check is the strongest one available: with the same keys and the same
encryption randomness (ACEHIP_SEED) every output CIPHERTEXT must be byte-identical to the plain run's -- written by the
ACEHIP_DUMP_OUTPUT hook of Set_output_data (csrc/rt/rt_io.cpp) in the ACEHCT01 container.

  * image batches (Acehip_rt_set_batch, BASELINE configs[3] throughput): B images per launch, keys / twiddles / bootstrap tables /
    weight plaintexts shared -- the GPU form of the reference's image loop (rtlib/ant/dataset/resnet_cifar.main.inc:77-116);
  * limb-sharded execution (BASELINE configs[4]): ACEHIP_SHARD_SIM=G runs G ranks on one GPU with separate per-rank buffers, the
    exchanges at Decomp_modup / Mod_down / Rescale / ModRaise / decode being device copies (RCCL broadcasts between processes
    otherwise: same code path above the exchange primitive).
"""
import glob
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

EX_DIR = os.path.join(ROOT, "workloads", "_gen", "examples")


def _run(exe, args, env_extra, tmp, tag, timeout=900):
    prefix = os.path.join(str(tmp), tag)
    env = dict(os.environ, ACEHIP_SEED="20261004", ACEHIP_DUMP_OUTPUT=prefix, **env_extra)
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dumps = sorted(glob.glob(prefix + ".*"))
    assert dumps, "no output ciphertext was dumped"
    return r.stdout, {os.path.basename(p)[len(tag) + 1:]: open(p, "rb").read() for p in dumps}


def _need(exe):
    if not os.path.exists(exe):
        pytest.fail("workloads/_gen/examples not built (needs /root/reference: make -C workloads) -- build outputs of the dev container that must travel with the snapshot")


@pytest.mark.parametrize("name", ["rotate", "relin", "conv2d", "bootstrap", "bootstrap_02"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_reference_example_sharded_is_bit_identical(name, world, tmp_path):
    """eg_fhertlib_* unchanged: G simulated ranks produce the output ciphertext of the unsharded run, byte for byte."""
    exe = os.path.join(EX_DIR, "eg_" + name)
    _need(exe)
    out0, plain = _run(exe, [], {}, tmp_path, "plain")
    out1, shard = _run(exe, [], {"ACEHIP_SHARD_SIM": str(world), "ACEHIP_PROFILE": "1"}, tmp_path, "shard%d" % world)
    assert "SUCESS!" in out0 and "SUCESS!" in out1
    assert plain.keys() == shard.keys()
    for k in plain:
        assert plain[k] == shard[k], "output ciphertext %s differs between the plain and the %d-rank run" % (k, world)
    assert "limb-sharded world %d (simulated)" % world in out1
    if name in ("rotate", "relin", "bootstrap", "bootstrap_02"):  # a key-switch exchanges limbs
        line = [ln for ln in out1.splitlines() if "limb exchanges:" in ln]
        assert line and int(line[0].split("limb exchanges:")[1].split()[0]) > 0, out1[-2000:]


def _logit_lines(stdout):
    return [ln.split("logits:")[1].split() for ln in stdout.splitlines() if "logits:" in ln]


def test_resnet20_image_batches_are_bit_identical(tmp_path):
    """The ACE-generated ResNet-20 (unchanged source, tools/model_main.c): 4 images one by one, as two batches of 2 and as one
    batch of 4 -- every image's output ciphertext byte-identical, and the printed logits with it."""
    exe = os.path.join(EX_DIR, "model_resnet20_cifar10_pre")
    _need(exe)
    env = {"ACEHIP_RT_DATA_SYNTH": "1"}
    out1, d1 = _run(exe, ["4"], env, tmp_path, "b1", timeout=1200)
    out2, d2 = _run(exe, ["4"], dict(env, MODEL_BATCH="2"), tmp_path, "b2", timeout=1200)
    out4, d4 = _run(exe, ["4"], dict(env, MODEL_BATCH="4"), tmp_path, "b4", timeout=1200)
    # one by one: call i, image 0; batches of 2: call i // 2, image i % 2; one batch of 4: call 0, image i
    for i in range(4):
        ref = d1["%d.0" % i]
        assert d2["%d.%d" % (i // 2, i % 2)] == ref, "image %d differs in batches of 2" % i
        assert d4["0.%d" % i] == ref, "image %d differs in a batch of 4" % i
    assert _logit_lines(out1) == _logit_lines(out2) == _logit_lines(out4) and len(_logit_lines(out1)) == 4


# (One image of the ACE-generated ResNet-110 -- BASELINE configs[4] -- unsharded and over 2 and 8 simulated ranks is checked against the REFERENCE's
# own output ciphertext since round 4: tests/test_gpu_gen_parity.py::test_resnet110_output_ciphertext_is_byte_identical_to_the_reference_cpu_run.)


def _mock_rccl(tmp):
    """tests/c/mock_rccl.c: RCCL's entry points on shared memory, so that the processes of a one-GPU box can be ranks"""
    so = os.path.join(str(tmp), "libmock_rccl.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c", "mock_rccl.c"),
                           "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib", "-o", so])
    return so


def _run_ranks(exe, args, world, env_extra, tmp, tag, timeout=1500, one_gpu_per_rank=False):
    """the program once per rank, concurrently, as a launcher would start it (ACEHIP_SHARD=1 + RANK / WORLD_SIZE / LOCAL_RANK);
    every process uses GPU 0 and exchanges limbs through the library named by ACEHIP_RCCL_LIB -- or, with one_gpu_per_rank, rank r
    uses GPU r and the real librccl"""
    port = str(29600 + os.getpid() % 300)
    id_file = os.path.join(str(tmp), tag + ".rccl_id")
    procs = []
    for r in range(world):
        prefix = os.path.join(str(tmp), "%s_r%d" % (tag, r))
        env = dict(os.environ, ACEHIP_SEED="20261004", ACEHIP_DUMP_OUTPUT=prefix, ACEHIP_SHARD="1", RANK=str(r), WORLD_SIZE=str(world),
                   LOCAL_RANK=str(r) if one_gpu_per_rank else "0", MASTER_PORT=port, ACEHIP_SHARD_ID_FILE=id_file, ACEHIP_PROFILE="1", **env_extra)
        procs.append((prefix, subprocess.Popen([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)))
    outs = []
    for prefix, p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()
            raise
        assert p.returncode == 0, so[-3000:] + se[-3000:]
        tagr = os.path.basename(prefix)
        dumps = sorted(glob.glob(prefix + ".*"))
        assert dumps, "rank wrote no output ciphertext"
        outs.append((so, {os.path.basename(d)[len(tagr) + 1:]: open(d, "rb").read() for d in dumps}))
    return outs


def _exchange_counts(out):
    """(steps, limbs, collectives) of the '[ACEHIP] limb exchanges:' line of a profiled run"""
    line = [ln for ln in out.splitlines() if "limb exchanges:" in ln]
    assert line, out[-2000:]
    w = line[0].split("limb exchanges:")[1].replace(",", " ").split()
    return int(w[0]), int(w[2]), int(w[w.index("collectives") - 1])


@pytest.mark.parametrize("name,world,packed", [("rotate", 2, "1"), ("relin", 3, "1"), ("bootstrap", 2, "1"), ("bootstrap_02", 3, "1"), ("rotate", 2, "0"),
                                               ("bootstrap", 3, "0")])
def test_reference_example_sharded_over_processes_is_bit_identical(name, world, packed, tmp_path):
    """The multi-process form (one rank per process, what `torchrun` starts per GPU): ACEHIP_SHARD=1, rank 0 publishes the
    communicator id, every rank joins and the exchanges of Decomp_modup / Mod_down / Rescale / ModRaise / decode go through the
    RCCL entry points on the exchange stream.  Round 6: every exchange step is ONE collective -- the owned limbs packed into a staging
    block, one ncclAllGather (one ncclBroadcast when a single rank owns the step's limbs), the other ranks' limbs copied out on arrival
    (packed = "1", the default; asserted: collectives == steps); packed = "0" keeps the round-5 form, one grouped ncclBroadcast per limb
    in place (collectives == limbs moved, sent or received).  A test box has one GPU and RCCL refuses two ranks on one device, so the entry points
    are served by tests/c/mock_rccl.c (shared memory; ranks must issue identical sequences or the run fails).  Every rank's output
    ciphertext equals the unsharded run's, byte for byte."""
    exe = os.path.join(EX_DIR, "eg_" + name)
    _need(exe)
    mock = _mock_rccl(tmp_path)
    out0, plain = _run(exe, [], {}, tmp_path, "plain")
    ranks = _run_ranks(exe, [], world, {"ACEHIP_RCCL_LIB": mock, "ACEHIP_SHARD_PACKED": packed}, tmp_path, "mp%d" % world)
    for r, (out, dumps) in enumerate(ranks):
        assert "SUCESS!" in out, out[-2000:]
        assert dumps.keys() == plain.keys()
        for k in plain:
            assert dumps[k] == plain[k], "rank %d of %d: output ciphertext %s differs from the unsharded run" % (r, world, k)
        assert "limb-sharded world %d" % world in out and "(simulated)" not in out
        steps, limbs, coll = _exchange_counts(out)
        assert steps > 0 and limbs > 0
        assert (coll == steps) if packed == "1" else (coll > steps and coll >= limbs), (steps, limbs, coll)


def test_resnet20_sharded_over_two_processes_is_bit_identical(tmp_path):
    """One image of the ACE-generated ResNet-20 with its limbs spread over two PROCESSES (see above for the exchange library):
    both ranks' output ciphertexts equal the unsharded run's."""
    exe = os.path.join(EX_DIR, "model_resnet20_cifar10_pre")
    _need(exe)
    mock = _mock_rccl(tmp_path)
    env = {"ACEHIP_RT_DATA_SYNTH": "1"}
    _, plain = _run(exe, ["2"], env, tmp_path, "plain", timeout=1200)
    ranks = _run_ranks(exe, ["1"], 2, dict(env, ACEHIP_RCCL_LIB=mock), tmp_path, "mp2")
    for r, (out, dumps) in enumerate(ranks):
        assert dumps["0.0"] == plain["0.0"], "rank %d: ResNet-20 output differs from the unsharded run" % r
        steps, limbs, coll = _exchange_counts(out)
        assert coll == steps and limbs > 3 * steps, (steps, limbs, coll)  # one collective per exchange step, several limbs per step
    # owner-only key limbs (opt-in, ACEHIP_SHARD_OWNER_LIMBS=1: HIP virtual memory management, include/acehip.h acehip_malloc_limbs): every
    # switch key keeps the reference's full layout in ADDRESSES, but only the limbs a rank owns are backed by memory of their own --
    # same bytes out, and the rank reports about half of the key limbs backed (two ranks)
    ranks = _run_ranks(exe, ["1"], 2, dict(env, ACEHIP_RCCL_LIB=mock, ACEHIP_SHARD_OWNER_LIMBS="1"), tmp_path, "mp2own")
    for r, (out, dumps) in enumerate(ranks):
        assert dumps["0.0"] == plain["0.0"], "rank %d: ResNet-20 output differs with owner-only key limbs" % r
        line = [ln for ln in out.splitlines() if "switch-key memory on this rank:" in ln]
        assert line, out[-2000:]
        w = line[0].split("rank:")[1].split()
        backed, addressed = float(w[0]), float(w[6])
        assert "owner-only" in line[0] and 0.4 * addressed < backed < 0.6 * addressed, line[0]
    # sharded AND batched: two images per launch on every rank (every exchange then moves both images' limbs)
    ranks = _run_ranks(exe, ["2"], 2, dict(env, ACEHIP_RCCL_LIB=mock, MODEL_BATCH="2"), tmp_path, "mp2b2")
    for r, (out, dumps) in enumerate(ranks):
        for i in range(2):
            assert dumps["0.%d" % i] == plain["%d.0" % i], "rank %d: image %d of a sharded batch differs from its unsharded single run" % (r, i)


def _gpu_count():
    import torch

    return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)


@pytest.mark.skipif(_gpu_count() < 2, reason="needs two GPUs: real RCCL refuses two ranks on one device (the one-GPU test box runs the "
                                              "same programs over tests/c/mock_rccl.c above)")
def test_two_real_rccl_ranks_are_bit_identical(tmp_path):
    """The day a node is there: eg_rotate, eg_bootstrap and one ResNet-20 image with their limbs spread over TWO GPUs, one process each,
    the exchanges of Decomp_modup / Mod_down / Rescale / ModRaise / decode as real ncclBroadcast calls over xGMI (root != self for half
    of the limbs) -- every rank's output ciphertext equals the unsharded run's, and bytes were received."""
    for name in ("rotate", "bootstrap"):
        exe = os.path.join(EX_DIR, "eg_" + name)
        _need(exe)
        _, plain = _run(exe, [], {}, tmp_path, "plain_" + name)
        for r, (out, dumps) in enumerate(_run_ranks(exe, [], 2, {}, tmp_path, "rccl2_" + name, one_gpu_per_rank=True)):
            assert "SUCESS!" in out and dumps == plain, "rank %d: eg_%s differs from the unsharded run" % (r, name)
            line = [ln for ln in out.splitlines() if "limb exchanges:" in ln]
            assert line and int(line[0].split("limb exchanges:")[1].split()[0]) > 0, out[-2000:]
    exe = os.path.join(EX_DIR, "model_resnet20_cifar10_pre")
    _need(exe)
    env = {"ACEHIP_RT_DATA_SYNTH": "1"}
    _, plain = _run(exe, ["1"], env, tmp_path, "plain_r20", timeout=1200)
    for r, (out, dumps) in enumerate(_run_ranks(exe, ["1"], 2, env, tmp_path, "rccl2_r20", one_gpu_per_rank=True)):
        assert dumps["0.0"] == plain["0.0"], "rank %d: ResNet-20 over two GPUs differs from the unsharded run" % r


def test_bench_shard_mode_one_rank_joins_rccl():
    """`bench.py --mode shard` (what torchrun starts once per GPU for BASELINE configs[4]) with a single rank: the rt_ant shim
    loads librccl, creates and joins its own communicator (ACEHIP_SHARD=1: id file, ncclCommInitRank), runs the generated
    ResNet-20 through the API and prints the JSON line with the rank count RCCL saw, the limbs each rank owns and the
    exchange counters (all zero: one rank owns every limb)."""
    import json
    import sys

    lib = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet20.so")
    if not os.path.exists(lib):
        pytest.fail("workloads/_gen/models not built (needs /root/reference: tools/build_models.py) -- build outputs of the dev container that must travel with the snapshot")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "shard", "--workload", "resnet20", "--batch", "2",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["unit"] == "images/s" and d["value"] > 0
    assert d["shard"]["rccl_ranks"] == 1 and d["shard"]["owned_limbs_per_rank"] == [45] and d["shard"]["bytes_received_per_image_all_ranks"] == 0


def test_bench_limb_sharded_leg_runs_after_the_headline_line():
    """`ACEHIP_BENCH_SHARD_LEG=1 bench.py --gpus N` (N > 1): AFTER the headline line is printed the ranks run one ResNet-20 image
    limb-sharded as child processes and report on stderr (never in the headline line, which a hang of the leg can then no longer
    delay).  One GPU here, so the leg is forced with a single rank (ACEHIP_BENCH_FORCE_SHARD_LEG=1): child start, environment, JSON
    hand-over, output digest and clean-up are the same code."""
    import json
    import sys

    lib = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet20.so")
    if not os.path.exists(lib):
        pytest.fail("workloads/_gen/models not built (needs /root/reference: tools/build_models.py) -- build outputs of the dev container that must travel with the snapshot")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["ACEHIP_BENCH_FORCE_SHARD_LEG"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--streams", "1", "--batch", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert "limb_sharded" not in d and d["value"] > 0
    line = [ln for ln in r.stderr.splitlines() if ln.startswith("[bench] limb_sharded ")]
    assert line, r.stderr[-3000:]
    leg = json.loads(line[-1][len("[bench] limb_sharded "):])
    assert "error" not in leg, leg
    assert leg["ranks_succeeded"] and leg["output_ciphertexts_identical_on_all_ranks"] and leg["images_per_s"] > 0
    assert leg["shard"]["rccl_ranks"] == 1 and len(leg["last_logits"]) == 10


@pytest.mark.parametrize("name", ["rotate", "relin", "conv2d", "relu", "bootstrap"])
def test_examples_under_poison_are_bit_identical(name, tmp_path):
    """ACEHIP_POISON=1 (csrc/rt/rt_poly.cpp): every limb whose zero fill is deferred and every block that returns to the pool is
    overwritten with a non-residue, and every launch with a declared operand list is checked against what the library says it
    touched.  A launch that reads an operand its list forgot, or freed memory, would compute with garbage: the output ciphertext of
    the unchanged reference programs must stay byte-identical to the normal run's."""
    exe = os.path.join(EX_DIR, "eg_" + name)
    _need(exe)
    out0, plain = _run(exe, [], {}, tmp_path, "plain")
    out1, pois = _run(exe, [], {"ACEHIP_POISON": "1"}, tmp_path, "poison")
    assert "SUCESS!" in out0 and "SUCESS!" in out1
    assert plain == pois


def test_resnet20_under_poison_is_bit_identical(tmp_path):
    """the same for one image of the generated ResNet-20 (every deferral path of the runtime: 170 k lazy fills, 2.5 M queued limb-ops,
    prefetched weight plaintexts) -- and with an image batch, whose shared blocks go through the same pool"""
    exe = os.path.join(EX_DIR, "model_resnet20_cifar10_pre")
    _need(exe)
    env = {"ACEHIP_RT_DATA_SYNTH": "1"}
    _, plain = _run(exe, ["2"], env, tmp_path, "plain", timeout=1200)
    _, pois = _run(exe, ["2"], dict(env, ACEHIP_POISON="1"), tmp_path, "poison", timeout=1200)
    _, pois_b2 = _run(exe, ["2"], dict(env, ACEHIP_POISON="1", MODEL_BATCH="2"), tmp_path, "poisonb2", timeout=1200)
    assert plain == pois
    assert pois_b2["0.0"] == plain["0.0"] and pois_b2["0.1"] == plain["1.0"]


def test_poison_mode_catches_an_undeclared_operand(tmp_path):
    """the check itself: with ACEHIP_POISON_SELFTEST=1 the paired Mod_down of a rotation leaves one input out of its declared list
    (csrc/rt/rt_poly.cpp Mod_down) -- under ACEHIP_POISON=1 the program must stop at that site instead of running on"""
    exe = os.path.join(EX_DIR, "eg_rotate")
    _need(exe)
    env = dict(os.environ, ACEHIP_POISON="1", ACEHIP_POISON_SELFTEST="1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "does not declare" in r.stderr and "rt_poly.cpp" in r.stderr, r.stdout[-1500:] + r.stderr[-1500:]
