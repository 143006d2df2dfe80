#!/usr/bin/env python3
"""Regenerates tests/golden/gen_parity.json by RUNNING THE REFERENCE rtlib under its own generated programs (dev container only):
oracle/_ref/examples/refgen_<name> = rtlib/ant/example/eg_fhertlib_<name>.c + .inc, unchanged, linked against
oracle/_ref/libref_rtlib.so together with tests/c/gen_parity_ref.c, which injects the key set and the encryption randomness the
PRODUCT derives from ACEHIP_SEED and writes every Set_output_data ciphertext in the product's dump layout (`make -C oracle refgen`).
The fixture holds sha256 digests of those files -- data only.  tests/test_gpu_gen_parity.py runs the same unchanged programs
against libFHErt_ant.so with the same seed (lazy queue on; image batches; poison mode) and must reproduce every digest.

  usage: gen_gen_parity.py [examples] [ksw] [extras] [resnet20] [resnet110]     (default: examples; the models take 0.5 h / 2.2 h of one core and
                                                                 44 GB, and are merged into the existing file)
"""
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden", "gen_parity.json")
EX = os.path.join(ROOT, "oracle", "_ref", "examples")
SEED = 20261004
EXAMPLES = ["add", "add_const", "mul_const", "rotate", "rotate_02", "relin", "relin_02", "gemm", "gemm_02", "conv2d", "avg_pool",
            "relu", "bootstrap", "bootstrap_02"]
MODELS = {"resnet20": ("resnet20_cifar10_pre", "resnet20_pt_entries.txt"), "resnet32": ("resnet32_cifar10_pre", "resnet32_pt_entries.txt"),
          "resnet32c100": ("resnet32_cifar100_pre", "resnet32c100_pt_entries.txt"), "resnet44": ("resnet44_cifar10_pre", "resnet44_pt_entries.txt"),
          "resnet56": ("resnet56_cifar10_pre", "resnet56_pt_entries.txt"), "resnet110": ("resnet110_cifar10_train", "resnet110_pt_entries.txt")}


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def digests(prefix):
    return {os.path.basename(p)[len(os.path.basename(prefix)) + 1:]: sha(p) for p in sorted(glob.glob(prefix + ".*"))}


def run_example(name, batch, skip, tmp):
    prefix = os.path.join(tmp, "%s_%d_%d" % (name, batch, skip))
    env = dict(os.environ, GEN_PARITY_SEED=str(SEED), GEN_PARITY_OUT=prefix, GEN_PARITY_BATCH=str(batch), GEN_PARITY_ENC_SKIP=str(skip))
    r = subprocess.run([os.path.join(EX, "refgen_" + name)], capture_output=True, text=True, env=env, timeout=3600)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    d = digests(prefix)
    assert d, "no output ciphertext"
    return d


KSW_SETS = {"n1024_l7": "1024 6 60 50 3", "n4096_resnet_primes": "4096 33 51 50 3", "n65536_resnet20": "65536 33 51 50 3"}


def run_ksw(args, tmp):
    """tests/c/ksw_variants.c against the reference: digests of the four BASE forms + which OPT forms have the base form's bytes"""
    d = os.path.join(tmp, "ksw_" + args.replace(" ", "_"))
    os.makedirs(d)
    env = dict(os.environ, GEN_PARITY_SEED=str(SEED))
    r = subprocess.run([os.path.join(EX, "refgen_ksw_variants"), d] + args.split(), capture_output=True, text=True, env=env, timeout=3600)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    verdicts = dict(re.findall(r"opt_vs_base (\w+) (EQUAL|DIFFERENT)", r.stdout))
    assert len(verdicts) == 4
    return {"args": args, "base": {os.path.basename(p)[:-3]: sha(p) for p in sorted(glob.glob(d + "/*.ct"))}, "opt_vs_base": verdicts}


EXTRAS_SETS = {"n1024_l7": "1024 6 60 50 3", "n65536_resnet20": "65536 33 51 50 3"}


def run_extras(args, tmp):
    """tests/c/api_extras.c against the reference: digests of the ciphertexts / plaintexts it writes + the text of its diagnostics"""
    d = os.path.join(tmp, "extras_" + args.replace(" ", "_"))
    os.makedirs(d)
    env = dict(os.environ, GEN_PARITY_SEED=str(SEED))
    r = subprocess.run([os.path.join(EX, "refgen_api_extras"), d] + args.split(), capture_output=True, text=True, env=env, timeout=3600)
    assert r.returncode == 0 and "SUCESS!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return {"args": args, "files": {os.path.basename(p)[:-3]: sha(p) for p in sorted(glob.glob(d + "/*.ct"))},
            "text": open(os.path.join(d, "text.txt")).read().splitlines(),
            "validate_stdout": [ln for ln in r.stdout.splitlines() if "internal validation" in ln],
            "validate_stderr": [ln for ln in r.stderr.splitlines() if ln.startswith(("ERROR: validation", "idx:", "res:", "std:"))]}


def run_model(key, tmp):
    """one image of the generated ResNet with the synthetic weight file of tools/make_weight_file.py (sigma from weights.json)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import model_weights

    name, _ = MODELS[key]
    wfile, wmeta = model_weights.ensure(key)
    prefix = os.path.join(tmp, key)
    env = dict(os.environ, GEN_PARITY_SEED=str(SEED), GEN_PARITY_OUT=prefix, MODEL_DATA_FILE=wfile, MODEL_ENC_SEED="1000",
               RTLIB_TIMING_OUTPUT="stdout")
    t0 = time.time()
    r = subprocess.run([os.path.join(EX, "refgen_model_" + name), "1"], capture_output=True, text=True, env=env)
    wall = time.time() - t0
    log = os.path.join(ROOT, "profiles", "r05_ref_%s_seeded.log" % key)
    open(log, "w").write(r.stdout + "\n--- stderr ---\n" + r.stderr)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    m = re.search(r"logits9:((?: -?\d+\.\d+)+)", r.stdout)
    return {"program": name, "weights": wmeta, "enc_seed": 1000, "outputs": digests(prefix), "logits9": [float(x) for x in m.group(1).split()],
            "reference_wall_s": round(wall, 1), "log": os.path.relpath(log, ROOT)}


def main():
    what = sys.argv[1:] or ["examples"]
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data["seed"] = SEED
    data["made_by"] = "tests/golden/gen_gen_parity.py (reference rtlib + tests/c/gen_parity_ref.c)"
    with tempfile.TemporaryDirectory() as tmp:
        if "examples" in what:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refgen"])
            data["examples"] = {}
            for name in EXAMPLES:
                # image k of a batch of B under an unchanged program = the k-th of B encryptions in a row from the thread's stream
                data["examples"][name] = {"single": run_example(name, 1, 0, tmp), "batch3": {str(k): run_example(name, 3, k, tmp) for k in range(3)}}
                print(name, data["examples"][name]["single"])
        if "ksw" in what:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refgen"])
            data["ksw_variants"] = {name: run_ksw(args, tmp) for name, args in KSW_SETS.items()}
            print(data["ksw_variants"])
        if "extras" in what:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "refgen"])
            data["api_extras"] = {name: run_extras(args, tmp) for name, args in EXTRAS_SETS.items()}
            print(data["api_extras"])
        if any(k in what for k in ("examples", "ksw", "extras")):
            json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)
            print("wrote", OUT)
        for key in MODELS:  # (hours of one core each: the fixture file is read again and written right after every model)
            if key in what:
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(EX, "refgen_model_" + MODELS[key][0])])
                res = run_model(key, tmp)
                data = json.load(open(OUT)) if os.path.exists(OUT) else data
                data.setdefault("models", {})[key] = res
                json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)
                print(key, res, "\nwrote", OUT)


if __name__ == "__main__":
    main()
