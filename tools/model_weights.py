"""The synthetic weight files of the generated ResNets: one place that says which file a test, the bench and the reference
run use.  The file itself never travels or gets committed (0.2-1.1 GB each) -- its md5 does (tests/golden/gen_parity.json) and it is
regenerated where it is needed by tools/make_weight_file.py.  Generator per model (GEN below; the fixture records which one the
reference run used): "ih12" = integer-only (SplitMix64 lanes summed to an Irwin-Hall variate; the same bytes under every numpy,
libm and CPU), "numpy" = default_rng(seed).standard_normal (rounds 1-4: the same bytes only where numpy's stream is the same; a model
moves to "ih12" when its reference run -- 0.6-2.6 h of one core and 44 GB -- has been repeated on the new file).

sigma: N(0, 0.05) (rounds 1-3) makes activations shrink layer by layer and the logits end up at 1e-3.  SIGMA below was picked
on the GPU (profiles/r04a_sigma_sweep.txt) as the largest value that keeps every bootstrap input of the 20-layer network inside
the range of its sine approximation: logits of order 0.1-1, so that a comparison of logits has real digits.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRIES = {"resnet20": "resnet20_pt_entries.txt", "resnet32": "resnet32_pt_entries.txt", "resnet32c100": "resnet32c100_pt_entries.txt",
           "resnet44": "resnet44_pt_entries.txt", "resnet56": "resnet56_pt_entries.txt", "resnet110": "resnet110_pt_entries.txt"}
# the generated program of each key (rtlib/ant/dataset/<name>.onnx.inc)
PROGRAM = {"resnet20": "resnet20_cifar10_pre", "resnet32": "resnet32_cifar10_pre", "resnet32c100": "resnet32_cifar100_pre",
           "resnet44": "resnet44_cifar10_pre", "resnet56": "resnet56_cifar10_pre", "resnet110": "resnet110_cifar10_train"}
# (the deeper the network, the smaller the largest sigma that stays in range: profiles/r04a_sigma_sweep.txt, r04ae_sigma_sweep_more_models.txt;
#  ResNet-110 leaves the range at every sigma tried, profiles/r04f_sigma_sweep_resnet110.txt)
SIGMA = {"resnet20": 0.2, "resnet32": 0.15, "resnet32c100": 0.2, "resnet44": 0.13, "resnet56": 0.08, "resnet110": 0.01}
SEED = 2
GEN = {k: "ih12" for k in ENTRIES}  # what the NEXT reference run of a model uses; a fixture entry records the generator it was made with


def path_of(key, sigma=None, gen=None):
    sigma = SIGMA[key] if sigma is None else sigma
    gen = GEN[key] if gen is None else gen
    tag = "" if gen == "numpy" else "_" + gen
    return os.path.join(ROOT, "workloads", "_gen", "weights", "%s_seed%d_sigma%g%s.msg" % (key, SEED, sigma, tag))


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def ensure(key, sigma=None, gen=None):
    """-> (path, {"sigma", "seed", "md5", "gen"}); writes the file when it is not there yet"""
    sigma = SIGMA[key] if sigma is None else sigma
    gen = GEN[key] if gen is None else gen
    p = path_of(key, sigma, gen)
    if not os.path.exists(p):
        os.makedirs(os.path.dirname(p), exist_ok=True)
        tmp = p + ".tmp.%d" % os.getpid()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_weight_file.py"), "--entries",
                               os.path.join(ROOT, "tests", "golden", ENTRIES[key]), "--out", tmp, "--seed", str(SEED), "--sigma", repr(sigma),
                               "--gen", gen],
                              stdout=subprocess.DEVNULL)
        os.replace(tmp, p)
    return p, {"sigma": sigma, "seed": SEED, "md5": md5(p), "gen": gen}


if __name__ == "__main__":
    k = sys.argv[1] if len(sys.argv) > 1 else "resnet20"
    print(ensure(k, float(sys.argv[2]) if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None))
