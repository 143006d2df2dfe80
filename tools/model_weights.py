"""The synthetic weight files of the generated ResNets: one place that says which file a test, the bench and the reference
run use (tools/make_weight_file.py: numpy default_rng(seed) * sigma, regenerated bit-identically wherever the same numpy runs,
so the file itself never travels or gets committed -- its md5 does, tests/golden/gen_parity.json).

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
SIGMA = {"resnet20": 0.2, "resnet32": 0.2, "resnet32c100": 0.2, "resnet44": 0.15, "resnet56": 0.12, "resnet110": 0.01}
SEED = 2


def path_of(key, sigma=None):
    sigma = SIGMA[key] if sigma is None else sigma
    return os.path.join(ROOT, "workloads", "_gen", "weights", "%s_seed%d_sigma%g.msg" % (key, SEED, sigma))


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def ensure(key, sigma=None):
    """-> (path, {"sigma", "seed", "md5"}); writes the file when it is not there yet"""
    sigma = SIGMA[key] if sigma is None else sigma
    p = path_of(key, sigma)
    if not os.path.exists(p):
        os.makedirs(os.path.dirname(p), exist_ok=True)
        tmp = p + ".tmp.%d" % os.getpid()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_weight_file.py"), "--entries",
                               os.path.join(ROOT, "tests", "golden", ENTRIES[key]), "--out", tmp, "--seed", str(SEED), "--sigma", repr(sigma)],
                              stdout=subprocess.DEVNULL)
        os.replace(tmp, p)
    return p, {"sigma": sigma, "seed": SEED, "md5": md5(p)}


if __name__ == "__main__":
    k = sys.argv[1] if len(sys.argv) > 1 else "resnet20"
    print(ensure(k, float(sys.argv[2]) if len(sys.argv) > 2 else None))
