#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CKKS runtime.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Headline workload (BASELINE.json metric, configs[3] = SURVEY 8d "C4"): encrypted inference of the
ACE-compiled ResNet-20/CIFAR-10 program (the reference's checked-in generated source, N=2^16, L=34,
dnum=3, 19 bootstraps) on a synthetic 3x32x32 image with synthetic weights.  A step = one image:
encode+encrypt, Main_graph, decrypt+decode, through the rt_ant drop-in API (libFHErt_ant.so over the
acehip C ABI).  The generated program is workloads/_gen/models/libmodel_resnet20.so (compiled unchanged from
/root/reference by tools/build_models.py in the dev container; it travels with the snapshot).  If that
library is absent the bench falls back to the largest configuration that needs no generated source:
C3, the full key-switch at N=2^16, L=25, dnum=4 (and says so in config.workload).
Each rank (one per GPU) runs independent images: replicas, weak scaling, no data-path collective
(the reference's own parallel axis is images: resnet_cifar.main.inc:77-116); RCCL only carries the
barrier and the max-reduce of the timing.

Same JSON line:
  roofline      the dominant kernel family (forward NTT, its two 8-stage pass kernels), HIP events on the
                launch stream over a 1024-limb batch (512 MiB > Infinity Cache): algorithmic 16*N B/limb
                (+ key_switch_ms / key_switch_frac / key_switch_batched_per_s: the C3 figures inside this object as well)
  key_switch    C3 key-switch, HIP-event timed (single operation, and 12 ciphertexts per launch set)
  latency_s_single_image / single_image_latency
                one image alone (B = 1, one stream) after the throughput run, verified against the reference's digest
  cpu_baseline.measured_pair (+ measured_program_s, gpu_program_s)
                a MEASURED like-for-like pair: tests/c/ct_parity.c (operator script + two bootstraps at the headline's ring) built
                against the reference rtlib (one core of this host) and against this runtime (the GPU), each side's own clock
  cpu_baseline  reference rtlib (oracle/_ref/ref_dump, kind "reference") on this host before the GPU is touched:
                one process per usable physical core of ONE socket, all at once (cores = how many; a cgroup CPU quota
                below the socket size is reported and the full socket given as an ideal-scaling extrapolation), with
                the 1-core figures beside it; bounded sample (NTT + key-switch micro-ops), scaled to images/s with
                the measured op mix of the full CPU run recorded in profiles/cpu_resnet20_devbox.json
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
N, L, Q0, SF, DNUM = 65536, 25, 60, 56, 4   # C2/C3 parameter set
# the reference's published ResNet-20/CIFAR-10 run, 1453.96 s per image on one Xeon 8369B core
# (scripts/ace_pre.log:28) -- the same metric on the reference's own hardware
BASELINE_IMAGES_PER_S = 1.0 / 1453.96
N_CT = 16                                    # ciphertext pairs in the resident NTT batch
MODEL_LIB = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet20.so")


def cpu_topology():
    """Physical cores of socket 0 usable by this process: one logical CPU per core (lscpu), intersected with the
    affinity mask, capped by the cgroup CPU quota (a container may see 256 CPUs and be allowed 16 of them)."""
    import math

    allowed = sorted(os.sched_getaffinity(0))
    per_core = {}
    sockets = set()
    try:
        out = subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, check=True).stdout
        for line in out.splitlines():
            if line.startswith("#") or not line.strip():
                continue
            cpu, core, sock = (int(x) for x in line.split(",")[:3])
            sockets.add(sock)
            if sock == min(sockets | {sock}) and cpu in allowed:
                per_core.setdefault((sock, core), cpu)
    except Exception:  # noqa: BLE001 -- no lscpu: treat every allowed CPU as a core
        per_core = {(0, c): c for c in allowed}
        sockets = {0}
    s0 = min(sockets) if sockets else 0
    cpus = [cpu for (sock, _), cpu in sorted(per_core.items()) if sock == s0]
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except Exception:  # noqa: BLE001
        pass
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:  # noqa: BLE001
        pass
    usable = len(cpus) if quota is None else max(1, min(len(cpus), int(math.floor(quota))))
    return {"cpu_model": model, "sockets": len(sockets), "cores_per_socket": len(cpus), "cgroup_cpu_quota": quota,
            "pin_cpus": cpus[:usable]}


def _ref_bench(ref, cpus, ks_reps, ntt_reps):
    """One `ref_dump bench` child per entry of cpus, started together, each pinned to its CPU; returns their JSON results."""
    procs = []
    for cpu in cpus:
        procs.append(subprocess.Popen([ref, "bench", str(N), str(L), str(Q0), str(SF), str(DNUM), str(L), str(ks_reps), str(ntt_reps)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      preexec_fn=(lambda c=cpu: os.sched_setaffinity(0, {c}))))
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=600)
        if p.returncode != 0:
            raise RuntimeError("ref_dump bench exited with %d" % p.returncode)
        res.append(json.loads(out.strip().splitlines()[-1]))
    return res


MIX = (65536, 34, 51, 50, 3, 20)   # the generated ResNets' parameter set (N, L, q0, Delta, dnum) and a mid-range level for the samples


def _ref_mix(ref, cpus, reps):
    """`ref_dump mix` (seconds per call of every primitive family of the reference at the workload's parameter set), one pinned
    child per entry of cpus, started together."""
    procs = [subprocess.Popen([ref, "mix"] + [str(x) for x in MIX] + [str(reps)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              preexec_fn=(lambda c=cpu: os.sched_setaffinity(0, {c}))) for cpu in cpus]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        if p.returncode != 0:
            raise RuntimeError("ref_dump mix exited with %d" % p.returncode)
        res.append(json.loads(out.strip().splitlines()[-1]))
    return res


def price_image(mix, st):
    """Seconds ONE core of this host needs for one image: the per-image call statistics of the run (st: family -> (calls, units,
    algorithmic bytes); the same data-oblivious program makes the same calls on the reference runtime) priced family by family with
    the reference's own primitives as timed by `ref_dump mix` at level 20 (costs scale with the limbs a call touches, i.e. with its
    algorithmic bytes)."""
    N, L, _, _, _, l = MIX
    K, nd = mix["K"], mix["num_decomp"]
    limb = 8.0 * N
    t_mul, t_add, t_rot = mix["hw_modmul_s"], mix["hw_modadd_s"], mix["hw_rotate_s"]
    t_ntt = 0.5 * (mix["ntt_fwd_s"] + mix["ntt_inv_s"])
    pair = 2 * t_mul + 2 * t_add                                   # one (limb, digit) of a key inner product: polynomial.c:148-183
    t_ks = mix["decomp_modup_all_digits_s"] + (l + K) * nd * pair + 2 * mix["mod_down_s"]
    by = lambda k: st.get(k, (0, 0, 0))[2]
    n_mul = st.get("elementwise_mul", (0, 0, 0))[1]
    parts = {
        "ntt": by("ntt") / (2 * limb) * t_ntt,
        "elementwise": n_mul * t_mul + max(0.0, by("elementwise") - 3 * limb * n_mul) / (3 * limb) * t_add,
        "rotate": by("rotate") / (2 * limb) * t_rot,
        "decomp_modup": by("decomp_modup") / (limb * (l + nd * (l + K))) * mix["decomp_modup_all_digits_s"],
        "key_inner_product": by("key_inner_product") / (limb * (l + K) * (3 * nd + 2)) * (l + K) * nd * pair,
        "mod_down": by("mod_down") / (limb * (2 * l + K)) * mix["mod_down_s"],
        "rescale": by("rescale") / (limb * (2 * l - 1)) * mix["rescale_s"],
        "key_switch": by("key_switch") / (limb * (l + 2 * nd * (l + K) + 2 * l)) * t_ks,
        "encode": by("encode") / (limb * l + 4.0 * N / 4) * mix["encode_s"],
    }
    return sum(parts.values()), parts


def profile_scaled_image(prof, dev, host):
    """Seconds ONE core of this host needs for one image, from the flat profile of the reference's full run on the dev container
    (profiles/cpu_resnet20_devbox.json r04_profile: seconds per function family over RTM_MAIN_GRAPH, tests/c/ref_sampler.c) -- every family
    moved to this host by the ratio of the reference primitive that IS that family's inner loop, timed on both machines by the same
    `ref_dump mix` (cold operands): transforms by Ftt_fwd / Ftt_inv, the multiply family by Hw_modmul, the add family by Hw_modadd, the
    permutations by Hw_rotate, the conversions by what Decompose_modup + Reduce_rns_base + Rescale_poly cost beyond their transforms,
    memset / memcpy by the measured rates, the encoder by Encode_at_level.  Exact on the dev container by construction; what it carries to
    another host is how that host's core runs each inner loop.  None when either side lacks a figure (an older ref_dump)."""
    need = ("ntt_fwd_s", "ntt_inv_s", "hw_modmul_s", "hw_modadd_s", "hw_rotate_s", "decomp_modup_all_digits_s", "mod_down_s", "rescale_s",
            "encode_s", "memset_GBs", "memcpy_GBs")
    if any(k not in dev or k not in host for k in need):
        return None, None
    l, K, nd = dev["level"], dev["K"], dev["num_decomp"]

    def conv_self(m):  # the three composites minus their transforms: nd(l+K) in Decompose_modup of every digit, K+l in Reduce_rns_base, l in Rescale_poly
        t = 0.5 * (m["ntt_fwd_s"] + m["ntt_inv_s"])
        return m["decomp_modup_all_digits_s"] + m["mod_down_s"] + m["rescale_s"] - (nd * (l + K) + (K + l) + l) * t

    ratio = {"ntt_fwd": host["ntt_fwd_s"] / dev["ntt_fwd_s"], "ntt_inv": host["ntt_inv_s"] / dev["ntt_inv_s"],
             "mul": host["hw_modmul_s"] / dev["hw_modmul_s"], "add": host["hw_modadd_s"] / dev["hw_modadd_s"],
             "permute": host["hw_rotate_s"] / dev["hw_rotate_s"], "conversion": max(0.05, conv_self(host)) / max(1e-9, conv_self(dev)),
             "memset": dev["memset_GBs"] / host["memset_GBs"], "memcpy": dev["memcpy_GBs"] / host["memcpy_GBs"],
             "encode": host["encode_s"] / dev["encode_s"]}
    parts = {f: sec * ratio.get(f, 1.0) for f, sec in prof["by_family_s"].items()}
    return sum(parts.values()), parts


def cpu_baseline(have_model):
    """Reference rtlib (oracle/_ref/ref_dump, built from /root/reference by oracle/Makefile) timed on this host BEFORE the
    GPU is touched: (1) one process on one core, (2) one process per usable physical core of one socket, all at once
    (the reference's throughput policy is one image per OpenMP thread on the cores of a socket, scripts/accuracy.sh:15-20,37,
    dataset/resnet_cifar.main.inc:77-116).  Bounded sample: NTTs of one limb and full C3 key-switches; scaled to images/s with
    the measured (CPU ResNet-20 s/image)/(CPU key-switch s) of profiles/cpu_resnet20_devbox.json."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    topo = cpu_topology()
    res = None
    pair = MeasuredPair(topo) if have_model else None  # (starts the reference program on a core of its own; that core is left out below)
    if os.path.exists(ref):
        try:
            one = _ref_bench(ref, topo["pin_cpus"][:1], 8, 3000)[0]
            many = _ref_bench(ref, topo["pin_cpus"], 8, 3000) if len(topo["pin_cpus"]) > 1 else [one]
            c = len(many)
            if have_model:  # the primitives of the reference at the workload's own parameter set (L=34, dnum=3), alone and under load
                mix_one = _ref_mix(ref, topo["pin_cpus"][:1], 6)[0]
                mix_many = _ref_mix(ref, topo["pin_cpus"], 6) if len(topo["pin_cpus"]) > 1 else [mix_one]
            agg_ks = sum(1.0 / r["key_switch_s"] for r in many)
            agg_ntt = sum(2 * 16 * N / (r["ntt_fwd_s"] + r["ntt_inv_s"]) for r in many) / 1e9
            res = {"cores": c, "kind": "reference", "cpu_model": topo["cpu_model"], "sockets": topo["sockets"],
                   "cores_per_socket": topo["cores_per_socket"], "cgroup_cpu_quota": topo["cgroup_cpu_quota"],
                   "key_switch_per_s": round(agg_ks, 4), "ntt_GBs": round(agg_ntt, 4),
                   "key_switch_s_per_core_loaded": round(c / agg_ks, 4),
                   "one_core": {"ntt_fwd_ms": round(one["ntt_fwd_s"] * 1e3, 4), "ntt_inv_ms": round(one["ntt_inv_s"] * 1e3, 4),
                                "key_switch_s": round(one["key_switch_s"], 4), "key_switch_per_s": round(1.0 / one["key_switch_s"], 4),
                                "ntt_GBs": round(2 * 16 * N / (one["ntt_fwd_s"] + one["ntt_inv_s"]) / 1e9, 4)},
                   "sample": "reference rtlib (gcc -O3): per process 3000 Ftt_fwd + 3000 Ftt_inv of one limb (N=2^16) and 8 full "
                             "key-switches (L=25, dnum=4); first 1 process alone, then %d processes at once, one pinned to each "
                             "usable physical core of socket 0" % c}
            if have_model:
                res["mix_one_core"] = mix_one
                res["mix_loaded"] = mix_many
            if topo["cgroup_cpu_quota"] is not None and c < topo["cores_per_socket"]:
                res["sample"] += (" (the container's cgroup allows %.0f CPUs of the socket's %d cores: the full socket cannot be "
                                  "loaded here; socket_extrapolated assumes ideal scaling from the measured %d)"
                                  % (topo["cgroup_cpu_quota"], topo["cores_per_socket"], c))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("reference baseline failed (%s); timing the oracle port instead\n" % e)
    if res is None:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O

        o = O.Oracle(N, L, Q0, SF, DNUM)
        x = o.uniform(1, 1, 1)
        reps = 3000
        t0 = time.perf_counter()
        for _ in range(reps):
            o.lib.orc_ntt_fwd(O.ptr(x[0]), o.prime_ptr(0), N)
        t1 = time.perf_counter()
        for _ in range(reps):
            o.lib.orc_ntt_inv(O.ptr(x[0]), o.prime_ptr(0), N)
        t2 = time.perf_counter()
        a, key = o.uniform(L, L, 1), o.make_key(101)
        t3 = time.perf_counter()
        for _ in range(4):
            o.key_switch(a, key, L)
        t4 = time.perf_counter()
        ks_s = (t4 - t3) / 4
        res = {"cores": 1, "kind": "port", "cpu_model": topo["cpu_model"], "sockets": topo["sockets"],
               "cores_per_socket": topo["cores_per_socket"], "cgroup_cpu_quota": topo["cgroup_cpu_quota"],
               "key_switch_per_s": round(1.0 / ks_s, 4), "ntt_GBs": round(2 * 16 * N * reps / (t2 - t0) / 1e9, 4),
               "key_switch_s_per_core_loaded": round(ks_s, 4),
               "sample": "3000 fwd + 3000 inv NTTs of one limb (N=2^16) and 4 key-switches by oracle/ckks_oracle.c, 1 thread"}
    if have_model and "mix_loaded" in res:
        res["unit"] = "images/s"
        res["value"] = None  # priced by finish_cpu_baseline() once the run's per-image call statistics are known
    elif have_model:
        # no reference build on this host: the port's key-switch scaled by the dev-box ratio (a fallback, labelled as such)
        dev = os.path.join(ROOT, "profiles", "cpu_resnet20_devbox.json")
        ratio, src = 1453.96 / 0.58, "BASELINE.md (published 1453.96 s/image; 0.58 s key-switch)"
        if os.path.exists(dev):
            d = json.load(open(dev))
            if d.get("image_s") and d.get("key_switch_s"):
                ratio, src = d["image_s"] / d["key_switch_s"], "profiles/cpu_resnet20_devbox.json"
        res["value"] = round(res["key_switch_per_s"] / ratio, 8)
        res["unit"] = "images/s"
        res["sample"] += "; scaled to images/s by (CPU ResNet-20 s/image) / (CPU key-switch s) = %.1f from %s" % (ratio, src)
    else:
        res["value"], res["unit"] = res["key_switch_per_s"], "key-switches/s"
    if res.get("value") is not None:
        _extrapolate_socket(res)
    if pair is not None:
        mp = pair.finish()
        if mp is not None:
            res["measured_pair"] = mp
            res["measured_program_s"], res["gpu_program_s"] = mp.get("cpu_script_wall_s"), mp.get("gpu_script_wall_s")
    return res


CT_PAIR_ARGS = "65536 33 51 50 3 192 4096 15 1 -5".split()  # the generated ResNet-20's ring: N = 2^16, L = 34, dnum = 3, bootstrap 2 -> 15 limbs


class MeasuredPair:
    """A like-for-like MEASURED pair beside the priced images/s: one program -- tests/c/ct_parity.c, the ciphertext-level operator script
    (HAdd, HSub, plaintext add / multiply, Rescale, HMul as tensor product + Relinearize, fused HMul, two Rotates at two levels,
    ModSwitch) and two Bootstraps at the headline's ring -- built twice from the same source: against the REFERENCE rtlib
    (oracle/_ref/ct_parity_ref, `dump` mode) and against this runtime (workloads/_gen/examples/ct_parity, `make` mode).  Both run with
    CT_PARITY_TIMING_ONLY=1 (nothing written or compared inside the span) and report the wall-clock seconds of the script span by
    their own clock.  The reference runs on ONE core of this host (the program is single-threaded, like one image thread of the
    reference's OpenMP loop), pinned, started first and left alone on its core while the primitive timings load the others; the
    product runs afterwards as a child on the idle GPU, before this process touches it.  Byte-level equality of the two builds'
    outputs on identical keys is tests/test_gpu_ct_parity.py::test_bootstrap_bit_exact_at_the_benchmark_ring; here each side uses its
    own keys and the decrypted messages are compared."""

    def __init__(self, topo):
        import tempfile

        self.ref = os.path.join(ROOT, "oracle", "_ref", "ct_parity_ref")
        self.gpu = os.path.join(ROOT, "workloads", "_gen", "examples", "ct_parity")
        self.proc = None
        self.cpu = None
        if not (os.path.exists(self.ref) and os.path.exists(self.gpu)) or os.environ.get("ACEHIP_BENCH_NO_PAIR"):
            return
        self.tmp = tempfile.mkdtemp(prefix="acehip_pair_")
        if len(topo["pin_cpus"]) > 2:  # a core of its own: the loaded-core samples take the others
            self.cpu = topo["pin_cpus"].pop()
        self.t0 = time.perf_counter()
        self.proc = subprocess.Popen([self.ref, "dump", self.tmp] + CT_PAIR_ARGS, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                     env=dict(os.environ, CT_PARITY_TIMING_ONLY="1"),
                                     preexec_fn=((lambda c=self.cpu: os.sched_setaffinity(0, {c})) if self.cpu is not None else None))

    @staticmethod
    def _parse(out):
        import re

        g = lambda k: (lambda m: float(m.group(1)) if m else None)(re.search(k + r" (\d+\.\d+)", out))  # noqa: E731
        msgs = {m.group(1): [float(v) for v in m.group(2).split()] for m in re.finditer(r"msg_(\w+)\[0\.\.3\] =((?: -?\d+\.\d+)+)", out)}
        return g("script_wall_s"), g("script_cpu_s"), msgs

    def finish(self):
        import shutil

        if self.proc is None:
            return None
        try:
            out, _ = self.proc.communicate(timeout=900)
            cpu_total = time.perf_counter() - self.t0
            if self.proc.returncode != 0:
                return {"error": "reference program exited with %d" % self.proc.returncode}
            cw, cc, cmsg = self._parse(out)
            tg = time.perf_counter()
            r = subprocess.run([self.gpu, "make", self.tmp] + CT_PAIR_ARGS, capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, CT_PARITY_TIMING_ONLY="1", ACEHIP_SEED="1"))
            gpu_total = time.perf_counter() - tg
            if r.returncode != 0:
                return {"error": "product program exited with %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:]), "cpu_script_wall_s": cw}
            gw, _, gmsg = self._parse(r.stdout)
            agree = bool(cmsg) and cmsg.keys() == gmsg.keys() and all(abs(a - b) < 1e-3 for k in cmsg for a, b in zip(cmsg[k], gmsg[k]))
            return {"program": "tests/c/ct_parity.c %s: operator script + two bootstraps at the headline's ring (N=2^16, L=34, dnum=3, 4096 slots, "
                               "bootstrap 2 -> 15 limbs); CT_PARITY_TIMING_ONLY=1" % " ".join(CT_PAIR_ARGS),
                    "cpu": "reference rtlib (oracle/_ref/ct_parity_ref dump), 1 thread, %s" % ("pinned to cpu %d" % self.cpu if self.cpu is not None else "not pinned"),
                    "cpu_cores": 1, "cpu_script_wall_s": cw, "cpu_script_cpu_s": cc, "cpu_process_wall_s": round(cpu_total, 1),
                    "gpu": "this runtime (workloads/_gen/examples/ct_parity make), 1 x MI355X, one stream, one ciphertext per launch",
                    "gpu_script_wall_s": gw, "gpu_process_wall_s": round(gpu_total, 1),
                    "gpu_over_one_cpu_core": (round(cw / gw, 1) if cw and gw else None),
                    "decrypted_messages_agree_to_1e-3": agree,
                    "note": "measured, not modelled: the same source on both runtimes, each side's own clock around the same span; latency form "
                            "(one ciphertext, no batch), so the GPU side is launch-bound -- the headline's batches are 12 images per launch"}
        except Exception as e:  # noqa: BLE001 -- a secondary measurement never fails the run
            try:
                self.proc.kill()
            except Exception:  # noqa: BLE001
                pass
            return {"error": repr(e)}
        finally:
            shutil.rmtree(self.tmp, ignore_errors=True)


def _extrapolate_socket(res):
    if res["cores"] < res["cores_per_socket"]:
        res["socket_extrapolated"] = {"cores": res["cores_per_socket"], "value": round(res["value"] * res["cores_per_socket"] / res["cores"], 8),
                                      "unit": res["unit"], "note": "EXTRAPOLATED, not measured: ideal linear scaling of the measured aggregate to the "
                                                                   "whole socket; an upper bound for the CPU"}


def finish_cpu_baseline(res, stats_per_image):
    """images/s of the measured cores: every loaded core's `ref_dump mix` rates price one image of THIS run's call statistics
    (price_image); the aggregate is the sum over the cores that were loaded at once.  Also: the same model with the dev container's
    rates against the full reference run measured there (profiles/cpu_resnet20_devbox.json) -- how far the pricing is off."""
    if res is None or "mix_loaded" not in res:
        return
    per_core = [price_image(m, stats_per_image)[0] for m in res["mix_loaded"]]
    s1, parts = price_image(res["mix_one_core"], stats_per_image)
    res["value"] = round(sum(1.0 / t for t in per_core), 8)
    res["image_s_per_core_loaded"] = round(sum(per_core) / len(per_core), 2)
    res["one_core"]["image_s"] = round(s1, 2)
    res["one_core"]["images_per_s"] = round(1.0 / s1, 8)
    res["one_core"]["image_s_by_family"] = {k: round(v, 2) for k, v in parts.items()}
    res["sample"] += ("; images/s: `ref_dump mix 65536 34 51 50 3 20` (the reference's NTT, Hw_modmul / Hw_modadd / Hw_rotate, Decompose_modup, "
                      "Reduce_rns_base, Rescale_poly, encode, memset and memcpy at the workload's own parameter set; ~10 s per process, alone and on all "
                      "measured cores at once; operands rotate through a 0.7 GB pool, as cold as in the program) pricing the per-image "
                      "call statistics of this run family by family")
    res["kind"] = "reference-primitives-priced"  # NOT a run of the program on this host: the reference's primitives timed here, priced
    dev = os.path.join(ROOT, "profiles", "cpu_resnet20_devbox.json")
    if os.path.exists(dev):
        d = json.load(open(dev))
        full = d.get("r04_full_run") or {}
        if d.get("mix_level20_cold") and full.get("image_s"):
            # how good is the pricing?  The same model with the dev container's own (cold-operand) primitive timings against the
            # full reference run of the unchanged generated ResNet-20 measured there (profiles/r04_ref_resnet20_seeded.log)
            pred = price_image(d["mix_level20_cold"], stats_per_image)[0]
            res["model_check"] = {"host": d.get("cpu"), "predicted_image_s": round(pred, 1), "measured_image_s": full["image_s"],
                                  "predicted_over_measured": round(pred / full["image_s"], 3),
                                  "note": "the same pricing with the dev container's primitive timings against the full reference run measured "
                                          "there; rounds 1-3 timed the primitives on cache-hot operands (0.634, now in hot_operand_pricing); "
                                          "the residual is work the call statistics do not carry (allocation, memset, evaluator bookkeeping "
                                          "inside Bootstrap)",
                                  "hot_operand_pricing_predicted_over_measured": (round(price_image(d["mix_level20"], stats_per_image)[0] / full["image_s"], 3)
                                                                                   if d.get("mix_level20") else None)}
            med = (d.get("mix_level20_cold_median") or {}).get("median")
            if med:  # single `mix` runs on the shared dev VM scatter by +-14 %: the same check with the per-key median of nine runs, and its range
                mc = res["model_check"]
                mc["with_median_of_nine_mix_runs"] = round(price_image(med, stats_per_image)[0] / full["image_s"], 3)
                mc["range_over_the_nine_runs"] = [round(price_image(dict(med, **d["mix_level20_cold_median"][k]), stats_per_image)[0] / full["image_s"], 3)
                                                  for k in ("min", "max")]
                prof = d.get("r04_profile")
                if prof:
                    mc["residual"] = ("what the call statistics of OUR run do not carry of the reference's work, from the flat profile of its full run "
                                      "(profiles/r04_ref_resnet20_profile.txt): it executes %d limb-transforms for the image, our runtime launches %d "
                                      "(digits raised once per rotated ciphertext, fused bootstrap), and spends %.1f %% of the image in libc's memset / memcpy "
                                      "(calloc'ed temporaries, polynomial copies)"
                                      % (prof["transforms_per_image"]["NTT"] + prof["transforms_per_image"]["INTT"],
                                         int(stats_per_image.get("ntt_launched", (0, 0, 0))[1]),
                                         100.0 * (prof["by_family_s"].get("memset", 0) + prof["by_family_s"].get("memcpy", 0)) / prof["main_graph_s"]))
                    # third estimate, the one reported as `value`: the PROFILE of the full run moved to this host family by family
                    per_core_p = [profile_scaled_image(prof, med, m)[0] for m in res["mix_loaded"]]
                    one_p, parts_p = profile_scaled_image(prof, med, res["mix_one_core"])
                    if one_p and all(per_core_p):
                        res["priced_by_call_statistics"] = {"value": res["value"], "unit": "images/s", "image_s_per_core_loaded": res["image_s_per_core_loaded"],
                                                            "note": "price_image(): this run's call statistics x the host's primitive timings; prices "
                                                                    "less work than the reference does (model_check), i.e. an upper bound for the CPU"}
                        res["value"] = round(sum(1.0 / t for t in per_core_p), 8)
                        res["image_s_per_core_loaded"] = round(sum(per_core_p) / len(per_core_p), 2)
                        res["one_core"]["image_s_profile_scaled"] = round(one_p, 2)
                        res["one_core"]["image_s_profile_scaled_by_family"] = {k: round(v, 1) for k, v in parts_p.items()}
                        res["value_method"] = ("flat profile of the reference's full run of the unchanged generated ResNet-20 on the dev container (%.0f s over "
                                               "Main_graph, seconds per function family) x host/dev ratio of the reference primitive that is each family's "
                                               "inner loop (`ref_dump mix`, cold operands, timed here on every loaded core and on the dev container), summed "
                                               "over the loaded cores" % prof["main_graph_s"])
            # second estimate, anchored to a RUN: the dev container's measured seconds per image moved to this host by the ratio of the
            # two hosts' priced images (the pricing only transfers, its absolute error cancels)
            res["anchored_to_full_run"] = {
                "one_core_image_s": round(full["image_s"] * s1 / pred, 1),
                "value": round(sum(1.0 / (full["image_s"] * t / pred) for t in per_core), 8), "unit": "images/s",
                "note": "measured dev-container seconds per image (1 thread) x priced(this host) / priced(dev container), summed over the "
                        "loaded cores (one ratio for the whole image; `value` moves every function family by its own)"}
    # the transfer model checked on the bench host itself (on demand: tools/cpu_model_check.py host, ~5 min of one core under gpurun):
    # a real reference program the host can finish -- operator script + two bootstraps at this ring -- timed there by the reference's own
    # clock, against profile_scaled_image() applied to that program's dev-container profile
    chk = os.path.join(ROOT, "profiles", "cpu_model_check_bench_host.json")
    if os.path.exists(chk):
        c = json.load(open(chk))
        res["model_check_on_bench_host"] = {k: c.get(k) for k in ("program", "host_cpu", "dev_cpu", "dev_measured_s", "predicted_s", "measured_s",
                                                                     "predicted_over_measured", "method")}
        res["model_check_on_bench_host"]["same_cpu_model_as_this_run"] = c.get("host_cpu") == res.get("cpu_model")
        res["model_check_on_bench_host"]["source"] = "profiles/cpu_model_check_bench_host.json (tools/cpu_model_check.py)"
    _extrapolate_socket(res)



GEN_PARITY = os.path.join(ROOT, "tests", "golden", "gen_parity.json")


def reference_fixture(key="resnet20"):
    """The committed digest of the REFERENCE rtlib's output ciphertext for one image of the generated ResNet (tests/golden/gen_parity.json,
    made in the dev container by tests/golden/gen_gen_parity.py: reference rtlib + the key set / encryption randomness of ACEHIP_SEED
    injected by tests/c/gen_parity_ref.c).  None when the fixture has no entry for this model."""
    try:
        fix = json.load(open(GEN_PARITY))
        m = fix["models"][key]
        return {"seed": fix["seed"], "enc_seed": m["enc_seed"], "weights": m["weights"], "digest": m["outputs"]["0.0"], "logits9": m.get("logits9")}
    except Exception:  # noqa: BLE001
        return None


def model_main_image0():
    """image 0 of tools/model_main.c (xorshift64, U(-1,1)): the image the reference run of the fixture was fed"""
    import numpy as np

    m, z, v = (1 << 64) - 1, 1, []
    for _ in range(3 * 32 * 32):
        z ^= (z << 13) & m
        z ^= z >> 7
        z ^= (z << 17) & m
        v.append((z >> 11) / 9007199254740992.0 * 2.0 - 1.0)
    return np.ascontiguousarray(np.array(v, dtype=np.float64))


def load_model_runtime(device, batch=1):
    """dlopen the generated ResNet-20 (+ the rt_ant drop-in it is linked against) and return (lib, step):
    step() pushes `batch` synthetic images through Prepare_input / Run_main_graph / Handle_output -- one Run_main_graph per
    batch (Acehip_rt_set_batch, include/rt_ant/rt_api.h: the images share every launch, key, twiddle and weight plaintext).
    step(verify=(slot, enc_seed, prefix)): the image in position `slot` of the batch is the fixture's image, encrypted with the
    fixture's randomness, and the batch's output ciphertexts are written to <prefix>.<image> (Acehip_rt_dump_next_output)."""
    import numpy as np

    os.environ["ACEHIP_DEVICE"] = str(device)
    if "ACEHIP_RT_DATA_FILE" not in os.environ:
        os.environ.setdefault("ACEHIP_RT_DATA_SYNTH", "1")
    # libmodel defines Main_graph + the Get_* callbacks and is linked against libFHErt_ant.so, so one
    # dlopen resolves both directions of the generated-code <-> runtime boundary
    fhe = C.CDLL(MODEL_LIB, mode=C.RTLD_GLOBAL)
    fhe.Alloc_tensor.restype = C.c_void_p
    fhe.Alloc_tensor.argtypes = [C.c_size_t] * 4 + [C.c_void_p]
    fhe.Prepare_input.argtypes = [C.c_void_p, C.c_char_p]
    fhe.Free_tensor.argtypes = [C.c_void_p]
    fhe.Handle_output.restype = C.POINTER(C.c_double)
    fhe.Handle_output.argtypes = [C.c_char_p]
    fhe.Acehip_rt_set_batch.argtypes = [C.c_uint32]
    fhe.Acehip_rt_select_image.argtypes = [C.c_uint32]
    fhe.Acehip_rt_seed_encryptor.argtypes = [C.c_uint64]
    fhe.Acehip_rt_seed_encryptor.restype = None
    fhe.Acehip_rt_dump_next_output.argtypes = [C.c_char_p]
    fhe.Acehip_rt_dump_next_output.restype = None
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    img_rng = np.random.default_rng(1)
    fixed = model_main_image0()

    def step(verify=None):
        for k in range(batch):
            img = np.ascontiguousarray(img_rng.uniform(-1.0, 1.0, size=3 * 32 * 32))
            if verify is not None and k == verify[0]:
                img = fixed
                fhe.Acehip_rt_seed_encryptor(verify[1])
            t = fhe.Alloc_tensor(1, 3, 32, 32, img.ctypes.data)
            if batch > 1:
                fhe.Acehip_rt_select_image(k)
            fhe.Prepare_input(t, b"input")
            fhe.Free_tensor(t)
        if verify is not None:
            fhe.Acehip_rt_dump_next_output(verify[2].encode())
        fhe.Run_main_graph()
        vals = None
        for k in range(batch):
            if batch > 1:
                fhe.Acehip_rt_select_image(k)
            out = fhe.Handle_output(b"output")
            vals = [out[i] for i in range(10)]
            libc.free(out)
        return vals

    return fhe, step


def shard_model_bench(args):
    """BASELINE configs[4]: ONE encrypted inference job whose RNS limbs are spread over the ranks (limb gi on rank gi % world),
    every rank running the same ACE-generated program behind the rt_ant API (SPMD; ACEHIP_SHARD=1, include/rt_ant/rt_api.h) and RCCL
    broadcasts over xGMI where limbs meet (Decomp_modup, Mod_down, Rescale, the ModRaise of Bootstrap, decode).  Launch:
      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N \
             --mode shard --workload resnet110 [--batch B]
    Strong scaling: the job is the same whatever N; value = images of the job / wall time (max over ranks)."""
    global MODEL_LIB
    r110 = args.workload != "resnet20"
    if r110:
        MODEL_LIB = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet110.so")
    if not os.path.exists(MODEL_LIB):
        raise SystemExit("bench --mode shard: %s missing (tools/build_models.py)" % MODEL_LIB)
    json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ["ACEHIP_SHARD"] = "1"             # the rt_ant shim joins the RCCL communicator of RANK / WORLD_SIZE in Prepare_context
    os.environ.setdefault("ACEHIP_SEED", "1")    # every rank derives the same keys and encryption randomness
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    import ace_compiler_amd  # noqa: F401
    from ace_compiler_amd.dist import Ranks

    ranks = Ranks()  # barrier + max over ranks of the timed region (torch.distributed "nccl" = RCCL); the data path has its own communicator
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    n_batch = max(args.batch, 1) if args.batch else 1
    fhe, step = load_model_runtime(local_rank, n_batch)
    fhe.Acehip_rt_shard_world.restype = C.c_uint32
    fhe.Acehip_rt_shard_rank.restype = C.c_uint32
    fhe.Acehip_rt_shard_traffic.restype = C.c_uint64
    fhe.Acehip_rt_shard_traffic.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    fhe.Acehip_rt_sync.restype = None
    t_ctx = time.perf_counter()
    fhe.Prepare_context()
    fhe.Acehip_rt_set_batch(n_batch)
    t_ctx = time.perf_counter() - t_ctx
    rccl_world = fhe.Acehip_rt_shard_world()
    assert rccl_world == world, "the runtime's RCCL communicator has %d ranks, the launcher %d" % (rccl_world, world)
    logits = None
    for _ in range(args.warmup):
        logits = step()
    fhe.Acehip_rt_sync()
    ranks.barrier()
    steps2 = (C.c_uint64 * 2)()
    fhe.Acehip_rt_shard_traffic(steps2, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step()
    fhe.Acehip_rt_sync()
    elapsed_local = time.perf_counter() - t0
    ranks.barrier()
    elapsed = ranks.max_over_ranks(elapsed_local)
    xbytes = fhe.Acehip_rt_shard_traffic(steps2, 0)
    n_images = n_batch * args.steps
    L_, K_ = 34, 11  # the generated ResNets: mul_depth 33, dnum 3
    owned = [sum(1 for gi in range(L_ + K_) if gi % world == r) for r in range(world)]
    all_x = ranks.sum_over_ranks(float(xbytes))
    if rank == 0:
        value = n_images / elapsed
        out = {
            "metric": "encrypted images/sec (ResNet-%s CIFAR-10, N=2^16), RNS limbs sharded over the GPUs" % ("110" if r110 else "20"),
            "value": round(value, 6), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": round(value / (1.0 / (7531.12 if r110 else 1453.96)), 1), "dtype": "u64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: ACE-compiled ResNet-%s/CIFAR-10 encrypted inference (unchanged generated source), one job "
                                   "whose RNS limbs are spread over %d rank(s) (limb gi on rank gi %% world), every rank running the same program "
                                   "behind the rt_ant API; RCCL broadcasts from the owning rank at Decomp_modup / Mod_down / Rescale / ModRaise / "
                                   "decode; synthetic image and weights; %d image(s) per batch" % ("110" if r110 else "20", world, n_batch),
                       "N": 65536, "images_per_step": n_batch, "parallelism": "limb-sharded: %d rank(s), one GPU each" % world},
            "shard": {"world": world, "rccl_ranks": int(rccl_world), "owned_limbs_per_rank": owned, "limbs_total": L_ + K_,
                      "exchange_steps_per_image": round(steps2[0] / n_images, 1),
                      "limbs_received_per_image_rank0": round(steps2[1] / n_images, 1),
                      "bytes_received_per_image_rank0": int(xbytes / n_images),
                      "bytes_received_per_image_all_ranks": int(all_x / n_images),
                      "prepare_context_s": round(t_ctx, 2)},
        }
        if logits is not None:
            out["config"]["last_logits"] = [round(v, 5) for v in logits]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    fhe.Finalize_context()
    ranks.close()


def limb_sharded_leg(ranks, timeout_s=240):
    """Secondary measurement of a multi-GPU run (N > 1), NOT the headline: after the replica measurement the same ranks run ONE
    ResNet-20 image with its RNS limbs spread over them (BASELINE configs[4]'s execution mode on the headline's network) --
    `bench.py --mode shard` as a child process per rank with its own rendezvous port and RCCL id file, so that a failure or a
    hang of this leg (bounded by a timeout) cannot take the headline line with it.  Reports images/s of the sharded job, the bytes
    every rank received over xGMI, and whether all ranks ended with the same output ciphertext (SHA-256 of the ACEHCT01 dumps)."""
    import glob
    import hashlib
    import subprocess
    import tempfile

    rank, world = ranks.rank, ranks.world
    nonce = int(ranks.max_over_ranks(float(int.from_bytes(os.urandom(3), "little")) if rank == 0 else 0.0))
    port = int(os.environ.get("MASTER_PORT", "29500")) + 1 + nonce % 200
    if port > 65000:
        port -= 2000
    prefix = os.path.join(tempfile.gettempdir(), "acehip_shardleg_%d_r%d" % (nonce, rank))
    # (the launcher's agent store lives on the launcher's port: the children make their own rendezvous, rank 0 hosting it)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC") and not k.startswith("ACEHIP_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ACEHIP_SEED="1", ACEHIP_DUMP_OUTPUT=prefix,
               ACEHIP_SHARD_ID_FILE=os.path.join(tempfile.gettempdir(), "acehip_rccl_%d_%d.id" % (port, nonce)))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--mode", "shard", "--workload", "resnet20", "--batch", "1",
           "--steps", "1", "--warmup", "1"]
    res, ok = None, 0.0
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        if r.returncode == 0:
            ok = 1.0
            if rank == 0:
                res = json.loads(r.stdout.strip().splitlines()[-1])
        elif rank == 0:
            res = {"error": "rank 0 exited with %d: %s" % (r.returncode, r.stderr[-400:])}
    except subprocess.TimeoutExpired:
        if rank == 0:
            res = {"error": "timed out after %d s" % timeout_s}
    except Exception as e:  # noqa: BLE001 -- a secondary measurement never fails the run
        if rank == 0:
            res = {"error": repr(e)}
    wall = time.perf_counter() - t0
    digest = 0.0
    dumps = sorted(glob.glob(prefix + ".*"))
    if ok and dumps:
        hsh = hashlib.sha256()
        for d in dumps:
            hsh.update(open(d, "rb").read())
        digest = float(int.from_bytes(hsh.digest()[:6], "little"))  # 48 bits: exact in a float64
    for d in dumps:
        os.remove(d)
    all_ok = -ranks.max_over_ranks(-ok)
    hi, lo = ranks.max_over_ranks(digest), -ranks.max_over_ranks(-digest)
    if rank != 0:
        return None
    out = {"what": "ONE ResNet-20 image, RNS limbs spread over the %d ranks (limb gi on rank gi %% %d), every rank running the unchanged "
                   "generated program; child processes of this run, RCCL broadcasts from the owning rank (DESIGN 6)" % (world, world),
           "ranks_succeeded": bool(all_ok), "leg_wall_s": round(wall, 1)}
    if isinstance(res, dict) and "error" in res:
        out["error"] = res["error"]
    elif isinstance(res, dict):
        out.update({"images_per_s": res.get("value"), "ms_per_image": res.get("ms_per_step"), "scaling": "strong", "shard": res.get("shard"),
                    "last_logits": (res.get("config") or {}).get("last_logits"),
                    "output_ciphertexts_identical_on_all_ranks": bool(all_ok and digest != 0.0 and hi == lo)})
    else:
        out["error"] = "no result line from rank 0"
    return out


def main():
    global MODEL_LIB
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="model workloads only: run although the timed images cannot be checked against the reference's CPU-run digest "
                         "(without it a run that cannot verify, or whose digests differ, prints its line and EXITS NON-ZERO)")
    ap.add_argument("--workload", choices=["auto", "resnet20", "resnet110", "keyswitch"], default="auto",
                    help="auto / resnet20: the headline (BASELINE configs[3]); resnet110: the workload of configs[4] (ACE-generated "
                         "ResNet-110) as replicas x image streams on each GPU -- a secondary measurement, not the headline; "
                         "keyswitch: configs[2] only")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas",
                    help="replicas (default, the headline): independent images per GPU; shard: latency of ONE limb-sharded "
                         "key-switch + rescale over the ranks (tools/shard_keyswitch_bench.py: acehip_shard_* phases + RCCL "
                         "all-gathers), the building block of BASELINE configs[4]")
    ap.add_argument("--roofline-only", action="store_true",
                    help="only the resident NTT batch of the roofline object (profiling aid: under rocprofv3 every ntt8_* "
                         "launch of the process then has the timed batch's size)")
    ap.add_argument("--roofline-batch", choices=["all", "mix", "scaling", "c3"], default="all",
                    help="with --roofline-only under a profiler: time only this one of the three 1024-limb NTT batches, so that every "
                         "ntt8_* launch of the process is the roofline object's own batch (mix), the scaling-prime batch or the C3 batch")
    ap.add_argument("--streams", type=int, default=3,
                    help="concurrent image streams per GPU for the ResNet headline: host threads of this process, each with "
                         "its own rt_ant context (keys, pool, queue) and HIP stream; 1 = a single stream")
    ap.add_argument("--no-shard-leg", action="store_true",
                    help="never run the secondary limb-sharded ResNet-20 image (it is opt-in anyway: ACEHIP_BENCH_SHARD_LEG=1 with "
                         "--gpus N > 1 runs it after the headline line has been printed; result on stderr)")
    ap.add_argument("--batch", type=int, default=12,
                    help="images per launch on every stream (Acehip_rt_set_batch): the images of a batch share launches, keys, "
                         "twiddles, bootstrap tables and encoded weight plaintexts; a step = one batch per stream")
    args = ap.parse_args()

    if args.mode == "shard":
        if args.workload == "keyswitch":  # the building block alone: one key-switch + rescale through the packed phases
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import shard_keyswitch_bench

            shard_keyswitch_bench.main(["--steps", str(max(args.steps, 20)), "--warmup", str(max(args.warmup, 3))])
            return
        shard_model_bench(args)
        return

    # ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with more streams than queues two image
    # streams serialise behind each other (measured: 4 image streams + this thread's own = 0.99 images/s with 4 queues,
    # 1.24 with 8).  Must be in the environment before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # the runtime library prints the reference's stdout contract ([RT_STAT] ..., ckks_param: ...) from C;
    # keep fd 1 clean for the ONE JSON line: route everything else to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    cpu_res = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline and not args.roofline_only:
        # timed first: the host cores are idle and no child process is started once the GPU is initialised
        want_model = args.workload != "keyswitch" and os.path.exists(MODEL_LIB)
        cpu_res = cpu_baseline(want_model)

    r110 = args.workload == "resnet110"
    if r110:
        MODEL_LIB = os.path.join(ROOT, "workloads", "_gen", "models", "libmodel_resnet110.so")
    import ace_compiler_amd as A
    from ace_compiler_amd.dist import Ranks

    ranks = Ranks()  # one process per GPU; RCCL ("nccl") carries only the barrier and the max over ranks
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world

    import numpy as np

    bmod = sys.modules["ace_compiler_amd.build"]
    use_model = (args.workload != "keyswitch" and not args.roofline_only and os.path.exists(MODEL_LIB) and
                 os.path.exists(bmod.RT_LIB))
    if args.workload in ("resnet20", "resnet110") and not use_model:
        raise SystemExit("bench: %s or libFHErt_ant.so missing (build with tools/build_models.py / __graft_entry__.build())" % MODEL_LIB)

    rt = A.AceHip(N, L, Q0, SF, DNUM, device=local_rank)  # raises without GPU / library: no fallback
    lib, h = rt.lib, rt.h

    def barrier():
        ranks.barrier()
        rt.sync()

    class Stat(C.Structure):
        _fields_ = [("calls", C.c_uint64), ("units", C.c_uint64), ("bytes", C.c_uint64)]

    def read_stats(reset):
        arr = (Stat * 16)()
        n = lib.acehip_stats(arr, 16, 1 if reset else 0)
        return {lib.acehip_stat_name(i).decode(): (arr[i].calls, arr[i].units, arr[i].bytes) for i in range(n)}

    # ---------------- headline: ResNet-20 images/s (or the C3 key-switch fallback) ----------------
    T = L + rt.K
    poly_words = T * N
    rng = np.random.default_rng(1234 + rank)
    host = np.empty((T, N), dtype=np.uint64)
    for l in range(T):
        host[l] = rng.integers(0, rt.primes[l], size=N, dtype=np.uint64)
    a = rt.buf(L * N)
    rt.check(lib.acehip_memcpy_h2d(a.ptr, host.ctypes.data, L * N * 8, None))
    key = rt.buf(DNUM * 2 * poly_words)
    for d in range(DNUM * 2):
        rt.check(lib.acehip_memcpy_h2d(key.at(d * poly_words), host.ctypes.data, poly_words * 8, None))
    o0, o1 = rt.buf(L * N), rt.buf(L * N)

    def ks():
        rt.check(lib.acehip_key_switch(h, o0.ptr, o1.ptr, a.ptr, key.ptr, L, None))

    logits = None
    fix, verify_prefix, verify_note, weights_note = None, None, None, "synthetic weights N(0,0.05)"
    n_streams = max(args.streams, 1) if use_model else 1
    n_batch = max(args.batch, 1) if use_model else 1
    if use_model:
        import threading

        # Image streams: the reference runs one OpenMP thread per image on one context (resnet_cifar.main.inc:77-116);
        # same structure here: the context (keys) is prepared once, every image thread attaches to it with its own
        # scratch, pool, queue and HIP stream, so the small dependent kernels of several images overlap on the GPU.
        # Verification inside the timed region (not a separate run): with the fixture's ACEHIP_SEED the context holds the key set
        # the REFERENCE rtlib was given for its CPU run of this program (tests/golden/gen_parity.json), the weights are that run's
        # file, and the first timed batch of every stream carries that run's image, encrypted with that run's randomness, in batch
        # position (stream index): its output ciphertext must hash to the reference's digest -- byte-identical to the CPU rtlib.
        vkey = "resnet110" if r110 else "resnet20"  # (both have a reference run in the fixture: 2.6 h / 0.6 h of one CPU core)
        if True:
            fix = reference_fixture(vkey)
            if fix is None:
                verify_note = "tests/golden/gen_parity.json has no %s entry" % vkey
            elif os.environ.get("ACEHIP_SEED", str(fix["seed"])) != str(fix["seed"]) or "ACEHIP_RT_DATA_FILE" in os.environ:
                fix, verify_note = None, "ACEHIP_SEED / ACEHIP_RT_DATA_FILE set by the caller: not the fixture's keys or weights"
            else:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import model_weights
                import tempfile

                wfile, wmeta = model_weights.ensure(vkey, fix["weights"]["sigma"], fix["weights"].get("gen", "numpy"))
                os.environ["ACEHIP_RT_DATA_FILE"] = wfile
                os.environ["ACEHIP_SEED"] = str(fix["seed"])
                weights_note = "synthetic weights N(0,%g) (tools/make_weight_file.py, md5 %s)" % (fix["weights"]["sigma"], wmeta["md5"][:8])
                if wmeta["md5"] != fix["weights"]["md5"]:
                    fix, verify_note = None, "generator %s writes a different weight file here than the fixture's (md5 %s vs %s)" % (wmeta["gen"], wmeta["md5"], fix["weights"]["md5"])
                else:
                    verify_prefix = os.path.join(tempfile.gettempdir(), "acehip_bench_verify_%d_r%d" % (os.getpid(), rank))
        fhe, _ = load_model_runtime(local_rank)
        fhe.Prepare_context()  # this thread owns the context: keys are generated once (on the device) and shared
        gate = threading.Barrier(n_streams + 1)
        cmd = {"op": None}
        stream_logits = [None] * n_streams
        stream_digests = [[] for _ in range(n_streams)]
        stream_stats = [None] * n_streams
        stream_err = []

        def stream_main(i):
            try:
                _, one_image = load_model_runtime(local_rank, n_batch)  # per-thread image generator on the shared library
                fhe.Prepare_context()                          # attaches: shared keys, own scratch / pool / queue / stream
                fhe.Acehip_rt_set_batch(n_batch)
                while True:
                    gate.wait()
                    op = cmd["op"]
                    if op == "step":
                        for j in range(cmd["n"]):               # images back to back: streams are not kept in lock step
                            # the FIRST batch of a verified region carries the fixture's image in position i of stream i's batch
                            # EVERY batch of a verified region carries the fixture's image, in position (stream + batch number) of the batch
                            slot = (i + j) % n_batch
                            v = (slot, fix["enc_seed"], "%s_s%d" % (verify_prefix, i)) if cmd.get("verify") else None
                            stream_logits[i] = one_image(v)
                            if v is not None:  # hash the image's output ciphertext now (the next batch reuses the file names)
                                import glob as _glob
                                import hashlib as _hashlib

                                path = "%s.%d" % (v[2], slot)
                                stream_digests[i].append(_hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None)
                                for f in _glob.glob(v[2] + ".*"):
                                    os.remove(f)
                    elif op == "reset":
                        read_stats(reset=True)                 # statistics are per thread as well
                    elif op == "stats":
                        stream_stats[i] = read_stats(reset=False)
                    elif op == "release":
                        fhe.Finalize_context()                 # a worker's Finalize only gives back its own state
                    elif op == "attach":
                        fhe.Prepare_context()
                        fhe.Acehip_rt_set_batch(n_batch)
                    gate.wait()
                    if op == "quit":
                        break
                fhe.Finalize_context()
            except BaseException as e:  # noqa: BLE001 -- report and release the barrier
                stream_err.append(repr(e))
                gate.abort()

        threads = [threading.Thread(target=stream_main, args=(i,), daemon=True) for i in range(n_streams)]
        for t in threads:
            t.start()

        def run_all(op, n=1, verify=False):
            cmd["op"], cmd["n"], cmd["verify"] = op, n, verify
            try:
                gate.wait()
                gate.wait()
            except threading.BrokenBarrierError:
                raise SystemExit("bench: an image stream failed: %s" % stream_err)

        def step(n=1, verify=False):  # n images on every stream of this GPU, concurrently
            run_all("step", n, verify)
            return stream_logits[0]

        unit, metric = "images/s", "encrypted images/sec (ResNet-20 CIFAR-10, N=2^16)"
        workload = ("C4 (BASELINE configs[3]): ACE-compiled ResNet-20/CIFAR-10 encrypted inference, N=2^16, L=34, dnum=3, "
                    "19 bootstraps, 227 rotation keys, 6044 weight plaintexts; synthetic image U(-1,1) and %s; "
                    "%d concurrent image stream(s) per GPU (host threads attached to one context and key set, one HIP stream "
                    "each) x batches of %d images per launch (the reference's own parallel axis is one OpenMP thread per image on "
                    "shared keys and weights: the images of a batch share launches, keys, twiddles and weight plaintexts); a step = "
                    "one batch per stream" % (weights_note, n_streams, n_batch))
        if r110:
            metric = "encrypted images/sec (ResNet-110 CIFAR-10, N=2^16) -- secondary measurement, not the BASELINE headline"
            workload = ("the workload of BASELINE configs[4] (ACE-compiled ResNet-110/CIFAR-10, resnet110_cifar10_train.onnx.inc: N=2^16, "
                        "36 464 weight plaintexts) run as REPLICAS: whole images per GPU, %d concurrent image streams per GPU x batches of %d images; "
                        "synthetic image and %s (with synthetic weights a 110-layer network leaves the range of the bootstrap on "
                        "both runtimes, so the logits are not meaningful: the work is measured, and the bytes are checked against the "
                        "reference's CPU run)" % (n_streams, n_batch, weights_note))
    elif args.roofline_only:
        def step():
            return None

        unit, metric = "steps/s", "none (--roofline-only: see roofline)"
        workload = "--roofline-only: no headline workload was run"
    else:
        def step():
            ks()
            return None

        unit, metric = "key-switches/s", "full key-switch throughput (N=2^16, L=25, dnum=4)"
        workload = ("C3 (BASELINE configs[2]) FALLBACK: workloads/_gen/models/libmodel_resnet20.so not present, so the headline ResNet-20 "
                    "workload cannot run; full key-switch N=2^16 L=25 dnum=4 K=7 on resident inputs")

    if use_model:
        if args.warmup:
            logits = step(args.warmup)
    else:
        for _ in range(args.warmup):
            logits = step()
    barrier()
    if use_model:
        run_all("reset")
    else:
        read_stats(reset=True)
    t0 = time.perf_counter()
    if use_model:
        logits = step(args.steps, verify=fix is not None)  # every stream runs its K batches back to back; the region ends when all are done
    else:
        for _ in range(args.steps):
            logits = step()
    rt.sync()
    elapsed_local = time.perf_counter() - t0
    if use_model:
        run_all("stats")
        stats = stream_stats[0]
    else:
        stats = read_stats(reset=False)
    barrier()
    elapsed = ranks.max_over_ranks(elapsed_local)
    verification = None
    if use_model:
        verification = {"verified": None, "note": verify_note}
        if fix is not None:
            import glob
            import hashlib

            got = [g for i in range(n_streams) for g in stream_digests[i]]
            ok_here = 1.0 if (len(got) == n_streams * args.steps and all(g == fix["digest"] for g in got)) else 0.0
            ok_all = -ranks.max_over_ranks(-ok_here)
            verification = {
                "verified": bool(ok_all),
                "what": "inside the timed region: EVERY batch of every image stream of every rank carries the fixture's image (batch "
                        "position = stream index + batch number, modulo the batch size) under the fixture's key set and encryption "
                        "randomness; sha256 of its output ciphertext (ACEHCT01) against the digest of the REFERENCE rtlib's CPU run of the "
                        "same unchanged program (tests/golden/gen_parity.json, tests/c/gen_parity_ref.c)",
                "reference_digest": fix["digest"], "batches_checked_rank0": len(got), "batches_matching_rank0": sum(g == fix["digest"] for g in got),
                "streams_per_rank": n_streams, "ranks": world}
    value = world * n_streams * n_batch * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    cache_run = None
    latency = None
    if use_model:
        if world == 1 and n_batch == 1:  # (an image batch shares each encode among its images already)
            # secondary, NOT the headline: the same streams with the encoded weight plaintexts kept in HBM
            # (ACEHIP_PT_CACHE=1, 12.3 GB shared by the streams; the reference's pre-encoded DE_PLAINTEXT mode)
            run_all("release")
            fhe.Finalize_context()
            os.environ["ACEHIP_PT_CACHE"] = "1"
            fhe.Prepare_context()
            run_all("attach")
            step(1)  # fills the cache
            tc = time.perf_counter()
            step(2)
            dt = (time.perf_counter() - tc) / 2
            os.environ["ACEHIP_PT_CACHE"] = "0"
            cache_run = {"images_per_s": round(n_streams * n_batch / dt, 6), "ms_per_step": round(dt * 1e3, 3), "steps": 2,
                         "streams_per_gpu": n_streams,
                         "note": "ACEHIP_PT_CACHE=1: weight plaintexts encoded once and kept resident (12.3 GB, shared by the "
                                 "image streams); reported beside the headline, which encodes all weight plaintexts for every "
                                 "image like the reference run does"}
        run_all("quit")  # the image streams leave the GPU before the micro workloads are timed
        for t in threads:
            t.join()
        fhe.Finalize_context()
        # ---- latency of ONE image (B = 1, one stream), beside the throughput headline: the reference's published figure is a per-image
        # latency (scripts/ace_pre.log:28-30, 1453.96 s).  A fresh context of this thread; one untimed image (the weight-plaintext prefetch
        # predicts from the first image's call trace), then one timed image -- the fixture's, verified against the reference's digest
        if world == 1 and not r110 and not os.environ.get("ACEHIP_BENCH_NO_LATENCY"):
            try:
                fhe.Prepare_context()
                _, one_lat = load_model_runtime(local_rank, 1)
                one_lat()
                vlat = (0, fix["enc_seed"], "%s_lat" % verify_prefix) if (fix is not None and verify_prefix) else None
                tl = time.perf_counter()
                one_lat(vlat)
                latency = {"latency_s_single_image": round(time.perf_counter() - tl, 4),
                           "what": "Prepare_input (encode + encrypt) + Run_main_graph + Handle_output (decrypt + decode) of one image, one "
                                   "stream, one image per launch, second image of a fresh context"}
                if vlat is not None:
                    import hashlib as _h

                    pth = "%s.0" % vlat[2]
                    latency["verified"] = os.path.exists(pth) and _h.sha256(open(pth, "rb").read()).hexdigest() == fix["digest"]
                    if os.path.exists(pth):
                        os.remove(pth)
                fhe.Finalize_context()
            except Exception as e:  # noqa: BLE001 -- secondary: never fails the headline
                latency = {"error": repr(e)}
    # ---------------- roofline of the dominant kernel family: batched forward NTT ----------------
    n_polys = 2 * N_CT
    batch = rt.buf(n_polys * poly_words)
    for p in range(n_polys):
        rt.check(lib.acehip_memcpy_h2d(batch.at(p * poly_words), host.ctypes.data, poly_words * 8, None))

    def fwd():
        rt.check(lib.acehip_ntt_batch(h, batch.ptr, poly_words, n_polys, L, 0, T, 0, None))

    def inv():
        rt.check(lib.acehip_ntt_batch(h, batch.ptr, poly_words, n_polys, L, 0, T, 1, None))

    reps = 10
    fwd_ms = inv_ms = float("nan")
    if args.roofline_batch in ("all", "c3"):
        for _ in range(2):
            fwd()
            inv()
        fwd_ms = inv_ms = 0.0
        for _ in range(reps):
            fwd_ms += rt.time_ms(fwd, 1)
            inv_ms += rt.time_ms(inv, 1)
        fwd_ms /= reps
        inv_ms /= reps
        back = np.empty((T, N), dtype=np.uint64)
        rt.check(lib.acehip_memcpy_d2h(back.ctypes.data, batch.at((n_polys - 1) * poly_words), poly_words * 8, None))
        if not os.environ.get("ACEHIP_BENCH_NO_VERIFY"):  # timing experiments with deliberately wrong kernels only
            assert np.array_equal(back, host), "NTT round trip over the timed batch is not the identity"
    limbs = n_polys * T
    bytes_per_dir = 16 * N * limbs  # algorithmic: read + write every limb once

    # the same two kernels on the HEADLINE workload's own primes (the generated ResNet-20: L=34, K=11, q0 51 bits, 33 scaling primes
    # of 50 bits, 60-bit P primes): 1024-limb batches again, (a) the limb mix of a key-switch in the middle of the network -- 32
    # PQ-extended polynomials at level 21: 20 scaling primes (FP64 butterflies, csrc/ntt_fp.hpp), limb 0 and 11 P-limbs (integer
    # classes) -- and (b) scaling primes only (32 polynomials x limbs 1..32)
    wl = None
    if True:
        rt2 = A.AceHip(65536, 34, 51, 50, 3, device=local_rank)
        T2 = 34 + rt2.K
        rng2 = np.random.default_rng(99 + rank)
        host2 = np.empty((T2, N), dtype=np.uint64)
        for l in range(T2):
            host2[l] = rng2.integers(0, rt2.primes[l], size=N, dtype=np.uint64)

        def time_batch(level, pos0, n_limbs, n_p=32):
            words = n_limbs * N
            buf = rt2.buf(n_p * words)
            rows = [pos0 + i if pos0 + i < level else 34 + (pos0 + i - level) for i in range(n_limbs)]  # prime of every position
            src = np.ascontiguousarray(host2[rows])
            for p_ in range(n_p):
                rt2.check(rt2.lib.acehip_memcpy_h2d(buf.at(p_ * words), src.ctypes.data, words * 8, None))
            base = buf.ptr - pos0 * N * 8
            f_ = lambda: rt2.check(rt2.lib.acehip_ntt_batch(rt2.h, base, words, n_p, level, pos0, n_limbs, 0, None))  # noqa: E731
            i_ = lambda: rt2.check(rt2.lib.acehip_ntt_batch(rt2.h, base, words, n_p, level, pos0, n_limbs, 1, None))  # noqa: E731
            for _ in range(2):
                f_()
                i_()
            tf = ti = 0.0
            for _ in range(reps):
                tf += rt2.time_ms(f_, 1)
                ti += rt2.time_ms(i_, 1)
            got = np.empty_like(src)
            rt2.check(rt2.lib.acehip_memcpy_d2h(got.ctypes.data, buf.at((n_p - 1) * words), words * 8, None))
            if not os.environ.get("ACEHIP_BENCH_NO_VERIFY"):
                assert np.array_equal(got, src), "NTT round trip over the timed batch is not the identity"
            buf.free()
            return tf / reps, ti / reps, n_p * n_limbs

        nan3 = (float("nan"), float("nan"), 1024)
        mix_f, mix_i, mix_limbs = time_batch(21, 0, 32) if args.roofline_batch in ("all", "mix") else nan3
        fp_f, fp_i, fp_limbs = time_batch(34, 1, 32) if args.roofline_batch in ("all", "scaling") else nan3
        wl = {"mix": (mix_f, mix_i, mix_limbs), "fp": (fp_f, fp_i, fp_limbs), "fp_on": os.environ.get("ACEHIP_NTT_FP", "1") != "0"}
        rt2.close()

    if not args.roofline_only:
        for _ in range(3):
            ks()
    ks_ms = rt.time_ms(ks, 20) if not args.roofline_only else float("nan")
    ks_bytes = lib.acehip_key_switch_bytes(h, L)
    # the same C3 key-switch as a THROUGHPUT figure: KS_BATCH independent ciphertexts per launch through the replica mechanism the
    # image batches use (acehip_ctx_set_arena / acehip_ctx_select, include/acehip.h: inputs, outputs and the pipeline's workspace
    # exist once per replica inside one arena, the switch key -- outside it -- is shared and read once per launch).  One launch set
    # then covers KS_BATCH x the limbs of the single operation, which separates the kernels' efficiency from launch latency.
    ks_batched = None
    if not args.roofline_only:
        from ace_compiler_amd.binding import ArenaCfg

        KS_BATCH = 12
        # (a context of its own: an arena that holds the pipeline's workspace stays with its context for good)
        rtb = A.AceHip(N, L, Q0, SF, DNUM, device=local_rank)
        hb = rtb.h
        lib.acehip_workspace_words.restype = C.c_size_t
        ws_words = lib.acehip_workspace_words(hb)
        gran = lambda w: (w + 31) // 32 * 32  # noqa: E731
        off_ws, off_sc = 0, gran(ws_words)
        off_a = off_sc + 2 * N
        off_o0 = off_a + gran(L * N)
        off_o1 = off_o0 + gran(L * N)
        rep_words = off_o1 + gran(L * N)
        arena = rtb.buf(rep_words * KS_BATCH)
        cfg = ArenaCfg(arena.ptr, rep_words * 8, rep_words * 8, KS_BATCH, arena.at(off_ws), arena.at(off_sc), 2)
        rtb.check(lib.acehip_ctx_set_arena(hb, C.byref(cfg)))
        rtb.check(lib.acehip_ctx_select(hb, 0, KS_BATCH))
        rtb.check(lib.acehip_upload(hb, arena.at(off_a), host.ctypes.data, L * N * 8, None))  # every selected replica gets the input

        def ks_b():
            rtb.check(lib.acehip_key_switch(hb, arena.at(off_o0), arena.at(off_o1), arena.at(off_a), key.ptr, L, None))

        for _ in range(3):
            ks_b()
        ksb_ms = rtb.time_ms(ks_b, 20)
        # every replica must hold the bits of the single operation (same input, same key)
        ref0 = np.empty(L * N, dtype=np.uint64)
        got = np.empty(L * N, dtype=np.uint64)
        ks()
        rt.check(lib.acehip_memcpy_d2h(ref0.ctypes.data, o0.ptr, L * N * 8, None))
        same = True
        for r in (0, KS_BATCH // 2, KS_BATCH - 1):
            rtb.check(lib.acehip_ctx_select(hb, r, 1))
            rtb.check(lib.acehip_download(hb, got.ctypes.data, arena.at(off_o0), L * N * 8, None))
            same = same and bool(np.array_equal(got, ref0))
        rtb.sync()
        arena.free()
        rtb.close()
        ks_batched = {"ciphertexts_per_launch": KS_BATCH, "ms_per_launch_set": round(ksb_ms, 4), "ms_per_key_switch": round(ksb_ms / KS_BATCH, 4),
                      "per_s": round(KS_BATCH * 1e3 / ksb_ms, 1),
                      # the key is read once per launch set: algorithmic bytes = B x (single - key) + key
                      "algorithmic_bytes": int(KS_BATCH * (ks_bytes - DNUM * 2 * poly_words * 8) + DNUM * 2 * poly_words * 8),
                      "outputs_equal_single_operation": same}
        ks_batched["achieved_GBs"] = round(ks_batched["algorithmic_bytes"] / (ksb_ms * 1e-3) / 1e9, 1)
        ks_batched["frac_of_hbm_peak"] = round(ks_batched["achieved_GBs"] / HBM_PEAK_GBS, 4)
        # (priced like the single operation -- every operation reading its own copy of the key -- the figure would be:)
        ks_batched["frac_of_hbm_peak_if_key_counted_per_operation"] = round(KS_BATCH * ks_bytes / (ksb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    if rank == 0:
        # the roofline object reports the batch with the workload's own limb mix; the C3-parameter batch (56-bit primes, integer
        # SMALL class: the figure of rounds 1-3) and the scaling-prime batch stay beside it
        c3_fwd_ms, c3_inv_ms, c3_limbs = fwd_ms, inv_ms, limbs
        fwd_ms, inv_ms, limbs = wl["mix"]
        bytes_per_dir = 16 * N * limbs
        achieved = bytes_per_dir / (fwd_ms * 1e-3) / 1e9
        traffic = image_traffic = ntt_kernel_s = None
        traffic_src = {"file": "profiles/traffic.json", "status": "absent"}
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):  # PMC passes (tools/pmc_roofline.sh, tools/pmc_image.sh) recorded under profiles/
            # counters are collected in separate rocprofv3 --pmc runs, not in this process: the file says which sources it was
            # measured on (tools/csrc_fingerprint.py) and for which batch; figures of other sources are NOT reported as measured
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import csrc_fingerprint

            tr = json.load(open(tr_path))
            now_fp = csrc_fingerprint.fingerprint()
            fresh = tr.get("csrc_fingerprint") == now_fp
            traffic_src = {"file": "profiles/traffic.json", "measured_on_csrc": tr.get("csrc_fingerprint"), "csrc_now": now_fp,
                           "measured_at_commit": tr.get("git_commit"), "images_per_batch_measured": tr.get("images_per_batch"),
                           "status": "current" if fresh else "stale: kernel / runtime sources changed since the counters were collected (figures withheld)"}
            if fresh:
                traffic = tr.get("ntt_forward_bytes_per_launch")
                if not r110 and tr.get("images_per_batch") == n_batch:
                    image_traffic = tr.get("resnet20_bytes_per_image")
                    ntt_kernel_s = (tr.get("resnet20_kernel_seconds_per_image") or {}).get("ntt")
                elif not r110:
                    traffic_src["status"] += "; whole-image traffic was measured with %s images per batch, this run uses %d" % (tr.get("images_per_batch"), n_batch)
        out = {
            "metric": metric, "value": round(value, 6), "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (round(value / (1.0 / 7531.12 if r110 else BASELINE_IMAGES_PER_S), 1) if use_model else None),  # ace_pre.log
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "N": 65536, "streams_per_gpu": n_streams, "images_per_batch": n_batch,
                       "images_per_step": world * n_streams * n_batch,
                       "parallelism": "replicas: %d GPU(s) x %d image stream(s) per GPU x %d images per batch" % (world, n_streams, n_batch)},
            "roofline": {"bound": "hbm",
                         "kernel": "ntt8_strided_kernel<fwd> + ntt8_contig_kernel<fwd> (one forward NTT launch = 2 passes of 8 radix-2 stages)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": round(fwd_ms, 4), "inverse_launch_ms": round(inv_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_dir,
                         "ntt_fwd_inv_GBs": round(2 * bytes_per_dir / ((fwd_ms + inv_ms) * 1e-3) / 1e9, 2),
                         "batch": "%d limbs, 512 MiB: 32 PQ-extended polynomials at level 21 of the headline workload's parameter set (generated "
                                  "ResNet-20: L=34, K=11) -- per polynomial 20 scaling primes of 50 bits (FP64 butterflies%s), limb 0 (51 bits) and "
                                  "11 P-limbs of 60 bits (integer butterflies)" % (limbs, "" if wl["fp_on"] else " SWITCHED OFF: ACEHIP_NTT_FP=0"),
                         "other_batches": {
                             "scaling_primes_only": {"what": "32 polynomials x limbs 1..32 of the same set (50-bit primes only)", "limbs": wl["fp"][2],
                                                     "launch_ms": round(wl["fp"][0], 4), "inverse_launch_ms": round(wl["fp"][1], 4),
                                                     "frac": round(16 * N * wl["fp"][2] / (wl["fp"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                             "c3_parameter_set": {"what": "%d limbs (%d PQ-extended ciphertext pairs of BASELINE configs[2]: L=25 K=7, 56-bit scaling primes: "
                                                          "integer butterflies; the roofline batch of rounds 1-3)" % (c3_limbs, N_CT), "limbs": c3_limbs,
                                                  "launch_ms": round(c3_fwd_ms, 4), "inverse_launch_ms": round(c3_inv_ms, 4),
                                                  "frac": round(16 * N * c3_limbs / (c3_fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}},
            "key_switch": {"workload": "C3 (BASELINE configs[2]): full key-switch N=2^16 L=25 dnum=4 K=7",
                           "ms": round(ks_ms, 4), "per_s": round(1e3 / ks_ms, 2), "algorithmic_bytes": int(ks_bytes),
                           "achieved_GBs": round(ks_bytes / (ks_ms * 1e-3) / 1e9, 2),
                           "frac_of_hbm_peak": round(ks_bytes / (ks_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "batched": ks_batched},
        }
        # whole-workload view: algorithmic bytes of every entry-point call of the timed region on this rank
        alg = sum(v[2] for k, v in stats.items() if k not in ("zero_fill_executed", "elementwise_mul", "ntt_launched"))  # (subsets / a launch counter)
        out["workload_roofline"] = {
            "algorithmic_bytes_per_image": int(alg / args.steps / n_batch),
            "algorithmic_GBs": round(n_streams * alg / elapsed_local / 1e9, 2),
            "algorithmic_frac_of_hbm_peak": round(n_streams * alg / elapsed_local / 1e9 / HBM_PEAK_GBS, 4),
            # counter-based: L2-miss traffic of one image (FETCH_SIZE x2 + WRITE_SIZE over every dispatch of a one-stream run,
            # profiles/traffic.json) times the measured images/s of this run; Infinity-Cache hits are inside this figure
            "measured_bytes_per_image": image_traffic,
            "measured_GBs": (round(image_traffic * value / world / 1e9, 2) if (image_traffic and use_model) else None),
            "measured_frac_of_hbm_peak": (round(image_traffic * value / world / 1e9 / HBM_PEAK_GBS, 4) if (image_traffic and use_model) else None),
            "calls_per_step": {k: round(v[0] / args.steps, 1) for k, v in stats.items() if v[0]},
            "GB_per_step": {k: round(v[2] / args.steps / 1e9, 2) for k, v in stats.items() if v[2]},
            "note": "algorithmic_*: sum over acehip_* calls of the SURVEY 8(d) per-call bytes (tables, scratch, re-reads excluded) of one "
                    "stream, times the streams of the GPU, / wall time -- NOT a memory-traffic figure (chains of per-limb ops keep "
                    "intermediates in registers, dead fills are dropped); measured_*: hardware counters"}
        if use_model:
            # the same kernels inside the workload: every limb-transform the library launched for one image (direct calls and the
            # ones inside ModUp / ModDown / Rescale / key-switch / encode), against their summed kernel time from the rocprofv3
            # --kernel-trace --stats run recorded in profiles/traffic.json (one stream; withheld when the sources changed since)
            n_lt = stats.get("ntt_launched", (0, 0, 0))[1] / (args.steps * n_batch)
            out["roofline"]["in_workload"] = {
                "limb_transforms_per_image": round(n_lt, 1),
                "kernel_seconds_per_image": ntt_kernel_s,
                "us_per_limb_transform": (round(ntt_kernel_s / n_lt * 1e6, 4) if (ntt_kernel_s and n_lt) else None),
                "best_case_batch_us_per_limb_transform": round((fwd_ms + inv_ms) / 2 * 1e3 / limbs, 4),
                "GBs_algorithmic": (round(n_lt * 16 * N / ntt_kernel_s / 1e9, 1) if (ntt_kernel_s and n_lt) else None)}
        if verification is not None:
            out["verified"] = verification["verified"]
            out["verification"] = verification
        if cache_run is not None:
            out["with_plaintext_cache"] = cache_run
        if latency is not None:
            out["latency_s_single_image"] = latency.get("latency_s_single_image")
            out["single_image_latency"] = latency
        if not args.roofline_only:
            # the C3 key-switch (BASELINE configs[2]) inside the roofline object as well (the driver's record keeps this object whole)
            out["roofline"]["key_switch_ms"] = out["key_switch"]["ms"]
            out["roofline"]["key_switch_frac"] = out["key_switch"]["frac_of_hbm_peak"]
            if ks_batched:
                out["roofline"]["key_switch_batched_per_s"] = ks_batched["per_s"]
                out["roofline"]["key_switch_batched_ms"] = ks_batched["ms_per_key_switch"]
                out["roofline"]["key_switch_batched_frac"] = ks_batched["frac_of_hbm_peak"]
        if logits is not None:
            out["config"]["last_logits"] = [round(v, 5) for v in logits]
        if args.roofline_only:
            out.pop("key_switch")
            out.pop("workload_roofline")
        if cpu_res is not None:
            if use_model:
                # (the images of a batch share their weight-plaintext encodes here; the reference encodes them for every image)
                finish_cpu_baseline(cpu_res, {k: tuple(x / (args.steps * (1 if k == "encode" else n_batch)) for x in v) for k, v in stats.items()})
            out["cpu_baseline"] = cpu_res
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            ref_v = cpu_res.get("socket_extrapolated", cpu_res)["value"]
            if cpu_res["unit"] == unit and ref_v > 0:
                out["cpu_baseline"]["gpu_over_cpu_measured"] = round(value / cpu_res["value"], 1)
                out["cpu_baseline"]["gpu_over_one_socket"] = round(value / ref_v, 1)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    # Secondary and opt-in, AFTER the headline line is on fd 1 (a failure or a hang of it can no longer delay or lose the line):
    # ACEHIP_BENCH_SHARD_LEG=1 with --gpus N > 1 runs one ResNet-20 image limb-sharded over the same ranks (child processes, own
    # communicator, bounded by a timeout).  Its result goes to stderr ("[bench] limb_sharded {...}") and, where gpurun_out/ exists,
    # to gpurun_out/limb_sharded_leg.json -- never into the headline line.
    force_leg = os.environ.get("ACEHIP_BENCH_FORCE_SHARD_LEG") == "1"  # test hook: exercise the leg's plumbing with one rank
    want_leg = (world > 1 and os.environ.get("ACEHIP_BENCH_SHARD_LEG") == "1" and not args.no_shard_leg) or force_leg
    if want_leg and use_model and not r110:
        shard_leg = limb_sharded_leg(ranks)  # (never raises before its three reductions are through: every failure path is an "error" entry)
        if rank == 0 and shard_leg is not None:
            sys.stderr.write("[bench] limb_sharded " + json.dumps(shard_leg) + "\n")
            sys.stderr.flush()
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                json.dump(shard_leg, open(os.path.join(ROOT, "gpurun_out", "limb_sharded_leg.json"), "w"), indent=1)
    ranks.close()
    rt.close()
    # a model run that did not prove its timed images byte-identical to the reference's CPU run is not a result (the line above says
    # why: "verified" / "verification.note"); --no-verify is for experiments that change keys, weights or kernels on purpose
    if use_model and not (args.no_verify or os.environ.get("ACEHIP_BENCH_NO_VERIFY")) and (verification is None or verification.get("verified") is not True):
        sys.stderr.write("[bench] NOT VERIFIED against the reference digest (%s): exit 3 (--no-verify to run anyway)\n" %
                         ((verification or {}).get("note") or "digests differ"))
        sys.exit(3)


if __name__ == "__main__":
    main()
