#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CKKS polynomial layer.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "C2" of SURVEY 8d): forward + inverse negacyclic NTT at
N = 2^16 over a resident batch of 1024 limbs (16 PQ-extended ciphertext pairs of the C3 parameter
set L=25, K=7: 16 x 2 polys x 32 limbs = 512 MiB, larger than the 256 MiB Infinity Cache so the
numbers are HBM numbers).  A step = one forward NTT and one inverse NTT of every limb of the batch.
metric = algorithmic NTT bandwidth: 16*N bytes per limb-transform * transforms / time.
Each rank (one per GPU) owns an independent batch: weak scaling, no data-path collective
(ciphertexts are independent; RCCL is only used for the barrier / max-reduce of the timing).

Extra, same JSON line:
  roofline      dominant kernel family (forward NTT = its two pass kernels) timed with HIP events on
                the launch stream inside the timed region
  key_switch    BASELINE.json configs[2] (C3): full key-switch at N=2^16, L=25, dnum=4, events-timed
  cpu_baseline  the reference rtlib (oracle/_ref/ref_dump bench, kind "reference") or the oracle
                port, one host thread, bounded sample; rank 0 at N=1 only
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
N, L, Q0, SF, DNUM = 65536, 25, 60, 56, 4
N_CT = 16  # ciphertext pairs in the resident batch


def cpu_baseline():
    """Reference rtlib timed on this host (1 thread): ~10-20 s of CPU work."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref):
        try:
            out = subprocess.run([ref, "bench", str(N), str(L), str(Q0), str(SF), str(DNUM), str(L), "8", "3000"],
                                 capture_output=True, text=True, timeout=300, check=True).stdout
            r = json.loads(out.strip().splitlines()[-1])
            per = r["ntt_fwd_s"] + r["ntt_inv_s"]
            return {"value": round(2 * 16 * N / per / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "reference",
                    "sample": "3000 Ftt_fwd + 3000 Ftt_inv of one limb (N=2^16) and 8 full key-switches "
                              "(L=25,dnum=4) by the reference rtlib (gcc -O3), 1 thread",
                    "ntt_fwd_ms": round(r["ntt_fwd_s"] * 1e3, 4), "ntt_inv_ms": round(r["ntt_inv_s"] * 1e3, 4),
                    "key_switch_s": round(r["key_switch_s"], 4),
                    "key_switch_per_s": round(1.0 / r["key_switch_s"], 4)}
        except Exception as e:  # fall through to the port
            sys.stderr.write("reference baseline failed (%s); timing the oracle port instead\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

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
    per = (t2 - t0) / reps
    return {"value": round(2 * 16 * N / per / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "3000 fwd + 3000 inv NTTs of one limb (N=2^16) and 4 key-switches by oracle/ckks_oracle.c, 1 thread",
            "ntt_fwd_ms": round((t1 - t0) / reps * 1e3, 4), "ntt_inv_ms": round((t2 - t1) / reps * 1e3, 4),
            "key_switch_s": round((t4 - t3) / 4, 4), "key_switch_per_s": round(4 / (t4 - t3), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import numpy as np

    import ace_compiler_amd as A

    rt = A.AceHip(N, L, Q0, SF, DNUM, device=local_rank)  # raises without GPU / library: no fallback
    lib, h = rt.lib, rt.h
    T = L + rt.K  # 32 limbs per PQ-extended polynomial
    n_polys = 2 * N_CT
    poly_words = T * N
    batch = rt.buf(n_polys * poly_words)
    # synthetic data: one random extended polynomial (canonical residues per limb), replicated
    rng = np.random.default_rng(1234 + rank)
    host = np.empty((T, N), dtype=np.uint64)
    for l in range(T):
        host[l] = rng.integers(0, rt.primes[l], size=N, dtype=np.uint64)
    for p in range(n_polys):
        rt.check(lib.acehip_memcpy_h2d(batch.at(p * poly_words), host.ctypes.data, poly_words * 8, None))

    def fwd():
        rt.check(lib.acehip_ntt_batch(h, batch.ptr, poly_words, n_polys, L, 0, T, 0, None))

    def inv():
        rt.check(lib.acehip_ntt_batch(h, batch.ptr, poly_words, n_polys, L, 0, T, 1, None))

    def barrier():
        if dist is not None:
            dist.barrier()
        rt.sync()

    for _ in range(args.warmup):
        fwd()
        inv()
    barrier()
    ev = [[lib.acehip_event_create() for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        lib.acehip_event_record(ev[k][0], None)
        fwd()
        lib.acehip_event_record(ev[k][1], None)
        inv()
        lib.acehip_event_record(ev[k][2], None)
    rt.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    import ctypes as C

    ms = C.c_float()
    fwd_ms = inv_ms = 0.0
    for k in range(args.steps):
        lib.acehip_event_elapsed_ms(ev[k][0], ev[k][1], C.byref(ms))
        fwd_ms += ms.value
        lib.acehip_event_elapsed_ms(ev[k][1], ev[k][2], C.byref(ms))
        inv_ms += ms.value
    fwd_ms /= args.steps
    inv_ms /= args.steps
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity spot-check of the timed buffers: fwd then inv returned the input exactly
    back = np.empty((T, N), dtype=np.uint64)
    rt.check(lib.acehip_memcpy_d2h(back.ctypes.data, batch.at((n_polys - 1) * poly_words), poly_words * 8, None))
    assert np.array_equal(back, host), "NTT round trip over the timed batch is not the identity"

    limbs = n_polys * T
    bytes_per_dir = 16 * N * limbs           # algorithmic: read + write every limb once
    step_bytes = 2 * bytes_per_dir
    ms_per_step = elapsed / args.steps * 1e3
    value = world * step_bytes / (elapsed / args.steps) / 1e9

    # C3 key-switch, events-timed (not part of the timed NTT region)
    a = rt.buf(L * N)
    rt.check(lib.acehip_memcpy_h2d(a.ptr, host.ctypes.data, L * N * 8, None))
    key = rt.buf(DNUM * 2 * poly_words)
    for d in range(DNUM * 2):
        rt.check(lib.acehip_memcpy_h2d(key.at(d * poly_words), host.ctypes.data, poly_words * 8, None))
    o0, o1 = rt.buf(L * N), rt.buf(L * N)

    def ks():
        rt.check(lib.acehip_key_switch(h, o0.ptr, o1.ptr, a.ptr, key.ptr, L, None))

    for _ in range(3):
        ks()
    ks_ms = rt.time_ms(ks, 20)
    ks_bytes = lib.acehip_key_switch_bytes(h, L)

    if rank == 0:
        achieved = bytes_per_dir / (fwd_ms * 1e-3) / 1e9
        traffic = None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            traffic = json.load(open(tr_path)).get("ntt_forward_bytes_per_launch")
        out = {
            "metric": "negacyclic NTT algorithmic bandwidth (fwd+inv, N=2^16, 64-bit primes)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "C2 (BASELINE configs[1]): forward+inverse negacyclic NTT, N=2^16, batch of %d limbs "
                                   "per GPU (%d PQ-extended ciphertext pairs, L=25 K=7), bit-exact vs CPU rtlib" % (limbs, N_CT),
                       "N": N, "limbs_per_gpu": limbs, "bytes_per_step_per_gpu": step_bytes, "parallelism": "replicas x%d" % world},
            "roofline": {"bound": "hbm", "kernel": "ntt8_strided_kernel<fwd> + ntt8_contig_kernel<fwd> (one forward NTT launch = 2 passes of 8 radix-2 stages)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_ms": round(fwd_ms, 4), "inverse_launch_ms": round(inv_ms, 4),
                         "algorithmic_bytes_per_launch": bytes_per_dir},
            "key_switch": {"workload": "C3 (BASELINE configs[2]): full key-switch N=2^16 L=25 dnum=4 K=7",
                           "ms": round(ks_ms, 4), "per_s": round(1e3 / ks_ms, 2),
                           "algorithmic_bytes": int(ks_bytes),
                           "achieved_GBs": round(ks_bytes / (ks_ms * 1e-3) / 1e9, 2),
                           "frac_of_hbm_peak": round(ks_bytes / (ks_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    rt.close()


if __name__ == "__main__":
    main()
