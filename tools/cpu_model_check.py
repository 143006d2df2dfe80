#!/usr/bin/env python3
"""Checks bench.py's CPU-baseline MODEL on the machine where it is applied (measurement infrastructure; uses oracle/_ref, the checker).

bench.py cannot run the reference's ResNet-20 on the bench host (550 s of key set-up + 1 500 s per image); `cpu_baseline.value` is the
flat profile of the reference's full run on the dev container moved to the bench host family by family with the host/dev ratio of the
reference primitive that is each family's inner loop (bench.py profile_scaled_image).  This tool applies THE SAME transfer to a real
reference program that the bench host CAN finish -- tests/c/ct_parity.c -DREF_BUILD (oracle/_ref/ct_parity_ref): key generation,
encryption, the ciphertext-level operator script and two bootstraps at the generated ResNet-20's ring (N = 2^16, L = 34, dnum = 3;
about 90 s of one core) -- and compares the prediction with the measured CPU seconds of the script span (`script_cpu_s`, the reference
binary's own clock):

  dev   (dev container)  runs the program under the sampler (tests/c/ref_sampler.c), buckets the span into function families
                         (tools/ref_profile_report.py) and writes profiles/r05_ct_parity_profile_dev.json together with the dev
                         container's primitive timings (`ref_dump mix`, median of three runs)
  host  (bench host; under gpurun, no GPU needed)  runs the program plainly, times the primitives with the same `ref_dump mix` on the
                         same core, predicts the span from the committed dev profile and prints / writes
                         {"predicted_s", "measured_s", "predicted_over_measured"}  ->  gpurun_out/cpu_model_check.json
                         (copy to profiles/cpu_model_check_bench_host.json: bench.py reports it as cpu_baseline.model_check_on_bench_host)
"""
import json
import os
import re
import statistics
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
ARGS = "65536 33 51 50 3 192 4096 15 1 -5".split()   # the gpu_slow parity case of tests/test_gpu_ct_parity.py
MIX = "65536 34 51 50 3 20 6".split()
DEV_JSON = os.path.join(ROOT, "profiles", "r05_ct_parity_profile_dev.json")


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "?"


def run_script(sampler_out=None):
    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ)
        if sampler_out:
            env["REF_SAMPLER_OUT"] = sampler_out
        r = subprocess.run([os.path.join(REF, "ct_parity_ref"), "dump", d] + ARGS, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        m = re.search(r"script_cpu_s (\d+\.\d+)", r.stdout)
        assert m, r.stdout[-2000:]
        return float(m.group(1))


def mix(n=1):
    runs = []
    for _ in range(n):
        out = subprocess.run([os.path.join(REF, "ref_dump"), "mix"] + MIX, capture_output=True, text=True, check=True).stdout
        runs.append(json.loads(out.strip().splitlines()[-1]))
    med = dict(runs[0])
    for k, v in runs[0].items():
        if isinstance(v, float):
            med[k] = statistics.median(r[k] for r in runs)
    return med


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "host"
    if mode == "dev":
        with tempfile.TemporaryDirectory() as d:
            samples = os.path.join(d, "samples.txt")
            cpu_s = run_script(samples)
            rep = os.path.join(d, "report.json")
            txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_profile_report.py"), samples, "--seconds", str(cpu_s), "--json", rep],
                                 capture_output=True, text=True, check=True).stdout
            prof = json.load(open(rep))
        out = {"what": "oracle/_ref/ct_parity_ref dump %s: the operator script + two bootstraps of the REFERENCE rtlib, sampled "
                       "(tests/c/ref_sampler.c); seconds per function family over the script span" % " ".join(ARGS),
               "cpu": cpu_model(), "script_cpu_s": cpu_s, "by_family_s": prof["by_family_s"], "by_function_s": prof["by_function_s"][:15],
               "mix_level20_cold_median_of_3": mix(3)}
        json.dump(out, open(DEV_JSON, "w"), indent=1)
        open(os.path.join(ROOT, "profiles", "r05_ct_parity_profile_dev.txt"), "w").write(txt)
        print(json.dumps({k: out[k] for k in ("cpu", "script_cpu_s", "by_family_s")}))
        return
    import bench  # profile_scaled_image: the transfer bench.py's cpu_baseline.value uses

    dev = json.load(open(DEV_JSON))
    host_mix = mix(3)
    measured = run_script()
    prof = {"by_family_s": dev["by_family_s"], "main_graph_s": dev["script_cpu_s"]}
    pred, parts = bench.profile_scaled_image(prof, dev["mix_level20_cold_median_of_3"], host_mix)
    res = {"program": "oracle/_ref/ct_parity_ref dump " + " ".join(ARGS) + " (reference rtlib: operator script + 2 bootstraps, N=2^16 L=34)",
           "host_cpu": cpu_model(), "dev_cpu": dev["cpu"], "dev_measured_s": dev["script_cpu_s"],
           "predicted_s": round(pred, 2), "measured_s": round(measured, 2), "predicted_over_measured": round(pred / measured, 3),
           "predicted_by_family_s": {k: round(v, 2) for k, v in parts.items()},
           "method": "bench.py profile_scaled_image: dev-container seconds per function family x host/dev ratio of the reference primitive that is "
                     "the family's inner loop (`ref_dump mix`, cold operands, median of 3 on one core of each machine)",
           "host_mix": host_mix}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "cpu_model_check.json"), "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("host_cpu", "predicted_s", "measured_s", "predicted_over_measured")}))


if __name__ == "__main__":
    main()
