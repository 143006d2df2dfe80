"""The CPU-baseline arithmetic of bench.py (no GPU, no reference needed): the profile of the reference's full run moved to another host
family by family (profile_scaled_image), the call-statistics pricing (price_image), and the sample bucketing of tools/ref_profile_report.py."""
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

DEV = json.load(open(os.path.join(ROOT, "profiles", "cpu_resnet20_devbox.json")))


def test_profile_scaled_image_is_exact_on_the_machine_the_profile_was_taken_on():
    prof, med = DEV["r04_profile"], DEV["mix_level20_cold_median"]["median"]
    total, parts = bench.profile_scaled_image(prof, med, med)
    assert abs(total - sum(prof["by_family_s"].values())) < 1e-6
    assert abs(total - prof["main_graph_s"]) < 1.0  # the families add up to the sampled span
    assert parts["mul"] > parts["ntt_fwd"] > parts["ntt_inv"]  # what the profile says: the multiply family is the largest


def test_profile_scaled_image_moves_each_family_by_its_own_primitive():
    prof, med = DEV["r04_profile"], DEV["mix_level20_cold_median"]["median"]
    host = dict(med)
    host["hw_modmul_s"] = med["hw_modmul_s"] / 2  # a host whose multiply loop is twice as fast, everything else equal
    total, parts = bench.profile_scaled_image(prof, med, host)
    assert abs(parts["mul"] - prof["by_family_s"]["mul"] / 2) < 1e-6
    assert abs(parts["ntt_fwd"] - prof["by_family_s"]["ntt_fwd"]) < 1e-6
    host = dict(med, memset_GBs=med["memset_GBs"] * 4)
    _, parts = bench.profile_scaled_image(prof, med, host)
    assert abs(parts["memset"] - prof["by_family_s"]["memset"] / 4) < 1e-6
    # an older ref_dump without the memset / memcpy figures: no estimate rather than a wrong one
    old = {k: v for k, v in med.items() if not k.startswith("mem")}
    assert bench.profile_scaled_image(prof, med, old) == (None, None)


def test_call_statistics_pricing_is_linear_in_the_statistics():
    med = DEV["mix_level20_cold_median"]["median"]
    limb = 8.0 * 65536
    st = {"mod_down": (10, 0, 10 * limb * (2 * 20 + 11)), "elementwise": (0, 0, 3 * limb * 100), "elementwise_mul": (0, 40, 3 * limb * 40)}
    one, parts = bench.price_image(med, st)
    assert abs(parts["mod_down"] - 10 * med["mod_down_s"]) < 1e-9   # ten Mod_downs at the level the primitive was timed at
    assert abs(parts["elementwise"] - (40 * med["hw_modmul_s"] + 60 * med["hw_modadd_s"])) < 1e-9
    two, _ = bench.price_image(med, {k: tuple(2 * x for x in v) for k, v in st.items()})
    assert abs(two - 2 * one) < 1e-9


def test_profile_report_buckets_samples_by_function_family(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_profile_report as R

    assert R.family_of("Forward_transform") == "ntt_fwd" and R.family_of("Inverse_transform") == "ntt_inv"
    assert R.family_of("Multiply_add") == "mul" and R.family_of("Hw_modmul") == "mul" and R.family_of("Add_poly") == "add"
    assert R.family_of("Automorphism_transform") == "permute" and R.family_of("Fast_base_conv") == "conversion"
    assert R.family_of("memset (rep stos)") == "memset" and R.family_of("memcpy (rep movs)") == "memcpy"
    assert R.family_of("Embedding_inv") == "encode" and R.family_of("main") == "other"
    # the committed samples of the full run reproduce the committed report
    samples = os.path.join(ROOT, "profiles", "r04_ref_resnet20.samples")
    out = tmp_path / "p.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ref_profile_report.py"), samples, "--seconds", "1534.1", "--json", str(out)],
                          stdout=subprocess.DEVNULL)
    got = json.load(open(out))
    assert got["samples"] == 384762
    # (modules that are absent on this machine fall back to the nearest exported name the sampler wrote: the big families do not depend on it)
    for fam in ("ntt_fwd", "ntt_inv"):
        assert abs(got["by_family_s"][fam] - DEV["r04_profile"]["by_family_s"][fam]) < 2.0
