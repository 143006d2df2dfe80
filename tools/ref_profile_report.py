#!/usr/bin/env python3
"""Flat profile of a REFERENCE run sampled by tests/c/ref_sampler.c (REF_SAMPLER_OUT): samples -> functions -> the families that
bench.py's price_image() prices.  Test / measurement infrastructure.

    python3 tools/ref_profile_report.py <samples file> [--json out.json] [--top 40]

Every line of the samples file is "<module> <offset> <count> <nearest exported symbol>".  Offsets are bucketed with the module's own
symbol table (`nm -n --defined-only`: static functions included; stripped system libraries fall back to the exported name the
sampler wrote).  1 sample = 1 ms of CPU time (ITIMER_PROF: user + system, so page faults count where they are taken).
"""
import bisect
import collections
import json
import os
import subprocess
import sys

# function -> family.  Names are the reference's (fhe-cmplr/rtlib/ant/{poly,util}/src); what a name does was read off its source.
FAMILIES = [
    ("ntt", ("Ftt_fwd", "Ftt_inv", "Ntt_", "Intt_", "Forward_transform", "Inverse_transform")),
    ("hw_elementwise", ("Hw_modadd", "Hw_modmul", "Hw_modsub", "Hw_rotate", "Add_poly", "Sub_poly", "Multiply_poly", "Mul_poly", "Scalars_integer_multiply",
                        "Rotate_poly", "Automorphism", "Add_int64", "Mul_int64", "Fast_mul", "Mod_mul", "Mod_add", "Poly_")),
    ("base_conversion", ("Fast_convert", "Base_conv", "Reduce_rns_base", "Raise_rns_base", "Decompose", "Rescale_poly", "Mod_down", "Mod_up", "Switch_modulus",
                         "Approx_switch", "Fast_base", "Precompute", "Barrett")),
    ("encode", ("Embedding", "Encode", "Fft_", "Transform_values_to_rns", "Reverse_bits", "Cexp", "cexp", "sincos", "Rotation_group")),
    ("memory", ("memset", "memcpy", "memmove", "malloc", "calloc", "free", "realloc", "mmap", "munmap", "brk", "_int_", "sysmalloc", "madvise", "Alloc_", "Free_", "Init_poly", "Copy_poly")),
]


def family_of(name):
    for fam, keys in FAMILIES:
        for k in keys:
            if name.startswith(k) or ("_" + k) in name or name.lstrip("_").startswith(k):
                return fam
    return "other"


def symtab(module):
    try:
        out = subprocess.run(["nm", "-n", "--defined-only", module], capture_output=True, text=True).stdout
    except OSError:
        return [], []
    addrs, names = [], []
    for ln in out.splitlines():
        p = ln.split()
        if len(p) == 3 and p[1] in "tTwW":
            addrs.append(int(p[0], 16))
            names.append(p[2])
    return addrs, names


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    tabs, by_func, total = {}, collections.Counter(), 0
    for ln in open(path):
        if ln.startswith("#"):
            continue
        mod, off, cnt, near = ln.split()
        off, cnt = int(off, 16), int(cnt)
        total += cnt
        name = None
        if mod != "?" and os.path.exists(mod):
            if mod not in tabs:
                tabs[mod] = symtab(os.path.realpath(mod))
            addrs, names = tabs[mod]
            if addrs:
                k = bisect.bisect_right(addrs, off) - 1
                if k >= 0:
                    name = names[k]
        if name is None:
            name = near if near != "?" else "?"
        by_func[(os.path.basename(mod), name)] += cnt
    fam = collections.Counter()
    for (mod, name), c in by_func.items():
        f = family_of(name)
        if f == "other" and mod.startswith("libc"):
            f = "memory" if any(k in name for k in ("mem", "alloc", "free", "brk", "map")) else "other"
        if f == "other" and mod.startswith("libm"):
            f = "encode"  # (the only libm callers on the path: the encoder's twiddles and the rounding of embedded values)
        fam[f] += c
    print("%d samples = %.1f s of CPU" % (total, total / 1e3))
    print("\nby family:")
    for f, c in fam.most_common():
        print("  %-18s %9.1f s  %5.1f %%" % (f, c / 1e3, 100.0 * c / total))
    print("\nby function:")
    for (mod, name), c in by_func.most_common(top):
        print("  %-22s %-44s %9.1f s  %5.1f %%  [%s]" % (mod[:22], name[:44], c / 1e3, 100.0 * c / total, family_of(name)))
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        json.dump({"samples": total, "seconds": total / 1e3, "by_family_s": {f: c / 1e3 for f, c in fam.most_common()},
                   "by_function_s": [{"module": m, "function": n, "s": c / 1e3, "family": family_of(n)} for (m, n), c in by_func.most_common(200)]},
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
