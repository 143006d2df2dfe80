#!/usr/bin/env python3
"""Flat profile of a REFERENCE run sampled by tests/c/ref_sampler.c (REF_SAMPLER_OUT): samples -> functions -> the families that
bench.py's price_image() prices.  Test / measurement infrastructure.

    python3 tools/ref_profile_report.py <samples file> [--seconds CPU_S] [--json out.json] [--top 40]

Every line of the samples file is "<module> <offset> <count> <nearest exported symbol>".  Offsets are bucketed with the module's own
symbol table (`nm -n --defined-only`: static functions included; stripped system libraries fall back to the exported name the
sampler wrote, hot spots of libc without one are named by the instruction they sit on).  ITIMER_PROF counts user + system time, so
page faults count where they are taken; the timer ticks with the kernel's HZ (4 ms here), so --seconds (the CPU time of the span, from
the run's own RTM_MAIN_GRAPH line) sets what a sample weighs.
"""
import bisect
import collections
import json
import os
import subprocess
import sys

# function -> family.  Names are the reference's (fhe-cmplr/rtlib/ant/{poly,util}/src); what a name does was read off its source.
FAMILIES = [
    ("ntt_fwd", ("Forward_transform", "Ftt_fwd", "Ntt_")),
    ("ntt_inv", ("Inverse_transform", "Ftt_inv", "Intt_")),
    ("mul", ("Multiply_add", "Multiply_ntt", "Hw_modmul", "Multiply_poly", "Mul_poly", "Scalars_integer_multiply", "Scalar_integer_multiply", "Mul_int64", "Fast_mul", "Mod_mul")),
    ("add", ("Add_poly", "Sub_poly", "Hw_modadd", "Hw_modsub", "Add_int64", "Mod_add")),
    ("permute", ("Automorphism", "Hw_rotate", "Rotate_poly")),
    ("conversion", ("Decompose_modup", "Fast_base_conv", "Rescale_poly", "Reduce_rns_base", "Raise_rns_base", "Base_conv", "Decompose", "Switch_modulus", "Approx_switch",
                    "Transform_values_from_level0")),
    ("encode", ("Embedding", "Encode", "Fft_", "Transform_values_to_rns", "Reverse_bits", "Bit_reverse", "__muldc3", "Rotation_group", "__log2", "lround", "sincos", "cexp")),
    ("memset", ("memset", "brk", "munmap", "mmap", "madvise", "calloc", "malloc", "free", "_int_", "sysmalloc")),
    ("memcpy", ("memcpy", "memmove", "Copy_poly")),
]


def family_of(name):
    bare = name.lstrip("_")
    for fam, keys in FAMILIES:
        for k in keys:
            if bare.startswith(k.lstrip("_")) or k in name:
                return fam
    return "other"


def libc_spot(module, off):
    """A PC inside a stripped libc: name it by the instruction it sits on (the string instructions of the memset / memcpy bodies)."""
    try:
        out = subprocess.run(["objdump", "-d", module, "--start-address=%#x" % off, "--stop-address=%#x" % (off + 8)], capture_output=True, text=True).stdout
    except OSError:
        return None
    if "rep stos" in out:
        return "memset (rep stos)"
    if "rep movs" in out:
        return "memcpy (rep movs)"
    return None


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
    total_hint = 50  # (only spots with more samples than this are disassembled)
    seconds = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else None
    for ln in open(path):
        if ln.startswith("#"):
            h = ln.split()
            if seconds is None and "cpu_s" in h:
                seconds = float(h[h.index("cpu_s") + 1])
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
        if name == "?" and os.path.basename(mod).startswith("libc") and cnt > total_hint:
            name = libc_spot(os.path.realpath(mod), off) or "?"
        by_func[(os.path.basename(mod), name)] += cnt
    fam = collections.Counter()
    for (mod, name), c in by_func.items():
        fam[family_of(name)] += c
    # ITIMER_PROF ticks with the kernel's timer (CONFIG_HZ), not with the requested period: --seconds gives the CPU time of the sampled span
    # (the run's own RTM_MAIN_GRAPH line) and every sample weighs seconds / samples
    w = (seconds / total) if seconds else 1e-3
    print("%d samples = %.1f s of CPU (%.3f ms per sample)" % (total, total * w, w * 1e3))
    print("\nby family:")
    for f, c in fam.most_common():
        print("  %-18s %9.1f s  %5.1f %%" % (f, c * w, 100.0 * c / total))
    print("\nby function:")
    for (mod, name), c in by_func.most_common(top):
        print("  %-22s %-44s %9.1f s  %5.1f %%  [%s]" % (mod[:22], name[:44], c * w, 100.0 * c / total, family_of(name)))
    if "--json" in sys.argv:
        out = sys.argv[sys.argv.index("--json") + 1]
        json.dump({"samples": total, "seconds": round(total * w, 1), "ms_per_sample": round(w * 1e3, 4),
                   "by_family_s": {f: round(c * w, 1) for f, c in fam.most_common()},
                   "by_function_s": [{"module": m, "function": n, "s": round(c * w, 1), "family": family_of(n)} for (m, n), c in by_func.most_common(40)]},
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
