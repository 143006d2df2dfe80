"""CPU tests (no GPU): the C-ABI library loads and exports every symbol of include/acehip.h, the
host-side table generation of the product (csrc/host_params.cpp) equals the oracle's and the
reference-generated golden tables, and launches on a GPU-less context fail loudly (no CPU fallback)."""
import glob
import json
import os
import re

import numpy as np
import pytest

import ace_compiler_amd as A
import _oracle as O
from conftest import GOLDEN, ROOT


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "acehip.h")).read()
    declared = set(re.findall(r"\b(acehip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"acehip_ctx", "acehip_stream"}
    bound = {s[0] for s in A.binding.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    lib = A.load_library()
    for name in declared:
        assert hasattr(lib, name)


PARAM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_params_*.json")))


@pytest.mark.parametrize("path", PARAM_FILES, ids=[os.path.basename(p)[11:-5] for p in PARAM_FILES])
def test_host_tables_match_reference(path):
    g = json.load(open(path))
    N, L, K = g["N"], g["L"], g["K"]
    rt = A.AceHip(N, L, g["q0_bits"], g["sf_bits"], g["dnum_req"], host_only=True)
    try:
        assert (rt.K, rt.alpha, rt.dnum) == (K, g["alpha"], g["dnum"])
        assert rt.primes == g["primes"]
        assert rt.table(0).tolist() == g["psi"]
        assert rt.table(1).tolist() == g["n_inv"]
        assert rt.table(2).tolist() == g["n_inv_prec"]
        assert rt.table(3).tolist() == g["prec128_lo"]
        assert rt.table(4).tolist() == g["prec128_hi"]
        for gi in range(L + K):
            r = g["rou"][gi]
            rou = rt.table(10, gi)
            assert O.sum64(rou) == r["sum64"] and O.xorw(rou) == r["xorw"]
            assert O.xorw(rt.table(11, gi)) == r["prec_xorw"]
            assert O.xorw(rt.table(12, gi)) == r["inv_xorw"]
            assert O.xorw(rt.table(13, gi)) == r["inv_prec_xorw"]
        assert rt.table(20).tolist() == g["phat_inv_modp"]
        assert rt.table(21).tolist() == g["phat_inv_modp_prec"]
        assert rt.table(22).tolist() == g["phat_modq"]
        assert rt.table(23).tolist() == g["pinv_modq"]
        for what, key in ((30, "ql_inv"), (31, "ql_inv_prec"), (32, "qlql"), (33, "qlql_prec")):
            t = rt.table(what).reshape(L, L)
            for k, row in enumerate(g["rescale"]):
                assert t[k, : k + 1].tolist() == row[key]
        for e in g["modup"]:
            n2, hat_inv, compl, hat_mod = rt.modup_tables(e["level"], e["digit"])
            assert n2 == e["n2"] and hat_inv.tolist() == e["hat_inv"]
            assert [rt.primes[i] for i in compl] == e["compl"]
            if "hat_mod" in e:
                assert hat_mod.reshape(-1).tolist() == e["hat_mod"]
            else:
                assert O.sum64(hat_mod) == e["hat_mod_sum64"] and O.xorw(hat_mod) == e["hat_mod_xorw"]
    finally:
        rt.close()


def test_automorphism_tables_match_oracle():
    o = O.Oracle(64, 3, 60, 50, 2)
    rt = A.AceHip(64, 3, 60, 50, 2, host_only=True)
    for rot in (1, -1, 5, 16, -7):
        k = rt.auto_index(rot)
        assert k == o.lib.orc_find_automorphism_index(rot, 64)
        assert rt.auto_order_host(k).tolist() == o.automorphism(k).tolist()
    assert rt.auto_order_host(127).tolist() == o.automorphism(127).tolist()  # conjugation
    rt.close()
    o.close()


def test_no_cpu_fallback():
    rt = A.AceHip(16, 3, 60, 50, 2, host_only=True)
    x = np.zeros(16, dtype=np.uint64)
    rc = rt.lib.acehip_ntt_forward(rt.h, x.ctypes.data, 3, 0, 1, None)
    assert rc == -3 and "no CPU fallback" in rt.err()
    rc = rt.lib.acehip_key_switch(rt.h, None, None, None, None, 3, None)
    assert rc == -3
    rt.close()
    if A.load_library().acehip_device_count() == 0:
        with pytest.raises(A.AceHipError):
            A.AceHip(16, 3, 60, 50, 2)


def test_digit_size_limit():
    """Base conversion sums alpha (ModUp) or K (ModDown) products of 122 bits exactly in 128 bits: up to 64 source limbs per
    digit are supported (the kernels take them through the registers in chunks of 16); more is refused when the context is
    created instead of dropping limbs silently (keyswitch.hip kMaxIn)."""
    import ace_compiler_amd as A

    rt = A.AceHip(64, 40, 60, 50, 2, host_only=True)   # alpha = 20, K = 17: fine
    assert (rt.alpha, rt.K) == (20, 17)
    rt.close()
    with pytest.raises(Exception, match="64 limbs"):
        A.AceHip(16, 70, 40, 30, 1, host_only=True)      # alpha = 70


def test_reference_link_line_with_archives(tmp_path):
    """scripts/perf.py:202-207 links a generated program as `cc model.c -I<inc> -I<inc>/rt_ant <lib>/libFHErt_ant.a
    <lib>/libFHErt_common.a -lgmp -lm`: the archive names must exist and that line must link unchanged (no GPU needed to
    link; -lgmp is left out because the image has no libgmp.so development symlink and nothing here needs GMP)."""
    import subprocess
    import sys

    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    for f in (bmod.RT_ARCHIVE, bmod.RT_OBJS_ARCHIVE, bmod.RT_COMMON_ARCHIVE):
        assert os.path.exists(f), f
    members = subprocess.run(["ar", "t", bmod.RT_OBJS_ARCHIVE], capture_output=True, text=True, check=True).stdout.split()
    assert "rt_rt_poly_cpp.o" in members and "ntt_fast_hip.o" in members and "api_ops_cpp.o" in members
    inc = os.path.join(ROOT, "include")
    exe = str(tmp_path / "dropin_c1_static")
    subprocess.check_call(["cc", os.path.join(ROOT, "tests", "c", "dropin_c1.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
                           bmod.RT_ARCHIVE, bmod.RT_COMMON_ARCHIVE, "-lm", "-o", exe])
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True, check=True).stdout
    assert "libFHErt_ant.so" not in needed and "libacehip.so" not in needed   # the runtime is inside the executable
    assert "libamdhip64" in needed


@pytest.mark.parametrize("params", [(1024, 24, 60, 50, 2), (2048, 9, 60, 50, 3), (65536, 34, 51, 50, 3)], ids=lambda p: "n%d_l%d_d%d" % (p[0], p[1], p[4]))
def test_matrix_core_conversion_tables_reproduce_the_exact_sums(params):
    """The base conversion on the matrix cores (keyswitch.hip base_conv_mfma_kernel) multiplies the BYTES of the source residues
    (sign bit flipped) with 7-bit digits of hat(i, j) * 2^(8a) mod t_j and starts its accumulators at an offset.  Everything the
    kernel needs comes from acehip_conv_mfma_tables; this test replays the kernel's arithmetic with Python integers on random
    residues and checks, for ModUp digits and ModDown at several levels, that (1) the constants are the reference's --
    hat(i, j) = prod of the other primes of the source basis mod t_j (crt.c:426-533) --, (2) every accumulator ends non-negative
    and below 2^23, and (3) the recombined 80-bit value is congruent to sum_i y_i * hat(i, j) mod t_j, the sum Reduce_rns_base
    (polynomial.c:928-967) reduces.  No GPU: host tables only."""
    import ctypes as C
    import random

    N, L, q0, sf, dnum = params
    rt = A.AceHip(N, L, q0, sf, dnum, host_only=True)
    rng = random.Random(5)
    try:
        primes, K, alpha = list(rt.primes), rt.K, rt.alpha
        for level in sorted({L, max(1, L // 2), 1}):
            nd = rt.lib.acehip_num_decomp(rt.h, level)
            assert nd == -(-level // alpha)
            for digit in list(range(nd)) + [-1]:
                dims = (C.c_uint32 * 4)()
                need = rt.lib.acehip_conv_mfma_tables(rt.h, level, digit, None, 0, None, 0, dims)
                assert need > 0, rt.err()
                n_in, n_out, steps, tiles = list(dims)
                frag = (C.c_uint8 * need)()
                off = (C.c_uint32 * (tiles * 16 * 9))()
                assert rt.lib.acehip_conv_mfma_tables(rt.h, level, digit, frag, need, off, len(off), dims) == need
                assert need == tiles * steps * 9 * 64 * 16
                # source / target primes of the problem
                if digit >= 0:
                    src = list(range(digit * alpha, min((digit + 1) * alpha, level)))
                    tgt = [g for g in list(range(level)) + list(range(L, L + K)) if g not in src]
                else:
                    src = list(range(L, L + K))
                    tgt = list(range(level))
                assert (n_in, n_out) == (len(src), len(tgt)), (digit, level, n_in, n_out, src, tgt)

                def dig(tile, step, b, lane, e):
                    return frag[((((tile * steps + step) * 9 + b) * 64 + lane) * 16) + e]

                y = [rng.randrange(primes[g]) for g in src]
                y[0] = primes[src[0]] - 1  # an extreme residue
                for j, gt in enumerate(tgt):
                    t = primes[gt]
                    tile, r = divmod(j, 16)
                    total = 0
                    acc = [off[j * 9 + b] for b in range(9)]
                    for i, gs in enumerate(src):
                        step, g, half = i // 8, (i % 8) // 2, i % 2
                        lane = g * 16 + r
                        # (1) the constant the digits of byte 0 spell is the reference's hat(i, j)
                        hat = sum(dig(tile, step, b, lane, half * 8) << (7 * b) for b in range(9))
                        want = 1
                        for k in src:
                            if k != gs:
                                want = want * primes[k] % t
                        assert hat == want, (level, digit, i, j)
                        for a8 in range(8):
                            u = (y[i] >> (8 * a8)) & 0xFF
                            for b in range(9):
                                d7 = dig(tile, step, b, lane, half * 8 + a8)
                                assert d7 < 128
                                acc[b] += (u - 128) * d7
                        total += y[i] * hat
                    assert all(0 <= v < (1 << 23) for v in acc), acc          # (2)
                    v = sum(c << (7 * b) for b, c in enumerate(acc))
                    assert v < (1 << 80) and v % t == total % t, (level, digit, j)  # (3)
                # columns past n_out carry zero constants and zero offsets
                for j in range(n_out, tiles * 16):
                    assert all(off[j * 9 + b] == 0 for b in range(9))
    finally:
        rt.close()


def test_library_fingerprint_covers_the_compile_flags(monkeypatch):
    """The fingerprint both libraries embed is the staleness gate of needs_build / load_library / Prepare_context.  It covers the sources AND
    the effective compile flags: a library an experiment script built with ACEHIP_EXTRA_HIPCC_FLAGS (kernels that give wrong results:
    -DNTT_EXP, -DACEHIP_ABLATION ...) must not pass for a clean build of the same sources in a process that does not carry those flags."""
    import sys

    bmod = sys.modules["ace_compiler_amd.build"]
    monkeypatch.delenv("ACEHIP_EXTRA_HIPCC_FLAGS", raising=False)
    clean = bmod.source_fingerprint()
    assert bmod.embedded_fingerprint(bmod.LIB) == clean and not bmod.needs_build()
    assert A.load_library().acehip_source_fingerprint().decode() == clean
    monkeypatch.setenv("ACEHIP_EXTRA_HIPCC_FLAGS", "-DNTT_EXP=1")
    assert bmod.source_fingerprint() != clean
    assert bmod.needs_build()  # the clean library in the tree is stale for a process that asks for experiment flags, and vice versa
