#!/bin/bash
# Regenerates the committed ciphertext-level parity fixtures (tests/golden/ct_parity_*.tar.gz) by RUNNING THE REFERENCE:
# oracle/_ref/ct_parity_ref is tests/c/ct_parity.c built with -DREF_BUILD against oracle/_ref/libref_rtlib.so
# (make -C oracle ref; dev container only).  Keys and encryption noise are random, so every regeneration gives new,
# equally valid fixtures.  Data only: key set, input ciphertexts, expected outputs.
set -e
cd "$(dirname "$0")/../.."
make -s -C oracle ref
gen() {  # name args...
  name=$1; shift
  d=$(mktemp -d)
  oracle/_ref/ct_parity_ref dump "$d" "$@" > "$d/dump.log"
  tar -C "$d" -czf "tests/golden/ct_parity_$name.tar.gz" .
  rm -rf "$d"
}
gen n16_full   16 25 60 51 3 0 8 3 1 -2 5
gen n16_sparse 16 25 60 51 2 0 4 4 1 3
# the generated models' prime sizes (mul_depth 33, q0 51; ResNet-20: Delta = 2^50, ResNet-110: Delta = 2^48; dnum 3), bootstrap from
# 2 limbs back to level_after 15 / 17, fully and sparsely packed: pins the bootstrap's level budget at L = 34 from a clean checkout
gen n64_r20_full    64 33 51 50 3 16 32 15 1 -3 16
gen n64_r20_sparse  64 33 51 50 3 16 8 15 2 -1 4
gen n64_r110_full   64 33 51 48 3 16 32 17 1 -3 16
gen n64_r110_sparse 64 33 51 48 3 16 8 17 2 -1 4
