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
