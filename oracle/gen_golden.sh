#!/bin/bash
# Regenerates tests/golden/ref_*.json by RUNNING THE REFERENCE (oracle/_ref/ref_dump).
# Dev-container only (needs /root/reference); the JSON files are committed.
set -e
cd "$(dirname "$0")"
make ref >/dev/null
G=../tests/golden
D=./_ref/ref_dump
mkdir -p $G
# name            N     L  q0 sf dnum
SETS=(
 "n8_l4_d2        8     4  60 56 2"
 "n16_l3_d2       16    3  60 50 2"
 "n16_l10_d3      16    10 60 59 3"
 "n32_l5_d0       32    5  33 30 0"
 "n64_l7_d3       64    7  60 51 3"
 "c1_n16384_l4    16384 4  60 50 2"
 "bl_n65536_l25   65536 25 60 56 4"
 "rn_n65536_l34   65536 34 51 50 3"
 "r110_n65536_l34 65536 34 51 48 3"
)
for s in "${SETS[@]}"; do
  set -- $s
  $D params $2 $3 $4 $5 $6 > $G/ref_params_$1.json
done
# ops: name N L q0 sf dnum level seed
OPS=(
 "n8_l4_d2_lv4       8     4  60 56 2 4  1"
 "n8_l4_d2_lv3       8     4  60 56 2 3  2"
 "n16_l3_d2_lv3      16    3  60 50 2 3  3"
 "n16_l10_d3_lv10    16    10 60 59 3 10 4"
 "n16_l10_d3_lv5     16    10 60 59 3 5  5"
 "n32_l5_d0_lv5      32    5  33 30 0 5  6"
 "n64_l7_d3_lv7      64    7  60 51 3 7  7"
 "n64_l7_d3_lv4      64    7  60 51 3 4  8"
 "n1024_l7_d3_lv6    1024  7  60 51 3 6  9"
 "c1_n16384_l4_lv4   16384 4  60 50 2 4  10"
 "bl_n65536_l25_lv25 65536 25 60 56 4 25 11"
 "bl_n65536_l25_lv17 65536 25 60 56 4 17 12"
 "rn_n65536_l34_lv34 65536 34 51 50 3 34 13"
 "r110_n65536_l34_lv21 65536 34 51 48 3 21 14"
)
for s in "${OPS[@]}"; do
  set -- $s
  $D ops $2 $3 $4 $5 $6 $7 $8 > $G/ref_ops_$1.json
done
# encode: name N L q0 sf dnum level seed
ENC=(
 "n16_l3_lv3        16    3  60 50 2 3  1"
 "n64_l7_lv4        64    7  60 51 3 4  2"
 "n1024_l7_lv6      1024  7  60 51 3 6  3"
 "bl_n65536_l25_lv20 65536 25 60 56 4 20 4"
 "rn_n65536_l34_lv12 65536 34 51 50 3 12 5"
)
for s in "${ENC[@]}"; do
  set -- $s
  $D encode $2 $3 $4 $5 $6 $7 $8 > $G/ref_encode_$1.json
done
ls -la $G
# pre-encoded plaintext data file (DE_PLAINTEXT; entries made by the reference's Encode_plain_buffer): Pt_get fixture
$D ptfile 64 5 60 50 2 3 $G/ref_ptfile_n64_l5_lv3.bin 3 1 77
