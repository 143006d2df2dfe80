#!/usr/bin/env python3
"""Write a synthetic `!ANTFHE` DE_MSG_F32 weight file (the container the ACE compiler emits next to a
generated model: header page, 2^5-byte aligned entries, lookup table at the end;
reference include/fhe/core/rt_data_def.h:16-29,90-109, rtlib/common/src/rt_data_file.c:26-126).

Entry sizes come from a (index, len) list: the Pt_from_msg call trace of the model
(tests/golden/resnet20_pt_entries.txt).  Values ~ N(0, 0.05), numpy default_rng(seed): the same file is
regenerated bit-identically wherever the same numpy runs, so it never needs to be shipped.
"""
import argparse
import struct

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--entries", required=True, help="text file: 'index len' per line")
ap.add_argument("--out", required=True)
ap.add_argument("--seed", type=int, default=2)
ap.add_argument("--sigma", type=float, default=0.05)
a = ap.parse_args()

sizes = {}
for line in open(a.entries):
    if line.strip():
        i, n = map(int, line.split())
        sizes[i] = max(sizes.get(i, 0), n)
count = max(sizes) + 1
rng = np.random.default_rng(a.seed)
PAGE, ALIGN = 4096, 32
ofs = PAGE
lut = []
with open(a.out, "wb") as f:
    f.write(b"\0" * PAGE)
    for i in range(count):
        n = sizes.get(i, 0)
        data = (rng.standard_normal(n) * a.sigma).astype(np.float32).tobytes()
        f.seek(ofs)
        f.write(data)
        lut.append((b"w%d" % i, i, len(data), ofs))
        ofs = (ofs + len(data) + ALIGN - 1) // ALIGN * ALIGN
    lut_ofs = ofs
    f.seek(lut_ofs)
    for name, i, sz, o in lut:  # struct DATA_LUT_ENTRY {char _name[16]; u32 _index; u32 _size; u64 _ent_ofst;}
        f.write(struct.pack("<16sIIQ", name, i, sz, o))
    # struct DATA_FILE_HDR {char magic[8]; u32 rt_ver; u16 flag; u8 ent_type; u8 ent_align; u64 ent_count;
    #                       u64 lut_ofst; struct timespec ctime; char model[48]; char uuid[40];}
    f.seek(0)
    f.write(struct.pack("<8sIHBBQQqq48s40s", b"!ANTFHE\0", 0, 0, 0, 5, count, lut_ofs, 0, 0, b"synthetic", b"synthetic"))
print("wrote %s: %d entries, %d bytes" % (a.out, count, lut_ofs + 32 * count))
