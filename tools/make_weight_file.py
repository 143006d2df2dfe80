#!/usr/bin/env python3
"""Write a synthetic `!ANTFHE` DE_MSG_F32 weight file (the container the ACE compiler emits next to a
generated model: header page, 2^5-byte aligned entries, lookup table at the end;
reference include/fhe/core/rt_data_def.h:16-29,90-109, rtlib/common/src/rt_data_file.c:26-126).

Entry sizes come from a (index, len) list: the Pt_from_msg call trace of the model
(tests/golden/resnet20_pt_entries.txt).

Two value generators:

  --gen ih12 (default)  INTEGER-ONLY and therefore the same bytes under every numpy / libm / CPU: element j of entry e is the sum of
                  twelve 16-bit lanes of three SplitMix64 words (an Irwin-Hall variate: mean 12 * 32767.5, variance 65536^2 to
                  2e-10), centred, divided by 2^16 (exact in float64), multiplied by sigma (one IEEE multiplication) and rounded
                  to float32 (one IEEE rounding).  `values_python` below is the same arithmetic on Python integers -- the
                  CPU test tests/test_weight_file.py compares the two and pins known answers.
  --gen numpy     numpy default_rng(seed).standard_normal * sigma (rounds 1-4; depends on numpy's stream staying the same)
"""
import argparse
import struct

import numpy as np

M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix_py(z):
    """SplitMix64 finaliser on a Python integer"""
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def entry_key(seed, e):
    """the stream of entry e: a function of the seed and the entry's index only (entries can be generated in any order)"""
    return mix_py(mix_py(seed) ^ mix_py(e + 1))


def values_python(seed, e, n, sigma):
    """reference form of the generator: Python integers, one element at a time (tests and documentation; slow)"""
    k = entry_key(seed, e)
    out = []
    for j in range(n):
        s = 0
        for t in range(3):
            w = mix_py(k + (3 * j + t + 1) * GOLDEN)
            s += (w & 0xFFFF) + ((w >> 16) & 0xFFFF) + ((w >> 32) & 0xFFFF) + (w >> 48)
        out.append(np.float32((s - 393210) / 65536.0 * sigma))
    return np.array(out, dtype=np.float32)


def _mix_np(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def values_ih12(seed, e, n, sigma):
    """the same values, vectorised: unsigned 64-bit wrap-around arithmetic, shifts, masks; one float64 multiply, one rounding"""
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    k = np.uint64(entry_key(seed, e))
    with np.errstate(over="ignore"):
        ctr = np.arange(1, 3 * n + 1, dtype=np.uint64) * np.uint64(GOLDEN) + k
        w = _mix_np(ctr)
    lanes = (w & np.uint64(0xFFFF)) + ((w >> np.uint64(16)) & np.uint64(0xFFFF)) + ((w >> np.uint64(32)) & np.uint64(0xFFFF)) + (w >> np.uint64(48))
    s = lanes.reshape(n, 3).sum(axis=1, dtype=np.uint64).astype(np.int64) - np.int64(393210)
    return (s.astype(np.float64) / 65536.0 * np.float64(sigma)).astype(np.float32)


def write_file(entries, out, seed, sigma, gen):
    sizes = {}
    for line in open(entries):
        if line.strip():
            i, n = map(int, line.split())
            sizes[i] = max(sizes.get(i, 0), n)
    count = max(sizes) + 1
    rng = np.random.default_rng(seed) if gen == "numpy" else None
    PAGE, ALIGN = 4096, 32
    ofs = PAGE
    lut = []
    with open(out, "wb") as f:
        f.write(b"\0" * PAGE)
        for i in range(count):
            n = sizes.get(i, 0)
            if gen == "numpy":
                data = (rng.standard_normal(n) * sigma).astype(np.float32).tobytes()
            else:
                data = values_ih12(seed, i, n, sigma).tobytes()
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
    return count, lut_ofs + 32 * count


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", required=True, help="text file: 'index len' per line")
    ap.add_argument("--out", required=True)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--sigma", type=float, default=0.05)
    ap.add_argument("--gen", choices=["ih12", "numpy"], default="ih12")
    a = ap.parse_args()
    cnt, size = write_file(a.entries, a.out, a.seed, a.sigma, a.gen)
    print("wrote %s: %d entries, %d bytes (%s)" % (a.out, cnt, size, a.gen))
