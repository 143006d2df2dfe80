#!/usr/bin/env python3
"""Analytic model of limb-sharded execution over G = 2 / 4 / 8 MI355X (BASELINE configs[4]) -- the expected curve, written down before
any hardware run exists (no multi-GPU node was available to the builder or the driver in rounds 1-6).  Every input is a measured
single-GPU figure or a published link rate; the arithmetic is below so that it can be checked line by line.

  python3 tools/shard_model.py            -> the table of DESIGN section 6 (also kept as profiles/r06_shard_model.txt)

Inputs
  * key-switches per level of one ResNet-20 image: profiles/r05s_kmac_pairs_and_levels.txt (Mod_down pairs = key-switches executed)
  * one image on one GPU: 0.39 s of kernels with 12 images per launch (one stream); 1.08 s as a single image (B = 1:
    Main_graph + decrypt of one image alone, profiles/r05b_ih12_logits.txt -- its small launches are latency-bound); 85 % of either
    belongs to the key-switch / rescale pipelines and is spread over the key-switches in proportion to their limb-transforms.
    OPTIMISTIC for B = 1: compute is divided by G, although a pass of a few limb rows does not get shorter than about 8 us
  * limb = N * 8 = 512 KiB; parameters of the generated ResNets: L = 34, K = 11, alpha = 12
  * xGMI: 7 links per GPU, 153.6 GB/s each both directions together = 76.8 GB/s per direction and peer (one link per peer on an 8-GPU
    node; RCCL all-gather on the fully connected mesh sends every peer its slice directly); on 2 / 4 GPUs of the node the same one link per peer
  * latency of one RCCL collective in a stream (launch + protocol, small message): 15 us assumed (unmeasured here)
  * exchanges per key-switch: 3 collectives (round 6: one packed all-gather per exchange step): the l coefficient-domain source limbs of
    ModUp, the K P-limbs of each accumulator at ModDown (c0's travel under c1's inverse transform)
"""
import math
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, K, ALPHA, N = 34, 11, 12, 65536
LIMB = N * 8
LINK = 76.8e9          # bytes / s per direction and peer
LAT = 15e-6            # s per collective
T_IMAGE = {1: 1.08, 12: 0.39}  # s per image on one GPU: alone / as one of 12 images per launch
KS_SHARE = 0.85


def levels():
    out = {}
    for ln in open(os.path.join(ROOT, "profiles", "r05s_kmac_pairs_and_levels.txt")):
        m = re.search(r"level\s+(\d+): ModUp (\d+) \(\+(\d+) reused\) Mod_down pairs (\d+)", ln)
        if m and int(m.group(4)):
            out[int(m.group(1))] = (int(m.group(4)), int(m.group(2)))  # key-switches, of which with a ModUp of their own
    return out


def lt(l, with_modup=True):
    beta = math.ceil(l / ALPHA)
    up = l + (beta * (l + K) - l) if with_modup else 0  # inverse of the sources + forward of every digit's complement
    return up + 2 * K + 2 * l                            # + ModDown: inverse of both accumulators' P-limbs, forward of their conversions


def main():
    lv = levels()
    total_lt = sum(n_up * lt(l) + (n - n_up) * lt(l, False) for l, (n, n_up) in lv.items())
    taus = {B: T_IMAGE[B] * KS_SHARE / total_lt for B in T_IMAGE}  # s per limb-transform-equivalent of pipeline work, per image
    print("# %d key-switches per image at levels %d..%d, %.0f limb-transforms; %.2f / %.2f us of pipeline time per limb-transform and image (B = 1 / 12)" %
          (sum(n for n, _ in lv.values()), min(lv), max(lv), total_lt, taus[1] * 1e6, taus[12] * 1e6))
    print("# per key-switch at level l, B = images per launch on every rank; times in us")
    print("# %-5s %-3s %-9s | %s" % ("level", "B", "1 GPU", " | ".join("G=%d: compute, exchange (bytes/link MB), no overlap, full overlap" % g for g in (2, 4, 8))))
    for l in (34, 21, 10, 3):
        for B in (1, 12):
            tau = taus[B]
            t1 = lt(l) * tau * B
            cells = []
            for G in (2, 4, 8):
                comp = t1 * math.ceil((l + K) / G) / (l + K)
                b_up = math.ceil(l / G) * LIMB * B
                b_dn = 2 * math.ceil(K / G) * LIMB * B
                x = (b_up + b_dn) / LINK + 3 * LAT
                cells.append("%6.0f, %6.0f (%5.1f), %6.0f, %6.0f" % (comp * 1e6, x * 1e6, (b_up + b_dn) / 1e6, (comp + x) * 1e6, max(comp, x) * 1e6))
            print("  %-5d %-3d %-9.0f | %s" % (l, B, t1 * 1e6, " | ".join(cells)))
    print("# whole image (every key-switch of profiles/r05s_* at its level), seconds per image and speed-up over one GPU")
    for B in (1, 12):
        row = []
        tau = taus[B]
        rest = T_IMAGE[B] * (1 - KS_SHARE)  # per-limb arithmetic outside the pipelines: divides by G (no exchange)
        for G in (1, 2, 4, 8):
            tot_no, tot_full = rest / G, rest / G
            for l, (n, n_up) in lv.items():
                for cnt, with_up in ((n_up, True), (n - n_up, False)):
                    comp = lt(l, with_up) * tau * math.ceil((l + K) / G) / (l + K)
                    if G == 1:
                        x = 0.0
                    else:
                        b = ((math.ceil(l / G) if with_up else 0) + 2 * math.ceil(K / G)) * LIMB * B
                        x = b / LINK / B + (3 if with_up else 2) * LAT / B  # per image of the batch
                    tot_no += cnt * (comp + x)
                    tot_full += cnt * max(comp, x)
            row.append((G, tot_no, tot_full))
        base = row[0][1]
        print("  B = %-2d " % B + "   ".join("G=%d: %.3f s (x%.2f) .. %.3f s (x%.2f)" % (g, a, base / a, b, base / b) for g, a, b in row))
    print("# reading: a level-3 key-switch -- 61 % of an image's key-switches -- has 3 + (14 - 3) + 22 + 6 = 42 limb-transforms of work to divide, but still")
    print("# moves ceil(3/G) + 2 ceil(11/G) limbs per peer and pays three collectives; at B = 1 the 15 us per collective decide, at B = 12 the link rate.")
    print("# Expected at 8 GPUs: x3.3-5.3 for one image IF its compute divided by G (it will not: launch-bound passes), x2.7-4.1 for batches of 12 -- against x8 for replicas (bench.py --gpus N, no data-path collective).")
    print("# Limb sharding buys latency and memory (keys / G) for the largest parameter sets; throughput is replicas.")


if __name__ == "__main__":
    main()
