"""Chain shapes of the per-limb launches (ACEHIP_HW_DUMP_EVERY=<n> output on stdin): how often each op sequence occurs as a chain
segment, and how many ops / stored results they carry.  Diagnostic for the planner (api_hw_batch.cpp)."""
import collections
import re
import sys

blocks = sys.stdin.read().split("[hw dump]")[1:]
half = blocks[len(blocks) // 2:]  # steady state: the later launches
shapes, ops_in = collections.Counter(), collections.Counter()
for b in half:
    for line in b.split("\n"):
        if "seg:" not in line:
            continue
        items = re.findall(r"L\d+=(\w+)(?:\([^)]*\))?(?:q\d+)?(~?)", line)
        names = [n if n != "0" else "zero" for n, _ in items]
        # run-length encode repeated (mul add) / (muladd) patterns
        key, i = [], 0
        while i < len(names):
            j = i
            while j < len(names) and names[j] == names[i]:
                j += 1
            key.append(names[i] + ("*%d" % (j - i) if j - i > 1 else ""))
            i = j
        k = " ".join(key)
        k = re.sub(r"(mul add )(?:mul add ?){2,}", lambda m: "(mul add)*%d " % (m.group(0).count("mul add")), k + " ").strip()
        shapes[k] += 1
        ops_in[k] += len(names)
tot = sum(ops_in.values())
print("launches sampled (steady half): %d, chain segments %d, ops %d" % (len(half), sum(shapes.values()), tot))
for k, v in sorted(shapes.items(), key=lambda kv: -ops_in[kv[0]])[:30]:
    print("%6d segs %7d ops (%4.1f %%)  %s" % (v, ops_in[k], 100.0 * ops_in[k] / tot, k[:150]))
