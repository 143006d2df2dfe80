#!/usr/bin/env python3
"""Register / spill / LDS metadata of the gfx950 kernels of one csrc/*.hip file, from the compiler's own notes (-save-temps):
  python3 tools/kernel_meta.py keyswitch.hip [name filter] [-D...]   ->  one line per kernel
(the product flags of ace-compiler_amd/build.py; nothing is written into the tree)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
per_file = {"keyswitch.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
with tempfile.TemporaryDirectory() as d:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-fgpu-default-stream=per-thread",
           "-save-temps", "-c", os.path.join(ROOT, "ace-compiler_amd", "csrc", src), "-o", "x.o"] + per_file.get(src, []) + extra
    subprocess.run(cmd, cwd=d, check=True, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
for blk in asm.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if filt and filt not in dem:
        continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))  # noqa: E731
    print("%-60s vgpr %3d (spilled %3d)  sgpr %3d (spilled %3d)  scratch %4d B  lds %6d B  max_wg %d" % (
        dem[-60:], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
        g("group_segment_fixed_size"), g("max_flat_workgroup_size")))
