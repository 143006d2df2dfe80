"""Build the native pieces in-tree (ace-compiler_amd/lib/, git-ignored; they travel to the GPU box with the gpurun snapshot):

  libacehip.so          HIP kernels + C ABI (include/acehip.h), hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  libFHErt_ant.so       the rt_ant drop-in shim (include/rt_ant, include/common), g++, links libacehip.so
  libFHErt_common.so    named by the reference link line too; everything lives in libFHErt_ant here
  libFHErt_ant.a, libFHErt_common.a
                        the ARCHIVE names of the reference link line (scripts/perf.py:202-207:
                        `cc model.c -I... rtlib/lib/libFHErt_ant.a rtlib/lib/libFHErt_common.a -lgmp -lm`).
                        libFHErt_ant_objs.a is the real static archive (every object of the shim and of the HIP library,
                        device code included); libFHErt_ant.a is a GNU ld script that pulls it in together with what a
                        plain `cc` link line does not name (the HIP runtime, libstdc++), so that the reference's line links
                        unchanged.  libFHErt_common.a is a real (one-object) archive.

Objects are compiled one per source into lib/obj/ and reused by the shared libraries and the archives.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libacehip.so")
SOURCES = ["kernels.hip", "ntt_fast.hip", "keyswitch.hip", "rt_kernels.hip", "embed.hip", "hw_batch.hip", "shard.hip", "api_core.cpp", "api_hw_batch.cpp", "api_ops.cpp", "api_shard.cpp", "host_params.cpp"]
HEADERS = ["kernels.hpp", "api_internal.hpp", "device_arith.hpp", "host_params.hpp", "rou_table.inc", os.path.join("..", "..", "include", "acehip.h")]
ROCM_LIB = "/opt/rocm/lib"
# keyswitch.hip: the matrix-core base conversion reads its MFMA results with VALU instructions right away; with the results in
# VGPRs (instead of the accumulator half of the register file) that needs no v_accvgpr_read per value
PER_FILE_FLAGS = {"keyswitch.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def _hip_objects(force, verbose):
    """one object per source (device code for gfx950 embedded); returns (objects, anything_rebuilt)"""
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    # -fgpu-default-stream=per-thread: the NULL stream of every entry point is the calling thread's own stream, so
    # host threads that each own a context (one image stream each) run concurrently on the GPU
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-fgpu-default-stream=per-thread", "-Wall",
             "-Wno-unused-function"] + os.environ.get("ACEHIP_EXTRA_HIPCC_FLAGS", "").split()  # extra flags: experiments only
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(OBJDIR, src.replace(".", "_") + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc()] + flags + PER_FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda c: _run(c, verbose), jobs))
    return objs, bool(jobs)


def needs_build():
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    return _newer(LIB, [os.path.join(CSRC, s) for s in SOURCES] + hdrs)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs, _ = _hip_objects(force, verbose)
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))


# ---- rt_ant drop-in shim (host C++ over the C ABI of libacehip.so) ----
RT_DIR = os.path.join(CSRC, "rt")
RT_LIB = os.path.join(LIBDIR, "libFHErt_ant.so")
RT_COMMON_LIB = os.path.join(LIBDIR, "libFHErt_common.so")
RT_ARCHIVE = os.path.join(LIBDIR, "libFHErt_ant.a")
RT_OBJS_ARCHIVE = os.path.join(LIBDIR, "libFHErt_ant_objs.a")
RT_COMMON_ARCHIVE = os.path.join(LIBDIR, "libFHErt_common.a")
RT_SOURCES = ["rt_poly.cpp", "rt_context.cpp", "rt_encode.cpp", "rt_io.cpp", "rt_eval.cpp", "rt_bootstrap.cpp", "rt_serial.cpp",
              "rt_timing.cpp", "rt_valid.cpp"]
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def build_rt(force=False, verbose=False):
    """libFHErt_ant.so / .a, libFHErt_common.so / .a: same link names as the reference provider library (scripts/perf.py:202-207)."""
    build(force=force, verbose=verbose)
    hip_objs, _ = _hip_objects(False, verbose)
    cxx, cc = shutil.which("g++") or "g++", shutil.which("gcc") or "gcc"
    hdrs = [os.path.join(RT_DIR, "rt_internal.hpp"), os.path.join(RT_DIR, "rt_ev.hpp"), os.path.join(RT_DIR, "bts_coeffs.inc"),
            os.path.join(INCLUDE, "acehip.h")] + [os.path.join(INCLUDE, d, f) for d in ("rt_ant", "common")
                                                   for f in os.listdir(os.path.join(INCLUDE, d))]
    jobs, rt_objs = [], []
    for src in RT_SOURCES:
        s, o = os.path.join(RT_DIR, src), os.path.join(OBJDIR, "rt_" + src.replace(".", "_") + ".o")
        rt_objs.append(o)
        if force or _newer(o, [s] + hdrs):
            # -ffp-contract=off: the FP64 canonical embedding must round like the reference's (no FMA fusion)
            jobs.append([cxx, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda c: _run(c, verbose), jobs))
    if force or _newer(RT_LIB, rt_objs + [LIB]):
        _run([cxx, "-shared", "-fPIC", "-o", RT_LIB] + rt_objs + ["-L", LIBDIR, "-lacehip", "-Wl,-rpath,$ORIGIN"], verbose)
    # libFHErt_common: the reference link line names it too; everything lives in libFHErt_ant here
    stub_c, stub_o = os.path.join(OBJDIR, "_common_stub.c"), os.path.join(OBJDIR, "_common_stub.o")
    if force or not os.path.exists(RT_COMMON_LIB) or not os.path.exists(RT_COMMON_ARCHIVE):
        with open(stub_c, "w") as f:
            f.write("const char* FHErt_common_provider(void) { return \"acehip\"; }\n")
        _run([cc, "-O2", "-fPIC", "-c", stub_c, "-o", stub_o], verbose)
        _run([cc, "-shared", "-fPIC", "-o", RT_COMMON_LIB, stub_o], verbose)
        if os.path.exists(RT_COMMON_ARCHIVE):
            os.remove(RT_COMMON_ARCHIVE)
        _run(["ar", "rcs", RT_COMMON_ARCHIVE, stub_o], verbose)
        os.remove(stub_c)
    # static archive of every object + the ld script under the reference's archive name
    if force or _newer(RT_OBJS_ARCHIVE, rt_objs + hip_objs) or not os.path.exists(RT_ARCHIVE):
        if os.path.exists(RT_OBJS_ARCHIVE):
            os.remove(RT_OBJS_ARCHIVE)
        _run(["ar", "rcs", RT_OBJS_ARCHIVE] + rt_objs + hip_objs, verbose)
        with open(RT_ARCHIVE, "w") as f:
            f.write("/* GNU ld script standing in for the archive name of the reference link line (scripts/perf.py:202-207):\n"
                    "   the objects are in libFHErt_ant_objs.a; a HIP program also needs the HIP runtime and the C++ runtime, which a\n"
                    "   plain `cc ... libFHErt_ant.a libFHErt_common.a -lgmp -lm` does not name.  Written by ace-compiler_amd/build.py. */\n"
                    "SEARCH_DIR ( %s )\nINPUT ( %s )\nINPUT ( -lamdhip64 -lstdc++ -lpthread -ldl -lm )\n" % (ROCM_LIB, RT_OBJS_ARCHIVE))
    return RT_LIB
