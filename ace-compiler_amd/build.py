"""Build libacehip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so lands in ace-compiler_amd/lib/ (git-ignored, but it
travels to the GPU box with the gpurun snapshot).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libacehip.so")
SOURCES = ["kernels.hip", "ntt_fast.hip", "keyswitch.hip", "rt_kernels.hip", "embed.hip", "hw_batch.hip", "api.cpp", "host_params.cpp"]
HEADERS = ["kernels.hpp", "device_arith.hpp", "host_params.hpp", "rou_table.inc", os.path.join("..", "..", "include", "acehip.h")]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # -fgpu-default-stream=per-thread: the NULL stream of every entry point is the calling thread's own stream, so
    # host threads that each own a context (one image stream each) run concurrently on the GPU
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-fgpu-default-stream=per-thread", "-Wall", "-Wno-unused-function", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += os.environ.get("ACEHIP_EXTRA_HIPCC_FLAGS", "").split()  # experiments only
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))


# ---- rt_ant drop-in shim (host C++ over the C ABI of libacehip.so) ----
RT_DIR = os.path.join(CSRC, "rt")
RT_LIB = os.path.join(LIBDIR, "libFHErt_ant.so")
RT_COMMON_LIB = os.path.join(LIBDIR, "libFHErt_common.so")
RT_SOURCES = ["rt_poly.cpp", "rt_context.cpp", "rt_encode.cpp", "rt_io.cpp", "rt_eval.cpp", "rt_bootstrap.cpp", "rt_serial.cpp"]
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def build_rt(force=False, verbose=False):
    """libFHErt_ant.so: same link name as the reference provider library (scripts/perf.py:202-207)."""
    build(force=force, verbose=verbose)
    srcs = [os.path.join(RT_DIR, s) for s in RT_SOURCES]
    deps = srcs + [os.path.join(RT_DIR, "rt_internal.hpp"), os.path.join(RT_DIR, "rt_ev.hpp"), os.path.join(RT_DIR, "bts_coeffs.inc"), os.path.join(INCLUDE, "rt_ant", "ant_api.h"), LIB]
    if not force and os.path.exists(RT_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(RT_LIB) for d in deps):
        return RT_LIB
    cxx = shutil.which("g++") or "g++"
    # -ffp-contract=off: the FP64 canonical embedding must round like the reference's (no FMA fusion)
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
           "-I", INCLUDE, "-o", RT_LIB] + srcs + ["-L", LIBDIR, "-lacehip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # libFHErt_common: the reference link line names it too; everything lives in libFHErt_ant here
    stub = os.path.join(LIBDIR, "_common_stub.c")
    with open(stub, "w") as f:
        f.write("const char* FHErt_common_provider(void) { return \"acehip\"; }\n")
    subprocess.check_call([shutil.which("gcc") or "gcc", "-O2", "-fPIC", "-shared", "-o", RT_COMMON_LIB, stub])
    os.remove(stub)
    return RT_LIB
