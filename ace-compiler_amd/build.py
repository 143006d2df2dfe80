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
import hashlib
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
# every header / include file of csrc/ (an object is rebuilt when ANY of them changes: cheaper than tracking who includes what)
HEADERS = sorted(n for n in os.listdir(CSRC) if n.endswith((".hpp", ".inc"))) + [os.path.join("..", "..", "include", "acehip.h")]
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


# ---- staleness is decided by CONTENT, not by mtime (the .so files are untracked build outputs that travel with a snapshot of the
# tree: a library older or newer than the sources beside it must never be loaded silently).
#  * every object has a sidecar <obj>.dep = sha256 of its source, the headers it may include and its command line; it is rebuilt
#    when that differs;
#  * both shared libraries embed the fingerprint of ALL sources (tools/csrc_fingerprint.py: csrc/**, include/**) AND of the effective
#    compile flags (the experiment scripts under tools/ build kernels that give wrong results into the in-tree library through
#    ACEHIP_EXTRA_HIPCC_FLAGS: such a library must never pass for a clean build of the same sources) as the string
#    "ACEHIP_SRC_FPR=<16 hex>" and export it (acehip_source_fingerprint / acehip_rt_source_fingerprint); needs_build() reads it out
#    of the file, binding.load_library() compares it with the sources after dlopen, and the shim's Prepare_context aborts when the
#    two libraries disagree.
FPR_TAG = b"ACEHIP_SRC_FPR="


def hip_flags():
    """the flags every HIP object is compiled with.  -fgpu-default-stream=per-thread: the NULL stream of every entry point is the
    calling thread's own stream, so host threads that each own a context (one image stream each) run concurrently on the GPU.
    ACEHIP_EXTRA_HIPCC_FLAGS: experiments only; part of the fingerprint, so a process that does not carry the same value refuses
    (rebuilds) a library built with it."""
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-fgpu-default-stream=per-thread", "-Wall",
            "-Wno-unused-function"] + os.environ.get("ACEHIP_EXTRA_HIPCC_FLAGS", "").split()


def source_fingerprint():
    root = os.path.dirname(HERE)
    h = hashlib.sha256()
    h.update(("flags: %s | %s\n" % (" ".join(hip_flags()), sorted(PER_FILE_FLAGS.items()))).encode())
    files = []
    for top in (os.path.join(root, "include"), CSRC):
        for d, _, names in os.walk(top):
            files += [os.path.join(d, n) for n in names if n.endswith((".h", ".hip", ".hpp", ".cpp", ".inc"))]
    for p in sorted(files):
        h.update(os.path.relpath(p, root).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def embedded_fingerprint(lib):
    """the fingerprint a built library carries (None: no library / an old one without the tag)"""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return None
    i = data.find(FPR_TAG)
    return data[i + len(FPR_TAG):i + len(FPR_TAG) + 16].decode("ascii", "replace") if i >= 0 else None


def _dep_hash(cmd, deps):
    h = hashlib.sha256(" ".join(cmd).encode())
    for d in deps:
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _stale(obj, cmd, deps):
    try:
        return not os.path.exists(obj) or open(obj + ".dep").read() != _dep_hash(cmd, deps)
    except OSError:
        return True


def _compile(cmd, obj, deps, verbose):
    _run(cmd, verbose)
    with open(obj + ".dep", "w") as f:
        f.write(_dep_hash(cmd, deps))


def _fingerprint_object(symbol, verbose):
    """one tiny C object that carries the fingerprint string and exports it through `symbol`"""
    fpr = source_fingerprint()
    src, obj = os.path.join(OBJDIR, "_%s.c" % symbol), os.path.join(OBJDIR, "_%s.o" % symbol)
    text = ('/* written by ace-compiler_amd/build.py */\nstatic const char fpr_[] = "%s%s";\n'
            'const char* %s(void) { return fpr_ + %d; }\n' % (FPR_TAG.decode(), fpr, symbol, len(FPR_TAG)))
    if not os.path.exists(obj) or not os.path.exists(src) or open(src).read() != text:
        with open(src, "w") as f:
            f.write(text)
        _run([shutil.which("gcc") or "gcc", "-O1", "-fPIC", "-c", src, "-o", obj], verbose)
    return obj


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def _hip_objects(force, verbose):
    """one object per source (device code for gfx950 embedded); returns (objects, anything_rebuilt)"""
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    flags = hip_flags()
    for src in SOURCES:
        s, o = os.path.join(CSRC, src), os.path.join(OBJDIR, src.replace(".", "_") + ".o")
        objs.append(o)
        cmd = [hipcc()] + flags + PER_FILE_FLAGS.get(src, []) + ["-c", s, "-o", o]
        if force or _stale(o, cmd, [s] + hdrs):
            jobs.append((cmd, o, [s] + hdrs))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    return objs, bool(jobs)


def needs_build():
    """the library is missing or was built from other sources than the ones beside it"""
    return embedded_fingerprint(LIB) != source_fingerprint()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    objs, _ = _hip_objects(force, verbose)
    fobj = _fingerprint_object("acehip_source_fingerprint", verbose)
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [fobj], verbose)
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
        # -ffp-contract=off: the FP64 canonical embedding must round like the reference's (no FMA fusion)
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-c", s, "-o", o]
        if force or _stale(o, cmd, [s] + hdrs):
            jobs.append((cmd, o, [s] + hdrs))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    rt_fobj = _fingerprint_object("acehip_rt_source_fingerprint", verbose)
    rt_objs.append(rt_fobj)
    if force or jobs or embedded_fingerprint(RT_LIB) != source_fingerprint() or _newer(RT_LIB, [LIB]):
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
    hip_fobj = os.path.join(OBJDIR, "_acehip_source_fingerprint.o")
    if force or _newer(RT_OBJS_ARCHIVE, rt_objs + hip_objs + [hip_fobj]) or not os.path.exists(RT_ARCHIVE):
        if os.path.exists(RT_OBJS_ARCHIVE):
            os.remove(RT_OBJS_ARCHIVE)
        _run(["ar", "rcs", RT_OBJS_ARCHIVE] + rt_objs + hip_objs + [hip_fobj], verbose)
        with open(RT_ARCHIVE, "w") as f:
            f.write("/* GNU ld script standing in for the archive name of the reference link line (scripts/perf.py:202-207):\n"
                    "   the objects are in libFHErt_ant_objs.a; a HIP program also needs the HIP runtime and the C++ runtime, which a\n"
                    "   plain `cc ... libFHErt_ant.a libFHErt_common.a -lgmp -lm` does not name.  Written by ace-compiler_amd/build.py. */\n"
                    "SEARCH_DIR ( %s )\nINPUT ( %s )\nINPUT ( -lamdhip64 -lstdc++ -lpthread -ldl -lm )\n" % (ROCM_LIB, RT_OBJS_ARCHIVE))
    return RT_LIB
