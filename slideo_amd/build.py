"""In-tree build of the native libraries (no pip, no JIT cache).

  slideo_amd/lib/libslideo_amd.so    HIP kernels + C ABI, hipcc --offload-arch=gfx950
  slideo_amd/lib/libslideo_synth.so  host-only synthetic-input generator, g++

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
build container; the built .so files travel to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(LIBDIR, "libslideo_amd.so")
SYNTH_LIB = os.path.join(LIBDIR, "libslideo_synth.so")

# one translation unit per stage; every kernel header (csrc/*.hip.h) is compiled by exactly one of them (csrc/runtime.hpp)
HIP_SOURCES = ["capi_runtime.hip", "capi_group.hip", "capi_taps.hip", "stage_orb.hip", "stage_knn.hip", "stage_verify.hip", "stage_sift.hip"]
# the kernel headers each unit includes (beyond runtime.hpp and the plain headers, which every unit depends on)
HIP_UNIT_HEADERS = {"stage_orb.hip": ["orb.hip.h", "cv_math.hip.h"],
                    "stage_knn.hip": ["knn.hip.h", "knn_tile.hip.h", "knn_tile1.hip.h", "knn_l2.hip.h", "knn_lsh.hip.h"],
                    "stage_verify.hip": ["verify.hip.h", "homography.hip.h"],
                    "stage_sift.hip": ["sift.hip.h", "cv_math.hip.h"]}
HIP_COMMON_HEADERS = ["runtime.hpp", "common.h", "geom.h", "types.h"]
OBJDIR = os.path.join(LIBDIR, "obj")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; cannot build the gfx950 library")


def build_hip(force=False, verbose=False, tag=None):
    """Compiles the changed units (in parallel) and links libslideo_amd.so.  tag: an experiment build (SLIDEO_HIP_EXTRA_FLAGS) into
    lib/variants/<tag>/ — loaded through SLIDEO_LIB_PATH, the product library stays as it is (tools/ab_variants.sh)."""
    global_lib, objdir = HIP_LIB, OBJDIR
    if tag:
        objdir = os.path.join(LIBDIR, "variants", tag)
        global_lib = os.path.join(objdir, "libslideo_amd.so")
    return _build_hip(force, verbose, global_lib, objdir)


def _build_hip(force, verbose, HIP_LIB, OBJDIR):
    os.makedirs(OBJDIR, exist_ok=True)
    common = [os.path.join(CSRC, h) for h in HIP_COMMON_HEADERS] + [os.path.join(INCLUDE, "slideo_amd.h")]
    extra = os.environ.get("SLIDEO_HIP_EXTRA_FLAGS", "").split()           # (experiments: -D switches of the kernels)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-gpu-rdc",
             "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC] + extra
    stamp = os.path.join(OBJDIR, "flags.txt")                              # another flag set = another build of every unit
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs, objs = [], []
    for src in HIP_SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [os.path.join(CSRC, src)] + common + [os.path.join(CSRC, h) for h in HIP_UNIT_HEADERS.get(src, [])]
        if force or _newer(obj, deps):
            jobs.append([_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj])
    if not jobs and not _newer(HIP_LIB, objs):
        return HIP_LIB
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    open(stamp, "w").write(" ".join(flags))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", HIP_LIB] + objs + ["-pthread"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return HIP_LIB


def build_synth(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(CSRC, "synth.cpp")
    if not (force or _newer(SYNTH_LIB, [src])):
        return SYNTH_LIB
    subprocess.check_call(["g++", "-O3", "-march=x86-64-v3", "-mtune=generic", "-std=c++17", "-fPIC",
                           "-shared", "-ffp-contract=off", "-pthread", "-o", SYNTH_LIB, src])
    return SYNTH_LIB


HOST_DEMO = os.path.join(LIBDIR, "host_demo")


def build_host_demo(force=False):
    """C++ host-side mirror of the trait surface (slideo_amd/host/matching.hpp) + its demo driver."""
    srcs = [os.path.join(HERE, "host", "host_demo.cpp"), os.path.join(HERE, "host", "matching.hpp"),
            os.path.join(HERE, "host", "png.hpp"), os.path.join(INCLUDE, "slideo_amd.h")]
    if not (force or _newer(HOST_DEMO, srcs + [HIP_LIB])):
        return HOST_DEMO
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", INCLUDE, "-I", os.path.join(HERE, "host"), "-o", HOST_DEMO, srcs[0],
                           "-L", LIBDIR, "-lslideo_amd", "-Wl,-rpath,$ORIGIN", "-L", "/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib",
                           "-lamdhip64", "-lz"])
    return HOST_DEMO


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force), build_host_demo(force)


if __name__ == "__main__":
    import sys
    if "--tag" in sys.argv:
        print(build_hip(verbose=False, tag=sys.argv[sys.argv.index("--tag") + 1]))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
