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

HIP_SOURCES = ["slideo_capi.hip"]
HIP_DEPS_GLOB = (".hip", ".h", ".hpp")


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


def build_hip(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(HIP_DEPS_GLOB)]
    deps.append(os.path.join(INCLUDE, "slideo_amd.h"))
    if not (force or _newer(HIP_LIB, deps)):
        return HIP_LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-fgpu-rdc" if False else "-fno-gpu-rdc",
           "-Wall", "-Wno-unused-function", "-I", INCLUDE, "-I", CSRC, "-o", HIP_LIB]
    cmd += os.environ.get("SLIDEO_HIP_EXTRA_FLAGS", "").split()           # (experiments: -D switches of the kernels)
    cmd += [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
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
    print(build_all(force="--force" in sys.argv, verbose=True))
