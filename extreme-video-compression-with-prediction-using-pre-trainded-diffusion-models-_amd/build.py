"""Ahead-of-time build of the two native libraries, in-tree (``<package>/lib/``):

* ``libevc_hip.so``  -- HIP kernels for gfx950 (hipcc --offload-arch=gfx950), C ABI include/evc_hip.h
* ``libevc_rans.so`` -- host range-ANS coder (g++), C ABI include/evc_rans.h

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting ``.so`` files
travel to the GPU box with the tree (they are git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
HIP_SOURCES = ["conv_igemm.hip", "attention.hip", "norm.hip", "fir.hip", "elementwise.hip", "gdn.hip", "stride2.hip", "lpips.hip", "frames.hip", "api.hip"]
ARCH = "gfx950"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))


def hipcc_path():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libevc_hip.so)")


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(INCLUDE, h) for h in ("evc_hip.h", "evc_rans.h")]
    hipcc = hipcc_path()
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE]
    flags += os.environ.get("EVC_HIPCC_FLAGS", "").split()     # e.g. -DEVC_CONV_PC=1 for A/B experiments

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(o, [s] + headers):
            if verbose:
                print("hipcc", src, flush=True)
            _run([hipcc] + flags + ["-c", s, "-o", o])
        return o

    with ThreadPoolExecutor(max_workers=min(6, len(HIP_SOURCES))) as ex:
        objs = list(ex.map(compile_one, HIP_SOURCES))
    hip_so = os.path.join(LIBDIR, "libevc_hip.so")
    if force or _newer(hip_so, objs):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", hip_so] + objs)
    rans_src = os.path.join(CSRC, "rans.cpp")
    rans_so = os.path.join(LIBDIR, "libevc_rans.so")
    if force or _newer(rans_so, [rans_src] + headers):
        _run([shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", INCLUDE,
              rans_src, "-o", rans_so])
    return hip_so, rans_so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
