"""Build the native libraries in-tree (hipcc cross-compiles gfx950 without a GPU).

  lib/libkwave_hip.so   device layer: HIP kernels + rocFFT wrapper + C-ABI (include/kwave_hip.h)
  lib/libkwave_host.so  C++ host mirror of the reference's Parameters/MatrixContainer/KSpaceFirstOrderSolver
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_DIR = os.path.join(PKG, "lib")
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INCLUDE = os.path.join(ROOT, "include")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

HIP_LIB = os.path.join(LIB_DIR, "libkwave_hip.so")
HOST_LIB = os.path.join(LIB_DIR, "libkwave_host.so")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))
    return r.stdout


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if force or _newer(HIP_LIB, deps):
        objs = []
        for s in srcs:
            o = os.path.join(LIB_DIR, os.path.basename(s) + ".o")
            if force or _newer(o, [s] + [d for d in deps if d.endswith(".h")]):
                cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                       "-I" + INCLUDE, "-I" + CSRC, "-I" + os.path.join(ROCM, "include"), "-c", s, "-o", o]
                if verbose:
                    cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                out = _run(cmd)
                if verbose:
                    print(out)
            objs.append(o)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs +
             ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    return HIP_LIB


def build_host(force: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    if not srcs:
        return ""
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if force or _newer(HOST_LIB, deps) or _newer(HOST_LIB, [HIP_LIB]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-I" + INCLUDE, "-I" + HOST, "-o", HOST_LIB] +
             srcs + ["-L" + LIB_DIR, "-lkwave_hip", "-Wl,-rpath,$ORIGIN", "-ldl"])
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False):
    build_hip(force, verbose)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print("built:", HIP_LIB, HOST_LIB if os.path.exists(HOST_LIB) else "")
