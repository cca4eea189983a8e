"""Build the native libraries in-tree (hipcc cross-compiles gfx950 without a GPU).

  lib/libkwave_hip.so   device layer: HIP kernels + rocFFT wrapper + C-ABI (include/kwave_hip.h)
  lib/libkwave_host.so  C++ host mirror of the reference's Parameters/MatrixContainer/KSpaceFirstOrderSolver
"""
from __future__ import annotations

import concurrent.futures
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

FUSED_TUS = (7, 8, 0, 2, 1, 5, 3, 6, 4)  # passes over kw_fused.hip, slowest first
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
        objs, jobs = [], []
        # kw_fused.hip is compiled in five passes (its x-inverse epilogue kernels in four of them, see the file's header)
        units = [(s, tu) for s in srcs for tu in (FUSED_TUS if os.path.basename(s) == "kw_fused.hip" else (None,))]
        for s, tu in units:
            o = os.path.join(LIB_DIR, os.path.basename(s) + (".o" if not tu else f".tu{tu}.o"))
            if force or _newer(o, [s] + [d for d in deps if d.endswith(".h")]):
                extra = os.environ.get("KW_HIPCC_EXTRA", "").split() + ([f"-DKW_FUSED_TU={tu}"] if tu else [])
                # -fno-slp-vectorize: packed-f32 pairing buys nothing here (same instruction count, more moves; measured +0.6 %)
                cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                       "-fno-slp-vectorize"] + extra + [
                    "-I" + INCLUDE, "-I" + CSRC, "-I" + os.path.join(ROCM, "include"), "-c", s, "-o", o]
                if verbose:
                    cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                jobs.append(cmd)
            objs.append(o)
        # the translation units compile side by side (kw_fused.hip alone takes minutes: one kernel set per line length)
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs)))) as pool:
            for out in pool.map(_run, jobs):
                if verbose:
                    print(out)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs +
             ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    return HIP_LIB


def build_hip_variant(name: str, extra_flags, only_length: int = 256) -> str:
    """Tuning build of the device library into ab/<name>/libkwave_hip.so (tools/ab.sh compares such builds on one GPU
    box): the fused pipeline restricted to one line length (compiles in seconds) plus extra compiler flags."""
    out_dir = os.path.join(ROOT, "ab", name)
    os.makedirs(out_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    units = [(s, tu) for s in srcs for tu in (FUSED_TUS if os.path.basename(s) == "kw_fused.hip" else (None,))]
    jobs, objs = [], []
    for s, tu in units:
        o = os.path.join(out_dir, os.path.basename(s) + (".o" if not tu else f".tu{tu}.o"))
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize"]
        cmd += list(extra_flags) + ([f"-DKW_FUSED_ONLY={only_length}"] if only_length else [])
        cmd += ([f"-DKW_FUSED_TU={tu}"] if tu else []) + ["-I" + INCLUDE, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
                                                          "-I" + os.path.join(ROCM, "include"), "-c", s, "-o", o]
        jobs.append(cmd)
        objs.append(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(_run, jobs))
    lib = os.path.join(out_dir, "libkwave_hip.so")
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
         ["-L" + os.path.join(ROCM, "lib"), "-lrocfft", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    for o in objs:
        os.remove(o)
    return lib


def build_host(force: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    if not srcs:
        return ""
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if force or _newer(HOST_LIB, deps) or _newer(HOST_LIB, [HIP_LIB]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-I" + INCLUDE, "-I" + HOST, "-o", HOST_LIB] +
             srcs + ["-L" + LIB_DIR, "-lkwave_hip", "-Wl,-rpath,$ORIGIN", "-ldl"])
    return HOST_LIB


HOST_H5_LIB = os.path.join(LIB_DIR, "libkwave_host_h5.so")
CLI_BIN = os.path.join(LIB_DIR, "kspaceFirstOrder-HIP")
HDF5_ROOT = os.environ.get("HDF5_ROOT", "/opt/conda")


def build_host_h5(force: bool = False) -> str:
    """Optional HDF5 component: the host layer + Hdf5File/Hdf5Input/output writer, and the command-line program."""
    if not os.path.exists(os.path.join(HDF5_ROOT, "include", "hdf5.h")):
        return ""
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp"))) + [os.path.join(HOST, "h5", "Hdf5File.cpp"),
                                                               os.path.join(HOST, "h5", "SeriesWriter.cpp"),
                                                               os.path.join(HOST, "h5", "h5_capi.cpp")]
    deps = srcs + glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(HOST, "h5", "*"))
    common = ["-O2", "-std=c++17", "-fPIC", "-fopenmp", "-I" + INCLUDE, "-I" + HOST, "-I" + os.path.join(HDF5_ROOT, "include")]
    # HDF5 by full path and the system library directory ahead of conda's in the run path: /opt/conda/lib also holds an
    # older libstdc++ that must not shadow the system one (rocFFT needs GLIBCXX_3.4.30)
    h5lib = os.path.join(HDF5_ROOT, "lib")
    link = ["-L" + LIB_DIR, "-lkwave_hip", os.path.join(h5lib, "libhdf5_hl.so"), os.path.join(h5lib, "libhdf5.so"),
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/usr/lib/x86_64-linux-gnu", "-Wl,-rpath," + h5lib, "-ldl"]
    if force or _newer(HOST_H5_LIB, deps) or _newer(HOST_H5_LIB, [HIP_LIB]):
        _run(["g++"] + common + ["-shared", "-o", HOST_H5_LIB] + srcs + link)
    if force or _newer(CLI_BIN, deps + [HOST_H5_LIB]):
        _run(["g++"] + common + ["-o", CLI_BIN, os.path.join(HOST, "h5", "main.cpp")] +
             ["-L" + LIB_DIR, "-lkwave_host_h5"] + link)
    return HOST_H5_LIB


SLAB_SELFTEST_BIN = os.path.join(LIB_DIR, "slab_selftest")


def build_native_drivers(force: bool = False) -> str:
    """tests/native/slab_selftest.c: the plain-C driver of the slab path over the library's own RCCL exchange."""
    src = os.path.join(ROOT, "tests", "native", "slab_selftest.c")
    if not (os.path.exists(src) and os.path.exists(HOST_H5_LIB)):
        return ""
    if force or _newer(SLAB_SELFTEST_BIN, [src, HOST_H5_LIB, HIP_LIB] + glob.glob(os.path.join(INCLUDE, "*.h"))):
        _run(["gcc", "-O1", "-std=c11", "-Wall", "-I" + INCLUDE, "-o", SLAB_SELFTEST_BIN, src, "-L" + LIB_DIR,
              "-lkwave_host_h5", "-lkwave_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/usr/lib/x86_64-linux-gnu"])
    return SLAB_SELFTEST_BIN


MOCK_RCCL_LIB = os.path.join(ROOT, "tests", "native", "libmock_rccl.so")  # test infrastructure: not in the product's lib/


def build_test_mocks(force: bool = False) -> str:
    """tests/native/mock_rccl.cpp: test infrastructure — a stand-in for librccl whose ranks are threads sharing one GPU
    (named through kw_comm_init_with by the test workers), so that the library's RCCL exchange path can be driven with
    several ranks on a one-GPU box."""
    src = os.path.join(ROOT, "tests", "native", "mock_rccl.cpp")
    if not os.path.exists(src):
        return ""
    if force or _newer(MOCK_RCCL_LIB, [src]):
        _run([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", MOCK_RCCL_LIB, src])
    return MOCK_RCCL_LIB


def build_all(force: bool = False, verbose: bool = False):
    build_hip(force, verbose)
    build_host(force)
    build_host_h5(force)
    build_native_drivers(force)
    build_test_mocks(force)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":  # build.py --variant <name> [flags...]
        print("built:", build_hip_variant(sys.argv[2], sys.argv[3:], int(os.environ.get("KW_VARIANT_LENGTH", "256"))))
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print("built:", HIP_LIB, HOST_LIB if os.path.exists(HOST_LIB) else "")
