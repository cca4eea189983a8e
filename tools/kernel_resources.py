#!/usr/bin/env python3
"""Registers / LDS / occupancy of the fused kernels at one line length, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage) on a KW_FUSED_ONLY build:  python tools/kernel_resources.py 500"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "k-wave-fluid-cuda_amd", "csrc")


def compile_tu(args):
    length, tu, tmp, extra = args
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-fno-slp-vectorize", f"-DKW_FUSED_ONLY={length}", "-Rpass-analysis=kernel-resource-usage", "-I" + CSRC,
           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-c", os.path.join(CSRC, "kw_fused.hip"), "-o",
           os.path.join(tmp, f"f{tu}.o")] + ([f"-DKW_FUSED_TU={tu}"] if tu else []) + extra
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    extra = sys.argv[2:]
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(8) as pool:
        texts = list(pool.map(compile_tu, [(length, tu, tmp, extra) for tu in (0, 1, 2, 3, 4, 5, 6, 7, 8)]))
    pat = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                     r"Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)
    for txt in texts:
        if "error:" in txt:
            print(txt[-3000:])
        for m in pat.finditer(txt):
            dn = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            dn = re.sub(r"\(.*\)$", "", dn.replace("(anonymous namespace)::", "").replace("void ", ""))
            print(f"{dn:52s} vgpr {m.group(2):>4s} agpr {m.group(3):>3s} scratch {m.group(4):>4s} occ {m.group(5)} lds {m.group(6)}")


if __name__ == "__main__":
    main()
