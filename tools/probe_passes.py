#!/usr/bin/env python3
"""Pass-level micro-benchmark of the fused pipeline (kw_fused_probe): HIP-event time per pass at n^3 and the
bandwidth it corresponds to (complex scratch array C read + written per pass)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import capi  # noqa: E402


def main(n=256, reps=30):
    dev = capi.Device()
    k = capi.Constants()
    k.nx = k.ny = k.nz = n
    k.n_elements = n ** 3
    k.nx_complex, k.ny_complex, k.nz_complex = n // 2 + 1, n, n
    k.n_elements_complex = (n // 2 + 1) * n * n
    k.fft_divider = 1.0 / n ** 3
    dev.set_constants(k)
    dev.call("fused_create")
    ne = C.c_size_t()
    capi.check(dev.L.kw_fused_reduced_elems(dev.ctx, C.byref(ne)))
    op = dev.array(np.ones(ne.value, dtype=np.float32))
    cbytes = 8 * (n // 2 + 1) * n * n
    names = {0: "y-pass (1 array)", 1: "line pass along z (1 array)", 2: "z-fused (1 array + operator)", 3: "y-pass (3 arrays)",
             10: "flat float4 copy in place", 11: "y tiles: load + store only", 12: "y tiles: + LDS exchange",
             13: "z tiles: load + store only", 14: "z tiles: + LDS exchange"}
    big = dev.array(np.ones(6 * (n ** 3 + 263168), dtype=np.float32))
    names[20] = "x-inverse + velocity epilogue pattern (15 units)"
    names[21], names[22], names[23] = "  same, arrays staggered by 4 KiB", "  same, staggered by 68 KiB", "  same, staggered by 1 MiB + 4 KiB"
    names[15], names[16] = "y tiles, 256-B segments: load + store", "z tiles, 256-B segments: load + store"
    for w, nm in enumerate(("y tiles 128 B", "z tiles 128 B", "y tiles 256 B", "z tiles 256 B", "y tiles 128 B, XCD-grouped",
                            "z tiles 128 B, XCD-grouped", "y tiles 256 B, XCD-grouped", "z tiles 256 B, XCD-grouped")):
        names[30 + w] = "rt: " + nm
    order = (10, 11, 15, 12, 0, 13, 16, 14, 1, 2, 3, 20) if n == 256 else (10, 0, 1, 2, 3)
    for which in order + tuple(range(30, 38)):
        for _ in range(3):
            dev.call("fused_probe", which, big if which >= 20 else op)
        e0, e1 = dev.event(), dev.event()
        dev.record(e0)
        for _ in range(reps):
            dev.call("fused_probe", which, big if which >= 20 else op)
        dev.record(e1)
        ms = dev.elapsed_ms(e0, e1) / reps
        arrays = 3 if which == 3 else 1
        traffic = 2 * cbytes * arrays + (ne.value * 4 if which == 2 else 0)
        if which >= 20:
            traffic = 6 * cbytes + 9 * 4 * n ** 3
        print(f"{names[which]:32s} {ms * 1e3:8.1f} us  {traffic / ms / 1e6:8.1f} GB/s")
    dev.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
