#!/usr/bin/env python3
"""Device-copy bandwidth of the box as bench.py reports it (kw_measure_copy_bandwidth: best of three float4 copy
shapes) at a few buffer sizes:   python tools/copy_bandwidth.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kwave_amd  # noqa: E402,F401
from kwave_amd import capi  # noqa: E402

dev = capi.Device()
g = C.c_double()
for mb in ([int(x) for x in sys.argv[1:]] or [256, 1024, 2048]):
    capi.check(dev.L.kw_measure_copy_bandwidth(dev.ctx, C.c_size_t(mb << 20), 50 if mb < 256 else 10, C.byref(g)))
    print(f"{mb} MiB each way: {g.value:.1f} GB/s")
dev.close()
