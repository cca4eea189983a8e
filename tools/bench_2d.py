#!/usr/bin/env python3
"""2-D simulations (Nz == 1): time-steps/s of the fused 2-D pipeline (x-pass, fused pass along y, x-inverse + epilogue)
against the rocFFT path, heterogeneous absorbing nonlinear medium:   python tools/bench_2d.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402

for n in ([int(x) for x in sys.argv[1:]] or [256, 512, 1024]):
    pr = synthetic.as_2d_file(synthetic.make_problem(n, n, 1, heterogeneous=True, nonlinear=True, absorbing=True, source="p0",
                                                     nt=2200, pml_size=10, sensor="random"))
    for fused in (True, False):
        g = HostSolver(pr, p_max=1, fused_kernels=fused)
        g.run(100)
        g.sync()
        t0 = time.perf_counter()
        g.run(2000)
        g.sync()
        dt = time.perf_counter() - t0
        print(f"{n} x {n}  {'fused' if fused else 'rocFFT'}: {2000 / dt:9.1f} steps/s  {1e6 * dt / 2000:7.1f} us/step")
        g.close()
