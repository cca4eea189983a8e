#!/usr/bin/env python3
"""Sustained run of the bench workload: N steps in legs, rate per leg and max|p| per leg (the PML absorbs the p0
pulse: the field must stay finite and decay).   python tools/soak.py [n = 256] [legs = 10] [steps per leg = 1000]"""
import sys
import os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402

n, legs, per = (int(a) for a in (sys.argv[1:4] + ["256", "10", "1000"][len(sys.argv) - 1:]))
pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=legs * per + 8)
sim = HostSolver(pr, p_max=1)
sim.run(5)
for leg in range(legs):
    ms = sim.time_steps(per)
    p = sim.field("p")
    print(f"steps {5 + (leg + 1) * per:6d}  {per / (ms * 1e-3):8.1f} steps/s  max|p| = {np.abs(p).max():.4e}  finite = {bool(np.isfinite(p).all())}",
          flush=True)
sim.close()
