#!/usr/bin/env python3
"""Per-step cost of the sampling streams of the bench configuration (one-plane mask, 65 536 points) at 256^3:
python tools/stream_cost.py  — run under KW_RAW_COPY / KW_RAW_ZERO_COPY to compare the ways a raw series reaches the host."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps, warm = (300, 20) if n >= 256 else (2000, 100)
pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=steps * 2 + warm + 8)
for name, opts in (("none", {}), ("p_max", dict(p_max=1)), ("p_raw", dict(p_raw=1)), ("p_raw+p_max", dict(p_raw=1, p_max=1))):
    sim = HostSolver(pr, **opts)
    sim.run(warm)
    sim.sync()
    ms = sim.time_steps(steps)
    ms2 = sim.time_steps(steps)
    sim.close()
    print(f"{name:12s} {ms / steps * 1e3:8.1f} us/step {ms2 / steps * 1e3:8.1f}", flush=True)
