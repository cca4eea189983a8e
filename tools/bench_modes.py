#!/usr/bin/env python3
"""Time-steps/s of the four equation-of-state modes ({linear, nonlinear} x {lossless, absorbing}) at n^3, heterogeneous
medium, p0 source, p_raw + p_max on one plane — the fast path of each mode next to the headline config of bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402


def main(n=256, steps=60, warm=8):
    for nonlinear in (False, True):
        for absorbing in (False, True):
            pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=nonlinear, absorbing=absorbing, source="p0",
                                        nt=steps + warm + 4)
            sim = HostSolver(pr, p_raw=1, p_max=1)
            sim.run(warm)
            sim.sync()
            ms = sim.time_steps(steps)
            sim.close()
            print(f"{n}^3 {'nonlinear' if nonlinear else 'linear':9s} {'absorbing' if absorbing else 'lossless':9s} "
                  f"{steps / (ms * 1e-3):8.1f} steps/s  {ms / steps:7.4f} ms/step", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
