#!/usr/bin/env python3
"""Cost of the sampling streams on top of the 256^3 bench loop (BASELINE config 5: on-the-fly compression + time-averaged
intensity): time-steps/s with a one-plane sensor mask (65 536 points) and different stream sets, tone-burst pressure source
(the compression period comes from it)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402


def main(n=256, steps=200, warm=20):
    pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=1,
                                nt=steps + warm + 4, nt_src=8)       # the source stops early: steady-state loop
    dt = float(pr["dt"].ravel()[0])
    comp = dict(period=1.0 / (1.0e6 * dt), mos=1, harmonics=2)
    cases = [("no streams", {}),
             ("p_raw + p_max", dict(p_raw=1, p_max=1)),
             ("p, u raw + rms + max + min", dict(p_raw=1, p_rms=1, p_max=1, p_min=1, u_raw=1, u_rms=1, u_max=1, u_min=1)),
             ("u_non_staggered_raw", dict(u_non_staggered_raw=1)),
             ("p_c + u_non_staggered_c + I_avg_c (config 5)", dict(p_c=1, u_non_staggered_c=1, i_avg_c=1, **comp)),
             ("Q_term_c", dict(q_term_c=1, **comp)),
             ("I_avg + Q_term (raw series kept for post-processing)", dict(i_avg=1, q_term=1))]
    for name, opts in cases:
        sim = HostSolver(pr, **opts)
        sim.run(warm)
        sim.sync()
        ms = sim.time_steps(steps)
        print(f"{n}^3 {name:55s} {steps / (ms * 1e-3):8.1f} steps/s  {ms / steps:7.4f} ms/step", flush=True)
        if "config 5" in name:  # what the sampling side of that configuration consists of, per step
            from kwave_amd import capi
            hip = capi.load()
            capi.check(hip.kw_profile_enable(sim.ctx, 1))
            sim.run(10)
            prof = capi.profile_collect(sim.ctx)
            capi.check(hip.kw_profile_enable(sim.ctx, 0))
            for kname, (calls, total_ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                if not kname.startswith(("k_xinv", "k_ypass", "k_zfused_pgrad", "k_zfused_vgrad", "k_zfused_absorb")):
                    print(f"      {kname:28s} {calls / 10:5.1f} calls/step  {1e3 * total_ms / max(calls, 1):8.1f} us each")
        sim.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
