"""Parity at the benchmark's full size (256^3, BASELINE configs 3 / 5 grid) through properties that do not need the CPU
oracle to finish a 256^3 run: the fp64 closed form of the lossless homogeneous problem (K1), agreement of the two
independent device implementations of the step (hand-written fused FFT passes vs rocFFT + one kernel per reference
kernel), linearity in the source amplitude for the linear medium, and bit-exact sampling of what is on the device."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5
N = 256


def gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


def test_k1_closed_form_256(syn):
    from oracle.kwave_np import closed_form_pressure
    pr = syn.make_problem(N, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0", nt=70)
    g = gpu(pr)
    g.run(61)
    assert rel_l2(g.field("p"), closed_form_pressure(pr, 60)) < TOL
    g.close()


def test_config3_fused_pipeline_equals_rocfft_path_256(syn):
    """heterogeneous c0 / rho0 / BonA / alpha_coeff, absorbing + nonlinear, the bench workload: 30 steps"""
    pr = syn.make_problem(N, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=40)
    a = gpu(pr, p_raw=1, p_max=1, fused_kernels=True)
    a.run(30)
    a.finish()
    fields_a = {f: a.field(f) for f in ("p", "ux", "uz", "rhoy")}
    series_a, max_a = a.stream("p"), a.stream("p_max")
    # bit-exact sampling of the device field (sensor = the z = N/2 plane, 65 536 points)
    mask = pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    assert np.array_equal(series_a[-1], fields_a["p"].reshape(-1)[mask])
    assert np.array_equal(max_a, series_a.max(axis=0))
    a.close()
    b = gpu(pr, p_raw=1, p_max=1, fused_kernels=False)
    b.run(30)
    b.finish()
    for f, va in fields_a.items():
        assert rel_l2(va, b.field(f)) < TOL, f
    assert rel_l2(series_a, b.stream("p")) < TOL
    b.close()


def test_config3_256_vs_oracle(orc, syn):
    """BASELINE config 3 at its real size, straight against the CPU oracle: 20 steps of the bench medium (heterogeneous
    c0 / rho0 / BonA / alpha_coeff, absorbing + nonlinear, p0 source) through the fused pipeline; state fields and the
    sensor series (one xy plane, 65 536 points) within 1e-5 relative L2, sampling bit-exact on the device field."""
    steps = 20
    pr = syn.make_problem(N, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=steps + 4)
    g = gpu(pr, p_raw=1, p_max=1)
    g.run(steps)
    g.finish()
    o = orc.OracleSim(pr)
    series = np.empty((steps, o.sensor_index.size), dtype=np.float32)
    for t in range(steps):
        o.step()
        series[t] = o.field("p").reshape(-1)[o.sensor_index]
    for f in ("p", "ux", "uy", "uz", "rhox", "rhoz"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    s = g.stream("p")
    assert s.shape == series.shape and rel_l2(s, series) < TOL
    assert np.array_equal(s[-1], g.field("p").reshape(-1)[o.sensor_index.astype(np.int64)])
    assert rel_l2(g.stream("p_max"), series.max(axis=0)) < TOL
    o.close()
    g.close()


def test_config5_256_vs_oracle(orc, syn):
    """BASELINE config 5 at its real size against the oracle: 256^3 config-3 medium driven by the 1 MHz p_source plane,
    --p_c --u_non_staggered_c --I_avg_c (+ the raw series they are built from).  The oracle runs the same 60 steps; its
    fields, its raw pressure series, its shifted-velocity restatement (computeShiftedVelocity, :2714-2735) and its
    compressor (IndexOutputStream.cpp:373-470, basis pinned to the compiled reference CompressHelper) are the checks."""
    pr = syn.make_problem(N, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=1, nt=64)
    dt = float(pr["dt"].ravel()[0])
    period, mos, harm = 1.0 / (1.0e6 * dt), 1, 1
    steps = 2 * int(period * mos) + 3  # two emitted frames
    assert steps <= 64
    g = gpu(pr, p_raw=1, u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, period=period, mos=mos, harmonics=harm)
    g.run(steps)
    g.finish()
    o = orc.OracleSim(pr)
    mask = o.sensor_index.astype(np.int64)
    p_series = np.empty((steps, mask.size), dtype=np.float32)
    ux_series = np.empty((steps, mask.size), dtype=np.float32)
    for t in range(steps):
        o.step()
        p_series[t] = o.field("p").reshape(-1)[mask]
        ux_series[t] = orc.shifted_velocity(o.field("ux"), pr["x_shift_neg_r"], 0).reshape(-1)[mask]
    for f in ("p", "ux", "uz", "rhoy"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    assert rel_l2(g.stream("p"), p_series) < TOL
    assert rel_l2(g.stream("ux_non_staggered"), ux_series) < TOL
    for axis, nm in enumerate(("ux", "uy", "uz")):
        ref = orc.shifted_velocity(o.field(nm), pr["xyz"[axis] + "_shift_neg_r"], axis)
        assert rel_l2(g.field(nm + "_shifted"), ref) < TOL, nm
        assert np.array_equal(g.stream(nm + "_non_staggered")[-1], g.field(nm + "_shifted").reshape(-1)[mask])
    # compression frames of the oracle's own series vs the device's accumulation of the device's series
    frames = {}
    for name, series, shifted in (("p_c", p_series, False), ("ux_non_staggered_c", ux_series, True)):
        comp = orc.Compressor(mask.size, period, mos, harm, shifted)
        for row in series:
            comp.step(row)
        got = g.stream(name).reshape(-1, mask.size, harm, 2)
        ref = np.array(comp.frames)
        assert got.shape == ref.shape and got.shape[0] == 2, name
        assert rel_l2(got, ref) < TOL, name
        frames[name] = ref[..., 0] + 1j * ref[..., 1]
    ref_i = (np.real(frames["p_c"] * np.conj(frames["ux_non_staggered_c"])).sum(axis=2) / 2.0).mean(axis=0)
    assert rel_l2(g.stream("Ix_avg_c"), ref_i) < TOL
    o.close()
    g.close()


def test_config5_streams_fused_shift_equals_rocfft_shift_256(syn):
    """BASELINE config 5 at full size: the non-staggered velocity of the one-kernel-per-axis shift against the
    R2C -> multiply -> C2R form (rocFFT 1-D transforms), and the compression / intensity streams built on it."""
    nt = 48
    pr = syn.make_problem(N, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=1, nt=nt)
    dt = float(pr["dt"].ravel()[0])
    opts = dict(u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, period=1.0 / (1.0e6 * dt), mos=1, harmonics=2)
    out = {}
    for fused in (True, False):
        g = gpu(pr, fused_kernels=fused, **opts)
        g.run(nt)
        g.finish()
        out[fused] = ({f: g.field(f) for f in ("ux_shifted", "uy_shifted", "uz_shifted")},
                      {s: g.stream(s) for s in ("ux_non_staggered", "uz_non_staggered", "p_c", "ux_non_staggered_c",
                                                "uy_non_staggered_c", "Ix_avg_c", "Iz_avg_c")})
        g.close()
    for f in out[True][0]:
        assert np.abs(out[False][0][f]).max() > 0 and rel_l2(out[True][0][f], out[False][0][f]) < TOL, f
    # on the sensor plane (z = N/2, a symmetry plane of the source) the y / z components are rounding noise: errors are
    # taken relative to the x component of the same kind
    scale = {"ux_non_staggered": "ux_non_staggered", "uz_non_staggered": "ux_non_staggered", "p_c": "p_c",
             "ux_non_staggered_c": "ux_non_staggered_c", "uy_non_staggered_c": "ux_non_staggered_c", "Ix_avg_c": "Ix_avg_c",
             "Iz_avg_c": "Ix_avg_c"}
    for s, ref in scale.items():
        a, b = out[True][1][s], out[False][1][s]
        assert a.shape == b.shape, s
        assert np.abs(a - b).max() < 2e-5 * np.abs(out[False][1][ref]).max(), s


def test_linearity_256(syn):
    """linear lossless heterogeneous medium: the field scales with the source amplitude"""
    pr = syn.make_problem(N, heterogeneous=True, nonlinear=False, absorbing=False, source="p0", nt=30)
    a = gpu(pr)
    a.run(20)
    pa = a.field("p")
    a.close()
    pr2 = dict(pr)
    pr2["p0_source_input"] = (0.25 * pr["p0_source_input"]).astype(np.float32)  # exact in fp32
    b = gpu(pr2)
    b.run(20)
    assert rel_l2(4.0 * b.field("p"), pa) < 1e-6
    b.close()


def test_config4_fused_pipeline_equals_rocfft_path_512(syn):
    """BASELINE config 4 grid (512^3): the 2 x 256 split y / z kernels and the 512-point x kernels against rocFFT"""
    n = 512
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=12, sensor="random")
    a = gpu(pr, p_raw=1, fused_kernels=True)
    a.run(8)
    a.finish()
    pa, ua, sa = a.field("p"), a.field("uy"), a.stream("p")
    a.close()
    b = gpu(pr, p_raw=1, fused_kernels=False)
    b.run(8)
    b.finish()
    assert rel_l2(pa, b.field("p")) < TOL and rel_l2(ua, b.field("uy")) < TOL
    assert rel_l2(sa, b.stream("p")) < TOL
    b.close()
