"""GPU parity tests proper: the C++ time loop (libkwave_host -> libkwave_hip, HIP kernels + rocFFT) against the CPU
oracle on the same seeded inputs.  Tolerance: relative L2 <= 1e-5 on pressure (BASELINE.json north_star);
sensor sampling bit-exact with respect to the sampled field."""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def make_gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


def compare(orc, pr, steps, fields=("p", "ux", "uy", "uz", "rhox"), **kw):
    g = make_gpu(pr, **kw)
    o = orc.OracleSim(pr)
    g.run(steps)
    o.step(steps)
    errs = {f: rel_l2(g.field(f), o.field(f)) for f in fields}
    g.close()
    o.close()
    return errs


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("nonlinear", [False, True])
@pytest.mark.parametrize("absorbing", [False, True])
@pytest.mark.parametrize("heterogeneous", [False, True])
def test_media_p0_32(orc, syn, fused, nonlinear, absorbing, heterogeneous):
    pr = syn.make_problem(32, heterogeneous=heterogeneous, nonlinear=nonlinear, absorbing=absorbing, source="p0")
    errs = compare(orc, pr, 40, fused_kernels=fused)
    assert max(errs.values()) < TOL, errs


def test_preprocessing_operators_match_oracle(orc, syn):
    """kappa / nabla / tau / eta / c2 / dt/rho0_sg are *computed* by the solver (KSpaceFirstOrderSolver.cpp:2404-2703)."""
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4)
    g, o = make_gpu(pr), orc.OracleSim(pr)
    g.run(1)
    for f in ("kappa", "nabla1", "nabla2", "tau", "eta", "c2", "dtrho0sgx", "dtrho0sgz"):
        assert rel_l2(g.field(f), o.field(f)) < 1e-6, f
    g.close()
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=True, source="p0", pml_size=4)
    g, o = make_gpu(pr), orc.OracleSim(pr)
    g.run(1)
    assert g.scalar("absorb_tau") == pytest.approx(o.scalar("tau"), rel=1e-6)
    assert g.scalar("absorb_eta") == pytest.approx(o.scalar("eta"), rel=1e-6)
    g.close()


def test_host_generators_are_bit_identical_to_the_frozen_arrays():
    """The pre-processing generators were rewritten around per-axis tables (round 2); their output must not move by a
    bit from what the statement-order round-1 build produced (tests/golden/host_generators.npz, written on an MI355X
    box by tests/golden/make_host_generators.py with the round-1 libraries)."""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_host_generators", os.path.join(here, "golden", "make_host_generators.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    frozen = np.load(os.path.join(here, "golden", "host_generators.npz"))
    now = mod.collect()
    assert sorted(now) == sorted(frozen.files)
    for name in frozen.files:
        assert np.array_equal(now[name], frozen[name]), name


def test_non_cubic_non_pow2(orc, syn):
    pr = syn.make_problem(24, 20, 18, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4)
    errs = compare(orc, pr, 30)
    assert max(errs.values()) < TOL, errs


def test_odd_sizes_scalar_path(orc, syn):
    """nx not a multiple of 4 and an odd half-spectrum plane: exercises the non-vectorised kernel variants."""
    pr = syn.make_problem(18, 15, 21, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=3)
    errs = compare(orc, pr, 20)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("subset", [
    {"c0": True, "rho0": False, "BonA": False, "alpha_coeff": False},
    {"c0": False, "rho0": True, "BonA": True, "alpha_coeff": False},
    {"c0": False, "rho0": False, "BonA": False, "alpha_coeff": True},
])
def test_mixed_scalar_array_medium(orc, syn, subset):
    pr = syn.make_problem(24, heterogeneous=False, nonlinear=True, absorbing=True, source="p0", hetero_subset=subset,
                          pml_size=4)
    errs = compare(orc, pr, 30)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("source", ["p_source", "u_source"])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("many", [0, 1])
def test_time_varying_sources(orc, syn, source, mode, many):
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=False, absorbing=False, source=source, source_mode=mode,
                          source_many=many, nt=40, pml_size=4)
    errs = compare(orc, pr, 40, fields=("p", "ux"))
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("source", ["p_source", "u_source"])
@pytest.mark.parametrize("mode", [0, 2])
def test_sources_on_fused_pipeline(orc, syn, source, mode):
    """Power-of-two grid: the fused pipeline handles the k-space corrected additive source (scaleSource) and, with an
    active pressure source, falls back to the stand-alone pressure-terms kernel after the injection."""
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source=source, source_mode=mode,
                          source_many=1, nt=40, nt_src=25, pml_size=4)
    errs = compare(orc, pr, 40, fields=("p", "ux", "rhoz"), fused_kernels=True)  # 15 steps past the end of the signal
    assert max(errs.values()) < TOL, errs


def test_fused_mixed_power_of_two_dims(orc, syn):
    """Nx != Ny != Nz (64 x 32 x 16): different line lengths per pass of the fused pipeline."""
    for kw in (dict(nonlinear=True, absorbing=True), dict(nonlinear=False, absorbing=False)):
        pr = syn.make_problem(64, 32, 16, heterogeneous=True, source="p0", pml_size=4, **kw)
        errs = compare(orc, pr, 30, fused_kernels=True)
        assert max(errs.values()) < TOL, (kw, errs)
    pr = syn.make_problem(16, 64, 128, heterogeneous=False, nonlinear=False, absorbing=True, source="p0", pml_size=4)
    errs = compare(orc, pr, 30, fused_kernels=True)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("dims", [(512, 16, 32), (16, 512, 16), (16, 32, 512)])
def test_fused_line_length_512(orc, syn, dims):
    """512-point lines: 8-lines-per-tile geometry (32-point register DFTs) along x, 2 x 256 split kernels along y / z."""
    pr = syn.make_problem(*dims, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4)
    errs = compare(orc, pr, 20, fused_kernels=True)
    assert max(errs.values()) < TOL, (dims, errs)


@pytest.mark.parametrize("dims", [
    (48, 16, 32), (96, 48, 16), (16, 96, 48), (192, 16, 16), (16, 192, 32), (32, 16, 192), (384, 16, 16), (16, 384, 16),
    (16, 16, 384), (48, 48, 48),                                                          # 3 * 2^m
    (72, 144, 16), (144, 16, 72), (16, 72, 144), (288, 16, 16), (16, 288, 16), (16, 16, 288), (576, 16, 16),
    (16, 576, 16), (16, 16, 576),                                                         # 9 * 2^m
    (80, 160, 16), (160, 16, 80), (16, 80, 160), (320, 16, 16), (16, 320, 16), (16, 16, 320), (640, 16, 16),
    (16, 640, 16), (16, 16, 640),                                                         # 5 * 2^m
    (120, 240, 16), (240, 16, 120), (16, 120, 240), (480, 16, 16), (16, 480, 16), (16, 16, 480),  # 15 * 2^m
    (768, 16, 16), (16, 768, 16), (16, 16, 768), (1024, 16, 16), (16, 1024, 16), (16, 16, 1024),  # the longest lines
    (100, 200, 16), (200, 16, 100), (16, 100, 200), (400, 16, 16), (16, 400, 16), (16, 16, 400),  # 25 * 2^m
    (108, 216, 16), (216, 16, 108), (16, 108, 216), (324, 16, 16), (16, 324, 16), (16, 16, 324), (432, 16, 16),
    (16, 432, 16), (16, 16, 432), (648, 16, 16), (16, 648, 16), (16, 16, 648),                    # 27 * 2^m, 81 * 4
    (300, 16, 16), (16, 300, 16), (16, 16, 300), (500, 16, 16), (16, 500, 16), (16, 16, 500), (600, 16, 16),
    (16, 600, 16), (16, 16, 600),                                                                 # 75 * 2^m, 125 * 4
    (180, 16, 16), (16, 180, 16), (16, 16, 180), (360, 16, 16), (16, 360, 16), (16, 16, 360), (540, 16, 16),
    (16, 540, 16), (16, 16, 540), (90 * 2, 360, 16),                                               # 45 * 2^m, 135 * 4
    (400, 100, 100), (500, 100, 108),       # Ny * Nz a multiple of 16 but not of 32: 16-row x tiles from Nx = 400 on
    (112, 224, 16), (224, 16, 112), (16, 112, 224), (448, 16, 16), (16, 448, 16), (16, 16, 448), (896, 16, 16),
    (16, 896, 16), (16, 16, 896),                                                                  # 7 * 2^m
    (168, 336, 16), (336, 16, 168), (16, 168, 336),                                                # 21 * 2^m
    (280, 16, 16), (16, 280, 16), (16, 16, 280), (560, 16, 16), (16, 560, 16), (16, 16, 560),      # 35 * 2^m
    (196, 392, 16), (392, 16, 196), (16, 196, 392),                                                # 49 * 2^m
    (140, 252, 16), (252, 16, 140), (16, 140, 252), (504, 16, 16), (16, 504, 16), (16, 16, 504),   # 35 * 4, 63 * 2^m
    (672, 16, 16), (16, 672, 16), (16, 16, 672), (784, 16, 16), (16, 784, 16), (16, 16, 784),      # 21 * 32, 49 * 16
    (420, 16, 16), (16, 420, 16), (16, 16, 420), (840, 16, 16), (16, 840, 16), (16, 16, 840),      # 105 * 2^m (14 x 30, 28 x 30)
    (720, 16, 16), (16, 720, 16), (16, 16, 720), (900, 16, 16), (16, 900, 16), (16, 16, 900),      # 24 x 30, 30 x 30
    (960, 16, 16), (16, 960, 16), (16, 16, 960),                                                   # 30 x 32
    (700, 16, 16), (16, 700, 16), (16, 16, 700), (756, 16, 16), (16, 756, 16), (16, 16, 756),      # L / 2 no multiple of the
    (800, 16, 16), (16, 800, 16), (16, 16, 800), (864, 16, 16), (16, 864, 16), (16, 16, 864),      # threads per line: x tiles
    (800, 48, 100), (864, 16, 48),                                                                # end in a partial round
])
def test_fused_line_lengths_mixed_radix(orc, syn, dims):
    """Line lengths with one radix-3, radix-5 or radix-7 stage (or two radix-3) inside the register DFTs: every
    supported length along every axis."""
    for kw in (dict(nonlinear=True, absorbing=True), dict(nonlinear=False, absorbing=False)):
        pr = syn.make_problem(*dims, heterogeneous=True, source="p0", pml_size=4, **kw)
        g, o = make_gpu(pr, fused_kernels=True), orc.OracleSim(pr)
        g.run(20)
        assert g.scalar("fused_pipeline") == 1.0
        o.step(20)
        errs = {f: rel_l2(g.field(f), o.field(f)) for f in ("p", "ux", "uy", "uz", "rhox", "rhoz")}
        g.close()
        o.close()
        assert max(errs.values()) < TOL, (dims, kw, errs)


@pytest.mark.parametrize("dims,opts", [
    ((300, 100, 100), {}),                                  # Ny * Nz = 10 000 = 312 tiles of 32 rows + 16 rows
    ((100, 100, 100), dict(u_non_staggered_raw=1)),         # cube of round 1's rocFFT-path list; the x-shift kernel too
    ((108, 100, 108), dict(source="p_source", source_mode=2)),  # 10 800 rows; k-space corrected source (EPI_STORE tail)
    ((64, 100, 300), {}),                                   # power-of-two x lines over a 4 * 25 by 4 * 75 plane
    ((240, 100, 100), {}),                                  # 24-row x tiles (12 line pairs): 416 tiles + 16 rows
    ((300, 108, 108), {}),                                  # 20-row x tiles (10 line pairs): 583 tiles + 4 rows
    ((160, 100, 108), dict(u_non_staggered_raw=1)),         # 16-row x tiles (8 line pairs): 675 tiles, no partial one
])
def test_grids_without_whole_x_tiles_stay_on_the_fused_pipeline(orc, syn, dims, opts):
    """The x kernels take tiles of 16 to 32 rows (per line length: nl_x in kw_fused.hip); a row count Ny * Nz that is no
    whole number of tiles ends in one masked tile (TAIL kernels) instead of sending the whole grid to the rocFFT path."""
    opts = dict(opts)
    source, mode = opts.pop("source", "p0"), opts.pop("source_mode", 0)
    ny, nz = dims[1], dims[2]
    pr = syn.make_problem(*dims, heterogeneous=True, nonlinear=True, absorbing=True, source=source, source_mode=mode,
                          pml_size=4 if min(dims) >= 20 else 2, sensor="random", nt=12)
    assert (ny * nz) % 16 in (0, 4)  # (the fused line lengths are multiples of 4)
    g, o = make_gpu(pr, fused_kernels=True, p_raw=1, **opts), orc.OracleSim(pr)
    g.run(10)
    assert g.scalar("fused_pipeline") == 1.0
    o.step(10)
    for f in ("p", "ux", "uz", "rhoy"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    if "u_non_staggered_raw" in opts:
        assert rel_l2(g.field("ux_shifted"), orc.shifted_velocity(g.field("ux"), pr["x_shift_neg_r"], 0)) < TOL
    g.close()
    o.close()


def test_transducer_source(orc, syn):
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=True, absorbing=True, source="transducer", nt=40,
                          pml_size=4)
    errs = compare(orc, pr, 40, fields=("p", "ux"))
    assert max(errs.values()) < TOL, errs


def test_k1_closed_form_on_gpu(syn):
    """Config 1 on the GPU against the fp64 closed form: p(n dt) = Fi{cos(c|k| n dt) F{p0}}."""
    from oracle.kwave_np import closed_form_pressure
    pr = syn.make_problem(64, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0")
    g = make_gpu(pr)
    g.run(1)
    assert np.array_equal(g.field("p"), pr["p0_source_input"])  # sample 0 == p0 bit-exactly
    g.run(100)
    assert rel_l2(g.field("p"), closed_form_pressure(pr, 100)) < TOL
    g.close()


def test_k8_absorbing_mode_recurrence_on_gpu(syn):
    """K8 on the GPU: the homogeneous absorbing linear medium against the fp64 per-mode recurrence (oracle-free pin of
    kappa / nabla1 / nabla2 / tau / eta and of the absorbing pressure sum), fused pipeline and rocFFT path."""
    from oracle.kwave_np import absorbing_mode_recurrence
    pr = syn.make_problem(64, heterogeneous=False, nonlinear=False, absorbing=True, pml_off=True, source="p0", nt=110)
    p, ux = absorbing_mode_recurrence(pr, 100)
    for fused in (True, False):
        g = make_gpu(pr, fused_kernels=fused)
        g.run(101)
        assert rel_l2(g.field("p"), p) < TOL and rel_l2(g.field("ux"), ux) < TOL, fused
        g.close()


def test_sensor_streams_bit_exact_and_delayed_flush(orc, syn):
    """p_raw / p_max / p_min / p_rms streams: raw samples equal the field at the mask bit-for-bit; aggregates equal the
    oracle's reduce operators applied to the GPU's own raw series bit-for-bit."""
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=20, sensor="random")
    g = make_gpu(pr, p_raw=1, p_max=1, p_min=1, p_rms=1, u_raw=1, p_max_all=1, sampling_start=3)
    mask = pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    snaps = []
    for t in range(20):
        g.run(1)
        if t >= 3:
            snaps.append(g.field("p").reshape(-1)[mask].copy())
    g.finish()
    raw = g.stream("p")
    assert raw.shape == (17, mask.size)
    assert np.array_equal(raw, np.array(snaps))
    mx = np.full(mask.size, -np.finfo(np.float32).max, dtype=np.float32)
    mn = np.full(mask.size, np.finfo(np.float32).max, dtype=np.float32)
    rms = np.zeros(mask.size, dtype=np.float32)
    ident = np.arange(mask.size, dtype=np.uint64)
    for row in raw:
        orc.sample_index(orc.OP_MAX, mx, row, ident)
        orc.sample_index(orc.OP_MIN, mn, row, ident)
        orc.sample_index(orc.OP_RMS, rms, row, ident)
    orc.post_rms(rms, 1.0 / (20 - 3))
    assert np.array_equal(g.stream("p_max"), mx)
    assert np.array_equal(g.stream("p_min"), mn)
    assert np.array_equal(g.stream("p_rms"), rms)
    assert g.stream("ux").shape == (17, mask.size)
    # whole-domain max is >= sensor max everywhere it overlaps
    assert np.all(g.stream("p_max_all")[mask] == mx)
    g.close()


def test_config2_128_heterogeneous(orc, syn):
    """BASELINE config 2: 128^3 heterogeneous; linear lossless and nonlinear absorbing, vs the CPU oracle."""
    for kw in (dict(nonlinear=False, absorbing=False), dict(nonlinear=True, absorbing=True)):
        pr = syn.make_problem(128, heterogeneous=True, source="p0", **kw)
        errs = compare(orc, pr, 30, fields=("p",))
        assert errs["p"] < TOL, (kw, errs)


def test_runs_stop_at_nt(syn):
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", nt=5, pml_size=2)
    g = make_gpu(pr)
    g.run(50)
    assert g.t == 5
    g.close()


def test_config5_compression_and_intensity_streams(orc, syn):
    """BASELINE config 5 at test size: p_c / u_non_staggered_c / I_avg_c streams (CompressHelper basis + correlation,
    IndexOutputStream.cpp:299-470) against the oracle's compressor fed with the GPU's own raw series, and the
    non-staggered velocity against the oracle's shifted-velocity restatement."""
    n, nt = 32, 130
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    period, mos, harm = 1.0 / (1.0e6 * dt), 1, 2
    g = make_gpu(pr, p_raw=1, u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, period=period, mos=mos,
                 harmonics=harm)
    g.run(nt)
    g.finish()
    mask = pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    # non-staggered velocity of the last step (KSpaceFirstOrderSolver.cpp:2714-2735)
    for axis, nm in enumerate(("ux", "uy", "uz")):
        shifted = orc.shifted_velocity(g.field(nm), pr["xyz"[axis] + "_shift_neg_r"], axis)
        assert rel_l2(g.field(nm + "_shifted"), shifted) < 1e-5
        assert np.array_equal(g.stream(nm + "_non_staggered")[-1], g.field(nm + "_shifted").reshape(-1)[mask])
    # compression frames
    raw = {"p_c": (g.stream("p"), False)}
    for nm in ("ux", "uy", "uz"):
        raw[nm + "_non_staggered_c"] = (g.stream(nm + "_non_staggered"), True)
    frames = {}
    for name, (series, shifted) in raw.items():
        comp = orc.Compressor(mask.size, period, mos, harm, shifted)
        for row in series:
            comp.step(row)
        got = g.stream(name).reshape(-1, mask.size, harm, 2)
        ref = np.array(comp.frames)
        assert got.shape == ref.shape and got.shape[0] == nt // int(period * mos)
        assert rel_l2(got, ref) < 2e-6, name
        frames[name] = got[..., 0] + 1j * got[..., 1]
    # time-averaged intensity from the coefficients: mean over frames of sum_h Re(P conj(U)) / 2
    for nm in ("x", "y", "z"):
        P, U = frames["p_c"], frames[f"u{nm}_non_staggered_c"]
        ref = (np.real(P * np.conj(U)).sum(axis=2) / 2.0).mean(axis=0)
        assert rel_l2(g.stream(f"I{nm}_avg_c"), ref) < 1e-5
    g.close()


def test_u_c_streams_and_frequency_option(orc, syn):
    """--u_c: the staggered velocities on the unshifted basis (OutputStreamContainer.cpp:133-142); --frequency f is
    --period 1 / (f dt) (Parameters.cpp:473-477) and excludes --period."""
    from kwave_amd import capi
    n, nt = 32, 100
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    period, harm = 1.0 / (1.0e6 * dt), 2
    g = make_gpu(pr, u_raw=1, u_c=1, p_c=1, frequency=1.0e6, mos=1, harmonics=harm)
    g.run(nt)
    g.finish()
    nsens = pr["sensor_mask_index"].size
    for nm in ("ux", "uy", "uz"):
        comp = orc.Compressor(nsens, period, 1, harm, False)
        for row in g.stream(nm):
            comp.step(row)
        got = g.stream(nm + "_c").reshape(-1, nsens, harm, 2)
        assert got.shape[0] == len(comp.frames) > 0
        assert rel_l2(got, np.array(comp.frames)) < 2e-6, nm
    pc = g.stream("p_c").copy()
    g.close()
    g = make_gpu(pr, p_c=1, period=period, mos=1, harmonics=harm)
    g.run(nt)
    g.finish()
    assert rel_l2(g.stream("p_c"), pc) < 1e-6   # same period up to the rounding of 1 / (f dt)
    g.close()
    with pytest.raises(capi.KWaveError):
        make_gpu(pr, p_c=1, period=period, frequency=1.0e6)


def test_compression_period_found_from_the_source_signal(syn):
    """No --period: the period comes from the pressure source signal (Parameters.cpp:488-512, CompressHelper::findPeriod);
    the run equals one given that period explicitly."""
    import ctypes as C
    from kwave_amd import solver
    n, nt = 32, 90
    pr = syn.make_problem(n, heterogeneous=False, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    sig = np.asarray(pr["p_source_input"], dtype=np.float32)
    nsrc = pr["p_source_index"].size if int(np.asarray(pr.get("p_source_many", 0)).ravel()[0]) else 1
    series = sig.reshape(-1, nsrc)[:, nsrc // 2]
    tail = np.ascontiguousarray(series[-min(500, series.size):])
    L = solver.load_host()
    L.kwh_find_period.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float)]
    per = C.c_float()
    assert L.kwh_find_period(tail.ctypes.data, tail.size, C.byref(per)) == 0
    dt = float(pr["dt"].ravel()[0])
    assert abs(per.value - 1.0 / (1.0e6 * dt)) < 0.02 / (1.0e6 * dt)  # the generator's 1 MHz tone
    auto = make_gpu(pr, p_c=1, harmonics=1)
    auto.run(nt)
    auto.finish()
    given = make_gpu(pr, p_c=1, harmonics=1, period=per.value)
    given.run(nt)
    given.finish()
    assert auto.stream("p_c").size > 0 and np.array_equal(auto.stream("p_c"), given.stream("p_c"))
    auto.close()
    given.close()


def test_cuboid_sensor_mask_streams(orc, syn):
    """sensor_mask_type = 1 (corners): two cuboids, concatenated in cuboid order, each x-fastest (CuboidOutputStream.cpp:263-345;
    dataset layout sensor_mask_corners = (6, nCuboids, 1), 1-based inclusive corners)."""
    n, nt = 32, 14
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=nt + 4, pml_size=4)
    pr = {k: v for k, v in pr.items() if k != "sensor_mask_index"}
    corners = np.array([[3, 4, 5, 10, 9, 7], [12, 2, 20, 12, 2, 20]], dtype=np.uint64)  # a box and a single voxel
    pr["sensor_mask_type"] = np.array([[[1]]], dtype=np.uint64)
    pr["sensor_mask_corners"] = corners.reshape(1, 2, 6)
    g = make_gpu(pr, p_raw=1, p_max=1, u_rms=1)
    g.run(nt)
    g.finish()

    def gather(field):
        out = []
        for x0, y0, z0, x1, y1, z1 in corners.astype(int):
            out.append(field[z0 - 1:z1, y0 - 1:y1, x0 - 1:x1].reshape(-1))
        return np.concatenate(out)

    raw = g.stream("p")
    npts = 8 * 6 * 3 + 1
    assert raw.shape == (nt, npts)
    assert np.array_equal(raw[-1], gather(g.field("p")))          # bit-exact sampling of the device field
    assert np.array_equal(g.stream("p_max"), raw.max(axis=0))
    o = orc.OracleSim(pr)
    acc = np.zeros(npts, dtype=np.float64)
    for step in range(nt):
        o.step()
        # the cuboids sit in the quiet part of the field: judge the error against the field's own scale
        assert np.abs(raw[step] - gather(o.field("p"))).max() < TOL * np.abs(o.field("p")).max(), step
        acc += gather(o.field("ux")).astype(np.float64) ** 2
    # RMS is scaled by 1 / (Nt - sampling start) of the input file, not by the steps actually run (BaseOutputStream.cpp:172-178)
    want = np.sqrt(acc / int(pr["Nt"].ravel()[0]))
    assert np.abs(g.stream("ux_rms") - want).max() < 1e-4 * want.max()
    g.close()
    o.close()


def test_step_graph_replay_is_bit_identical(syn):
    """kwh_options::step_graph: the steady-state step replayed from a recorded graph (kw_graph_*) gives the same bits as eager
    launches, including across a source that stops mid-run (eager while it is active, graph afterwards)."""
    for kw in (dict(source="p0"), dict(source="u_source", source_mode=1, nt_src=6)):
        pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, nt=30, pml_size=4,
                              sensor="random", **kw)
        a = make_gpu(pr, p_raw=1, p_max=1, step_graph=0)
        a.run(7)
        a.run(17)
        a.finish()
        pa, sa, ma = a.field("p"), a.stream("p"), a.stream("p_max")
        a.close()
        b = make_gpu(pr, p_raw=1, p_max=1, step_graph=1)
        b.run(7)
        b.run(17)
        b.finish()
        assert np.array_equal(b.field("p"), pa) and np.array_equal(b.stream("p"), sa) and np.array_equal(b.stream("p_max"), ma)
        b.close()


@pytest.mark.parametrize("no_overlap", [0, 1])
def test_40bit_complex_compression_streams(orc, syn, no_overlap):
    """--40-bit_complex (IndexOutputStream.cpp:410-436; codec CompressHelper.cpp:224-389): the accumulators are decoded,
    updated and re-encoded as 5-byte complex numbers at every sampled step, and the frames are stored that way (dataset
    width ceil(1.25 Nsens) * harmonics floats).  Emulated on the host with the codec that is pinned bit-for-bit to the
    compiled reference (tests/test_compress_host.py), fed with the GPU's own raw series."""
    import ctypes as C
    from kwave_amd import solver
    L = solver.load_host()
    L.kwh_pack_complex_40b.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32]
    L.kwh_unpack_complex_40b.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32]
    n, nt, harm = 32, 70, 2
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    period = 1.0 / (1.0e6 * dt) / 2.0
    g = make_gpu(pr, p_raw=1, u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, period=period, mos=1,
                 harmonics=harm, complex_40bit=1, no_overlap=no_overlap)
    g.run(nt)
    g.finish()
    nsens = pr["sensor_mask_index"].size
    width = int(np.ceil(nsens * 1.25)) * harm  # floats per stored frame

    def unpack(buf, e):
        out = np.zeros((nsens * harm, 2), dtype=np.float32)
        assert L.kwh_unpack_complex_40b(buf.ctypes.data, nsens * harm, out.ctypes.data, e) == 0
        return out

    def pack(vals, buf, e):
        assert L.kwh_pack_complex_40b(np.ascontiguousarray(vals, dtype=np.float32).ctypes.data, nsens * harm, buf.ctypes.data, e) == 0

    def emulate(series, shifted, e):
        bE, bE1 = orc.compress_basis(period, 1, harm, shifted)  # [harm][bSize][2]
        o_size, b_size = int(orc.lib().kwo_compress_osize(period, 1)), int(orc.lib().kwo_compress_bsize(period, 1))
        c1 = np.zeros(width * 4, dtype=np.uint8)
        c2 = c1 if no_overlap else np.zeros(width * 4, dtype=np.uint8)
        frames, compressed = [], 0
        for s, row in enumerate(series):
            local = s % (b_size - 1)
            save = (local + 1) % o_size == 0
            odd = (compressed + 1) % 2 == 0
            mirror = compressed == 0 and save and not no_overlap
            x = np.repeat(row.astype(np.float32), harm)[:, None]                     # [point*harm][1]
            b0 = np.tile(bE[:, local, :], (nsens, 1)).astype(np.float32)             # [point*harm][2]
            b1 = np.tile(bE1[:, local, :], (nsens, 1)).astype(np.float32)
            cc1 = unpack(c1, e)
            if no_overlap:
                cc1 = cc1 + (b0 * x + b1 * x)
                pack(cc1, c1, e)
            else:
                cc2 = unpack(c2, e)
                cc1 = cc1 + b0 * x
                cc2 = cc2 + b1 * x
                pack(cc1, c1, e)
                if mirror:
                    cc2 = cc2 + cc1
                pack(cc2, c2, e)
            if save:
                cur = c1 if odd else c2
                frames.append(cur.copy())
                cur[:] = 0
                compressed += 1
        return np.array(frames)

    for name, raw, shifted, e in (("p_c", "p", False, 138), ("ux_non_staggered_c", "ux_non_staggered", True, 114)):
        got = g.stream(name)
        assert got.shape[1] == width and got.shape[0] == nt // int(period)
        ref = emulate(g.stream(raw), shifted, e)
        got_b = np.ascontiguousarray(got).view(np.uint8).reshape(got.shape[0], -1)
        # decoded values agree to the 17-bit mantissa (a fused multiply-add on the device may move a sum by one fp32 ulp
        # and thereby a 40-bit code by one step); most codes are identical
        same = (got_b == ref).mean()
        assert same > 0.97, (name, same)
        for f in range(got.shape[0]):
            a, b = unpack(got_b[f], e), unpack(ref[f], e)
            assert np.abs(a - b).max() <= 4e-5 * np.abs(b).max(), (name, f)
    # the intensity built from the 40-bit frames: mean over frames of sum_h Re(P conj(U)) / 2 on the decoded frames
    P = [unpack(np.ascontiguousarray(fr).view(np.uint8), 138) for fr in g.stream("p_c")]
    U = [unpack(np.ascontiguousarray(fr).view(np.uint8), 114) for fr in g.stream("ux_non_staggered_c")]
    ref_i = np.mean([((p[:, 0] * u[:, 0] + p[:, 1] * u[:, 1]) / 2.0).reshape(nsens, harm).sum(axis=1) for p, u in zip(P, U)], axis=0)
    assert rel_l2(g.stream("Ix_avg_c"), ref_i) < 1e-5
    g.close()
