"""2-D simulations (Nz == 1; the reference's SD::k2D instantiations): a 2-D input file carries no z
datasets; the arithmetic differs from 3-D in the initial density split (p0 / (2 c^2), SolverCudaKernels.cu:873-876) and
in sources going to rho_x, rho_y only (:588-622, :795-807)."""
import os

import numpy as np
import pytest

from conftest import rel_l2

TOL = 1e-5


def problem2d(syn, nx=48, ny=32, **kw):
    kw.setdefault("nt", 40)
    kw.setdefault("pml_size", 6)
    kw.setdefault("sensor", "random")
    pr = syn.as_2d_file(syn.make_problem(nx, ny, 1, **kw))
    for name in syn.Z_ONLY_DATASETS:
        assert name not in pr
    return pr


# ---- CPU: the oracle's own pins in 2-D ---------------------------------------------------------------------------------
def test_oracle_2d_closed_form(orc, syn):
    """K1 in two dimensions: p(n dt) = Fi{cos(c |k| n dt) F{p0}} for the homogeneous lossless periodic box."""
    from oracle.kwave_np import closed_form_pressure, complete_2d
    pr = problem2d(syn, 64, 64, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0", nt=120)
    o = orc.OracleSim(pr)
    o.step(1)
    assert np.array_equal(o.field("p").reshape(-1), pr["p0_source_input"].reshape(-1))
    # step-0 identity with the 2-D split: (rho_x + rho_y) c^2 == p0, rho_z == 0
    c2 = float(pr["c0"].ravel()[0]) ** 2
    assert rel_l2((o.field("rhox") + o.field("rhoy")) * c2, pr["p0_source_input"]) < 1e-6
    assert not o.field("rhoz").any() and not o.field("uz").any()
    o.step(100)
    assert rel_l2(o.field("p"), closed_form_pressure(complete_2d(pr), 100)) < TOL
    o.close()


@pytest.mark.parametrize("kw", [
    dict(nonlinear=True, absorbing=True, source="p0"),
    dict(nonlinear=False, absorbing=False, source="p_source", source_mode=0, source_many=1),
    dict(nonlinear=True, absorbing=False, source="p_source", source_mode=2),
    dict(nonlinear=False, absorbing=True, source="u_source", source_mode=1),
])
def test_oracle_2d_c_vs_numpy(orc, syn, kw):
    from oracle.kwave_np import NumpySim
    pr = problem2d(syn, heterogeneous=True, **kw)
    o, n = orc.OracleSim(pr), NumpySim(pr)
    for _ in range(25):
        o.step()
        n.step()
    assert rel_l2(o.field("p"), n.p) < 5e-6
    assert rel_l2(o.field("ux"), n.u[0]) < 5e-6 and rel_l2(o.field("rhoy"), n.rho[1]) < 5e-6
    assert not o.field("uz").any() and not o.field("rhoz").any()
    o.close()


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(heterogeneous=True, nonlinear=True, absorbing=True, source="p0"),
    dict(heterogeneous=False, nonlinear=False, absorbing=False, source="p0"),
    dict(heterogeneous=True, nonlinear=False, absorbing=True, source="p_source", source_mode=1, source_many=1),
    dict(heterogeneous=True, nonlinear=True, absorbing=False, source="p_source", source_mode=2),
    dict(heterogeneous=True, nonlinear=True, absorbing=True, source="u_source", source_mode=0),
])
@pytest.mark.parametrize("fused", [True, False])
def test_gpu_2d_matches_oracle(orc, syn, kw, fused):
    """fused: x-pass, fused pass along y (forward y transform, spectral multiply, inverse y transform in one kernel),
    x-inverse + epilogue; not fused: rocFFT + one kernel per reference kernel"""
    from kwave_amd.solver import HostSolver
    pr = problem2d(syn, **kw)
    g = HostSolver(pr, p_raw=1, p_max=1, u_raw=1, u_rms=1, p_final=1, fused_kernels=fused)
    o = orc.OracleSim(pr)
    series = []
    for _ in range(30):
        o.step()
        series.append(o.field("p").reshape(-1)[o.sensor_index].copy())
    g.run(30)
    g.finish()
    for f in ("p", "ux", "uy", "rhox", "rhoy"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    assert not g.field("uz").any() and not g.field("rhoz").any()
    assert rel_l2(g.stream("p"), np.array(series)) < TOL
    # no z-velocity streams in 2-D (OutputStreamContainer.cpp:117-250)
    names = g.stream_names()
    assert "ux" in names and "uy" in names and "uz" not in names and "uz_rms" not in names
    g.close()
    o.close()


@pytest.mark.gpu
def test_gpu_2d_closed_form_and_rejections(syn):
    from oracle.kwave_np import closed_form_pressure, complete_2d
    from kwave_amd import capi
    from kwave_amd.solver import HostSolver
    pr = problem2d(syn, 64, 64, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0", nt=120)
    g = HostSolver(pr)
    g.run(101)
    assert rel_l2(g.field("p"), closed_form_pressure(complete_2d(pr), 100)) < TOL
    g.close()


@pytest.mark.gpu
def test_gpu_2d_non_staggered_compression_and_intensity_streams(orc, syn):
    """2-D runs carry the x / y members of the non-staggered, compression and intensity streams (no z member), and the
    Q term is the 2-D divergence (KSpaceFirstOrderSolver.cpp:2714-2735 k2D, :2014-2026)."""
    from kwave_amd.solver import HostSolver
    nt = 130
    pr = problem2d(syn, 48, 32, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1, nt=nt)
    dt = float(pr["dt"].ravel()[0])
    g = HostSolver(pr, u_non_staggered_raw=1, i_avg=1, q_term=1, i_avg_c=1, q_term_c=1, period=1.0 / (1.0e6 * dt),
                   mos=1, harmonics=2)
    g.run(nt)
    g.finish()
    names = g.stream_names()
    for nm in ("p", "ux_non_staggered", "uy_non_staggered", "Ix_avg", "Iy_avg", "Q_term", "Ix_avg_c", "Iy_avg_c", "Q_term_c"):
        assert nm in names, (nm, names)
    assert not [nm for nm in g.stream_names(include_hidden=True) if "z" in nm]
    mask = pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    for axis, nm in enumerate(("ux", "uy")):
        shifted = orc.shifted_velocity(g.field(nm), pr["xy"[axis] + "_shift_neg_r"], axis)
        assert rel_l2(g.field(nm + "_shifted"), shifted) < TOL
        assert np.array_equal(g.stream(nm + "_non_staggered")[-1], g.field(nm + "_shifted").reshape(-1)[mask])
    p = g.stream("p").reshape(nt, -1)
    dims = (48, 32, 1)
    spacing = tuple(float(pr[k].ravel()[0]) for k in ("dx", "dy", "dx"))  # no dz in a 2-D file; the z term is zero
    for suffix in ("", "_c"):
        inten = [g.stream(f"I{a}_avg{suffix}").reshape(-1) for a in "xy"]
        if suffix == "":
            for a, got in zip("xy", inten):
                ref = orc.average_intensity(p, g.stream(f"u{a}_non_staggered").reshape(nt, -1))
                assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
        ref = orc.q_term(inten[0], inten[1], 0 * inten[0], mask, dims, spacing)
        assert np.abs(ref).max() > 0
        assert np.abs(g.stream("Q_term" + suffix).reshape(-1) - ref).max() < 1e-5 * np.abs(ref).max()
    g.close()


@pytest.mark.gpu
def test_gpu_2d_from_file(syn, tmp_path):
    import kwave_amd  # noqa: F401
    from kwave_amd import h5io
    from kwave_amd.solver import HostSolver
    if not os.path.exists(h5io.H5_LIB_PATH):
        pytest.skip("HDF5 component not built")
    pr = problem2d(syn, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=20)
    path_in, path_out = str(tmp_path / "in2d.h5"), str(tmp_path / "out2d.h5")
    h5io.write_input_file(pr, path_in)
    mem = HostSolver(pr, p_raw=1, u_final=1)
    mem.run(20)
    mem.finish()
    fs = h5io.FileSolver(path_in, p_raw=1, u_final=1)
    fs.run(20)
    fs.finish()
    assert np.array_equal(fs.field("p"), mem.field("p")) and np.array_equal(fs.stream("p"), mem.stream("p"))
    fs.write_output(path_out)
    assert h5io.dataset_info(path_out, "ux_final")[0] == (48, 32, 1)
    with pytest.raises(Exception):
        h5io.dataset_info(path_out, "uz_final")
    fs.close()
    mem.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(256, 512), (500, 240), (100, 72), (1024, 16)])
def test_gpu_2d_fused_pass_on_other_line_lengths(orc, syn, dims):
    """the fused 2-D pipeline on 512-point y lines (2 x 256 split kernels), mixed-radix lengths and a masked last x tile"""
    from kwave_amd.solver import HostSolver
    pr = problem2d(syn, dims[0], dims[1], heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=16)
    g = HostSolver(pr, p_raw=1)
    o = orc.OracleSim(pr)
    o.step(14)
    g.run(14)
    for f in ("p", "ux", "uy", "rhox", "rhoy"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, (f, dims)
    assert not g.field("uz").any() and not g.field("rhoz").any()
    g.close()
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,fused", [((720, 48), True), ((720, 100), False), ((800, 64), True), ((448, 100), True)])
def test_gpu_2d_long_lines_without_masked_kernels(orc, syn, dims, fused):
    """the ten longest lengths of round 3 carry no masked (TAIL) x kernels: a 2-D grid whose Ny rows are no whole number of
    16-row x tiles runs on the rocFFT path there (720 x 100), any other one on the fused pipeline; older lengths keep their
    masked kernels (448 x 100: 6 tiles + 4 rows)"""
    from kwave_amd.solver import HostSolver
    pr = problem2d(syn, dims[0], dims[1], heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=14)
    g = HostSolver(pr, p_raw=1)
    o = orc.OracleSim(pr)
    o.step(12)
    g.run(12)
    assert (g.scalar("fused_pipeline") == 1.0) == fused, dims
    for f in ("p", "ux", "uy", "rhoy"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, (f, dims)
    g.close()
    o.close()

