"""Post-processing after the last step (KSpaceFirstOrderSolver.cpp:977-1024): time-averaged intensity from the stored
p / u_non_staggered series (--I_avg, :1231-1534) and the volume rate of heat deposition Q = -div(I_avg) (--Q_term,
--Q_term_c, :1783-2080).  CPU part: the numpy oracle against closed forms.  GPU part: the device implementation
against the oracle fed with the run's own stored series (tolerance: 1e-5 of the largest magnitude, fp32 FFTs)."""
import ctypes as C
import os

import numpy as np
import pytest


def err_rel_max(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---- oracle against closed forms (CPU) ------------------------------------------------------------------------------
@pytest.mark.parametrize("steps", [16, 25, 64, 101])
def test_oracle_half_step_shift_of_periodic_series(orc, steps):
    k = np.arange(steps)[:, None]
    m = np.array([1, 2, 3])[None, :]
    phase = np.array([0.3, 1.1, 2.0])[None, :]
    series = np.sin(2 * np.pi * m * k / steps + phase)
    want = np.sin(2 * np.pi * m * (k + 0.5) / steps + phase)
    assert np.abs(orc.time_shift_half_step(series) - want).max() < 1e-12


def test_oracle_average_intensity_of_a_plane_wave(orc):
    """p = cos(w t_k), u sampled half a step earlier with amplitude A and phase lag phi: I = A cos(phi) / 2."""
    steps, A, phi = 48, 0.7, 0.4
    k = np.arange(steps)[:, None]
    w = 2 * np.pi * 3 / steps
    p = np.cos(w * k) * np.ones((1, 5))
    u = A * np.cos(w * (k - 0.5) - phi) * np.ones((1, 5))
    assert np.abs(orc.average_intensity(p, u) - A * np.cos(phi) / 2).max() < 1e-12


def test_oracle_q_term_of_trigonometric_intensity(orc):
    nx, ny, nz = 16, 12, 10
    dx, dy, dz = 1e-3, 2e-3, 1.5e-3
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ix = np.sin(2 * np.pi * 2 * x / nx) * np.cos(2 * np.pi * y / ny)
    iy = np.cos(2 * np.pi * 3 * y / ny + 0.2)
    iz = np.sin(2 * np.pi * z / nz) * np.sin(2 * np.pi * x / nx)
    want = -((2 * np.pi * 2 / (nx * dx)) * np.cos(2 * np.pi * 2 * x / nx) * np.cos(2 * np.pi * y / ny)
             - (2 * np.pi * 3 / (ny * dy)) * np.sin(2 * np.pi * 3 * y / ny + 0.2)
             + (2 * np.pi / (nz * dz)) * np.cos(2 * np.pi * z / nz) * np.sin(2 * np.pi * x / nx))
    idx = np.arange(nx * ny * nz)
    got = orc.q_term(ix.reshape(-1), iy.reshape(-1), iz.reshape(-1), idx, (nx, ny, nz), (dx, dy, dz))
    assert np.abs(got - want.reshape(-1)).max() < 1e-9 * np.abs(want).max()
    # a sparse mask differentiates the zero-filled grid: the gather of the full-grid answer for that grid
    sub = idx[::7]
    full = np.zeros(nx * ny * nz)
    full[sub] = ix.reshape(-1)[sub]
    ref = orc.q_term(full, 0 * full, 0 * full, idx, (nx, ny, nz), (dx, dy, dz))[sub]
    got = orc.q_term(ix.reshape(-1)[sub], 0 * sub, 0 * sub, sub, (nx, ny, nz), (dx, dy, dz))
    assert np.abs(got - ref).max() < 1e-12 * max(np.abs(ref).max(), 1.0)


# ---- device implementation (GPU) ------------------------------------------------------------------------------------
def make_gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("steps,n", [(16, 1000), (37, 513), (128, 70000)])
def test_time_shift_and_intensity_kernels(orc, steps, n):
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    rng = np.random.default_rng(steps)
    u = rng.standard_normal((steps, n)).astype(np.float32)
    p = rng.standard_normal((steps, n)).astype(np.float32)
    shift = np.exp(1j * np.pi * (((np.arange(steps // 2 + 1) + steps // 2) % steps) - steps // 2) / steps).astype(np.complex64)
    dev = capi.Device()
    du, dp, ds = dev.array(u), dev.array(p), dev.array(shift.view(np.float32))
    di = dev.array(np.zeros(n, dtype=np.float32))
    dev.call("time_shift_series", du, ds, steps, n)
    dev.call("intensity_avg", di, dp, du, steps, n)
    assert err_rel_max(du.download(), orc.time_shift_half_step(u)) < 1e-5
    assert err_rel_max(di.download(), orc.average_intensity(p, u)) < 1e-5
    dev.close()


def grid_indices(pr):
    if "sensor_mask_index" in pr:
        return pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    nx, ny = (int(pr[k].ravel()[0]) for k in ("Nx", "Ny"))
    out = []
    for x0, y0, z0, x1, y1, z1 in pr["sensor_mask_corners"].reshape(-1, 6).astype(int):
        z, y, x = np.meshgrid(np.arange(z0 - 1, z1), np.arange(y0 - 1, y1), np.arange(x0 - 1, x1), indexing="ij")
        out.append(((z * ny + y) * nx + x).reshape(-1))
    return np.concatenate(out)


def check_intensity_and_q(orc, pr, g, suffix=""):
    dims = tuple(int(pr[k].ravel()[0]) for k in ("Nx", "Ny", "Nz"))
    spacing = tuple(float(pr[k].ravel()[0]) for k in ("dx", "dy", "dz"))
    inten = [g.stream(f"I{a}_avg{suffix}").reshape(-1) for a in "xyz"]
    ref = orc.q_term(*inten, grid_indices(pr), dims, spacing)
    assert np.abs(ref).max() > 0
    assert err_rel_max(g.stream("Q_term" + suffix), ref) < 1e-5
    return inten


@pytest.mark.gpu
@pytest.mark.parametrize("nt", [40, 33])
def test_average_intensity_and_q_term_index_mask(orc, syn, nt):
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=True, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    g = make_gpu(pr, i_avg=1, q_term=1)
    g.run(nt)
    g.finish()
    names = g.stream_names()
    # the series the intensities come from are stored with them (OutputStreamContainer.cpp:230-246)
    for nm in ("p", "ux_non_staggered", "uy_non_staggered", "uz_non_staggered", "Ix_avg", "Iy_avg", "Iz_avg", "Q_term"):
        assert nm in names, (nm, names)
    p = g.stream("p").reshape(nt, -1)
    for a in "xyz":
        u = g.stream(f"u{a}_non_staggered").reshape(nt, -1)
        ref = orc.average_intensity(p, u)
        assert np.abs(ref).max() > 0
        assert err_rel_max(g.stream(f"I{a}_avg"), ref) < 1e-5
    check_intensity_and_q(orc, pr, g)
    g.close()


@pytest.mark.gpu
def test_q_term_alone_keeps_the_intensities_out_of_the_output(orc, syn):
    nt = 24
    pr = syn.make_problem(32, 16, 48, heterogeneous=False, nonlinear=False, absorbing=False, source="p_source",
                          source_mode=1, nt=nt, pml_size=4, sensor="random")
    g = make_gpu(pr, q_term=1, sampling_start=4)
    g.run(nt)
    g.finish()
    names = g.stream_names()
    assert "Q_term" in names and "Ix_avg" not in names and "p" in names
    assert g.stream("p").size == (nt - 4) * pr["sensor_mask_index"].size
    check_intensity_and_q(orc, pr, g)   # hidden streams can still be read by name
    g.close()


@pytest.mark.gpu
def test_average_intensity_and_q_term_cuboid_mask(orc, syn):
    nt = 20
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=nt, pml_size=4)
    pr = {k: v for k, v in pr.items() if k != "sensor_mask_index"}
    corners = np.array([[3, 4, 5, 20, 9, 17], [12, 2, 20, 12, 2, 20]], dtype=np.uint64)
    pr["sensor_mask_type"] = np.array([[[1]]], dtype=np.uint64)
    pr["sensor_mask_corners"] = corners.reshape(1, 2, 6)
    g = make_gpu(pr, i_avg=1, q_term=1)
    g.run(nt)
    g.finish()
    p = g.stream("p").reshape(nt, -1)
    for a in "xyz":
        ref = orc.average_intensity(p, g.stream(f"u{a}_non_staggered").reshape(nt, -1))
        assert err_rel_max(g.stream(f"I{a}_avg"), ref) < 1e-5
    check_intensity_and_q(orc, pr, g)
    g.close()


@pytest.mark.gpu
def test_q_term_from_compressed_intensity(orc, syn):
    n, nt = 32, 130
    pr = syn.make_problem(n, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    g = make_gpu(pr, q_term_c=1, period=1.0 / (1.0e6 * dt), mos=1, harmonics=2)
    g.run(nt)
    g.finish()
    names = g.stream_names()
    assert "Q_term_c" in names and "Ix_avg_c" not in names and "p_c" not in names
    check_intensity_and_q(orc, pr, g, suffix="_c")
    g.close()
    g = make_gpu(pr, q_term_c=1, i_avg_c=1, p_c=1, period=1.0 / (1.0e6 * dt), mos=1, harmonics=2)
    g.run(nt)
    g.finish()
    names = g.stream_names()
    assert "Q_term_c" in names and "Ix_avg_c" in names and "p_c" in names and "ux_non_staggered_c" not in names
    g.close()


@pytest.mark.gpu
def test_command_line_i_avg_and_q_term(orc, syn, tmp_path):
    """kspaceFirstOrder-HIP ... --I_avg --Q_term: the output file holds p, u_non_staggered (stored because the
    intensities are computed from them), Ix/Iy/Iz_avg and Q_term as (Nsens, 1, 1) datasets."""
    import os
    import subprocess
    import kwave_amd  # noqa: F401
    from kwave_amd import capi, h5io
    if not os.path.exists(h5io.H5_LIB_PATH):
        pytest.skip("HDF5 component not built")
    nt = 30
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=True, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    path_in, path_out = str(tmp_path / "in.h5"), str(tmp_path / "out.h5")
    h5io.write_input_file(pr, path_in)
    exe = os.path.join(capi.PKG, "lib", "kspaceFirstOrder-HIP")
    r = subprocess.run([exe, "-i", path_in, "-o", path_out, "--I_avg", "--Q_term"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    nsens = pr["sensor_mask_index"].size
    assert h5io.dataset_info(path_out, "p")[0] == (nsens, nt, 1)
    assert h5io.dataset_info(path_out, "Q_term")[0] == (nsens, 1, 1)
    p = h5io.read_dataset(path_out, "p").reshape(nt, nsens)
    inten = []
    for a in "xyz":
        u = h5io.read_dataset(path_out, f"u{a}_non_staggered").reshape(nt, nsens)
        got = h5io.read_dataset(path_out, f"I{a}_avg").reshape(-1)
        assert err_rel_max(got, orc.average_intensity(p, u)) < 1e-5
        inten.append(got)
    dims = tuple(int(pr[k].ravel()[0]) for k in ("Nx", "Ny", "Nz"))
    spacing = tuple(float(pr[k].ravel()[0]) for k in ("dx", "dy", "dz"))
    ref = orc.q_term(*inten, grid_indices(pr), dims, spacing)
    assert err_rel_max(h5io.read_dataset(path_out, "Q_term"), ref) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("streamed", [False, True])
def test_only_post_processing_of_an_existing_output_file(syn, tmp_path, streamed):
    """--post (KSpaceFirstOrderSolver.cpp:373-415, :977-1024): a first run stores the raw p / u_non_staggered series and
    the compression coefficients; a second invocation without any time loop computes I_avg, Q_term (from the series) and
    I_avg_c, Q_term_c (from the coefficient frames) out of that file and adds them to it — the same values a single run
    with those flags produces.  Through the C++ entry point and through the command-line program."""
    import subprocess
    import kwave_amd  # noqa: F401
    from kwave_amd import capi, h5io
    if not os.path.exists(h5io.H5_LIB_PATH):
        pytest.skip("HDF5 component not built")
    nt = 60
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    comp = dict(period=1.0 / (1.0e6 * dt) / 2.0, harmonics=2)
    path_in, direct, staged, cli = (str(tmp_path / n) for n in ("in.h5", "direct.h5", "staged.h5", "cli.h5"))
    h5io.write_input_file(pr, path_in)
    post = dict(i_avg=1, q_term=1, i_avg_c=1, q_term_c=1)
    fs = h5io.FileSolver(path_in, **post, **comp)  # the reference result: everything in one run
    fs.run(nt)
    fs.finish()
    fs.write_output(direct)
    fs.close()
    for target in (staged, cli):
        fs = h5io.FileSolver(path_in, output=target if streamed else None, p_raw=1, u_non_staggered_raw=1, p_c=1,
                             u_non_staggered_c=1, **comp)
        fs.run(nt)
        fs.finish()
        fs.write_output(target)
        fs.close()
        assert not h5io.dataset_exists(target, "Ix_avg") and not h5io.dataset_exists(target, "Q_term_c")
    ps = h5io.FileSolver(path_in, only_post_processing=1, **post, **comp)
    ps.post_process(staged)
    ps.close()
    exe = os.path.join(capi.PKG, "lib", "kspaceFirstOrder-HIP")
    r = subprocess.run([exe, "-i", path_in, "-o", cli, "--post", "--I_avg", "--Q_term", "--I_avg_c", "--Q_term_c", "--period",
                        repr(comp["period"]), "--harmonics", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout
    for out in (staged, cli):
        for name in ("Ix_avg", "Iy_avg", "Iz_avg", "Q_term", "Ix_avg_c", "Iz_avg_c", "Q_term_c"):
            a, b = h5io.read_dataset(out, name), h5io.read_dataset(direct, name)
            assert a.shape == b.shape and np.abs(b).max() > 0, name
            assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), name
        assert np.array_equal(h5io.read_dataset(out, "p"), h5io.read_dataset(direct, "p"))  # the series are untouched
    # a second --post replaces the earlier results instead of failing on the existing names
    ps = h5io.FileSolver(path_in, only_post_processing=1, i_avg=1, **comp)
    ps.post_process(staged)
    ps.close()
    assert np.array_equal(h5io.read_dataset(staged, "Ix_avg"), h5io.read_dataset(cli, "Ix_avg"))
