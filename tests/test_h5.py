"""HDF5 file-format layer: input files the reference could read, output files with the
reference's dataset shapes; the C++ loop driven from a file gives the same bits as from memory."""
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def h5io():
    import __graft_entry__ as ge
    ge.build()
    import kwave_amd  # noqa: F401
    from kwave_amd import h5io as m
    if not os.path.exists(m.H5_LIB_PATH):
        pytest.skip("HDF5 component not built (no HDF5 in this image)")
    return m


def test_input_file_roundtrip_and_attributes(h5io, syn, tmp_path):
    pr = syn.make_problem(16, 12, 8, nt=6, pml_size=2, source="p_source", source_many=1, sensor="random")
    path = str(tmp_path / "in.h5")
    h5io.write_input_file(pr, path)
    assert h5io.read_attribute(path, "/", "file_type") == "input"
    assert h5io.read_attribute(path, "/", "major_version") == "1" and h5io.read_attribute(path, "/", "minor_version") == "1"
    # 3-D real array: dims (x,y,z), float/real
    assert h5io.dataset_info(path, "c0") == ((16, 12, 8), "float", "real")
    # scalars are (1,1,1); flags / sizes are "long"
    assert h5io.dataset_info(path, "Nx") == ((1, 1, 1), "long", "real")
    assert h5io.dataset_info(path, "dt") == ((1, 1, 1), "float", "real")
    # complex operators: interleaved, doubled fastest dimension, domain_type complex (Hdf5File.cpp:898-915)
    assert h5io.dataset_info(path, "ddx_k_shift_pos_r") == ((2 * 9, 1, 1), "float", "complex")
    assert h5io.dataset_info(path, "ddy_k_shift_neg") == ((2, 12, 1), "float", "complex")
    assert h5io.dataset_info(path, "sensor_mask_index")[1] == "long"
    for name, a in pr.items():
        got = h5io.read_dataset(path, name)
        assert got.size == a.size and np.array_equal(got.ravel(), np.asarray(a).ravel()), name


def test_wrong_file_type_is_rejected(h5io, syn, tmp_path):
    """Reader checks like the reference's (KSpaceFirstOrderSolver.cpp:2797-2891): an output file is not an input file."""
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    path = str(tmp_path / "not_input.h5")
    pr = syn.make_problem(16, nt=4, pml_size=2)
    h5io.write_input_file(pr, path)
    with pytest.raises(capi.KWaveError):
        h5io.read_dataset(path, "no_such_dataset")
    with pytest.raises(capi.KWaveError):
        h5io.FileSolver(str(tmp_path / "missing.h5"))


@pytest.mark.gpu
def test_file_driven_run_equals_memory_run_and_output_file(h5io, syn, tmp_path):
    from kwave_amd.solver import HostSolver
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=15, pml_size=4,
                          sensor="random")
    path_in, path_out = str(tmp_path / "in.h5"), str(tmp_path / "out.h5")
    h5io.write_input_file(pr, path_in)
    mem = HostSolver(pr, p_raw=1, p_max=1, p_final=1)
    mem.run(15)
    mem.finish()
    fs = h5io.FileSolver(path_in, p_raw=1, p_max=1, p_final=1)
    fs.run(15)
    fs.finish()
    assert np.array_equal(fs.field("p"), mem.field("p"))
    assert np.array_equal(fs.stream("p"), mem.stream("p"))
    fs.write_output(path_out)
    nsens = pr["sensor_mask_index"].size
    assert h5io.read_attribute(path_out, "/", "file_type") == "output"
    assert h5io.dataset_info(path_out, "p") == ((nsens, 15, 1), "float", "real")      # (Nsens, Nt - s, 1)
    assert h5io.dataset_info(path_out, "p_max") == ((nsens, 1, 1), "float", "real")
    assert h5io.dataset_info(path_out, "p_final") == ((32, 32, 32), "float", "real")
    assert np.array_equal(h5io.read_dataset(path_out, "p").reshape(15, nsens), mem.stream("p"))
    assert np.array_equal(h5io.read_dataset(path_out, "p_final"), mem.field("p"))
    assert int(h5io.read_dataset(path_out, "t_index").ravel()[0]) == 15
    for attr in ("created_by", "creation_date", "file_description", "major_version", "minor_version", "host_names",
                 "number_of_cpu_cores", "total_memory_in_use", "peak_core_memory_in_use", "total_execution_time",
                 "data_loading_phase_execution_time", "pre-processing_phase_execution_time",
                 "simulation_phase_execution_time", "post-processing_phase_execution_time"):
        assert h5io.read_attribute(path_out, "/", attr) != "", attr         # Hdf5FileHeader.cpp:71-87
    assert h5io.read_attribute(path_out, "/", "simulation_phase_execution_time").strip().endswith("s")
    # the simulation flags and the PML description travel from the input to the output file (Parameters.cpp:559-647)
    for name in ("Nx", "Ny", "Nz", "Nt", "dt", "dx", "dy", "dz", "c_ref", "pml_x_size", "pml_y_size", "pml_z_size",
                 "pml_x_alpha", "pml_y_alpha", "pml_z_alpha", "p_source_flag", "p0_source_flag", "transducer_source_flag",
                 "ux_source_flag", "uy_source_flag", "uz_source_flag", "nonuniform_grid_flag", "absorbing_flag",
                 "nonlinear_flag", "alpha_power"):
        want = np.asarray(pr[name]).ravel()[0] if name in pr else 0
        assert h5io.read_dataset(path_out, name).ravel()[0] == want, name
    fs.close()
    mem.close()


@pytest.mark.gpu
def test_command_line_program(h5io, syn, tmp_path):
    """kspaceFirstOrder-HIP -i in.h5 -o out.h5 --p_raw --p_final -s 3 --benchmark 12"""
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    from kwave_amd.solver import HostSolver
    pr = syn.make_problem(32, heterogeneous=False, nonlinear=False, absorbing=True, source="p0", nt=40, pml_size=4)
    path_in, path_out = str(tmp_path / "in.h5"), str(tmp_path / "out.h5")
    h5io.write_input_file(pr, path_in)
    exe = os.path.join(capi.PKG, "lib", "kspaceFirstOrder-HIP")
    r = subprocess.run([exe, "-i", path_in, "-o", path_out, "--p_raw", "--p_final", "-s", "3", "--benchmark", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    mem = HostSolver(pr, p_raw=1, sampling_start=2, benchmark_steps=12)
    mem.run(12)
    mem.finish()
    nsens = pr["sensor_mask_index"].size
    assert h5io.dataset_info(path_out, "p")[0] == (nsens, 10, 1)
    assert np.array_equal(h5io.read_dataset(path_out, "p").reshape(10, nsens), mem.stream("p"))
    assert np.array_equal(h5io.read_dataset(path_out, "p_final"), mem.field("p"))
    mem.close()
    r = subprocess.run([exe, "-i", str(tmp_path / "missing.h5"), "-o", path_out], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode != 0 and "Error" in r.stdout


@pytest.mark.gpu
def test_output_compression_level_and_copied_sensor_mask(h5io, syn, tmp_path):
    """-c <level> and --copy_sensor_mask (KSpaceFirstOrderSolver.cpp:1036-1052): same data, smaller file, mask 1-based"""
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=False, source="p0", nt=12, pml_size=4,
                          sensor="random")
    path_in = str(tmp_path / "in.h5")
    h5io.write_input_file(pr, path_in)
    fs = h5io.FileSolver(path_in, p_raw=1, p_final=1, p_max_all=1)
    fs.run(12)
    fs.finish()
    plain, packed = str(tmp_path / "plain.h5"), str(tmp_path / "packed.h5")
    fs.write_output(plain)
    fs.write_output(packed, compression_level=6, copy_sensor_mask=True)
    fs.close()
    for name in ("p", "p_final", "p_max_all", "t_index"):
        assert np.array_equal(h5io.read_dataset(plain, name), h5io.read_dataset(packed, name)), name
    assert os.path.getsize(packed) < os.path.getsize(plain)
    assert np.array_equal(h5io.read_dataset(packed, "sensor_mask_index").ravel(), pr["sensor_mask_index"].ravel())
    with pytest.raises(capi.KWaveError):
        h5io.read_dataset(plain, "sensor_mask_index")
    # the command line spells them -c / --copy_sensor_mask
    exe = os.path.join(capi.PKG, "lib", "kspaceFirstOrder-HIP")
    cli = str(tmp_path / "cli.h5")
    r = subprocess.run([exe, "-i", path_in, "-o", cli, "--p_raw", "--p_final", "--p_max_all", "-c", "6", "--copy_sensor_mask"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert np.array_equal(h5io.read_dataset(cli, "p"), h5io.read_dataset(plain, "p"))
    assert np.array_equal(h5io.read_dataset(cli, "sensor_mask_index").ravel(), pr["sensor_mask_index"].ravel())


@pytest.mark.gpu
def test_corners_mask_output_is_one_group_per_stream_with_a_dataset_per_cuboid(h5io, syn, tmp_path):
    """sensor_mask_type = corners: "/p/1", "/p/2" ... — (nx, ny, nz, Nt - s) for series, (nx, ny, nz) for aggregates and the
    post-processed intensities (CuboidOutputStream.cpp:95-140, :656-722); whole-domain streams stay plain datasets."""
    nt, start = 14, 3
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=True, source="p0", nt=nt, pml_size=4)
    pr = {k: v for k, v in pr.items() if k != "sensor_mask_index"}
    corners = np.array([[3, 4, 5, 10, 9, 7], [12, 2, 20, 12, 2, 20]], dtype=np.uint64)
    pr["sensor_mask_type"] = np.array([[[1]]], dtype=np.uint64)
    pr["sensor_mask_corners"] = corners.reshape(1, 2, 6)
    path_in, path_out = str(tmp_path / "in.h5"), str(tmp_path / "out.h5")
    h5io.write_input_file(pr, path_in)
    fs = h5io.FileSolver(path_in, p_raw=1, p_max=1, u_rms=1, p_max_all=1, i_avg=1, sampling_start=start)
    fs.run(nt)
    fs.finish()
    fs.write_output(path_out, compression_level=4, copy_sensor_mask=True)
    series, pmax, iavg = fs.stream("p"), fs.stream("p_max").reshape(-1), fs.stream("Ix_avg").reshape(-1)
    offset = 0
    for c, (x0, y0, z0, x1, y1, z1) in enumerate(corners.astype(int), start=1):
        shape = (z1 - z0 + 1, y1 - y0 + 1, x1 - x0 + 1)
        n = int(np.prod(shape))
        assert h5io.dataset_info_4d(path_out, f"p/{c}") == ((shape[2], shape[1], shape[0], nt - start), "float", "real")
        assert h5io.dataset_info_4d(path_out, f"p_max/{c}")[0] == (shape[2], shape[1], shape[0], 0)
        got = h5io.read_dataset(path_out, f"p/{c}")
        assert np.array_equal(got.reshape(nt - start, n), series[:, offset:offset + n])
        assert np.array_equal(h5io.read_dataset(path_out, f"p_max/{c}").reshape(-1), pmax[offset:offset + n])
        assert np.array_equal(h5io.read_dataset(path_out, f"Ix_avg/{c}").reshape(-1), iavg[offset:offset + n])
        assert h5io.dataset_info_4d(path_out, f"ux_non_staggered/{c}")[0][3] == nt - start
        offset += n
    # the first step of cuboid 1 is the field sampled in x-fastest order
    assert h5io.dataset_info(path_out, "p_max_all")[0] == (32, 32, 32)
    assert np.array_equal(h5io.read_dataset(path_out, "sensor_mask_corners").reshape(-1), corners.reshape(-1))
    fs.close()


@pytest.mark.gpu
def test_compression_datasets_carry_their_parameters(h5io, syn, tmp_path):
    """c_harmonics / c_type / c_period / c_mos / c_shift / c_complex_size / c_max_exp (IndexOutputStream.cpp:146-157)"""
    nt = 100
    pr = syn.make_problem(32, heterogeneous=False, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    path_in, path_out = str(tmp_path / "in.h5"), str(tmp_path / "out.h5")
    h5io.write_input_file(pr, path_in)
    fs = h5io.FileSolver(path_in, p_c=1, u_non_staggered_c=1, frequency=1.0e6, mos=1, harmonics=2)
    fs.run(nt)
    fs.finish()
    fs.write_output(path_out)
    frames = fs.stream("p_c").reshape(-1, pr["sensor_mask_index"].size * 2 * 2).shape[0]
    fs.close()
    assert h5io.dataset_info(path_out, "p_c")[0] == (pr["sensor_mask_index"].size * 2 * 2, frames, 1)
    assert h5io.read_attribute(path_out, "p_c", "c_type") == "c"
    period = 1.0 / (1.0e6 * dt)
    for name, shift, e in (("p_c", 0, 138), ("ux_non_staggered_c", 1, 114)):
        assert h5io.read_numeric_attribute(path_out, name, "c_harmonics") == 2
        assert h5io.read_numeric_attribute(path_out, name, "c_mos") == 1
        assert h5io.read_numeric_attribute(path_out, name, "c_shift") == shift
        assert h5io.read_numeric_attribute(path_out, name, "c_max_exp") == e
        assert h5io.read_numeric_attribute(path_out, name, "c_complex_size") == 2.0
        assert h5io.read_numeric_attribute(path_out, name, "c_period") == pytest.approx(period, rel=1e-6)


# ---- independent look at the files (h5dump, not this repo's reader) ------------------------------------------------------
def _h5dump():
    import h5dump_util
    if not h5dump_util.available():
        pytest.skip("h5dump not found")
    return h5dump_util


def test_input_file_as_h5dump_sees_it(h5io, syn, tmp_path):
    """What the reference's reader relies on (Hdf5File.cpp:59-68, 345-352, 767-815, 898-915, 1023-1034;
    Hdf5FileHeader.cpp:62-87), checked with HDF5's own tool: F32LE / U64LE element types, 3-D dataspaces in (z, y, x)
    order, scalars as (1,1,1), complex data with a doubled fastest dimension, fixed-length NUL-terminated ASCII string
    attributes data_type / domain_type on every dataset and the six header attributes on the root group."""
    u = _h5dump()
    pr = syn.make_problem(16, 12, 8, nt=6, pml_size=2, source="p_source", source_many=1, sensor="random")
    path = str(tmp_path / "in.h5")
    h5io.write_input_file(pr, path)
    d = u.describe(path)
    assert u.check_kwave_conventions(d, "input") == []
    ds = d["datasets"]
    assert ds["/c0"]["dims"] == (8, 12, 16) and ds["/c0"]["datatype"] == "H5T_IEEE_F32LE"
    assert ds["/Nx"]["dims"] == (1, 1, 1) and ds["/Nx"]["datatype"] == "H5T_STD_U64LE"
    assert ds["/ddx_k_shift_pos_r"]["dims"] == (1, 1, 18) and ds["/ddx_k_shift_pos_r"]["attrs"]["domain_type"]["value"] == "complex"
    assert ds["/ddz_k_shift_neg"]["dims"] == (8, 1, 2)
    assert ds["/sensor_mask_index"]["attrs"]["data_type"]["value"] == "long"
    assert ds["/p_source_input"]["dims"][0] == 1  # (1, Nt_src, Nsrc): time-major series of a many-series source
    assert set(name.lstrip("/") for name in ds) == set(pr)


@pytest.mark.gpu
def test_output_and_checkpoint_files_as_h5dump_sees_them(h5io, syn, tmp_path):
    """Output file (in-memory and streamed variants) and checkpoint file through h5dump: same conventions; output
    datasets chunked like the reference's — series one time step per chunk (IndexOutputStream.cpp:119-126), grid-sized
    arrays one z-plane per chunk (RealMatrix.cpp:88-121), deflate at the -c level; scalars contiguous."""
    u = _h5dump()
    nt = 12
    pr = syn.make_problem(32, 24, 16, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    nsens = pr["sensor_mask_index"].size
    path_in = str(tmp_path / "in.h5")
    h5io.write_input_file(pr, path_in)
    flags = dict(p_raw=1, p_max=1, p_final=1, u_raw=1, p_max_all=1)
    for streamed in (False, True):
        out = str(tmp_path / f"out{int(streamed)}.h5")
        fs = h5io.FileSolver(path_in, output=out if streamed else None, compression_level=4, **flags)
        fs.run(nt // 2)
        ckpt = str(tmp_path / f"ckpt{int(streamed)}.h5")
        fs.write_checkpoint(ckpt)
        fs.run(nt - nt // 2)
        fs.finish()
        fs.write_output(out, compression_level=4)
        fs.close()
        d = u.describe(out)
        assert u.check_kwave_conventions(d, "output") == [], streamed
        ds = d["datasets"]
        assert ds["/p"]["dims"] == (1, nt, nsens) and ds["/ux"]["dims"] == (1, nt, nsens)
        if streamed:
            assert ds["/p"]["chunk"] == (1, 1, nsens) and ds["/p"]["deflate"] == 4
        assert ds["/p_max"]["dims"] == (1, 1, nsens)
        assert ds["/p_final"]["dims"] == (16, 24, 32) and ds["/p_final"]["chunk"] == (1, 24, 32) and ds["/p_final"]["deflate"] == 4
        assert ds["/p_max_all"]["dims"] == (16, 24, 32)
        assert ds["/Nx"]["dims"] == (1, 1, 1) and ds["/Nx"]["chunk"] is None and ds["/t_index"]["datatype"] == "H5T_STD_U64LE"
        for attr in ("host_names", "number_of_cpu_cores", "total_execution_time", "simulation_phase_execution_time"):
            assert d["attrs"][attr]["string"] and d["attrs"][attr]["nullterm"], attr
        c = u.describe(ckpt)
        assert u.check_kwave_conventions(c, "checkpoint") == []
        for name in ("p", "rhox", "rhoy", "rhoz", "ux_sgx", "uy_sgy", "uz_sgz"):
            assert c["datasets"]["/" + name]["dims"] == (16, 24, 32), name
        assert c["datasets"]["/t_index"]["dims"] == (1, 1, 1)


@pytest.mark.gpu
def test_streamed_output_equals_the_output_written_at_the_end(h5io, syn, tmp_path):
    """Per-step output (IndexOutputStream.cpp:348-372, OutputStreamContainer.cpp:380-403): with the output file open
    from the start every series row is appended as its step is flushed; the finished file holds the same bits as the one
    written from memory at the end — raw and compressed series, aggregates, intensities computed from the re-read series."""
    nt = 40
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=1,
                          nt=nt, pml_size=4, sensor="random")
    dt = float(pr["dt"].ravel()[0])
    path_in = str(tmp_path / "in.h5")
    h5io.write_input_file(pr, path_in)
    flags = dict(p_raw=1, p_max=1, u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, i_avg=1,
                 period=1.0 / (1.0e6 * dt) / 2.0, harmonics=2, sampling_start=3)
    outs = []
    for streamed in (False, True):
        out = str(tmp_path / f"o{int(streamed)}.h5")
        fs = h5io.FileSolver(path_in, output=out if streamed else None, **flags)
        fs.run(nt)
        if streamed:  # a stream can still be read back through the API (from the file)
            series = fs.stream("p")
        fs.finish()
        fs.write_output(out)
        fs.close()
        outs.append(out)
    names = ("p", "p_max", "ux_non_staggered", "uz_non_staggered", "p_c", "uy_non_staggered_c", "Ix_avg_c", "Ix_avg", "Iz_avg")
    for name in names:
        assert h5io.dataset_info(outs[1], name) == h5io.dataset_info(outs[0], name), name
        a, b = h5io.read_dataset(outs[1], name), h5io.read_dataset(outs[0], name)
        assert a.size > 0 and np.array_equal(a, b), name
    assert np.array_equal(series.reshape(-1), h5io.read_dataset(outs[0], "p").reshape(-1)[: series.size])
    assert h5io.read_numeric_attribute(outs[1], "p_c", "c_harmonics") == 2


@pytest.mark.gpu
def test_streamed_output_survives_and_continues_after_a_checkpoint(h5io, syn, tmp_path):
    """A streamed run that stops with a checkpoint leaves the rows sampled so far in the output file (no series in the
    checkpoint); a new process re-opens the file, recovers and completes it — identical to the uninterrupted output.
    Cuboid sensor mask: one 4-D dataset per cuboid, one time step per hyperslab (CuboidOutputStream.cpp:439-470)."""
    nt, split = 20, 9
    pr = syn.make_problem(32, 24, 20, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=nt, pml_size=4)
    corners = np.array([[3, 4, 5, 10, 9, 8], [12, 2, 1, 20, 20, 6]], dtype=np.uint64)  # 1-based x1 y1 z1 x2 y2 z2
    pr.pop("sensor_mask_index")
    pr["sensor_mask_type"] = np.array([[[1]]], dtype=np.uint64)
    pr["sensor_mask_corners"] = corners.reshape(1, 2, 6)
    path_in, whole, legs, ckpt = (str(tmp_path / n) for n in ("in.h5", "whole.h5", "legs.h5", "ckpt.h5"))
    h5io.write_input_file(pr, path_in)
    flags = dict(p_raw=1, p_rms=1, u_raw=1, p_final=1)
    fs = h5io.FileSolver(path_in, output=whole, **flags)
    fs.run(nt)
    fs.finish()
    fs.write_output(whole)
    fs.close()
    a = h5io.FileSolver(path_in, output=legs, **flags)
    a.run(split)
    a.write_checkpoint(ckpt)
    a.close()  # no write_output: the process "dies" here
    assert not h5io.dataset_exists(ckpt, "stream_p")  # the series lives in the output file, not in the checkpoint
    first = h5io.read_dataset(legs, "p/1")
    assert first.shape[0] == nt and np.array_equal(first[:split], h5io.read_dataset(whole, "p/1")[:split])
    assert not first[split:].any()
    b = h5io.FileSolver(path_in, output=legs, reopen_output=True, **flags)
    b.read_checkpoint(ckpt)
    assert b.t == split
    b.run(nt)
    b.finish()
    b.write_output(legs)
    b.close()
    for name in ("p/1", "p/2", "ux/2", "p_rms/1", "p_final"):
        assert np.array_equal(h5io.read_dataset(legs, name), h5io.read_dataset(whole, name)), name
