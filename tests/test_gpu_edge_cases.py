"""Edge cases of the time loop and the sampling streams: shortest runs, a single sensor / source point, a sampling
start at the last step, zero-length legs, finishing twice, runs that ask for more steps than Nt, thin grids."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


def one_point(pr, name, flat_index):
    out = dict(pr)
    out[name] = np.array([flat_index + 1], dtype=np.uint64).reshape(1, 1, 1)
    return out


@pytest.mark.parametrize("fused", [True, False])
def test_single_time_step_and_single_sensor_point(orc, syn, fused):
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=1, pml_size=4)
    pr = one_point(pr, "sensor_mask_index", 16 * 32 * 32 + 16 * 32 + 16)
    g = gpu(pr, fused_kernels=fused, p_raw=1, p_max=1, p_rms=1, u_raw=1, p_final=1)
    g.run(5)                       # stops at Nt = 1
    assert g.t == 1
    g.finish()
    o = orc.OracleSim(pr)
    o.step(1)
    assert rel_l2(g.field("p"), o.field("p")) < TOL
    p = g.stream("p").reshape(1, -1)
    assert p.shape == (1, 1) and p[0, 0] == g.field("p").reshape(-1)[16 * 32 * 32 + 16 * 32 + 16]
    assert g.stream("p_max").reshape(-1)[0] == p[0, 0]
    assert g.stream("p_rms").reshape(-1)[0] == pytest.approx(abs(p[0, 0]), rel=1e-6)
    g.finish()                     # a second finish is harmless
    g.close()
    o.close()


def test_sampling_starts_at_the_last_step_and_zero_length_legs(orc, syn):
    nt = 12
    pr = syn.make_problem(32, 16, 16, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", nt=nt,
                          pml_size=4, sensor="random")
    g = gpu(pr, p_raw=1, p_min=1, u_max=1, sampling_start=nt - 1)
    g.run(0)
    assert g.t == 0
    g.run(7)
    g.run(0)
    g.run(nt)                      # asks for more than is left
    assert g.t == nt
    g.finish()
    mask = pr["sensor_mask_index"].reshape(-1).astype(np.int64) - 1
    last = g.field("p").reshape(-1)[mask]
    assert g.stream("p").size == mask.size and np.array_equal(g.stream("p").reshape(-1), last)
    assert np.array_equal(g.stream("p_min").reshape(-1), last)
    assert np.array_equal(g.stream("ux_max").reshape(-1), g.field("ux").reshape(-1)[mask])
    o = orc.OracleSim(pr)
    o.step(nt)
    assert rel_l2(g.field("p"), o.field("p")) < TOL
    g.close()
    o.close()


def test_sampling_start_at_the_end_stores_nothing_and_beyond_is_refused(syn):
    """-s Nt + 1 (0-based index Nt) is the last start the reference accepts (Parameters.cpp:135-139: index > Nt is
    "out of the simulation time span"); nothing is sampled then.  Anything later — or -s 0, which wraps around — is an error."""
    from kwave_amd import capi
    nt = 6
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", nt=nt, pml_size=2,
                          sensor="random")
    g = gpu(pr, p_raw=1, p_max=1, i_avg=1, sampling_start=nt)
    g.run(nt)
    g.finish()
    assert g.stream("p").size == 0
    assert not g.stream("Ix_avg").any()
    g.close()
    for bad in (nt + 1, 2 ** 64 - 1):
        with pytest.raises(capi.KWaveError, match="beginning of data sampling"):
            gpu(pr, p_raw=1, sampling_start=bad)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_single_source_point(orc, syn, mode):
    nt = 30
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=False, absorbing=True, source="p_source", source_mode=mode,
                          source_many=0, nt=nt, pml_size=4, sensor="random")
    pr = one_point(pr, "p_source_index", 10 * 32 * 32 + 11 * 32 + 12)
    pr["p_source_input"] = np.ascontiguousarray(pr["p_source_input"].reshape(-1)[:nt]).reshape(1, nt, 1)
    g, o = gpu(pr), orc.OracleSim(pr)
    g.run(nt)
    o.step(nt)
    assert np.abs(o.field("p")).max() > 0
    for f in ("p", "ux", "rhoz"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    g.close()
    o.close()


@pytest.mark.parametrize("dims", [(256, 16, 16), (16, 256, 16), (16, 16, 256), (64, 16, 128)])
def test_thin_grids_on_the_fused_pipeline(orc, syn, dims):
    pr = syn.make_problem(*dims, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=16, pml_size=4,
                          sensor="random")
    g, o = gpu(pr, fused_kernels=True, p_raw=1), orc.OracleSim(pr)
    g.run(16)
    g.finish()
    assert g.scalar("fused_pipeline") == 1.0
    o.step(16)
    for f in ("p", "uy", "rhox"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, (dims, f)
    g.close()
    o.close()


def test_bad_inputs_are_reported_not_computed(syn):
    from kwave_amd import capi
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", nt=4, pml_size=2)
    bad = {k: v for k, v in pr.items() if k != "dt"}
    with pytest.raises(capi.KWaveError):
        gpu(bad)
    bad = dict(pr)
    bad["c0"] = np.ones((3, 3, 3), dtype=np.float32)          # neither scalar nor grid-sized
    with pytest.raises(capi.KWaveError):
        gpu(bad)
    bad = dict(pr)
    bad["sensor_mask_index"] = np.array([16 ** 3 + 5], dtype=np.uint64).reshape(1, 1, 1)   # outside the grid
    with pytest.raises(capi.KWaveError):
        gpu(bad, p_raw=1).run(1)       # indices are checked when the run is prepared
    bad = dict(pr)
    bad["sensor_mask_type"] = np.array([[[1]]], dtype=np.uint64)
    del bad["sensor_mask_index"]
    bad["sensor_mask_corners"] = np.array([[3, 3, 3, 17, 4, 4]], dtype=np.uint64).reshape(1, 1, 6)   # x end beyond Nx = 16
    with pytest.raises(capi.KWaveError):
        gpu(bad, p_raw=1).run(1)
    with pytest.raises(TypeError):
        gpu(pr, no_such_option=1)


@pytest.mark.parametrize("dims", [(32, 32, 32), (64, 64, 64), (64, 64, 16), (32, 32, 48)])
@pytest.mark.parametrize("medium", ["111", "100", "001"])
def test_whole_plane_kernels_give_the_bits_of_the_three_launch_form(syn, dims, medium):
    """Grids with square planes of 32 / 64 points run a stage's tail as ONE launch whose blocks take whole z-planes and
    do the y transforms themselves (k_xinv PLANE): same small DFTs, twiddles and operation order as the separate y-passes,
    so every field must come out bit-identical to kw_tuning::plane_kernels = 0 — heterogeneous absorbing nonlinear, linear
    lossless (equation of state inside the density kernel) and homogeneous absorbing media, with a p0 and a velocity source."""
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    nx, ny, nz = dims
    het, nonlin, absorb = (c == "1" for c in medium)
    for source, mode in (("p0", 0), ("u_source", 2)):
        pr = syn.make_problem(nx, ny, nz, heterogeneous=het, nonlinear=nonlin, absorbing=absorb, source=source, source_mode=mode,
                              source_many=1 if source != "p0" else 0, nt=16, pml_size=4, sensor="random")
        out = {}
        for plane in (1, 0):
            g = HostSolver(pr, p_raw=1, p_max=1, tuning={"plane_kernels": plane})
            g.run(12)
            g.finish()
            out[plane] = {f: g.field(f) for f in ("p", "ux", "uy", "uz", "rhox", "rhoy", "rhoz")}
            out[plane]["series"] = g.stream("p")
            g.close()
        for f, v in out[1].items():
            assert np.abs(v).max() > 0 and np.array_equal(v, out[0][f]), (f, source)
