"""Non-uniform grids (nonuniform_grid_flag): dt/rho0_sg carries the staggered-grid derivative scaling
(KSpaceFirstOrderSolver.cpp:2650-2685 heterogeneous, SolverCudaKernels.cu:372-410,1061-1083 homogeneous) and the velocity
gradient the regular-grid one (.cu:1285-1301)."""
import numpy as np
import pytest

from conftest import rel_l2

TOL = 1e-5
CASES = [
    dict(heterogeneous=True, nonlinear=True, absorbing=True, source="p0"),
    dict(heterogeneous=False, nonlinear=False, absorbing=False, source="p0"),
    dict(heterogeneous=False, nonlinear=True, absorbing=True, source="u_source", source_mode=1),
    dict(heterogeneous=True, nonlinear=False, absorbing=False, source="p_source", source_mode=2),
]


@pytest.mark.parametrize("kw", CASES)
def test_oracle_nonuniform_c_vs_numpy(orc, syn, kw):
    from oracle.kwave_np import NumpySim
    pr = syn.make_problem(32, 24, 16, nt=30, pml_size=4, sensor="random", nonuniform=True, **kw)
    o, n = orc.OracleSim(pr), NumpySim(pr)
    for _ in range(25):
        o.step()
        n.step()
    assert rel_l2(o.field("p"), n.p) < 5e-6 and rel_l2(o.field("uz"), n.u[2]) < 5e-6
    # the scalings matter: the same problem on the uniform grid is a different field
    uni = dict(pr)
    uni["nonuniform_grid_flag"] = np.array([[[0]]], dtype=np.uint64)
    u = orc.OracleSim(uni)
    u.step(25)
    assert rel_l2(o.field("p"), u.field("p")) > 1e-2
    o.close()
    u.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CASES)
def test_gpu_nonuniform_matches_oracle(orc, syn, kw):
    from kwave_amd.solver import HostSolver
    pr = syn.make_problem(32, 24, 16, nt=30, pml_size=4, sensor="random", nonuniform=True, **kw)
    g = HostSolver(pr, p_raw=1)
    o = orc.OracleSim(pr)
    g.run(25)
    o.step(25)
    for f in ("p", "ux", "uy", "uz", "rhox", "rhoz"):
        assert rel_l2(g.field(f), o.field(f)) < TOL, f
    g.close()
    o.close()


@pytest.mark.gpu
def test_gpu_nonuniform_power_of_two_grid_and_rejections(orc, syn):
    """a grid the fused pipeline would take: non-uniform grids run the launch-per-kernel path and still match"""
    from kwave_amd import capi
    from kwave_amd.solver import HostSolver
    pr = syn.make_problem(32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=30, pml_size=4,
                          nonuniform=True)
    g = HostSolver(pr, fused_kernels=True)
    o = orc.OracleSim(pr)
    g.run(20)
    o.step(20)
    assert rel_l2(g.field("p"), o.field("p")) < TOL and rel_l2(g.field("duxdx"), o.field("duxdx")) < TOL
    g.close()
    o.close()
    pr2 = syn.make_problem(32, 32, 1, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", nt=10,
                           pml_size=4, nonuniform=True)
    with pytest.raises(capi.KWaveError):
        HostSolver(pr2)
