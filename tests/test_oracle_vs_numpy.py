"""K4: fp32 C oracle vs the independent fp64 NumPy restatement, all four
{linear,nonlinear} x {lossless,absorbing} media, scalar/array mixes and every source mode."""
import numpy as np
import pytest

from conftest import rel_l2
from oracle.kwave_np import NumpySim

TOL = 1e-5  # BASELINE.json: pressure within 1e-5 relative L2


def run_pair(orc, pr, steps):
    o, s = orc.OracleSim(pr), NumpySim(pr)
    for _ in range(steps):
        o.step()
        s.step()
    out = {"p": rel_l2(o.field("p"), s.p), "ux": rel_l2(o.field("ux"), s.u[0]), "uz": rel_l2(o.field("uz"), s.u[2]),
           "rhoy": rel_l2(o.field("rhoy"), s.rho[1])}
    assert np.abs(s.p).max() > 0
    o.close()
    return out


@pytest.mark.parametrize("nonlinear", [False, True])
@pytest.mark.parametrize("absorbing", [False, True])
@pytest.mark.parametrize("heterogeneous", [False, True])
def test_media_p0(orc, syn, nonlinear, absorbing, heterogeneous):
    pr = syn.make_problem(32, heterogeneous=heterogeneous, nonlinear=nonlinear, absorbing=absorbing, source="p0")
    errs = run_pair(orc, pr, 40)
    assert max(errs.values()) < TOL, errs


def test_non_cubic_non_pow2(orc, syn):
    pr = syn.make_problem(24, 20, 18, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4)
    errs = run_pair(orc, pr, 30)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("subset", [
    {"c0": True, "rho0": False, "BonA": False, "alpha_coeff": False},
    {"c0": False, "rho0": True, "BonA": True, "alpha_coeff": False},
    {"c0": False, "rho0": False, "BonA": False, "alpha_coeff": True},
])
def test_mixed_scalar_array_medium(orc, syn, subset):
    pr = syn.make_problem(24, heterogeneous=False, nonlinear=True, absorbing=True, source="p0", hetero_subset=subset,
                          pml_size=4)
    errs = run_pair(orc, pr, 30)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("source", ["p_source", "u_source"])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("many", [0, 1])
def test_time_varying_sources(orc, syn, source, mode, many):
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=False, absorbing=False, source=source, source_mode=mode,
                          source_many=many, nt=40, pml_size=4)
    errs = run_pair(orc, pr, 40)
    assert errs["p"] < TOL and errs["ux"] < TOL, errs


def test_transducer_source(orc, syn):
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=True, absorbing=True, source="transducer", nt=40,
                          pml_size=4)
    errs = run_pair(orc, pr, 40)
    assert errs["p"] < TOL and errs["ux"] < TOL, errs


def test_source_stops_after_flag(orc, syn):
    """Sources are applied only while flag > t (KSpaceFirstOrderSolver.cpp:2258,2314)."""
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=False, source="p_source", nt=10,
                          pml_size=2)
    errs = run_pair(orc, pr, 25)  # 15 steps beyond the end of the signal: must not read past it
    assert errs["p"] < TOL
