"""Known-answer tests that pin the oracle.  The reference ships no tests or
fixtures ("parity unpinned"), so the pins are analytic properties of the scheme it implements."""
import math

import numpy as np
import pytest

from conftest import rel_l2
from oracle.kwave_np import NumpySim, closed_form_pressure


@pytest.fixture(scope="module")
def k1(syn):
    return syn.make_problem(64, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0")


def test_k1_closed_form_config1(orc, k1):
    """BASELINE config 1: 64^3 homogeneous linear lossless, 100 steps.  p(n dt) = Fi{cos(c|k| n dt) F{p0}}."""
    o = orc.OracleSim(k1)
    o.step()
    # step 0: p == p0 bit-exactly (SolverCudaKernels.cu:871)
    assert np.array_equal(o.field("p"), k1["p0_source_input"])
    for n in range(1, 101):
        o.step()
        if n in (1, 10, 100):
            assert rel_l2(o.field("p"), closed_form_pressure(k1, n)) < 1e-5
    o.close()


def test_k8_absorbing_mode_recurrence(orc, syn):
    """K8: homogeneous absorbing linear medium in a periodic box — the scheme closes per Fourier mode on (p^, sum rho^,
    div u^), iterated in fp64 without any of the restatements' FFT / kernel / generator code: pins kappa, nabla1,
    nabla2, tau, eta and the absorbing branch of the pressure sum (SolverCudaKernels.cu:1724-1742, 1812-1820,
    1966-1980; generators KSpaceFirstOrderSolver.cpp:2514-2643).  The absorption itself moves p by 1.6e-2 here."""
    from oracle.kwave_np import absorbing_mode_recurrence
    pr = syn.make_problem(64, heterogeneous=False, nonlinear=False, absorbing=True, pml_off=True, source="p0", nt=110)
    o = orc.OracleSim(pr)
    o.step(1)
    for n in (10, 100):
        o.step(n - (o.t - 1))
        p, ux = absorbing_mode_recurrence(pr, n)
        assert rel_l2(o.field("p"), p) < 1e-5 and rel_l2(o.field("ux"), ux) < 1e-5, n
    assert rel_l2(p, closed_form_pressure(pr, 100)) > 1e-3  # a lossless run would not pass
    s = NumpySim(syn.make_problem(32, heterogeneous=False, nonlinear=False, absorbing=True, pml_off=True, source="p0"))
    for _ in range(41):
        s.step()
    assert rel_l2(s.p, absorbing_mode_recurrence(s.pr, 40)[0]) < 1e-6
    o.close()


def test_k1_numpy_fp64_closed_form(k1, syn):
    pr = syn.make_problem(32, heterogeneous=False, nonlinear=False, absorbing=False, pml_off=True, source="p0")
    s = NumpySim(pr)
    for _ in range(41):
        s.step()
    # limited only by the fp32 storage of dt / operators in the problem
    assert rel_l2(s.p, closed_form_pressure(pr, 40)) < 1e-6


def test_k2_operator_spot_values(orc, syn):
    """kappa / nabla at k=0 and tau/eta formula (KSpaceFirstOrderSolver.cpp:2514-2643)."""
    pr = syn.make_problem(16, heterogeneous=False, nonlinear=False, absorbing=True, source="p0")
    o = orc.OracleSim(pr)
    assert o.field("kappa")[0, 0, 0] == 1.0
    assert o.field("nabla1")[0, 0, 0] == 0.0  # k^(y-2) = inf -> 0
    assert o.field("nabla2")[0, 0, 0] == 0.0  # 0^(0.5) = 0
    y, alpha, c0 = 1.5, 0.75, 1500.0
    a_np = 100.0 * alpha * (1e-6 / (2 * math.pi)) ** y / (20 * math.log10(math.e))
    assert o.scalar("tau") == pytest.approx(-2 * a_np * c0 ** (y - 1), rel=2e-6)
    assert o.scalar("eta") == pytest.approx(2 * a_np * c0 ** y * math.tan(math.pi * y / 2), rel=2e-6)
    # Nyquist corner: k = 2 pi sqrt(3)/(2 dx)
    dx, dt, c_ref = (float(pr[k].ravel()[0]) for k in ("dx", "dt", "c_ref"))
    k = 2 * math.pi * math.sqrt(3 * 0.25 / dx ** 2)
    arg = c_ref * dt / 2 * k
    assert o.field("kappa")[8, 8, 8] == pytest.approx(math.sin(arg) / arg, rel=1e-5)
    assert o.field("nabla1")[8, 8, 8] == pytest.approx(k ** (y - 2), rel=1e-5)
    assert o.field("nabla2")[8, 8, 8] == pytest.approx(k ** (y - 1), rel=1e-5)
    o.close()


def test_k3_p0_step0_identities(orc, syn):
    """After step 0 with a p0 source: sum(rho)*c2 == p0, u == +0.5 dt/rho0 * grad p0
    (KSpaceFirstOrderSolver.cpp:2359-2396)."""
    pr = syn.make_problem(24, heterogeneous=True, nonlinear=False, absorbing=False, source="p0")
    o = orc.OracleSim(pr)
    o.step()
    p0 = pr["p0_source_input"]
    c2 = o.field("c2")
    s = (o.field("rhox") + o.field("rhoy") + o.field("rhoz")) * c2
    assert rel_l2(s, p0) < 1e-6
    # gradient via fp64 numpy
    s64 = NumpySim(pr)
    e = s64.F(p0.astype(np.float64)) * s64.kappa
    gx = s64.Fi(e * s64.ddx_pos) / s64.N
    ux_ref = 0.5 * s64.dtrho[0] * gx
    assert rel_l2(o.field("ux"), ux_ref) < 2e-6
    o.close()


def test_k7_pml_absorbs_energy(orc, syn):
    """Energy sanity: with the PML on, the field decays once the wave has reached the boundary."""
    pr = syn.make_problem(32, heterogeneous=False, nonlinear=False, absorbing=False, source="p0", pml_size=8)
    o = orc.OracleSim(pr)
    o.step(2)
    e0 = float(np.sum(o.field("p").astype(np.float64) ** 2))
    o.step(300)
    e1 = float(np.sum(o.field("p").astype(np.float64) ** 2))
    assert e1 < 0.02 * e0
    assert np.isfinite(e1)
    o.close()
