"""Randomised differential test of the fused pipeline: random combinations of medium flags, source kinds / modes /
durations (sources that stop mid-run flip the stage-chaining conditions), stream sets and fast-path grids, each
compared with the CPU oracle and with the launch-per-kernel path.  Seeds are fixed: the cases are reproducible."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def gpu(pr, **kw):
    import kwave_amd  # noqa: F401
    from kwave_amd.solver import HostSolver
    return HostSolver(pr, **kw)


@pytest.mark.parametrize("seed", range(44))
def test_random_configuration(orc, syn, seed):
    rng = np.random.default_rng(1000 + seed)
    # seeds 24..35: line lengths with radix-3 / radix-5 stages; 36..43: radix-7 stages, large-factor-first pairs (108) and
    # x tiles of 8 line pairs (100, 108)
    sizes = [16, 32, 64] if seed < 24 else [16, 48, 72, 80, 96] if seed < 36 else [16, 48, 100, 108, 112]
    dims = [int(rng.choice(sizes)) for _ in range(3)]
    if seed % 5 == 0:
        dims[int(rng.integers(3))] = 128 if seed < 24 else 120
    source = str(rng.choice(["p0", "p_source", "u_source", "transducer"]))
    steps = int(rng.integers(12, 26))
    kw = dict(heterogeneous=bool(rng.integers(2)), nonlinear=bool(rng.integers(2)), absorbing=bool(rng.integers(2)),
              source=source, pml_size=4, sensor="random", nt=steps + 2)
    if source in ("p_source", "u_source"):
        kw["source_mode"] = int(rng.integers(3))
        kw["source_many"] = int(rng.integers(2))
        kw["nt_src"] = int(rng.integers(3, steps))      # the source ends before the run does
    if source == "transducer":
        kw["nt_src"] = int(rng.integers(3, steps))
    if kw["heterogeneous"] and rng.integers(2):
        kw["hetero_subset"] = {k: bool(rng.integers(2)) for k in ("c0", "rho0", "BonA", "alpha_coeff")}
    pr = syn.make_problem(*dims, **kw)
    streams = dict(p_raw=1, p_max=int(rng.integers(2)), p_rms=int(rng.integers(2)), u_raw=int(rng.integers(2)),
                   u_max=int(rng.integers(2)), p_min_all=int(rng.integers(2)))
    # run in uneven legs: chaining state must survive (or be rebuilt across) kwh_run boundaries
    legs = [int(x) for x in rng.multinomial(steps, [0.3, 0.5, 0.2]) if x > 0]
    fields = ("p", "ux", "uy", "uz", "rhox", "rhoz")

    def run(fused):
        g = gpu(pr, fused_kernels=fused, **streams)
        for n in legs:
            g.run(n)
        g.finish()
        out = ({f: g.field(f) for f in fields}, {name: g.stream(name) for name in g.stream_names()})
        g.close()
        return out

    fa, sa = run(True)
    fb, sb = run(False)
    o = orc.OracleSim(pr)
    o.step(steps)
    for f in fields:
        ref = o.field(f)
        if not ref.any():
            assert not fa[f].any(), (f, kw)
            continue
        assert rel_l2(fa[f], ref) < TOL, (f, dims, kw)
        assert rel_l2(fa[f], fb[f]) < TOL, (f, dims, kw)
    assert sa.keys() == sb.keys()
    for name in sa:
        assert sa[name].shape == sb[name].shape, name
        assert np.abs(sa[name] - sb[name]).max() <= 2e-5 * max(np.abs(sb[name]).max(), 1e-30), (name, dims, kw)
    o.close()


def test_two_solvers_live_in_one_process(orc, syn):
    """Every solver handle owns its parameter set, device context and compression basis (Parameters::Scope binds them to
    the calling thread per entry point; the reference keeps one set per process, Parameters.h:90-96): two simulations of
    different grids, media and stream sets, stepped in turn, give exactly what each gives alone."""
    pa = syn.make_problem(32, 64, 16, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4,
                          sensor="random", nt=20)
    pb = syn.make_problem(48, 16, 80, heterogeneous=False, nonlinear=False, absorbing=False, source="p_source", nt_src=9,
                          pml_size=4, sensor="random", nt=20)
    ka, kb = dict(p_raw=1, p_max=1), dict(p_raw=1, u_raw=1, p_c=1, period=5.0, harmonics=2)

    def solo(pr, kw):
        g = gpu(pr, **kw)
        g.run(14)
        g.finish()
        out = ({f: g.field(f) for f in ("p", "ux", "rhoz")}, {n: g.stream(n) for n in g.stream_names()})
        g.close()
        return out

    fa, sa = solo(pa, ka)
    fb, sb = solo(pb, kb)
    a, b = gpu(pa, **ka), gpu(pb, **kb)
    for na, nb in ((3, 5), (4, 2), (7, 7)):
        a.run(na)
        b.run(nb)
    b.finish()
    a.finish()
    for f in fa:
        assert np.array_equal(a.field(f), fa[f]), f
        assert np.array_equal(b.field(f), fb[f]), f
    assert set(a.stream_names()) == set(sa) and set(b.stream_names()) == set(sb)
    for n in sa:
        assert np.array_equal(a.stream(n), sa[n]), n
    for n in sb:
        assert np.array_equal(b.stream(n), sb[n]), n
    a.close()
    assert np.array_equal(b.field("p"), fb["p"])  # b is untouched by a's release
    b.close()
    o = orc.OracleSim(pa)
    o.step(14)
    assert rel_l2(fa["p"], o.field("p")) < TOL
    o.close()


def test_two_solvers_on_two_threads(syn):
    """The parameter set is bound per thread and per entry point: two solvers stepped concurrently from two host threads
    (ctypes releases the GIL inside kwh_run) give what each gives alone."""
    import threading
    pa = syn.make_problem(64, 32, 32, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", pml_size=4,
                          sensor="random", nt=40)
    pb = syn.make_problem(32, 48, 16, heterogeneous=True, nonlinear=False, absorbing=True, source="u_source", nt_src=12,
                          pml_size=4, sensor="random", nt=40)

    def solo(pr):
        g = gpu(pr, p_raw=1)
        for _ in range(30):
            g.run(1)
        g.finish()
        out = (g.field("p"), g.stream("p_raw") if "p_raw" in g.stream_names() else g.stream(g.stream_names()[0]))
        g.close()
        return out

    ref = [solo(pa), solo(pb)]
    got, errs = [None, None], []

    def worker(i, pr):
        try:
            got[i] = solo(pr)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(i, pr)) for i, pr in enumerate((pa, pb))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert np.array_equal(got[i][0], ref[i][0]) and np.array_equal(got[i][1], ref[i][1]), i
