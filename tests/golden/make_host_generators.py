"""Dumps the operators the host layer *computes* (kappa, nabla1/2, source_kappa, tau, eta, c^2, dt/rho0_sg) for a few
small problems -> tests/golden/host_generators.npz.

Run once on a GPU box (the host layer refuses to start without a device) with the library build whose generator
output is to be frozen:   python tests/golden/make_host_generators.py [LIB_DIR] [OUT]
The committed file was written by the round-1 build (statement-order generators) right before the generators were
rewritten around per-axis tables; tests/test_gpu_solver.py::test_host_generators_are_bit_identical_to_the_frozen_arrays
holds every later build to the same bits.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import capi, solver, synthetic  # noqa: E402

CASES = {
    "abs24": dict(nx=24, ny=20, nz=18, heterogeneous=True, nonlinear=True, absorbing=True, source="p_source", source_mode=2,
                  pml_size=4),
    "abs32": dict(nx=32, heterogeneous=True, nonlinear=False, absorbing=True, source="p0", pml_size=4,
                  hetero_subset={"alpha_coeff": False}),
    "lossless40": dict(nx=40, ny=16, nz=48, heterogeneous=True, nonlinear=True, absorbing=False, source="u_source",
                       source_mode=2, pml_size=4),
}
FIELDS = ("kappa", "nabla1", "nabla2", "source_kappa", "tau", "eta", "c2", "dtrho0sgx", "dtrho0sgy", "dtrho0sgz")


def collect():
    out = {}
    for name, kw in CASES.items():
        kw = dict(kw)
        pr = synthetic.make_problem(kw.pop("nx"), kw.pop("ny", None), kw.pop("nz", None), nt=4, **kw)
        g = solver.HostSolver(pr)
        g.run(1)
        for f in FIELDS:
            try:
                out[f"{name}/{f}"] = g.field(f).copy()
            except capi.KWaveError:
                pass  # this medium has no such operator
        g.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1:
        capi.LIB_PATH = os.path.join(sys.argv[1], "libkwave_hip.so")
        solver.HOST_LIB_PATH = os.path.join(sys.argv[1], "libkwave_host.so")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_generators.npz")
    arrays = collect()
    np.savez_compressed(dst, **arrays)
    print("wrote", dst, sorted(arrays))
