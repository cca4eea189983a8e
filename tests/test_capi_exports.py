"""CPU-side checks of the drop-in boundary: the C-ABI libraries load and export every symbol the headers declare.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    import kwave_amd  # noqa: F401
    from kwave_amd import capi, solver
    return capi, solver


def test_hip_library_exports_every_declared_symbol(built):
    capi, _ = built
    L = capi.load()
    declared = list(capi.declared_symbols())
    assert len(declared) >= 55
    for sym in declared:
        assert hasattr(L, sym), f"include/kwave_hip.h declares {sym} but libkwave_hip.so does not export it"
    # every declared function has a ctypes signature (so tests call with checked argument counts)
    unbound = [s for s in declared if s not in capi._SIG and s not in ("kw_last_error", "kw_get_stream")]
    assert not unbound, unbound


def test_host_library_exports_every_declared_symbol(built):
    _, solver = built
    L = solver.load_host()
    txt = open(os.path.join(ROOT, "include", "kwave_host.h")).read()
    declared = sorted(set(re.findall(r"KWH_API\s+[\w\s\*]+?\b(kwh_\w+)\s*\(", txt)))
    assert len(declared) >= 12
    for sym in declared:
        assert hasattr(L, sym), sym


def test_h5_library_exports_every_declared_symbol(built):
    from kwave_amd import h5io
    if not os.path.exists(h5io.H5_LIB_PATH):
        pytest.skip("HDF5 component not built (no libhdf5)")
    L = h5io.load_h5()
    txt = open(os.path.join(ROOT, "include", "kwave_host_h5.h")).read()
    declared = sorted(set(re.findall(r"KWH_API\s+[\w\s\*]+?\b(kwh_\w+)\s*\(", txt)))
    assert len(declared) >= 8
    for sym in declared:
        assert hasattr(L, sym), sym


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "kwave_hip.h")).read()
    # each kernel entry names the SolverCudaKernels / OutputStreamsCudaKernels lines it replaces
    assert txt.count(".cu:") >= 25 and txt.count(".cuh:") >= 25


def test_constants_struct_layout_matches_header(built):
    capi, _ = built
    # 8 u32 + 15 f32 + 6 u32 = 116 bytes, no padding
    assert ctypes.sizeof(capi.Constants) == 116


def test_no_device_fails_loudly(built):
    """Without a gfx950 device the product path must raise — there is no CPU fallback."""
    capi, _ = built
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.KWaveError):
        capi.Device()


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under the product package or include/ may mention it."""
    bad = []
    for base in ("k-wave-fluid-cuda_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    s = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"\boracle\b|kwo_|kwave_oracle", s):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
