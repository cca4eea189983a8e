"""Host-side compression helpers against known answers of the reference's own CompressHelper.cpp,
captured by tests/golden/make_golden.py from the compiled reference (oracle/_ref) into tests/golden/compress_ref.npz:
the period finder and the 40-bit packing of complex coefficients, bit for bit.  CPU only: these run on the host in the
reference too (Compression/CompressHelper.cpp:146-389)."""
import ctypes as C
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as ge
    ge.build()
    import kwave_amd  # noqa: F401
    from kwave_amd import solver
    L = solver.load_host()
    L.kwh_find_period.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float)]
    L.kwh_pack_complex_40b.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32]
    L.kwh_unpack_complex_40b.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32]
    return L


def find_period(L, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = C.c_float()
    assert L.kwh_find_period(x.ctypes.data, x.size, C.byref(out)) == 0, L.kwh_last_error()
    return out.value


def test_find_period_matches_reference_bit_for_bit(host):
    g = np.load(os.path.join(GOLD, "compress_ref.npz"))
    for per, want in zip(g["find_period_in"], g["find_period_out"]):
        x = np.sin(2 * np.pi * np.arange(400) / per).astype(np.float32)  # the signal make_golden.py fed the reference
        got = find_period(host, x)
        assert np.float32(got) == np.float32(want), (per, got, want)


def test_find_period_properties(host):
    # noise-free tone bursts of other periods; amplitude and offset do not matter; small ripples below half-max are ignored
    t = np.arange(500)
    for per in (6.0, 17.3, 41.0):
        x = 3.0e5 * np.sin(2 * np.pi * t / per + 0.7)
        assert abs(find_period(host, x) - per) < 0.02 * per
        x2 = x + 0.05 * 3.0e5 * np.sin(2 * np.pi * t / (per / 7.3))
        assert abs(find_period(host, x2) - per) < 0.05 * per
    # fewer than two peaks: an error, not a garbage period
    out = C.c_float()
    flat = np.zeros(50, dtype=np.float32)
    assert host.kwh_find_period(flat.ctypes.data, flat.size, C.byref(out)) != 0
    assert b"peaks" in host.kwh_last_error()


@pytest.mark.parametrize("e", [138, 114])
def test_40_bit_codec_matches_reference_bit_for_bit(host, e):
    g = np.load(os.path.join(GOLD, "compress_ref.npz"))
    vin, packed_ref, vout_ref = g[f"codec{e}_in"], g[f"codec{e}_packed"], g[f"codec{e}_out"]
    n = vin.shape[0]
    packed = np.zeros((n, 5), dtype=np.uint8)
    assert host.kwh_pack_complex_40b(np.ascontiguousarray(vin).ctypes.data, n, packed.ctypes.data, e) == 0
    assert np.array_equal(packed, packed_ref)
    back = np.zeros((n, 2), dtype=np.float32)
    assert host.kwh_unpack_complex_40b(packed_ref.ctypes.data, n, back.ctypes.data, e) == 0
    assert np.array_equal(back.view(np.uint32), vout_ref.view(np.uint32))
    # idempotence: what was decoded packs to the same five bytes
    again = np.zeros((n, 5), dtype=np.uint8)
    host.kwh_pack_complex_40b(back.ctypes.data, n, again.ctypes.data, e)
    assert np.array_equal(again, packed_ref)


def test_40_bit_codec_edge_values(host):
    """zero, values below the representable range (flushed towards zero), values above it (saturated), mixed magnitudes"""
    for e, big, tiny in ((138, 1.0e9, 1.0e-3), (114, 100.0, 1.0e-11)):
        vals = np.array([[0.0, 0.0], [big, -big], [tiny, tiny], [1.0, 0.0], [0.0, -1.0], [123.456, -0.001]], dtype=np.float32)
        packed = np.zeros((len(vals), 5), dtype=np.uint8)
        back = np.zeros_like(vals)
        host.kwh_pack_complex_40b(vals.ctypes.data, len(vals), packed.ctypes.data, e)
        host.kwh_unpack_complex_40b(packed.ctypes.data, len(vals), back.ctypes.data, e)
        assert np.all(np.isfinite(back))
        assert np.all(np.sign(back[1]) == np.sign(vals[1])) and np.all(np.abs(back[1]) <= big)   # saturation keeps signs
        assert np.all(np.abs(back[2]) <= 2 * tiny)                                                # never amplified
        if e == 138:
            assert abs(back[5, 0] - 123.456) < 1e-3 * 123.456
    if os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libcompress_ref.so")):
        # build container only: the compiled reference on the same edge values
        R = C.CDLL(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libcompress_ref.so"))
        R.cref_to40b.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int]
        rng = np.random.default_rng(7)
        vals = (rng.standard_normal((2000, 2)) * 10.0 ** rng.uniform(-12, 9, size=(2000, 2))).astype(np.float32)
        vals[::50] = 0.0
        for e in (138, 114):
            mine = np.zeros((len(vals), 5), dtype=np.uint8)
            host.kwh_pack_complex_40b(vals.ctypes.data, len(vals), mine.ctypes.data, e)
            ref = np.zeros(5, dtype=np.uint8)
            for j in range(len(vals)):
                R.cref_to40b(float(vals[j, 0]), float(vals[j, 1]), ref.ctypes.data, e)
                assert np.array_equal(mine[j], ref), (e, vals[j], mine[j], ref)
