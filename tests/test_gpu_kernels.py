"""GPU parity tests, kernel level, through the C-ABI (include/kwave_hip.h): rocFFT wrapper and sampling kernels
against the CPU oracle.  Sampling must be bit-exact (BASELINE.json); FFT within fp32 round-off."""
import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    d = capi.Device()
    yield d
    d.close()


def set_dims(dev, nx, ny, nz):
    from kwave_amd import capi
    k = capi.Constants()
    k.nx, k.ny, k.nz, k.n_elements = nx, ny, nz, nx * ny * nz
    k.nx_complex, k.ny_complex, k.nz_complex = nx // 2 + 1, ny, nz
    k.n_elements_complex = (nx // 2 + 1) * ny * nz
    k.fft_divider = 1.0 / (nx * ny * nz)
    k.fft_divider_x, k.fft_divider_y, k.fft_divider_z = 1.0 / nx, 1.0 / ny, 1.0 / nz
    dev.set_constants(k)
    return k


@pytest.mark.parametrize("shape", [(32, 32, 32), (16, 12, 10), (18, 20, 24), (64, 32, 16), (15, 9, 21)])
def test_fft_3d_matches_oracle(dev, orc, shape):
    nz, ny, nx = shape
    set_dims(dev, nx, ny, nz)
    dev.call("fft_create_plans_3d")
    a = np.random.default_rng(nx).standard_normal(shape).astype(np.float32)
    d_in, d_out = dev.array(a), dev.empty((nz, ny, nx // 2 + 1, 2))
    dev.call("fft_r2c_3d", d_in, d_out)
    F = d_out.download()
    Fg = F[..., 0] + 1j * F[..., 1]
    Fo = orc.fft_r2c_3d(a)
    assert rel_l2(Fg, Fo) < 1e-6
    d_back = dev.empty(shape)
    dev.call("fft_c2r_3d", d_out, d_back)
    assert rel_l2(d_back.download() / a.size, a) < 1e-6


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_shifted_velocity_matches_oracle(dev, orc, syn, axis):
    nx, ny, nz = 32, 24, 16
    set_dims(dev, nx, ny, nz)
    dev.call("fft_create_plans_1d", axis)
    ops = syn.kspace_operators(nx, ny, nz, 1e-4, 1e-4, 1e-4)
    sh = ops["xyz"[axis] + "_shift_neg_r"]
    u = np.random.default_rng(axis).standard_normal((nz, ny, nx)).astype(np.float32)
    dims = [nx, ny, nz]
    dims[axis] = dims[axis] // 2 + 1
    d_u, d_t, d_o, d_s = dev.array(u), dev.empty((dims[2], dims[1], dims[0], 2)), dev.empty(u.shape), dev.array(sh)
    dev.call("fft_r2c_1d", axis, d_u, d_t)
    dev.call("compute_velocity_shift", axis, d_t, d_s)
    dev.call("fft_c2r_1d", axis, d_t, d_o)
    assert rel_l2(d_o.download(), orc.shifted_velocity(u, sh, axis)) < 1e-6


def hermitian_filter(half, n):
    """Full-length filter of kw_fused_shift_velocity from the reference's half-length shift vector (see kwave_hip.h)."""
    half = np.asarray(half, dtype=np.float32).reshape(-1, 2)
    h = half[:, 0].astype(np.complex64) + 1j * half[:, 1].astype(np.complex64)
    out = np.zeros(n, dtype=np.complex64)
    out[0] = h[0].real
    for k in range(1, (n + 1) // 2):
        out[k], out[n - k] = h[k], np.conj(h[k])
    if n % 2 == 0:
        out[n // 2] = h[n // 2].real
    return (out / np.float32(n)).astype(np.complex64)


@pytest.mark.parametrize("dims", [(32, 48, 16), (64, 16, 96), (256, 32, 16), (16, 256, 32), (32, 16, 512), (80, 72, 120), (108, 300, 216), (500, 16, 648),
                                  (224, 112, 168), (392, 16, 280),
                                  (420, 16, 252), (16, 720, 140), (800, 16, 48), (756, 48, 16)])
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_fused_shift_velocity_matches_oracle(orc, syn, dims, axis):
    """One kernel per axis (two real x-neighbours or rows packed into one complex line) against the oracle's
    R2C -> shift -> C2R restatement, for line lengths of every factorisation family."""
    import kwave_amd  # noqa: F401
    from kwave_amd import capi
    nx, ny, nz = dims
    d = capi.Device()
    set_dims(d, nx, ny, nz)
    d.call("fused_create")
    ops = syn.kspace_operators(nx, ny, nz, 1e-4, 1e-4, 1e-4)
    sh = ops["xyz"[axis] + "_shift_neg_r"]
    u = np.random.default_rng(axis + nx).standard_normal((nz, ny, nx)).astype(np.float32)
    d_u, d_o = d.array(u), d.zeros(u.shape)
    d_h = d.array(hermitian_filter(sh, dims[axis]).view(np.float32))
    d.call("fused_shift_velocity", axis, d_u, d_o, d_h)
    assert rel_l2(d_o.download(), orc.shifted_velocity(u, sh, axis)) < 1e-6, (dims, axis)
    d.call("fused_shift_velocity", axis, d_u, d_u, d_h)          # in place
    assert rel_l2(d_u.download(), orc.shifted_velocity(u, sh, axis)) < 1e-6, (dims, axis)
    d.close()


@pytest.mark.parametrize("op", [0, 1, 2, 3])
def test_sample_index_bit_exact(dev, orc, op):
    rng = np.random.default_rng(op)
    src = rng.standard_normal(64 * 48 * 40).astype(np.float32)
    mask = np.sort(rng.choice(src.size, size=5000, replace=False)).astype(np.uint64)
    init = {0: 0.0, 1: 0.0, 2: -np.finfo(np.float32).max, 3: np.finfo(np.float32).max}[op]
    ref = np.full(mask.size, init, dtype=np.float32)
    d_src, d_mask, d_buf = dev.array(src), dev.array(mask), dev.array(ref.copy())
    for rep in range(3):  # accumulate over three "time steps" with different fields
        s = (src * np.float32(1 + rep)).astype(np.float32)
        d_src.upload(s)
        dev.call("sample_index", op, d_buf, d_src, d_mask, mask.size)
        orc.sample_index(op, ref, s, mask)
    assert np.array_equal(d_buf.download(), ref)


def test_sample_index_multi_equals_separate_launches(dev, orc):
    """raw + rms + max + min of one field over one mask in one launch == four sampleIndex<op> launches, bit for bit"""
    import ctypes as C
    rng = np.random.default_rng(11)
    src = rng.standard_normal(48 * 40 * 32).astype(np.float32)
    mask = np.sort(rng.choice(src.size, size=7001, replace=False)).astype(np.uint64)
    inits = [0.0, 0.0, -np.finfo(np.float32).max, np.finfo(np.float32).max]
    refs = [np.full(mask.size, v, dtype=np.float32) for v in inits]
    d_src, d_mask = dev.array(src), dev.array(mask)
    d_bufs = [dev.array(r.copy()) for r in refs]
    ops = (C.c_int * 4)(0, 1, 2, 3)
    ptrs = (C.c_void_p * 4)(*[b.ptr for b in d_bufs])
    for rep in range(3):
        s = (src * np.float32(1.5 - rep)).astype(np.float32)
        d_src.upload(s)
        dev.call("sample_index_multi", 4, ops, ptrs, d_src, d_mask, mask.size)
        for op in range(4):
            orc.sample_index(op, refs[op], s, mask)
    for op in range(4):
        assert np.array_equal(d_bufs[op].download(), refs[op]), op


def test_sample_index_empty_mask_is_noop(dev):
    d = dev.array(np.ones(8, dtype=np.float32))
    dev.call("sample_index", 0, d, d, None, 0)  # n == 0: must not touch pointers
    assert np.array_equal(d.download(), np.ones(8, dtype=np.float32))


@pytest.mark.parametrize("op", [0, 1, 2, 3])
def test_sample_cuboid_bit_exact(dev, orc, op):
    nx, ny, nz = 40, 36, 28
    rng = np.random.default_rng(10 + op)
    src = rng.standard_normal(nx * ny * nz).astype(np.float32)
    tl, br = np.array([3, 5, 7], dtype=np.uint32), np.array([30, 20, 19], dtype=np.uint32)
    n = int(np.prod(br - tl + 1))
    init = {0: 0.0, 1: 0.0, 2: -np.finfo(np.float32).max, 3: np.finfo(np.float32).max}[op]
    ref = np.full(n, init, dtype=np.float32)
    d_src, d_buf = dev.array(src), dev.array(ref.copy())
    size = np.array([nx, ny, nz], dtype=np.uint32)
    dev.call("sample_cuboid", op, d_buf, d_src, tl.ctypes.data, br.ctypes.data, size.ctypes.data, n)
    orc.sample_cuboid(op, ref, src, tl, br, size)
    assert np.array_equal(d_buf.download(), ref)


@pytest.mark.parametrize("op", [1, 2, 3])
def test_sample_all_and_rms_bit_exact(dev, orc, op):
    rng = np.random.default_rng(20 + op)
    src = rng.standard_normal(100003).astype(np.float32)  # ragged size
    init = {1: 0.0, 2: -np.finfo(np.float32).max, 3: np.finfo(np.float32).max}[op]
    ref = np.full(src.size, init, dtype=np.float32)
    d_src, d_buf = dev.array(src), dev.array(ref.copy())
    for rep in range(2):
        dev.call("sample_all", op, d_buf, d_src, src.size)
        orc.sample_all(op, ref, src)
    if op == 1:
        dev.call("post_processing_rms", d_buf, np.float32(0.5), src.size)
        orc.post_rms(ref, 0.5)
    assert np.array_equal(d_buf.download(), ref)


def test_invalid_arguments_return_errors(dev):
    from kwave_amd import capi
    set_dims(dev, 16, 16, 16)
    with pytest.raises(capi.KWaveError):
        dev.call("compute_pressure_gradient", None, None, None, None, None, None, None)
    with pytest.raises(capi.KWaveError):
        dev.call("sample_index", 7, dev.zeros(4), dev.zeros(4), dev.zeros(4, np.uint64), 4)  # unknown operator
