"""Oracle FFT vs numpy.fft in fp64: pins the cuFFT R2C/C2R contract the oracle restates
(/root/reference/MatrixClasses/CufftComplexMatrix.cpp:82-130,508-534: unnormalised, forward sign -i,
Nx/2+1 bins along the fastest axis)."""
import numpy as np
import pytest

from conftest import rel_l2

SHAPES = [(8, 8, 8), (16, 12, 10), (5, 6, 7), (32, 32, 32), (18, 20, 24), (1, 16, 16), (64, 32, 16), (13, 11, 17)]


@pytest.mark.parametrize("shape", SHAPES)
def test_r2c_c2r_3d(orc, shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.standard_normal(shape).astype(np.float32)
    F = orc.fft_r2c_3d(a)
    Fr = np.fft.rfftn(a.astype(np.float64), axes=(0, 1, 2))
    assert rel_l2(F, Fr) < 5e-7
    b = orc.fft_c2r_3d(Fr.astype(np.complex64), shape[2])
    br = np.fft.irfftn(Fr, s=shape, axes=(0, 1, 2)) * a.size  # unnormalised
    assert rel_l2(b, br) < 5e-7


def test_roundtrip_scales_by_n(orc):
    a = np.random.default_rng(3).standard_normal((12, 16, 20)).astype(np.float32)
    b = orc.fft_c2r_3d(orc.fft_r2c_3d(a).astype(np.complex64), 20)
    assert rel_l2(b / a.size, a) < 5e-7


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_shifted_velocity_matches_numpy(orc, syn, axis):
    """KSpaceFirstOrderSolver.cpp:2714-2735: u_shifted = real(ifft(shift_neg .* fft(u, [], axis), [], axis))."""
    nx, ny, nz = 16, 12, 20
    ops = syn.kspace_operators(nx, ny, nz, 1e-4, 1e-4, 1e-4)
    u = np.random.default_rng(axis).standard_normal((nz, ny, nx)).astype(np.float32)
    name = "xyz"[axis] + "_shift_neg_r"
    sh = ops[name]
    out = orc.shifted_velocity(u, sh, axis)
    np_axis = 2 - axis
    n = u.shape[np_axis]
    shc = (sh[..., 0] + 1j * sh[..., 1]).astype(np.complex128)
    shape = [1, 1, 1]
    shape[np_axis] = -1
    ref = np.fft.irfft(np.fft.rfft(u.astype(np.float64), axis=np_axis) * shc.reshape(shape), n=n, axis=np_axis)
    assert rel_l2(out, ref) < 5e-7
