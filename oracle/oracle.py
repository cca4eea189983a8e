"""ctypes front-end of the CPU oracle (oracle/kwave_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the product package.  Parity status: "parity unpinned" by the reference (no tests or
fixtures exist upstream); pinned by tests/test_oracle_*.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

def host_threads(cap: int = 16) -> int:
    """Threads to use on this host: CPU affinity, capped (a GPU box shows every hardware thread of the node but a
    one-GPU job owns a 16-core share; oversubscribed OpenMP spin-waits are pathologically slow)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(cap, n))


# must be set before libgomp initialises (first dlopen of an OpenMP library in this process)
os.environ.setdefault("OMP_NUM_THREADS", str(host_threads()))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libkwave_oracle.so")
REF_LIB_PATH = os.path.join(HERE, "_ref", "libcompress_ref.so")

OP_NONE, OP_RMS, OP_MAX, OP_MIN = 0, 1, 2, 3


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(HERE, "kwave_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "kwave_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "libkwave_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/Compression/CompressHelper.cpp") and (force or not os.path.exists(REF_LIB_PATH)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class Problem(C.Structure):
    _fields_ = [
        ("nx", C.c_uint64), ("ny", C.c_uint64), ("nz", C.c_uint64),
        ("c0", C.c_void_p), ("rho0", C.c_void_p), ("rho0_sgx", C.c_void_p), ("rho0_sgy", C.c_void_p),
        ("rho0_sgz", C.c_void_p), ("bona", C.c_void_p), ("alpha_coeff", C.c_void_p),
        ("ddx_k_shift_pos", C.c_void_p), ("ddy_k_shift_pos", C.c_void_p), ("ddz_k_shift_pos", C.c_void_p),
        ("ddx_k_shift_neg", C.c_void_p), ("ddy_k_shift_neg", C.c_void_p), ("ddz_k_shift_neg", C.c_void_p),
        ("pml_x", C.c_void_p), ("pml_y", C.c_void_p), ("pml_z", C.c_void_p),
        ("pml_x_sgx", C.c_void_p), ("pml_y_sgy", C.c_void_p), ("pml_z_sgz", C.c_void_p),
        ("p0_source_input", C.c_void_p), ("p_source_index", C.c_void_p), ("p_source_input", C.c_void_p),
        ("u_source_index", C.c_void_p), ("ux_source_input", C.c_void_p), ("uy_source_input", C.c_void_p),
        ("uz_source_input", C.c_void_p), ("transducer_source_input", C.c_void_p), ("delay_mask", C.c_void_p),
        ("p_source_n", C.c_uint64), ("u_source_n", C.c_uint64),
        ("p_source_flag", C.c_uint64), ("ux_source_flag", C.c_uint64), ("uy_source_flag", C.c_uint64),
        ("uz_source_flag", C.c_uint64), ("transducer_source_flag", C.c_uint64), ("p0_source_flag", C.c_uint64),
        ("dt", C.c_float), ("dx", C.c_float), ("dy", C.c_float), ("dz", C.c_float), ("c_ref", C.c_float),
        ("alpha_power", C.c_float),
        ("c0_s", C.c_float), ("rho0_s", C.c_float), ("rho0_sgx_s", C.c_float), ("rho0_sgy_s", C.c_float),
        ("rho0_sgz_s", C.c_float), ("bona_s", C.c_float), ("alpha_coeff_s", C.c_float),
        ("nonlinear_flag", C.c_int32), ("absorbing_flag", C.c_int32),
        ("p_source_mode", C.c_int32), ("p_source_many", C.c_int32), ("u_source_mode", C.c_int32),
        ("u_source_many", C.c_int32),
        ("dxudxn", C.c_void_p), ("dyudyn", C.c_void_p), ("dzudzn", C.c_void_p),
        ("dxudxn_sgx", C.c_void_p), ("dyudyn_sgy", C.c_void_p), ("dzudzn_sgz", C.c_void_p),
    ]


class CompressState(C.Structure):
    _fields_ = [("n_sens", C.c_uint64), ("harmonics", C.c_uint64), ("o_size", C.c_uint64), ("b_size", C.c_uint64),
                ("sampled_step", C.c_uint64), ("compressed_step", C.c_uint64), ("no_overlap", C.c_int32),
                ("pad_", C.c_int32), ("c1", C.c_void_p), ("c2", C.c_void_p)]


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.kwo_create.restype = C.c_void_p
        L.kwo_create.argtypes = [C.POINTER(Problem)]
        L.kwo_destroy.argtypes = [C.c_void_p]
        L.kwo_step.argtypes = [C.c_void_p]
        L.kwo_time_index.restype = C.c_uint64
        L.kwo_time_index.argtypes = [C.c_void_p]
        L.kwo_field.restype = C.POINTER(C.c_float)
        L.kwo_field.argtypes = [C.c_void_p, C.c_char_p]
        L.kwo_scalar.restype = C.c_float
        L.kwo_scalar.argtypes = [C.c_void_p, C.c_char_p]
        for nm in ("kwo_fft_r2c_3d", "kwo_fft_c2r_3d"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for nm in ("kwo_fft_r2c_1d", "kwo_fft_c2r_1d"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
        L.kwo_sample_index.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.kwo_sample_cuboid.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint64]
        L.kwo_sample_all.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.kwo_post_rms.argtypes = [C.c_void_p, C.c_float, C.c_uint64]
        L.kwo_shifted_velocity.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                           C.c_int]
        L.kwo_compress_basis.argtypes = [C.c_float, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.kwo_compress_osize.restype = C.c_uint64
        L.kwo_compress_osize.argtypes = [C.c_float, C.c_uint64]
        L.kwo_compress_bsize.restype = C.c_uint64
        L.kwo_compress_bsize.argtypes = [C.c_float, C.c_uint64]
        L.kwo_compress_step.restype = C.c_int
        L.kwo_compress_step.argtypes = [C.POINTER(CompressState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p]
        _lib = L
    return _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _sc(a) -> float:
    return float(np.asarray(a).reshape(-1)[0])


class OracleSim:
    """One simulation on the CPU oracle, built from a problem dict (HDF5 dataset names, 1-based indices)."""

    FIELD_NAMES = ("p", "ux", "uy", "uz", "rhox", "rhoy", "rhoz", "duxdx", "duydy", "duzdz")

    def __init__(self, pr: Dict[str, np.ndarray]):
        L = lib()
        self._keep = []
        P = Problem()
        from .kwave_np import complete_2d
        pr = complete_2d(pr)  # Nz == 1: the z datasets a 2-D input does not carry
        nx, ny, nz = int(_sc(pr["Nx"])), int(_sc(pr["Ny"])), int(_sc(pr["Nz"]))
        self.nx, self.ny, self.nz = nx, ny, nz
        self.n = nx * ny * nz
        self.nc = (nx // 2 + 1) * ny * nz
        P.nx, P.ny, P.nz = nx, ny, nz

        def arr(name, dtype=np.float32):
            a = np.ascontiguousarray(pr[name], dtype=dtype)
            self._keep.append(a)
            return a

        def medium(field, key, scalar_field):
            if key not in pr:
                setattr(P, scalar_field, 0.0)
                return
            a = pr[key]
            if a.size == 1:
                setattr(P, scalar_field, _sc(a))
            else:
                setattr(P, field, _ptr(arr(key)))

        medium("c0", "c0", "c0_s")
        medium("rho0", "rho0", "rho0_s")
        medium("rho0_sgx", "rho0_sgx", "rho0_sgx_s")
        medium("rho0_sgy", "rho0_sgy", "rho0_sgy_s")
        medium("rho0_sgz", "rho0_sgz", "rho0_sgz_s")
        medium("bona", "BonA", "bona_s")
        medium("alpha_coeff", "alpha_coeff", "alpha_coeff_s")
        P.ddx_k_shift_pos = _ptr(arr("ddx_k_shift_pos_r"))
        P.ddy_k_shift_pos = _ptr(arr("ddy_k_shift_pos"))
        P.ddz_k_shift_pos = _ptr(arr("ddz_k_shift_pos"))
        P.ddx_k_shift_neg = _ptr(arr("ddx_k_shift_neg_r"))
        P.ddy_k_shift_neg = _ptr(arr("ddy_k_shift_neg"))
        P.ddz_k_shift_neg = _ptr(arr("ddz_k_shift_neg"))
        for nm in ("pml_x", "pml_y", "pml_z", "pml_x_sgx", "pml_y_sgy", "pml_z_sgz"):
            setattr(P, nm, _ptr(arr(nm)))
        for nm in ("dt", "dx", "dy", "dz", "c_ref"):
            setattr(P, nm, _sc(pr[nm]))
        if int(_sc(pr.get("nonuniform_grid_flag", 0))):
            for nm in ("dxudxn", "dyudyn", "dzudzn", "dxudxn_sgx", "dyudyn_sgy", "dzudzn_sgz"):
                setattr(P, nm, _ptr(arr(nm)))
        P.alpha_power = _sc(pr["alpha_power"]) if "alpha_power" in pr else 0.0
        P.nonlinear_flag = int(_sc(pr["nonlinear_flag"]))
        P.absorbing_flag = int(_sc(pr["absorbing_flag"]))
        for nm in ("p_source_flag", "ux_source_flag", "uy_source_flag", "uz_source_flag", "transducer_source_flag",
                   "p0_source_flag"):
            setattr(P, nm, int(_sc(pr.get(nm, 0))))
        if P.p0_source_flag:
            P.p0_source_input = _ptr(arr("p0_source_input"))
        if P.p_source_flag:
            idx = (np.ascontiguousarray(pr["p_source_index"], dtype=np.uint64).reshape(-1) - np.uint64(1))
            self._keep.append(idx)
            P.p_source_index = _ptr(idx)
            P.p_source_n = idx.size
            P.p_source_input = _ptr(arr("p_source_input"))
            P.p_source_mode = int(_sc(pr["p_source_mode"]))
            P.p_source_many = int(_sc(pr["p_source_many"]))
        if P.ux_source_flag or P.uy_source_flag or P.uz_source_flag or P.transducer_source_flag:
            idx = (np.ascontiguousarray(pr["u_source_index"], dtype=np.uint64).reshape(-1) - np.uint64(1))
            self._keep.append(idx)
            P.u_source_index = _ptr(idx)
            P.u_source_n = idx.size
            P.u_source_mode = int(_sc(pr["u_source_mode"]))
            P.u_source_many = int(_sc(pr["u_source_many"]))
            if P.ux_source_flag:
                P.ux_source_input = _ptr(arr("ux_source_input"))
            if P.uy_source_flag:
                P.uy_source_input = _ptr(arr("uy_source_input"))
            if P.uz_source_flag:
                P.uz_source_input = _ptr(arr("uz_source_input"))
            if P.transducer_source_flag:
                P.transducer_source_input = _ptr(arr("transducer_source_input"))
                dm = (np.ascontiguousarray(pr["delay_mask"], dtype=np.uint64).reshape(-1) - np.uint64(1))
                self._keep.append(dm)
                P.delay_mask = _ptr(dm)
        self._P = P
        self._h = L.kwo_create(C.byref(P))
        if "sensor_mask_index" in pr:
            self.sensor_index = (np.ascontiguousarray(pr["sensor_mask_index"], dtype=np.uint64).reshape(-1)
                                 - np.uint64(1))
        else:
            self.sensor_index = None

    def close(self):
        if self._h:
            lib().kwo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, n: int = 1):
        L = lib()
        for _ in range(n):
            L.kwo_step(self._h)

    @property
    def t(self) -> int:
        return int(lib().kwo_time_index(self._h))

    def field(self, name: str) -> Optional[np.ndarray]:
        """View (no copy) of a state / operator array; None when it is a scalar for this medium."""
        p = lib().kwo_field(self._h, name.encode())
        if not p:
            return None
        reduced = name in ("kappa", "nabla1", "nabla2", "source_kappa")
        shape = (self.nz, self.ny, self.nx // 2 + 1) if reduced else (self.nz, self.ny, self.nx)
        return np.ctypeslib.as_array(p, shape=shape)

    def scalar(self, name: str) -> float:
        return float(lib().kwo_scalar(self._h, name.encode()))

    def sample(self, op: int, buf: np.ndarray, field: str = "p"):
        src = self.field(field)
        lib().kwo_sample_index(op, buf.ctypes.data, src.ctypes.data, self.sensor_index.ctypes.data,
                               self.sensor_index.size)


# ---- stand-alone helpers ------------------------------------------------------------------------
def fft_r2c_3d(a: np.ndarray) -> np.ndarray:
    a = _f32(a)
    nz, ny, nx = a.shape
    out = np.empty((nz, ny, nx // 2 + 1, 2), dtype=np.float32)
    lib().kwo_fft_r2c_3d(a.ctypes.data, out.ctypes.data, nx, ny, nz)
    return out[..., 0] + 1j * out[..., 1]


def fft_c2r_3d(c: np.ndarray, nx: int) -> np.ndarray:
    nz, ny, nxc = c.shape
    assert nxc == nx // 2 + 1
    cin = np.empty((nz, ny, nxc, 2), dtype=np.float32)
    cin[..., 0] = c.real
    cin[..., 1] = c.imag
    out = np.empty((nz, ny, nx), dtype=np.float32)
    lib().kwo_fft_c2r_3d(cin.ctypes.data, out.ctypes.data, nx, ny, nz)
    return out


def sample_index(op, buf, src, mask):
    lib().kwo_sample_index(op, buf.ctypes.data, src.ctypes.data, mask.ctypes.data, mask.size)


def sample_cuboid(op, buf, src, tl, br, size):
    tl = np.asarray(tl, dtype=np.uint32)
    br = np.asarray(br, dtype=np.uint32)
    size = np.asarray(size, dtype=np.uint32)
    lib().kwo_sample_cuboid(op, buf.ctypes.data, src.ctypes.data, tl.ctypes.data, br.ctypes.data, size.ctypes.data,
                            buf.size)


def sample_all(op, buf, src):
    lib().kwo_sample_all(op, buf.ctypes.data, src.ctypes.data, buf.size)


def post_rms(buf, scale):
    lib().kwo_post_rms(buf.ctypes.data, float(scale), buf.size)


def shifted_velocity(u: np.ndarray, shift_neg_r: np.ndarray, axis: int) -> np.ndarray:
    u = _f32(u)
    nz, ny, nx = u.shape
    out = np.empty_like(u)
    sh = _f32(shift_neg_r)
    lib().kwo_shifted_velocity(u.ctypes.data, out.ctypes.data, sh.ctypes.data, nx, ny, nz, axis)
    return out


def compress_basis(period: float, mos: int, harmonics: int, shifted: bool):
    L = lib()
    bs = int(L.kwo_compress_bsize(period, mos))
    bE = np.empty((harmonics, bs, 2), dtype=np.float32)
    bE1 = np.empty((harmonics, bs, 2), dtype=np.float32)
    L.kwo_compress_basis(period, mos, harmonics, int(shifted), bE.ctypes.data, bE1.ctypes.data)
    return bE, bE1


class Compressor:
    """Compression accumulation of one stream (IndexOutputStream.cpp:373-470 restated in the oracle)."""

    def __init__(self, n_sens: int, period: float, mos: int, harmonics: int, shifted: bool, no_overlap=False):
        L = lib()
        self.bE, self.bE1 = compress_basis(period, mos, harmonics, shifted)
        self.c1 = np.zeros((n_sens, harmonics, 2), dtype=np.float32)
        self.c2 = np.zeros((n_sens, harmonics, 2), dtype=np.float32)
        st = CompressState()
        st.n_sens, st.harmonics = n_sens, harmonics
        st.o_size = int(L.kwo_compress_osize(period, mos))
        st.b_size = int(L.kwo_compress_bsize(period, mos))
        st.no_overlap = int(no_overlap)
        st.c1, st.c2 = self.c1.ctypes.data, self.c2.ctypes.data
        self.st = st
        self.frames = []

    def step(self, x: np.ndarray, is_last: bool = False):
        x = _f32(x)
        frame = np.empty_like(self.c1)
        if lib().kwo_compress_step(C.byref(self.st), self.bE.ctypes.data, self.bE1.ctypes.data, x.ctypes.data,
                                   int(is_last), frame.ctypes.data):
            self.frames.append(frame)
            return frame
        return None


# ---- post-processing of stored series (test oracle only; numpy, fp64 inside) ---------------------------------------
def _signed_bins(n: int) -> np.ndarray:
    """shift(i) = (i + n/2) % n - n/2 for the half spectrum i = 0..n/2 (KSpaceFirstOrderSolver.cpp:1257, :1907):
    the Nyquist bin of an even length gets the negative frequency."""
    i = np.arange(n // 2 + 1)
    return ((i + n // 2) % n) - n // 2


def time_shift_half_step(series: np.ndarray) -> np.ndarray:
    """Sampled series [steps][points] advanced by half a time step through its spectrum along time
    (computeAverageIntensities, KSpaceFirstOrderSolver.cpp:1253-1260, :1434-1449): X[k] *= exp(i*pi*shift(k)/steps),
    unnormalised R2C / C2R pair with the 1/steps folded into the multiply."""
    s = np.asarray(series, dtype=np.float64)
    steps = s.shape[0]
    kx = np.exp(1j * np.pi * _signed_bins(steps) / steps)
    return np.fft.irfft(np.fft.rfft(s, axis=0) * kx[:, None], n=steps, axis=0)


def average_intensity(p_series: np.ndarray, u_series: np.ndarray) -> np.ndarray:
    """I_avg per sensor point: mean over the stored steps of p * (u shifted by half a step) (:1492-1513)."""
    return (time_shift_half_step(u_series) * np.asarray(p_series, dtype=np.float64)).mean(axis=0)


def q_term(ix, iy, iz, grid_index, dims, spacing) -> np.ndarray:
    """Q = -(dIx/dx + dIy/dy + dIz/dz) at the sensor points (computeQTerm, :1783-2080): the intensities are scattered
    into a zero grid (dims = (nx, ny, nz), grid_index = 0-based linear indices, x fastest), each is differentiated
    spectrally along its own axis (multiply by i*2*pi/d*shift(k)/n, :1905-1921) and the sum is gathered back."""
    nx, ny, nz = dims
    idx = np.asarray(grid_index, dtype=np.int64).reshape(-1)
    total = np.zeros(nx * ny * nz, dtype=np.float64)
    for comp, (n, d, ax) in zip((ix, iy, iz), ((nx, spacing[0], 2), (ny, spacing[1], 1), (nz, spacing[2], 0))):
        g = np.zeros(nx * ny * nz, dtype=np.float64)
        g[idx] = np.asarray(comp, dtype=np.float64).reshape(-1)
        g = g.reshape(nz, ny, nx)
        k = 1j * (2.0 * np.pi / d) * _signed_bins(n) / n
        shape = [1, 1, 1]
        shape[ax] = n // 2 + 1
        total += np.fft.irfft(np.fft.rfft(g, axis=ax) * k.reshape(shape), n=n, axis=ax).reshape(-1)
    return -total[idx]
