"""ctypes binding of the device-layer C-ABI (include/kwave_hip.h -> lib/libkwave_hip.so).

There is no CPU fallback: importing works anywhere (so `-m "not gpu"` tests can check the exported
symbols), but every compute entry point needs a gfx950 device and raises KWaveError otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, Iterable, Optional

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, "lib", "libkwave_hip.so")
HEADER_PATH = os.path.join(ROOT, "include", "kwave_hip.h")

OP_NONE, OP_RMS, OP_MAX, OP_MIN = 0, 1, 2, 3
SRC_DIRICHLET, SRC_ADDITIVE_NO_CORRECTION, SRC_ADDITIVE = 0, 1, 2


class KWaveError(RuntimeError):
    """Non-zero kw_status; message = kw_last_error() (the reference throws std::runtime_error)."""


class Constants(C.Structure):
    """struct kw_constants (mirror of CudaDeviceConstants, Parameters/CudaDeviceConstants.cuh:44-116)."""
    _fields_ = [(n, C.c_uint32) for n in ("nx", "ny", "nz", "n_elements", "nx_complex", "ny_complex", "nz_complex",
                                          "n_elements_complex")] + \
               [(n, C.c_float) for n in ("fft_divider", "fft_divider_x", "fft_divider_y", "fft_divider_z", "dt",
                                         "dt_by_2", "c2", "rho0", "dt_rho0", "dt_rho0_sgx", "dt_rho0_sgy",
                                         "dt_rho0_sgz", "b_on_a", "absorb_tau", "absorb_eta")] + \
               [(n, C.c_uint32) for n in ("velocity_source_size", "velocity_source_mode", "velocity_source_many",
                                          "pressure_source_size", "pressure_source_mode", "pressure_source_many")]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("calls", C.c_uint64), ("total_ms", C.c_double)]


def profile_collect(ctx) -> Dict[str, tuple]:
    """{entry point: (calls, total_ms)} since the last collect (kw_profile_collect)."""
    L = load()
    buf = (ProfileEntry * 64)()
    n = C.c_size_t()
    check(L.kw_profile_collect(ctx, buf, 64, C.byref(n)))
    return {buf[i].name.decode(): (int(buf[i].calls), float(buf[i].total_ms)) for i in range(min(n.value, 64))}


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("device_id", C.c_int32),
                ("compute_units", C.c_int32), ("wavefront_size", C.c_int32), ("clock_mhz", C.c_int32),
                ("total_mem", C.c_uint64), ("free_mem", C.c_uint64), ("lds_per_cu", C.c_uint64),
                ("l2_bytes", C.c_uint64)]


def declared_symbols(header: str = HEADER_PATH) -> Iterable[str]:
    """Every KW_API function name declared in include/kwave_hip.h."""
    txt = open(header).read()
    return sorted(set(re.findall(r"KW_API\s+[\w\s\*]+?\b(kw_\w+)\s*\(", txt)))


_lib: Optional[C.CDLL] = None

_P = C.c_void_p
_U64 = C.c_uint64
_SIG: Dict[str, list] = {
    "kw_init": [C.c_int, C.POINTER(_P)],
    "kw_destroy": [_P],
    "kw_device_info_get": [_P, C.POINTER(DeviceInfo)],
    "kw_set_stream": [_P, _P],
    "kw_sync": [_P],
    "kw_graph_begin": [_P],
    "kw_graph_end": [_P, _P],
    "kw_graph_launch": [_P, _P],
    "kw_graph_destroy": [_P, _P],
    "kw_event_create": [_P, C.POINTER(_P)],
    "kw_event_record": [_P, _P],
    "kw_event_synchronize": [_P, _P],
    "kw_event_elapsed_ms": [_P, _P, _P, C.POINTER(C.c_float)],
    "kw_event_destroy": [_P, _P],
    "kw_profile_enable": [_P, C.c_int],
    "kw_profile_enabled": [_P],
    "kw_profile_collect": [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)],
    "kw_malloc": [_P, C.c_size_t, C.POINTER(_P)],
    "kw_free": [_P, _P],
    "kw_memcpy_h2d": [_P, _P, _P, C.c_size_t],
    "kw_memcpy_d2h": [_P, _P, _P, C.c_size_t],
    "kw_memcpy_d2d": [_P, _P, _P, C.c_size_t],
    "kw_memcpy_d2h_async": [_P, _P, _P, C.c_size_t],
    "kw_memset": [_P, _P, C.c_int, C.c_size_t],
    "kw_memcpy_d2h_overlapped": [_P, _P, _P, C.c_size_t, _P],
    "kw_host_alloc": [_P, C.c_size_t, C.POINTER(_P)],
    "kw_host_free": [_P, _P],
    "kw_set_constants": [_P, C.POINTER(Constants)],
    "kw_get_constants": [_P, C.POINTER(Constants)],
    "kw_fft_create_plans_3d": [_P],
    "kw_fft_create_plans_1d": [_P, C.c_int],
    "kw_fft_destroy_plans": [_P],
    "kw_fft_r2c_3d": [_P, _P, _P],
    "kw_fft_c2r_3d": [_P, _P, _P],
    "kw_fft_r2c_1d": [_P, C.c_int, _P, _P],
    "kw_fft_c2r_1d": [_P, C.c_int, _P, _P],
    "kw_compute_velocity": [_P] + [_P] * 12,
    "kw_add_transducer_source": [_P, _P, _P, _P, _P, _U64],
    "kw_add_velocity_source": [_P, _P, _P, _P, _U64],
    "kw_add_pressure_source": [_P, _P, _P, _P, _P, _P, _U64],
    "kw_insert_source_into_scaling_matrix": [_P, _P, _P, _P, _U64, C.c_int, _U64],
    "kw_compute_source_gradient": [_P, _P, _P],
    "kw_add_velocity_scaled_source": [_P, _P, _P],
    "kw_add_pressure_scaled_source": [_P, _P, _P, _P, _P],
    "kw_add_initial_pressure_source": [_P] + [_P] * 6,
    "kw_compute_initial_velocity": [_P] + [_P] * 6,
    "kw_compute_pressure_gradient": [_P] + [_P] * 7,
    "kw_compute_velocity_gradient": [_P] + [_P] * 7,
    "kw_compute_velocity_gradient_shift_nonuniform": [_P] + [_P] * 6,
    "kw_compute_density_nonlinear": [_P] + [_P] * 10,
    "kw_compute_density_linear": [_P] + [_P] * 10,
    "kw_compute_pressure_terms_nonlinear": [_P] + [_P] * 11,
    "kw_compute_pressure_terms_linear": [_P] + [_P] * 9,
    "kw_compute_absorbtion_term": [_P] + [_P] * 4,
    "kw_sum_pressure_terms_nonlinear": [_P] + [_P] * 7,
    "kw_sum_pressure_terms_linear": [_P] + [_P] * 7,
    "kw_sum_pressure_nonlinear_lossless": [_P] + [_P] * 7,
    "kw_sum_pressure_linear_lossless": [_P] + [_P] * 5,
    "kw_compute_velocity_shift": [_P, C.c_int, _P, _P],
    "kw_measure_copy_bandwidth": [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double)],
    "kw_comm_unique_id": [_P, C.c_size_t],
    "kw_comm_init": [_P, C.c_uint32, C.c_uint32, _P],
    "kw_comm_destroy": [_P],
    "kw_comm_info": [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)],
    "kw_comm_unique_id_from": [C.c_char_p, _P, C.c_size_t],
    "kw_comm_init_with": [_P, C.c_char_p, C.c_uint32, C.c_uint32, _P],
    "kw_comm_init_p2p": [_P, C.c_uint32, C.c_uint32],
    "kw_comm_p2p_export": [_P, _P, C.c_size_t],
    "kw_comm_p2p_connect": [_P, _P],
    "kw_comm_p2p_emulate": [_P, C.c_float, C.c_float],
    "kw_comm_transport": [_P, C.POINTER(C.c_int)],
    "kw_get_tuning": [_P, _P],
    "kw_set_tuning": [_P, _P],
    "kw_fused_set_slab": [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P],
    "kw_fused_set_slab_async": [_P, _P, _P],
    "kw_fused_set_slab_pieces": [_P, _P, _P],
    "kw_fused_scratch_bytes": [_P, C.POINTER(C.c_size_t)],
    "kw_fused_create_with_scratch": [_P, _P, _P],
    "kw_fused_supported": [_P, C.POINTER(C.c_int)],
    "kw_fused_shift_velocity": [_P, C.c_int, _P, _P, _P],
    "kw_fused_create": [_P],
    "kw_fused_destroy": [_P],
    "kw_fused_reduced_elems": [_P, C.POINTER(C.c_size_t)],
    "kw_fused_import_reduced": [_P, _P, _P],
    "kw_fused_velocity": [_P] + [_P] * 14 + [C.c_int],
    "kw_fused_initial_velocity": [_P] + [_P] * 11,
    "kw_fused_density": [_P, C.c_int] + [_P] * 14 + [_P] * 3 + [C.c_int, _P, _P, _P, _P, C.c_int],
    "kw_fused_velocity_gradient": [_P] + [_P] * 10 + [C.c_int],
    "kw_fused_absorption_pressure": [_P] + [_P] * 9 + [C.c_int],
    "kw_fused_scale_source": [_P, _P, _P],
    "kw_fused_probe": [_P, C.c_int, _P],
    "kw_sample_index": [_P, C.c_int, _P, _P, _P, _U64],
    "kw_sample_index_multi": [_P, C.c_int, _P, _P, _P, _P, C.c_uint64],
    "kw_sample_cuboid": [_P, C.c_int, _P, _P, _P, _P, _P, _U64],
    "kw_sample_all": [_P, C.c_int, _P, _P, _U64],
    "kw_post_processing_rms": [_P, _P, C.c_float, _U64],
    "kw_sample_index_compress": [_P, _P, _P, _P, _P, _U64, C.c_uint32, _P, _P, C.c_uint32, C.c_uint32, C.c_int],
    "kw_sample_index_compress_40b": [_P, _P, _P, _P, _P, _U64, C.c_uint32, _P, _P, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int],
    "kw_intensity_avg_c_accumulate_40b": [_P, _P, _P, _P, _U64, C.c_uint32, C.c_int, C.c_int],
    "kw_intensity_avg_c_accumulate": [_P, _P, _P, _P, _U64, C.c_uint32],
    "kw_time_shift_series": [_P, _P, _P, _U64, _U64],
    "kw_intensity_avg": [_P, _P, _P, _P, _U64, _U64],
    "kw_q_term_sum": [_P, _P, _P, _P, _P, _U64],
    "kw_divide": [_P, _P, C.c_float, _U64],
}


def load() -> C.CDLL:
    """dlopen lib/libkwave_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KWaveError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first")
        L = C.CDLL(LIB_PATH)
        L.kw_last_error.restype = C.c_char_p
        L.kw_last_error.argtypes = []
        L.kw_get_stream.restype = _P
        L.kw_get_stream.argtypes = [_P]
        for name, args in _SIG.items():
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != 0:
        raise KWaveError(f"[kw_status {status}] " + load().kw_last_error().decode(errors="replace"))


class DeviceArray:
    """A raw device buffer with numpy shape/dtype metadata."""

    def __init__(self, dev: "Device", shape, dtype):
        self.dev = dev
        self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = _P()
        check(dev.L.kw_malloc(dev.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        check(self.dev.L.kw_memcpy_h2d(self.dev.ctx, self.ptr, a.ctypes.data, self.nbytes))
        return self

    def download(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.dev.L.kw_memcpy_d2h(self.dev.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def zero(self):
        check(self.dev.L.kw_memset(self.dev.ctx, self.ptr, 0, self.nbytes))

    def fill_bytes(self, value: int):
        check(self.dev.L.kw_memset(self.dev.ctx, self.ptr, value, self.nbytes))

    def free(self):
        if self.ptr:
            self.dev.L.kw_free(self.dev.ctx, self.ptr)
            self.ptr = None


class Device:
    """Owner of one kw_ctx."""

    def __init__(self, device_id: int = -1):
        self.L = load()
        ctx = _P()
        check(self.L.kw_init(device_id, C.byref(ctx)))
        self.ctx = ctx
        self._arrays = []

    def info(self) -> DeviceInfo:
        inf = DeviceInfo()
        check(self.L.kw_device_info_get(self.ctx, C.byref(inf)))
        return inf

    def empty(self, shape, dtype=np.float32) -> DeviceArray:
        a = DeviceArray(self, shape, dtype)
        self._arrays.append(a)
        return a

    def zeros(self, shape, dtype=np.float32) -> DeviceArray:
        a = self.empty(shape, dtype)
        a.zero()
        return a

    def array(self, host: np.ndarray, dtype=None) -> DeviceArray:
        host = np.ascontiguousarray(host, dtype=dtype or host.dtype)
        return self.empty(host.shape, host.dtype).upload(host)

    def set_constants(self, k: Constants):
        check(self.L.kw_set_constants(self.ctx, C.byref(k)))

    def sync(self):
        check(self.L.kw_sync(self.ctx))

    def call(self, name: str, *args):
        """Call kw_<name>(ctx, *args); DeviceArray arguments are passed as pointers, None as NULL."""
        conv = [a.ptr if isinstance(a, DeviceArray) else a for a in args]
        check(getattr(self.L, "kw_" + name)(self.ctx, *conv))

    # events
    def event(self):
        ev = _P()
        check(self.L.kw_event_create(self.ctx, C.byref(ev)))
        return ev

    def record(self, ev):
        check(self.L.kw_event_record(self.ctx, ev))

    def elapsed_ms(self, ev0, ev1) -> float:
        check(self.L.kw_event_synchronize(self.ctx, ev1))
        ms = C.c_float()
        check(self.L.kw_event_elapsed_ms(self.ctx, ev0, ev1, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if self.ctx:
            for a in self._arrays:
                a.free()
            self._arrays = []
            self.L.kw_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128
COMM_P2P_BLOB_BYTES = 1024
TRANSPORTS = {-1: "none", 0: "rccl", 1: "p2p (not connected)", 2: "p2p", 3: "p2p link model"}


class Tuning(C.Structure):
    """kw_tuning of include/kwave_hip.h: every schedule parameter of the device library"""
    _fields_ = [("struct_bytes", C.c_uint32), ("side_array", C.c_int32), ("tail_chunks", C.c_int32), ("split512", C.c_int32),
                ("slab_pipeline", C.c_int32), ("slab_chunks", C.c_int32), ("slab_batch", C.c_int32),
                ("p2p_blocks_per_peer", C.c_int32), ("p2p_timeout_s", C.c_float), ("plane_kernels", C.c_int32)]


def default_tuning() -> Tuning:
    """the library's defaults (what a fresh context reports through kw_get_tuning)"""
    return Tuning(C.sizeof(Tuning), 1, 0, 1, 1, 1, -1, 4, 20.0, 1)


def make_tuning(spec=None) -> Tuning:
    """Tuning from a dict or a "key=value,key=value" string over the defaults; the KW_TUNING environment variable, read
    HERE (tools' A/B scripts), is applied first — the library itself reads no environment."""
    t = default_tuning()
    for src in (os.environ.get("KW_TUNING"), spec):
        if not src:
            continue
        items = src.items() if isinstance(src, dict) else (kv.split("=", 1) for kv in str(src).replace(" ", ",").split(",") if kv)
        for k, v in items:
            if k not in dict(Tuning._fields_) or k == "struct_bytes":
                raise KWaveError(f"unknown tuning parameter {k!r}")
            setattr(t, k, float(v) if k == "p2p_timeout_s" else int(v))
    return t


def comm_unique_id(rccl_library=None) -> bytes:
    """ncclGetUniqueId through the device library (kw_comm_unique_id): call on rank 0, hand the bytes to every rank."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    lib = rccl_library.encode() if rccl_library else None
    check(load().kw_comm_unique_id_from(lib, buf, COMM_ID_BYTES))
    return buf.raw


def comm_transport(ctx) -> str:
    v = C.c_int()
    check(load().kw_comm_transport(ctx, C.byref(v)))
    return TRANSPORTS.get(v.value, str(v.value))


def comm_exchanges(ctx) -> int:
    """exchanges started so far on the context's communicator (0 without one)"""
    n = C.c_uint64()
    check(load().kw_comm_info(ctx, None, None, C.byref(n)))
    return int(n.value)
