"""ctypes front-end of the C++ host layer (include/kwave_host.h -> lib/libkwave_host.so).

`HostSolver` runs the C++ `KSpaceFirstOrderSolver` time loop (k-wave-fluid-cuda_amd/host/) on an MI355X from a
problem given as a dict of NumPy arrays keyed by the k-Wave HDF5 dataset names.  No CPU fallback: a missing
library or device raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import capi

HOST_LIB_PATH = os.path.join(capi.PKG, "lib", "libkwave_host.so")

# host pre-processing uses OpenMP: keep it to this job's CPU share (a GPU box exposes every hardware thread)
try:
    _ncpu = len(os.sched_getaffinity(0))
except AttributeError:  # pragma: no cover
    _ncpu = os.cpu_count() or 1
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, _ncpu))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


class Dataset(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("pad_", C.c_int32),
                ("nx", C.c_uint64), ("ny", C.c_uint64), ("nz", C.c_uint64)]


class Options(C.Structure):
    _fields_ = [("device_idx", C.c_int32), ("fused_kernels", C.c_int32),
                ("sampling_start_time_index", C.c_uint64), ("benchmark_time_steps", C.c_uint64)] + \
               [(n, C.c_int32) for n in ("p_raw", "p_rms", "p_max", "p_min", "p_max_all", "p_min_all", "p_final",
                                         "u_raw", "u_rms", "u_max", "u_min", "u_max_all", "u_min_all", "u_final",
                                         "u_non_staggered_raw", "p_c", "u_non_staggered_c", "i_avg_c", "no_overlap")] + \
               [("period", C.c_float), ("mos", C.c_uint64), ("harmonics", C.c_uint64),
                ("slab_ranks", C.c_uint64), ("slab_rank", C.c_uint64), ("nz_global", C.c_uint64),
                ("exchange_fn", C.c_void_p), ("exchange_user", C.c_void_p),
                ("exchange_start_fn", C.c_void_p), ("exchange_wait_fn", C.c_void_p), ("scratch", C.c_void_p * 6),
                ("i_avg", C.c_int32), ("q_term", C.c_int32), ("q_term_c", C.c_int32), ("u_c", C.c_int32),
                ("frequency", C.c_float), ("only_post_processing", C.c_int32), ("comm_unique_id", C.c_void_p),
                ("complex_40bit", C.c_int32), ("reserved_", C.c_int32), ("exchange_piece_fn", C.c_void_p),
                ("tuning", C.c_void_p), ("step_graph", C.c_int32), ("comm_p2p", C.c_int32),
                ("comm_allgather_fn", C.c_void_p), ("comm_allgather_user", C.c_void_p), ("rccl_library", C.c_char_p),
                ("p2p_emulate_link_gbs", C.c_float), ("p2p_emulate_latency_us", C.c_float)]

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)  # kwh_allgather_fn


_hlib: Optional[C.CDLL] = None


def load_host() -> C.CDLL:
    global _hlib
    if _hlib is None:
        capi.load()  # libkwave_hip.so first (RTLD_GLOBAL not needed: host lib links it via rpath $ORIGIN)
        if not os.path.exists(HOST_LIB_PATH):
            raise capi.KWaveError(f"{HOST_LIB_PATH} is missing: run build() first")
        L = C.CDLL(HOST_LIB_PATH)
        L.kwh_last_error.restype = C.c_char_p
        L.kwh_create.argtypes = [C.POINTER(Dataset), C.c_size_t, C.POINTER(Options), C.POINTER(C.c_void_p)]
        L.kwh_destroy.argtypes = [C.c_void_p]
        L.kwh_run.argtypes = [C.c_void_p, C.c_uint64]
        L.kwh_finish.argtypes = [C.c_void_p]
        L.kwh_sync.argtypes = [C.c_void_p]
        L.kwh_time_index.restype = C.c_uint64
        L.kwh_time_index.argtypes = [C.c_void_p]
        L.kwh_context.restype = C.c_void_p
        L.kwh_context.argtypes = [C.c_void_p]
        L.kwh_get_matrix.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.kwh_matrix_size.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        L.kwh_get_scalar.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]
        L.kwh_stream_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.kwh_stream_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.kwh_set_matrix.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.kwh_set_time_index.argtypes = [C.c_void_p, C.c_uint64]
        L.kwh_stream_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.kwh_stream_name.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.kwh_stream_count_all.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.kwh_stream_name_all.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.kwh_stream_checkpoint.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]
        L.kwh_stream_restore.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
        _hlib = L
    return _hlib


def _check(rc: int):
    if rc != 0:
        raise capi.KWaveError(load_host().kwh_last_error().decode(errors="replace"))


# reference-style matrix names (MatrixContainer records) for the state arrays
STATE_NAMES = {"p": "p", "ux": "ux_sgx", "uy": "uy_sgy", "uz": "uz_sgz", "rhox": "rhox", "rhoy": "rhoy",
               "rhoz": "rhoz", "duxdx": "duxdx", "duydy": "duydy", "duzdz": "duzdz", "kappa": "kappa_r",
               "nabla1": "absorb_nabla1_r", "nabla2": "absorb_nabla2_r", "source_kappa": "source_kappa_r",
               "tau": "absorb_tau", "eta": "absorb_eta", "c2": "c0", "dtrho0sgx": "rho0_sgx", "dtrho0sgy": "rho0_sgy",
               "dtrho0sgz": "rho0_sgz", "ux_shifted": "ux_shifted", "uy_shifted": "uy_shifted",
               "uz_shifted": "uz_shifted"}


class HostSolver:
    """One simulation on the GPU through the C++ host layer."""

    def __init__(self, pr: Dict[str, np.ndarray], **opts):
        L = load_host()
        self._keep = []
        sets = (Dataset * len(pr))()
        for i, (name, a) in enumerate(pr.items()):
            if a.dtype == np.uint64:
                arr, dt = np.ascontiguousarray(a, dtype=np.uint64), 1
            else:
                arr, dt = np.ascontiguousarray(a, dtype=np.float32), 0
            self._keep.append(arr)
            shp = list(arr.shape)[::-1]
            while len(shp) < 3:
                shp.append(1)
            if len(shp) > 3:
                shp = shp[:2] + [int(np.prod(shp[2:]))]
            nm = name.encode()
            self._keep.append(nm)
            sets[i].name, sets[i].data, sets[i].dtype = nm, arr.ctypes.data, dt
            sets[i].nx, sets[i].ny, sets[i].nz = shp
        o = Options()
        o.device_idx = opts.pop("device_idx", -1)
        o.fused_kernels = int(opts.pop("fused_kernels", True))
        o.sampling_start_time_index = opts.pop("sampling_start", 0)
        o.benchmark_time_steps = opts.pop("benchmark_steps", 0)
        o.period = float(opts.pop("period", 0.0))
        o.frequency = float(opts.pop("frequency", 0.0))
        o.mos = int(opts.pop("mos", 1))
        o.harmonics = int(opts.pop("harmonics", 1))
        # Z-slab decomposition (see dist.py): the problem dict is this rank's slab
        o.slab_ranks = int(opts.pop("slab_ranks", 1))
        o.slab_rank = int(opts.pop("slab_rank", 0))
        o.nz_global = int(opts.pop("nz_global", 0))
        fn = opts.pop("exchange_fn", None)
        if fn is not None:
            self._keep.append(fn)  # keep the ctypes callback alive
            o.exchange_fn = C.cast(fn, C.c_void_p)
        for key in ("exchange_start_fn", "exchange_wait_fn", "exchange_piece_fn"):
            fn = opts.pop(key, None)
            if fn is not None:
                self._keep.append(fn)
                setattr(o, key, C.cast(fn, C.c_void_p))
        comm_id = opts.pop("comm_unique_id", None)
        if comm_id is not None:  # bytes from capi.comm_unique_id(): the library's own RCCL exchange
            buf = C.create_string_buffer(bytes(comm_id), len(comm_id))
            self._keep.append(buf)
            o.comm_unique_id = C.cast(buf, C.c_void_p)
        # schedule parameters of the device library: tuning={"tail_chunks": 4, ...} or "key=value,..." (+ KW_TUNING, a
        # tool-side convenience read by capi.make_tuning — the libraries read no environment)
        tuning = opts.pop("tuning", None)
        if tuning is not None or os.environ.get("KW_TUNING"):
            t = capi.make_tuning(tuning)
            self._keep.append(t)
            o.tuning = C.cast(C.pointer(t), C.c_void_p)
        # slab exchange over the library's P2P transport: allgather(bytes mine) -> bytes of all ranks, in rank order
        allgather = opts.pop("comm_allgather", None)
        if opts.pop("comm_p2p", False):
            o.comm_p2p = 1
            if allgather is not None:
                def _gather(user, mine, out, n, _fn=allgather):
                    try:
                        data = _fn(C.string_at(mine, n))
                        if len(data) != n * int(o.slab_ranks):
                            return 2
                        C.memmove(out, data, len(data))
                        return 0
                    except BaseException:  # noqa: BLE001 - must not unwind through the C frames
                        return 1
                cb = ALLGATHER_FN(_gather)
                self._keep.append(cb)
                o.comm_allgather_fn = C.cast(cb, C.c_void_p)
        emu = opts.pop("p2p_emulate", None)  # (link GB/s, latency us): tools/emulate_rank.py
        if emu is not None:
            o.p2p_emulate_link_gbs, o.p2p_emulate_latency_us = float(emu[0]), float(emu[1])
        lib = opts.pop("rccl_library", None)
        if lib:
            o.rccl_library = lib.encode()
        scratch = opts.pop("scratch", None)
        if scratch is not None:
            for i, ptr in enumerate(scratch):
                o.scratch[i] = ptr
        for k, v in opts.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k}")
            setattr(o, k, int(v))
        self.nx, self.ny, self.nz = (int(np.asarray(pr[k]).ravel()[0]) for k in ("Nx", "Ny", "Nz"))
        h = C.c_void_p()
        _check(L.kwh_create(sets, len(pr), C.byref(o), C.byref(h)))
        self._h = h
        self.L = L

    def run(self, n_steps: int):
        _check(self.L.kwh_run(self._h, n_steps))

    def step(self, n: int = 1):
        self.run(n)

    def finish(self):
        _check(self.L.kwh_finish(self._h))

    def sync(self):
        _check(self.L.kwh_sync(self._h))

    @property
    def t(self) -> int:
        return int(self.L.kwh_time_index(self._h))

    @property
    def ctx(self):
        return C.c_void_p(self.L.kwh_context(self._h))

    def field(self, name: str) -> np.ndarray:
        ref = STATE_NAMES.get(name, name).encode()
        n = C.c_uint64()
        _check(self.L.kwh_matrix_size(self._h, ref, C.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        _check(self.L.kwh_get_matrix(self._h, ref, out.ctypes.data, n.value))
        full, red = self.nx * self.ny * self.nz, (self.nx // 2 + 1) * self.ny * self.nz
        if n.value == full:
            return out.reshape(self.nz, self.ny, self.nx)
        if n.value == red:
            return out.reshape(self.nz, self.ny, self.nx // 2 + 1)
        return out

    def scalar(self, name: str) -> float:
        v = C.c_float()
        _check(self.L.kwh_get_scalar(self._h, name.encode(), C.byref(v)))
        return float(v.value)

    def stream(self, name: str) -> np.ndarray:
        size, steps = C.c_uint64(), C.c_uint64()
        _check(self.L.kwh_stream_info(self._h, name.encode(), C.byref(size), C.byref(steps)))
        out = np.empty(size.value * steps.value, dtype=np.float32)
        _check(self.L.kwh_stream_read(self._h, name.encode(), out.ctypes.data, out.size))
        return out.reshape(steps.value, size.value) if steps.value != 1 else out

    # ---- checkpoint / restart through the plain host API (the HDF5 checkpoint file is h5io.FileSolver's) ----
    CHECKPOINT_MATRICES = ("p", "rhox", "rhoy", "rhoz", "ux_sgx", "uy_sgy", "uz_sgz")

    def stream_names(self, include_hidden: bool = False):
        """Streams of the output; include_hidden adds those that only feed others (e.g. p_c behind --Q_term_c alone)."""
        count, name = (self.L.kwh_stream_count_all, self.L.kwh_stream_name_all) if include_hidden else \
                      (self.L.kwh_stream_count, self.L.kwh_stream_name)
        n = C.c_uint64()
        _check(count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            buf = C.create_string_buffer(128)
            _check(name(self._h, i, buf, 128))
            out.append(buf.value.decode())
        return out

    def checkpoint_state(self) -> dict:
        """{"t_index", "matrices": {name: array}, "streams": {name: (array, sampled_steps)}} of the run so far."""
        st = {"t_index": self.t, "matrices": {}, "streams": {}}
        for name in self.CHECKPOINT_MATRICES:
            st["matrices"][name] = self.field(name).copy()
        for name in self.stream_names(include_hidden=True):
            n, steps = C.c_uint64(), C.c_uint64()
            _check(self.L.kwh_stream_checkpoint(self._h, name.encode(), None, 0, C.byref(n), C.byref(steps)))
            a = np.empty(n.value, dtype=np.float32)
            _check(self.L.kwh_stream_checkpoint(self._h, name.encode(), a.ctypes.data, a.size, C.byref(n), C.byref(steps)))
            st["streams"][name] = (a, steps.value)
        return st

    def restore_state(self, st: dict):
        for name, a in st["matrices"].items():
            a = np.ascontiguousarray(a, dtype=np.float32)
            _check(self.L.kwh_set_matrix(self._h, name.encode(), a.ctypes.data, a.size))
        for name, (a, steps) in st["streams"].items():
            a = np.ascontiguousarray(a, dtype=np.float32)
            _check(self.L.kwh_stream_restore(self._h, name.encode(), a.ctypes.data, a.size, steps))
        _check(self.L.kwh_set_time_index(self._h, st["t_index"]))

    # HIP-event timing of n steps on the solver's stream
    def time_steps(self, n_steps: int) -> float:
        hip = capi.load()
        ctx = self.ctx
        e0, e1 = C.c_void_p(), C.c_void_p()
        capi.check(hip.kw_event_create(ctx, C.byref(e0)))
        capi.check(hip.kw_event_create(ctx, C.byref(e1)))
        capi.check(hip.kw_event_record(ctx, e0))
        self.run(n_steps)
        capi.check(hip.kw_event_record(ctx, e1))
        capi.check(hip.kw_event_synchronize(ctx, e1))
        ms = C.c_float()
        capi.check(hip.kw_event_elapsed_ms(ctx, e0, e1, C.byref(ms)))
        hip.kw_event_destroy(ctx, e0)
        hip.kw_event_destroy(ctx, e1)
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            self.L.kwh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
