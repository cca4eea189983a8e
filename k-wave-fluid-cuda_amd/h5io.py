"""k-Wave HDF5 files (file format 1.1) on either side of the hot path: optional component backed by
lib/libkwave_host_h5.so (C++ `Hdf5File`, k-wave-fluid-cuda_amd/host/h5/).

  write_input_file(problem_dict, path)   synthetic problem -> input file the reference could read (main.cpp:446-563)
  FileSolver(path, **options)            the C++ time loop driven from an input file (kwh_create_from_file)
  FileSolver.write_output(path)          sampled streams / final fields -> output file
  read_dataset / dataset_info / read_attribute   small readers for tests and post-processing
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import capi
from .solver import Dataset, HostSolver, Options, _check, load_host

H5_LIB_PATH = os.path.join(capi.PKG, "lib", "libkwave_host_h5.so")
_h5 = None

# datasets stored as interleaved complex (domain_type "complex")
COMPLEX_DATASETS = {"ddx_k_shift_pos_r", "ddx_k_shift_neg_r", "ddy_k_shift_pos", "ddy_k_shift_neg", "ddz_k_shift_pos",
                    "ddz_k_shift_neg", "x_shift_neg_r", "y_shift_neg_r", "z_shift_neg_r"}


def load_h5() -> C.CDLL:
    global _h5
    if _h5 is None:
        capi.load()
        if not os.path.exists(H5_LIB_PATH):
            raise capi.KWaveError(f"{H5_LIB_PATH} is missing (HDF5 component not built)")
        L = C.CDLL(H5_LIB_PATH)
        L.kwh_last_error.restype = C.c_char_p
        L.kwh_create_from_file.argtypes = [C.c_char_p, C.POINTER(Options), C.POINTER(C.c_void_p)]
        L.kwh_write_output_file.argtypes = [C.c_void_p, C.c_char_p]
        L.kwh_write_input_file.argtypes = [C.c_char_p, C.POINTER(Dataset), C.c_size_t, C.POINTER(C.c_int32)]
        L.kwh_h5_dataset_info_4d.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64 * 4), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32)]
        L.kwh_h5_dataset_info.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32)]
        L.kwh_h5_read.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int32]
        L.kwh_h5_read_attribute.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
        L.kwh_write_output_file_ex.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int32]
        L.kwh_checkpoint_write.argtypes = [C.c_void_p, C.c_char_p]
        L.kwh_checkpoint_read.argtypes = [C.c_void_p, C.c_char_p]
        _h5 = L
    return _h5


def _h5check(rc: int):
    if rc != 0:
        raise capi.KWaveError(load_h5().kwh_last_error().decode(errors="replace"))


VECTOR_AXIS = {"ddx_k_shift_pos_r": "x", "ddx_k_shift_neg_r": "x", "x_shift_neg_r": "x", "pml_x": "x", "pml_x_sgx": "x",
               "dxudxn": "x", "dxudxn_sgx": "x",
               "ddy_k_shift_pos": "y", "ddy_k_shift_neg": "y", "y_shift_neg_r": "y", "pml_y": "y", "pml_y_sgy": "y",
               "dyudyn": "y", "dyudyn_sgy": "y",
               "ddz_k_shift_pos": "z", "ddz_k_shift_neg": "z", "z_shift_neg_r": "z", "pml_z": "z", "pml_z_sgz": "z",
               "dzudzn": "z", "dzudzn_sgz": "z"}


def write_input_file(pr: Dict[str, np.ndarray], path: str) -> None:
    """Write a problem dict (HDF5 dataset names, 1-based indices) as a k-Wave input file."""
    L = load_h5()
    keep = []
    sets = (Dataset * len(pr))()
    cplx = (C.c_int32 * len(pr))()
    for i, (name, a) in enumerate(pr.items()):
        arr = np.ascontiguousarray(a, dtype=np.uint64 if a.dtype == np.uint64 else np.float32)
        keep.append(arr)
        is_c = name in COMPLEX_DATASETS
        shp = list(arr.shape)
        # 1-D operators / PML / grid-derivative vectors keep their axis in the file, as k-Wave writes them: an x vector is
        # (n,1,1), a y vector (1,n,1), a z vector (1,1,n) in (x,y,z) order; complex ones double the fastest (x) dimension
        axis = VECTOR_AXIS.get(name)
        if axis is not None:
            n = arr.size // (2 if is_c else 1)
            shp = {"x": [1, 1, n * (2 if is_c else 1)], "y": [1, n, 2 if is_c else 1], "z": [n, 1, 2 if is_c else 1]}[axis]
        shp = shp[::-1]
        while len(shp) < 3:
            shp.append(1)
        nm = name.encode()
        keep.append(nm)
        sets[i].name, sets[i].data, sets[i].dtype = nm, arr.ctypes.data, (1 if arr.dtype == np.uint64 else 0)
        sets[i].nx, sets[i].ny, sets[i].nz = shp
        cplx[i] = int(is_c)
    _h5check(L.kwh_write_input_file(path.encode(), sets, len(pr), cplx))


def dataset_info(path: str, name: str):
    dims = (C.c_uint64 * 3)()
    dt, cx = C.c_int32(), C.c_int32()
    _h5check(load_h5().kwh_h5_dataset_info(path.encode(), name.encode(), C.byref(dims), C.byref(dt), C.byref(cx)))
    return tuple(int(d) for d in dims), ("long" if dt.value else "float"), ("complex" if cx.value else "real")


def dataset_info_4d(path: str, name: str):
    """((nx, ny, nz, nt), type, domain); nt = 0 for a 3-D dataset."""
    dims = (C.c_uint64 * 4)()
    dt, cx = C.c_int32(), C.c_int32()
    _h5check(load_h5().kwh_h5_dataset_info_4d(path.encode(), name.encode(), C.byref(dims), C.byref(dt), C.byref(cx)))
    return tuple(int(d) for d in dims), ("long" if dt.value else "float"), ("complex" if cx.value else "real")


def read_dataset(path: str, name: str) -> np.ndarray:
    """3-D datasets as [nz][ny][nx]; the per-cuboid series of a corners mask ("p/1") as [nt][nz][ny][nx]."""
    (nx, ny, nz, nt), dtype, _ = dataset_info_4d(path, name)
    out = np.empty((nt, nz, ny, nx) if nt else (nz, ny, nx), dtype=np.uint64 if dtype == "long" else np.float32)
    _h5check(load_h5().kwh_h5_read(path.encode(), name.encode(), out.ctypes.data, out.size, 1 if dtype == "long" else 0))
    return out


# every dataset a k-Wave 1.1 input file may hold (main.cpp:446-563, Utils/MatrixNames.h:48-275)
INPUT_DATASETS = (
    "Nx", "Ny", "Nz", "Nt", "dt", "dx", "dy", "dz", "c_ref", "pml_x_size", "pml_y_size", "pml_z_size", "pml_x_alpha",
    "pml_y_alpha", "pml_z_alpha", "ux_source_flag", "uy_source_flag", "uz_source_flag", "p_source_flag", "p0_source_flag",
    "transducer_source_flag", "nonuniform_grid_flag", "nonlinear_flag", "absorbing_flag", "sensor_mask_type",
    "u_source_mode", "p_source_mode", "u_source_many", "p_source_many", "alpha_power",
    "c0", "rho0", "rho0_sgx", "rho0_sgy", "rho0_sgz", "BonA", "alpha_coeff",
    "ddx_k_shift_pos_r", "ddx_k_shift_neg_r", "ddy_k_shift_pos", "ddy_k_shift_neg", "ddz_k_shift_pos", "ddz_k_shift_neg",
    "x_shift_neg_r", "y_shift_neg_r", "z_shift_neg_r", "pml_x", "pml_x_sgx", "pml_y", "pml_y_sgy", "pml_z", "pml_z_sgz",
    "dxudxn", "dyudyn", "dzudzn", "dxudxn_sgx", "dyudyn_sgy", "dzudzn_sgz",
    "p0_source_input", "p_source_index", "p_source_input", "u_source_index", "ux_source_input", "uy_source_input",
    "uz_source_input", "transducer_source_input", "delay_mask", "sensor_mask_index", "sensor_mask_corners")


def dataset_exists(path: str, name: str) -> bool:
    L = load_h5()
    L.kwh_h5_dataset_exists.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32)]
    v = C.c_int32()
    _h5check(L.kwh_h5_dataset_exists(path.encode(), name.encode(), C.byref(v)))
    return bool(v.value)


def read_planes(path: str, name: str, z0: int, z1: int) -> np.ndarray:
    """Planes [z0, z1) of a grid-sized float dataset as [z1 - z0][ny][nx]."""
    (nx, ny, _nz), _, _ = dataset_info(path, name)
    out = np.empty((z1 - z0, ny, nx), dtype=np.float32)
    L = load_h5()
    L.kwh_h5_read_planes.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p]
    _h5check(L.kwh_h5_read_planes(path.encode(), name.encode(), z0, z1 - z0, out.ctypes.data))
    return out


def read_problem(path: str, zslab=None) -> Dict[str, np.ndarray]:
    """The datasets of an input file as a problem dict (what HostSolver / dist.partition_problem take).  zslab = (z0, z1)
    reads only those planes of the grid-sized arrays (one rank of a slab-decomposed run; pass the result to
    partition_problem(..., arrays_are_local=True))."""
    pr: Dict[str, np.ndarray] = {}
    dims = tuple(int(read_dataset(path, k).ravel()[0]) for k in ("Nx", "Ny", "Nz"))
    for name in INPUT_DATASETS:
        if not dataset_exists(path, name):
            continue
        shape, dtype, _ = dataset_info(path, name)
        if zslab is not None and shape == dims and dtype == "float" and dims[2] > 1:
            pr[name] = read_planes(path, name, int(zslab[0]), int(zslab[1]))
        else:
            pr[name] = read_dataset(path, name)
    return pr


def write_file(datasets: Dict[str, np.ndarray], path: str, file_type: str = "output",
               description: str = "k-Wave output written by kwave_amd") -> None:
    """Write named arrays (shape = HDF5 dims, i.e. (z, y, x) order; float32 or uint64) as a k-Wave file of the given type."""
    L = load_h5()
    L.kwh_write_file.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Dataset), C.c_size_t, C.POINTER(C.c_int32)]
    keep = []
    sets = (Dataset * len(datasets))()
    cplx = (C.c_int32 * len(datasets))()
    for i, (name, a) in enumerate(datasets.items()):
        a = np.asarray(a)
        arr = np.ascontiguousarray(a, dtype=np.uint64 if a.dtype == np.uint64 else np.float32)
        shp = list(arr.shape)[::-1]
        while len(shp) < 3:
            shp.append(1)
        if len(shp) != 3:
            raise ValueError(f"{name}: at most 3 dimensions")
        nm = name.encode()
        keep += [arr, nm]
        sets[i].name, sets[i].data, sets[i].dtype = nm, arr.ctypes.data, (1 if arr.dtype == np.uint64 else 0)
        sets[i].nx, sets[i].ny, sets[i].nz = shp
    _h5check(L.kwh_write_file(path.encode(), file_type.encode(), description.encode(), sets, len(datasets), cplx))


def append_cuboid(path: str, group: str, index: int, data: np.ndarray, series: bool) -> None:
    """Add dataset "<group>/<index>" (1-based) to an existing output file: data shaped (steps, nz, ny, nx) for a series,
    (nz, ny, nx) for an aggregate — the per-cuboid layout of a corner sensor mask (CuboidOutputStream.cpp:95-140)."""
    L = load_h5()
    L.kwh_h5_append_cuboid.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64 * 4), C.c_void_p]
    a = np.ascontiguousarray(data, dtype=np.float32)
    nz, ny, nx = a.shape[-3:]
    dims = (C.c_uint64 * 4)(nx, ny, nz, a.shape[0] if series else 0)
    _h5check(L.kwh_h5_append_cuboid(path.encode(), group.encode(), index, C.byref(dims), a.ctypes.data))


def read_attribute(path: str, dataset: str, attr: str) -> str:
    buf = C.create_string_buffer(256)
    _h5check(load_h5().kwh_h5_read_attribute(path.encode(), dataset.encode(), attr.encode(), buf, 256))
    return buf.value.decode()


def read_numeric_attribute(path: str, dataset: str, attr: str) -> float:
    v = C.c_double()
    L = load_h5()
    L.kwh_h5_read_numeric_attribute.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]
    _h5check(L.kwh_h5_read_numeric_attribute(path.encode(), dataset.encode(), attr.encode(), C.byref(v)))
    return float(v.value)


class FileSolver(HostSolver):
    """HostSolver created from a k-Wave HDF5 input file instead of in-memory datasets.

    output=path opens the output file before the first step: every stored time series is appended to it step by step
    (the reference's per-step hyperslab writes) instead of being kept in host memory; write_output(path) completes it.
    reopen_output=True continues the output file of a checkpointed run (then call read_checkpoint)."""

    def __init__(self, path: str, output: Optional[str] = None, compression_level: int = 0, reopen_output: bool = False,
                 **opts):
        L = load_h5()
        self._keep = []
        o = Options()
        o.device_idx = opts.pop("device_idx", -1)
        o.fused_kernels = int(opts.pop("fused_kernels", True))
        o.sampling_start_time_index = opts.pop("sampling_start", 0)
        o.benchmark_time_steps = opts.pop("benchmark_steps", 0)
        o.period = float(opts.pop("period", 0.0))
        o.mos = int(opts.pop("mos", 1))
        o.harmonics = int(opts.pop("harmonics", 1))
        for k, v in opts.items():
            setattr(o, k, int(v))
        self.nx, self.ny, self.nz = (int(read_dataset(path, k).ravel()[0]) for k in ("Nx", "Ny", "Nz"))
        h = C.c_void_p()
        _h5check(L.kwh_create_from_file(path.encode(), C.byref(o), C.byref(h)))
        self._h = h
        self.L = L  # the h5 library exports the whole kwh_* API
        for fn, res, args in (("kwh_time_index", C.c_uint64, [C.c_void_p]), ("kwh_context", C.c_void_p, [C.c_void_p])):
            getattr(L, fn).restype = res
            getattr(L, fn).argtypes = args
        L.kwh_run.argtypes = [C.c_void_p, C.c_uint64]
        L.kwh_get_matrix.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        L.kwh_matrix_size.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64)]
        L.kwh_get_scalar.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float)]
        L.kwh_stream_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.kwh_stream_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]
        for fn in ("kwh_destroy", "kwh_finish", "kwh_sync"):
            getattr(L, fn).argtypes = [C.c_void_p]
        self.output = output
        if output is not None:
            L.kwh_open_output_file.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_int32]
            _h5check(L.kwh_open_output_file(self._h, output.encode(), compression_level, int(reopen_output)))

    def write_output(self, path: str, compression_level: int = 0, copy_sensor_mask: bool = False):
        _h5check(self.L.kwh_write_output_file_ex(self._h, path.encode(), compression_level, int(copy_sensor_mask)))

    def post_process(self, path: str):
        """--post: compute I_avg / I_avg_c / Q_term / Q_term_c from the series stored in the output file `path` and add
        them to it (the solver must have been created with only_post_processing=1)."""
        _h5check(self.L.kwh_post_process_output_file(self._h, path.encode()))

    def write_checkpoint(self, path: str):
        """State arrays, time index and stream states -> checkpoint file (KSpaceFirstOrderSolver.cpp:1176-1224)."""
        _h5check(self.L.kwh_checkpoint_write(self._h, path.encode()))

    def read_checkpoint(self, path: str):
        """Recover from a checkpoint file; run() then continues from its time index (…Solver.cpp:186-228)."""
        _h5check(self.L.kwh_checkpoint_read(self._h, path.encode()))
