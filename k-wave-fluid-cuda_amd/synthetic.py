"""Synthetic k-Wave input generator (file format 1.1 datasets).

Produces, as a dict of NumPy arrays keyed by the reference's HDF5 dataset names
(/root/reference/Utils/MatrixNames.h:48-275, main.cpp:446-563), exactly what the reference would
read from an input file: grid scalars, medium, k-space derivative/shift operators, PML vectors,
sources and the sensor mask.  Index datasets are 1-based like in the file
(/root/reference/MatrixClasses/IndexMatrix.cpp:161-168 converts them at load time).

The operator / PML formulas are the standard k-Wave MATLAB conventions; they are *data* for the
solver (the reference reads them from the file and never computes them).

Arrays are stored [z][y][x] (x fastest), i.e. HDF5 dims (Nz, Ny, Nx).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

F32 = np.float32
U64 = np.uint64


def _kvec(n: int, d: float) -> np.ndarray:
    """k-Wave wavenumber vector in FFT order (float64)."""
    if n == 1:
        return np.zeros(1)
    if n % 2 == 0:
        idx = np.arange(-n // 2, n // 2)
    else:
        idx = np.arange(-(n - 1) // 2, (n - 1) // 2 + 1)
    k = (2.0 * math.pi / (n * d)) * idx
    return np.fft.ifftshift(k)


def _cplx_to_f32_pairs(c: np.ndarray) -> np.ndarray:
    out = np.empty(c.shape + (2,), dtype=F32)
    out[..., 0] = c.real
    out[..., 1] = c.imag
    return out


def kspace_operators(nx: int, ny: int, nz: int, dx: float, dy: float, dz: float) -> Dict[str, np.ndarray]:
    """ddx/ddy/ddz_k_shift_{pos,neg}[_r] and {x,y,z}_shift_neg_r (interleaved complex float32)."""
    ops: Dict[str, np.ndarray] = {}
    for ax, n, d in (("x", nx, dx), ("y", ny, dy), ("z", nz, dz)):
        k = _kvec(n, d)
        pos = 1j * k * np.exp(1j * k * d / 2.0)
        neg = 1j * k * np.exp(-1j * k * d / 2.0)
        shift = np.exp(-1j * k * d / 2.0)
        nr = n // 2 + 1
        if ax == "x":
            ops["ddx_k_shift_pos_r"] = _cplx_to_f32_pairs(pos[:nr])
            ops["ddx_k_shift_neg_r"] = _cplx_to_f32_pairs(neg[:nr])
        else:
            ops[f"dd{ax}_k_shift_pos"] = _cplx_to_f32_pairs(pos)
            ops[f"dd{ax}_k_shift_neg"] = _cplx_to_f32_pairs(neg)
        ops[f"{ax}_shift_neg_r"] = _cplx_to_f32_pairs(shift[:nr])
    return ops


def pml_vectors(n: int, d: float, dt: float, c_ref: float, size: int, alpha: float, staggered: bool) -> np.ndarray:
    """k-Wave getPML (PML inside the grid); returns float32 [n]."""
    pml = np.ones(n, dtype=np.float64)
    if size <= 0:
        return pml.astype(F32)
    x = np.arange(1, size + 1, dtype=np.float64)
    if staggered:
        left = alpha * (c_ref / d) * (((x + 0.5) - size - 1.0) / (0.0 - size)) ** 4
        right = alpha * (c_ref / d) * ((x + 0.5) / size) ** 4
    else:
        left = alpha * (c_ref / d) * ((x - size - 1.0) / (0.0 - size)) ** 4
        right = alpha * (c_ref / d) * (x / size) ** 4
    pml[:size] = np.exp(-left * dt / 2.0)
    pml[n - size:] = np.exp(-right * dt / 2.0)
    return pml.astype(F32)


def _grid(nx, ny, nz, zlo=0, zhi=None):
    """Open (broadcastable) coordinate grids x[1,1,nx], y[1,ny,1], z[nzl,1,1]: every closed form below is evaluated on
    the axes it depends on and expanded by _full(); elementwise this is the arithmetic of dense grids (same operations,
    same order, hence the same bits) at a fraction of the time and memory for 512^3."""
    zhi = nz if zhi is None else zhi
    z, y, x = np.meshgrid(np.arange(zlo, zhi, dtype=np.float64), np.arange(ny, dtype=np.float64),
                          np.arange(nx, dtype=np.float64), indexing="ij", sparse=True)
    return x, y, z


def _full(a: np.ndarray, shape, nzl: int) -> np.ndarray:
    """float32 dense array [nzl][ny][nx] of a broadcastable float64 field"""
    return np.broadcast_to(a, shape)[:nzl].astype(F32)


def _c0_field(x, y, z, nx, ny, nz):
    two_pi = 2.0 * math.pi
    n = max(nx, ny, nz)
    c0 = 1500.0 * (1.0 + 0.05 * np.sin(two_pi * 3 * x / nx) * np.cos(two_pi * 2 * y / ny) * np.cos(two_pi * z / nz))
    r2 = (x - nx // 2) ** 2 + (y - ny // 2) ** 2 + (z - nz // 2) ** 2
    return np.where(r2 <= (n / 6.0) ** 2, 1600.0, c0)


def _by_planes(fn, x, y, z, nzl: int, planes: int = 4) -> np.ndarray:
    """float32 [nzl][ny][nx] of a genuinely 3-D closed form fn(x, y, z) -> float64, evaluated a few planes at a time so
    that the float64 temporaries stay cache-sized (elementwise: same bits as one dense evaluation)."""
    out = np.empty((nzl, y.shape[1], x.shape[2]), dtype=F32)
    for a in range(0, nzl, planes):
        b = min(a + planes, nzl)
        out[a:b] = fn(x, y, z[a:b])
    return out


def _sg_mean(a: np.ndarray, axis: int) -> np.ndarray:
    """staggered-grid value = mean of the two neighbours along axis (edge replicated)."""
    if a.shape[axis] == 1:  # the field does not depend on this axis: 0.5 * (a + a) == a exactly
        return a
    nxt = np.concatenate([np.take(a, range(1, a.shape[axis]), axis=axis),
                          np.take(a, [a.shape[axis] - 1], axis=axis)], axis=axis)
    return 0.5 * (a + nxt)


def make_problem(nx: int, ny: Optional[int] = None, nz: Optional[int] = None, *,
                 heterogeneous: bool = True, nonlinear: bool = True, absorbing: bool = True,
                 source: str = "p0", nt: int = 110, pml_size: int = 10, pml_alpha: float = 2.0,
                 pml_off: bool = False, dx: float = 2.0e-4, cfl: float = 0.3,
                 source_mode: int = 0, source_many: int = 0, sensor: str = "plane",
                 hetero_subset: Optional[dict] = None, seed: int = 0x5EED1234,
                 nt_src: Optional[int] = None, zslab: Optional[tuple] = None,
                 nonuniform: bool = False) -> Dict[str, np.ndarray]:
    """Build one synthetic problem.

    source: "p0" (1 MPa Gaussian ball), "p_source" (1 MHz tone burst on plane x=12),
            "u_source" (velocity source on the same plane, ux only), "transducer", or "none".
    hetero_subset: optionally {"c0": bool, "rho0": bool, "BonA": bool, "alpha_coeff": bool} to mix
            scalar / array medium parameters (detected per dataset by shape in the reference:
            /root/reference/Parameters/Parameters.cpp:426-459).
    zslab:  (z_lo, z_hi): build the 3-D arrays only for planes z_lo <= z < z_hi (multi-GPU runs generate their own
            slab; every other dataset — scalars, operators, PML vectors, index masks — stays global and is cut to
            the slab by dist.partition_problem(..., arrays_are_local=True)).
    """
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    dy = dz = dx
    zlo, zhi = (0, nz) if zslab is None else (int(zslab[0]), int(zslab[1]))
    zext = min(zhi + 1, nz)  # one extra plane for the staggered-grid mean along z
    x, y, z = _grid(nx, ny, nz, zlo, zext)
    nzl = zhi - zlo
    shape = (zext - zlo, ny, nx)
    two_pi = 2.0 * math.pi
    het = {"c0": heterogeneous, "rho0": heterogeneous, "BonA": heterogeneous, "alpha_coeff": heterogeneous}
    if hetero_subset:
        het.update(hetero_subset)

    pr: Dict[str, np.ndarray] = {}

    def scalar_f(v):
        return np.array([[[v]]], dtype=F32)

    def scalar_u(v):
        return np.array([[[v]]], dtype=U64)

    # ---- medium -------------------------------------------------------------------------------
    if het["c0"]:
        pr["c0"] = _by_planes(lambda xx, yy, zz: _c0_field(xx, yy, zz, nx, ny, nz), x, y, z, nzl)
        if zslab is None:
            c_ref = float(pr["c0"].max())
        else:  # c_ref is the global maximum: evaluate the closed form plane by plane
            c_ref = 0.0
            for zz in range(nz):
                xp, yp, zp = _grid(nx, ny, nz, zz, zz + 1)
                c_ref = max(c_ref, float(_c0_field(xp, yp, zp, nx, ny, nz).astype(F32).max()))
    else:
        pr["c0"] = scalar_f(1500.0)
        c_ref = 1500.0
    if het["rho0"]:
        rho0 = 1000.0 * (1.0 + 0.04 * np.cos(two_pi * 2 * x / nx) * np.sin(two_pi * 3 * z / nz))
        pr["rho0"] = _full(rho0, shape, nzl)
        pr["rho0_sgx"] = _full(_sg_mean(rho0, 2), shape, nzl)
        pr["rho0_sgy"] = _full(_sg_mean(rho0, 1), shape, nzl)
        pr["rho0_sgz"] = _full(_sg_mean(rho0, 0), shape, nzl)
    else:
        for nm in ("rho0", "rho0_sgx", "rho0_sgy", "rho0_sgz"):
            pr[nm] = scalar_f(1000.0)
    if nonlinear:
        pr["BonA"] = _full(6.0 + 2.0 * np.sin(two_pi * y / ny), shape, nzl) if het["BonA"] else scalar_f(6.0)
    if absorbing:
        pr["alpha_coeff"] = (_full(0.75 + 0.25 * np.cos(two_pi * x / nx), shape, nzl)
                             if het["alpha_coeff"] else scalar_f(0.75))
        pr["alpha_power"] = scalar_f(1.5)

    dt = float(F32(cfl * dx / c_ref))

    # ---- grid scalars / flags -----------------------------------------------------------------
    pr["Nx"], pr["Ny"], pr["Nz"], pr["Nt"] = scalar_u(nx), scalar_u(ny), scalar_u(nz), scalar_u(nt)
    pr["dt"], pr["dx"], pr["dy"], pr["dz"] = scalar_f(dt), scalar_f(dx), scalar_f(dy), scalar_f(dz)
    pr["c_ref"] = scalar_f(c_ref)
    for ax in "xyz":
        pr[f"pml_{ax}_size"] = scalar_u(0 if pml_off else pml_size)
        pr[f"pml_{ax}_alpha"] = scalar_f(pml_alpha)
    pr["nonuniform_grid_flag"] = scalar_u(int(nonuniform))
    if nonuniform:
        # derivative scalings of a smoothly stretched grid (k-Wave's makeGrid/setNUGrid writes d(uniform)/d(non-uniform) on
        # the regular and on the staggered points); any positive vectors exercise the kernels
        for ax, nn in (("x", nx), ("y", ny), ("z", nz)):
            t = np.arange(nn, dtype=np.float64)
            shape = {"x": (1, 1, nn), "y": (1, nn, 1), "z": (nn, 1, 1)}[ax]
            pr[f"d{ax}ud{ax}n"] = (1.0 + 0.15 * np.sin(two_pi * t / nn)).astype(F32).reshape(shape)
            pr[f"d{ax}ud{ax}n_sg{ax}"] = (1.0 + 0.15 * np.sin(two_pi * (t + 0.5) / nn)).astype(F32).reshape(shape)
    pr["nonlinear_flag"] = scalar_u(int(nonlinear))
    pr["absorbing_flag"] = scalar_u(int(absorbing))

    # ---- operators + PML ----------------------------------------------------------------------
    pr.update(kspace_operators(nx, ny, nz, dx, dy, dz))
    eff = 0 if pml_off else pml_size
    for ax, nn, dd in (("x", nx, dx), ("y", ny, dy), ("z", nz, dz)):
        size = 0 if nn == 1 else eff  # a 2-D grid (Nz == 1) has no PML along z
        pr[f"pml_{ax}"] = pml_vectors(nn, dd, dt, c_ref, size, pml_alpha, False)
        pr[f"pml_{ax}_sg{ax}"] = pml_vectors(nn, dd, dt, c_ref, size, pml_alpha, True)

    # ---- sources ------------------------------------------------------------------------------
    for nm in ("ux_source_flag", "uy_source_flag", "uz_source_flag", "p_source_flag", "p0_source_flag",
               "transducer_source_flag"):
        pr[nm] = scalar_u(0)
    rng = np.random.default_rng(seed)
    if source == "p0":
        sigma = 4.0

        def ball(xx, yy, zz):
            r2 = (xx - nx // 2) ** 2 + (yy - ny // 2) ** 2 + (zz - nz // 2) ** 2
            return 1.0e6 * np.exp(-r2 / (2.0 * sigma * sigma))
        pr["p0_source_input"] = _by_planes(ball, x, y, z, nzl)
        pr["p0_source_flag"] = scalar_u(1)
    elif source in ("p_source", "u_source", "transducer"):
        xs = min(12, nx - 1)
        yy, zz = np.meshgrid(np.arange(ny), np.arange(nz), indexing="xy")
        # linear (MATLAB, 1-based) indices of plane x = xs, restricted to a centred patch
        y0, y1 = ny // 4, ny - ny // 4
        z0, z1 = nz // 4, nz - nz // 4
        sel = (yy >= y0) & (yy < y1) & (zz >= z0) & (zz < z1)
        lin = (zz[sel].astype(np.int64) * ny + yy[sel].astype(np.int64)) * nx + xs
        lin = np.sort(lin).astype(U64) + U64(1)
        nsrc = lin.size
        nt_src = nt if nt_src is None else nt_src
        t = np.arange(nt_src, dtype=np.float64) * dt
        f0 = 1.0e6
        env = np.minimum(1.0, t * f0 / 3.0)  # 3-cycle ramp
        sig = np.sin(two_pi * f0 * t) * env
        if source == "p_source":
            amp = 1.0e5 / (3.0 * c_ref * c_ref) if source_mode != 0 else 1.0e5 / (3.0 * c_ref * c_ref)
            base = (amp * sig).astype(F32)
            if source_many:
                w = (1.0 + 0.1 * rng.standard_normal(nsrc)).astype(F32)
                pr["p_source_input"] = (base[:, None] * w[None, :]).reshape(1, nt_src, nsrc)
            else:
                pr["p_source_input"] = base.reshape(1, nt_src, 1)
            pr["p_source_index"] = lin.reshape(1, 1, nsrc)
            pr["p_source_flag"] = scalar_u(nt_src)
            pr["p_source_mode"] = scalar_u(source_mode)
            pr["p_source_many"] = scalar_u(source_many)
        elif source == "u_source":
            base = (0.05 * sig).astype(F32)
            if source_many:
                w = (1.0 + 0.1 * rng.standard_normal(nsrc)).astype(F32)
                pr["ux_source_input"] = (base[:, None] * w[None, :]).reshape(1, nt_src, nsrc)
            else:
                pr["ux_source_input"] = base.reshape(1, nt_src, 1)
            pr["u_source_index"] = lin.reshape(1, 1, nsrc)
            pr["ux_source_flag"] = scalar_u(nt_src)
            pr["u_source_mode"] = scalar_u(source_mode)
            pr["u_source_many"] = scalar_u(source_many)
        else:  # transducer
            delays = (np.abs(np.arange(nsrc) % 16 - 8)).astype(U64)  # focusing-like delay pattern
            nsig = nt_src + int(delays.max()) + 1
            tt = np.arange(nsig, dtype=np.float64) * dt
            sig_t = np.sin(two_pi * f0 * tt) * np.minimum(1.0, tt * f0 / 3.0)
            pr["transducer_source_input"] = (0.05 * sig_t).astype(F32).reshape(1, 1, nsig)
            pr["delay_mask"] = (delays + U64(1)).reshape(1, 1, nsrc)
            pr["u_source_index"] = lin.reshape(1, 1, nsrc)
            pr["transducer_source_flag"] = scalar_u(nt_src)
            pr["u_source_mode"] = scalar_u(0)
            pr["u_source_many"] = scalar_u(0)
    elif source != "none":
        raise ValueError(source)

    # ---- sensor -------------------------------------------------------------------------------
    pr["sensor_mask_type"] = scalar_u(0)
    if sensor == "plane":
        zs = nz // 2
        yy, xx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        lin = (zs * ny + yy.ravel().astype(np.int64)) * nx + xx.ravel().astype(np.int64)
    elif sensor == "random":
        lin = np.sort(rng.choice(nx * ny * nz, size=min(4096, nx * ny * nz // 4), replace=False))
    else:
        raise ValueError(sensor)
    pr["sensor_mask_index"] = (lin.astype(U64) + U64(1)).reshape(1, 1, -1)
    return pr


def is_scalar(a: np.ndarray) -> bool:
    return a.size == 1


# datasets a k-Wave 2-D input file (Nz == 1) does not contain (MatrixContainer.cpp:102-200 creates them only for 3-D)
Z_ONLY_DATASETS = ("ddz_k_shift_pos", "ddz_k_shift_neg", "z_shift_neg_r", "pml_z", "pml_z_sgz", "rho0_sgz",
                   "uz_source_flag", "transducer_source_flag", "pml_z_size", "pml_z_alpha", "dz")


def as_2d_file(pr: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """A problem built with nz == 1, reduced to what a 2-D input file holds (no z dataset at all, dz included:
    Parameters/Parameters.cpp:240-259 reads dz, pml_z_size and pml_z_alpha only for 3-D)."""
    if int(np.asarray(pr["Nz"]).ravel()[0]) != 1:
        raise ValueError("as_2d_file needs a problem with Nz == 1")
    return {k: v for k, v in pr.items() if k not in Z_ONLY_DATASETS}
