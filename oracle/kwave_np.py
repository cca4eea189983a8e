"""Independent fp64 NumPy restatement of the per-step algorithm.

TEST INFRASTRUCTURE ONLY.  Its purpose is to bound the error of the fp32 C oracle
(oracle/kwave_oracle.c) — test K4 of SURVEY.md §8c — and to provide the closed-form K1 check.
It follows the same reference lines as the C oracle (KSpaceSolver/KSpaceFirstOrderSolver.cpp:864-943,
2087-2396, 2404-2703; KSpaceSolver/SolverCudaKernels.cu per kernel) but is written against
numpy.fft.rfftn/irfftn in float64, with whole-array expressions, so an indexing or ordering slip in one
restatement does not repeat in the other.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np


def _sc(a) -> float:
    return float(np.asarray(a).reshape(-1)[0])


def _c(pairs: np.ndarray) -> np.ndarray:
    p = np.asarray(pairs, dtype=np.float64).reshape(-1, 2)
    return p[:, 0] + 1j * p[:, 1]


def complete_2d(pr: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """A 2-D problem (Nz == 1) as the degenerate 3-D one both restatements run: the z datasets a 2-D input file does not
    have, with the values that make the 3-D formulas reduce to the reference's SD::k2D ones (zero z-gradient operators,
    unit z-PML and z-shift, rho0_sgz = rho0_sgx; u_z and rho_z then stay zero)."""
    if int(_sc(pr["Nz"])) != 1:
        return pr
    out = dict(pr)
    zero_c = np.zeros((1, 1, 2), dtype=np.float32)
    one = np.ones((1, 1, 1), dtype=np.float32)
    out.setdefault("ddz_k_shift_pos", zero_c)
    out.setdefault("ddz_k_shift_neg", zero_c)
    out.setdefault("z_shift_neg_r", np.array([[[1.0, 0.0]]], dtype=np.float32))
    out.setdefault("pml_z", one)
    out.setdefault("pml_z_sgz", one)
    out.setdefault("rho0_sgz", pr["rho0_sgx"])
    out.setdefault("dz", pr["dx"])  # never used: the only z bin is 0 (the reference sets 1/dz^2 = 0 in 2-D)
    return out


class NumpySim:
    def __init__(self, pr: Dict[str, np.ndarray]):
        pr = complete_2d(pr)
        self.pr = pr
        nx, ny, nz = (int(_sc(pr[k])) for k in ("Nx", "Ny", "Nz"))
        self.nx, self.ny, self.nz = nx, ny, nz
        self.shape = (nz, ny, nx)
        self.N = nx * ny * nz
        self.dt = _sc(pr["dt"])
        self.nonlinear = int(_sc(pr["nonlinear_flag"]))
        self.absorbing = int(_sc(pr["absorbing_flag"]))
        f8 = lambda k: np.asarray(pr[k], dtype=np.float64)
        self.c0 = f8("c0") if pr["c0"].size > 1 else _sc(pr["c0"])
        self.rho0 = f8("rho0") if pr["rho0"].size > 1 else _sc(pr["rho0"])
        self.dtrho = [self.dt / (f8(k) if pr[k].size > 1 else _sc(pr[k])) for k in ("rho0_sgx", "rho0_sgy", "rho0_sgz")]
        self.bona = (f8("BonA") if pr["BonA"].size > 1 else _sc(pr["BonA"])) if "BonA" in pr else 0.0
        # non-uniform grid (MatrixContainer.cpp:301-329): dt/rho0_sg carries the staggered-grid derivative scaling
        # (KSpaceFirstOrderSolver.cpp:2650-2685 / SolverCudaKernels.cu:372-410), the velocity gradient the regular one
        self.nonuniform = int(_sc(pr.get("nonuniform_grid_flag", 0)))
        self.dudn = None
        if self.nonuniform:
            sg = [f8("dxudxn_sgx").reshape(1, 1, -1), f8("dyudyn_sgy").reshape(1, -1, 1), f8("dzudzn_sgz").reshape(-1, 1, 1)]
            self.dtrho = [self.dtrho[a] * sg[a] for a in range(3)]
            self.dudn = [f8("dxudxn").reshape(1, 1, -1), f8("dyudyn").reshape(1, -1, 1), f8("dzudzn").reshape(-1, 1, 1)]
        # operators, broadcast shapes
        self.ddx_pos = _c(pr["ddx_k_shift_pos_r"]).reshape(1, 1, -1)
        self.ddy_pos = _c(pr["ddy_k_shift_pos"]).reshape(1, -1, 1)
        self.ddz_pos = _c(pr["ddz_k_shift_pos"]).reshape(-1, 1, 1)
        self.ddx_neg = _c(pr["ddx_k_shift_neg_r"]).reshape(1, 1, -1)
        self.ddy_neg = _c(pr["ddy_k_shift_neg"]).reshape(1, -1, 1)
        self.ddz_neg = _c(pr["ddz_k_shift_neg"]).reshape(-1, 1, 1)
        self.pml = [f8("pml_x").reshape(1, 1, -1), f8("pml_y").reshape(1, -1, 1), f8("pml_z").reshape(-1, 1, 1)]
        self.pml_sg = [f8("pml_x_sgx").reshape(1, 1, -1), f8("pml_y_sgy").reshape(1, -1, 1),
                       f8("pml_z_sgz").reshape(-1, 1, 1)]
        # generators (KSpaceFirstOrderSolver.cpp:2404-2643)
        dx, dy, dz, c_ref = (_sc(pr[k]) for k in ("dx", "dy", "dz", "c_ref"))
        fx = 0.5 - np.abs(0.5 - np.arange(nx // 2 + 1) / nx)
        fy = 0.5 - np.abs(0.5 - np.arange(ny) / ny)
        fz = 0.5 - np.abs(0.5 - np.arange(nz) / nz)
        kk = np.sqrt((fz ** 2 / dz ** 2).reshape(-1, 1, 1) + (fy ** 2 / dy ** 2).reshape(1, -1, 1)
                     + (fx ** 2 / dx ** 2).reshape(1, 1, -1))
        arg = c_ref * self.dt * math.pi * kk
        with np.errstate(divide="ignore", invalid="ignore"):
            self.kappa = np.where(arg == 0.0, 1.0, np.sin(arg) / arg)
        self.source_kappa = np.cos(arg)
        if self.absorbing:
            y = _sc(pr["alpha_power"])
            k2pi = 2.0 * math.pi * kk
            with np.errstate(divide="ignore"):
                n1 = np.power(k2pi, y - 2.0)
                n2 = np.power(k2pi, y - 1.0)
            n1[np.isinf(n1)] = 0.0
            n2[np.isinf(n2)] = 0.0
            self.nabla1, self.nabla2 = n1, n2
            a_np = 100.0 * (1.0e-6 / (2.0 * math.pi)) ** y / (20.0 * math.log10(math.e))
            alpha = f8("alpha_coeff") if pr["alpha_coeff"].size > 1 else _sc(pr["alpha_coeff"])
            a2 = 2.0 * a_np * alpha
            self.tau = -a2 * np.power(self.c0, y - 1.0)
            self.eta = a2 * np.power(self.c0, y) * math.tan(math.pi * y / 2.0)
        self.c2 = self.c0 * self.c0
        z = lambda: np.zeros(self.shape)
        self.p = z()
        self.u = [z(), z(), z()]
        self.rho = [z(), z(), z()]
        self.du = [z(), z(), z()]
        self.t = 0

    # helpers
    def F(self, a):
        return np.fft.rfftn(a, axes=(0, 1, 2))

    def Fi(self, c):
        # unnormalised inverse (cuFFT C2R), the 1/N lives in the consumers
        return np.fft.irfftn(c, s=self.shape, axes=(0, 1, 2)) * self.N

    def _src_values(self, inp, n, many):
        inp = np.asarray(inp, dtype=np.float64).reshape(-1)
        return inp[self.t * n:(self.t + 1) * n] if many else np.full(n, inp[self.t])

    def _scaled(self, inp, idx, many):
        T = np.zeros(self.N)
        T[idx] = self._src_values(inp, idx.size, many)
        S = self.F(T.reshape(self.shape)) * self.source_kappa / self.N
        return self.Fi(S)

    def step(self):
        pr, d = self.pr, 1.0 / self.N
        # A1-A4
        e = self.F(self.p) * self.kappa
        g = [self.Fi(e * self.ddx_pos), self.Fi(e * self.ddy_pos), self.Fi(e * self.ddz_pos)]
        for a in range(3):
            self.u[a] = (self.u[a] * self.pml_sg[a] - d * g[a] * self.dtrho[a]) * self.pml_sg[a]
        # A5
        names = ("ux", "uy", "uz")
        if any(int(_sc(pr.get(f"{nm}_source_flag", 0))) for nm in names) or int(_sc(pr.get("transducer_source_flag", 0))):
            idx = np.asarray(pr["u_source_index"], dtype=np.int64).reshape(-1) - 1
            mode = int(_sc(pr["u_source_mode"]))
            many = int(_sc(pr["u_source_many"]))
            for a, nm in enumerate(names):
                if int(_sc(pr.get(f"{nm}_source_flag", 0))) > self.t:
                    flat = self.u[a].reshape(-1)
                    if mode == 2:
                        self.u[a] = self.u[a] + self._scaled(pr[f"{nm}_source_input"], idx, many)
                    else:
                        v = self._src_values(pr[f"{nm}_source_input"], idx.size, many)
                        if mode == 0:
                            flat[idx] = v
                        else:
                            np.add.at(flat, idx, v)
            if int(_sc(pr.get("transducer_source_flag", 0))) > self.t:
                dm = np.asarray(pr["delay_mask"], dtype=np.int64).reshape(-1) - 1
                sig = np.asarray(pr["transducer_source_input"], dtype=np.float64).reshape(-1)
                np.add.at(self.u[0].reshape(-1), idx, sig[dm + self.t])
        # A6-A8
        kd = self.kappa * d
        self.du[0] = self.Fi(self.F(self.u[0]) * kd * self.ddx_neg)
        self.du[1] = self.Fi(self.F(self.u[1]) * kd * self.ddy_neg)
        self.du[2] = self.Fi(self.F(self.u[2]) * kd * self.ddz_neg)
        if self.dudn is not None:  # SolverCudaKernels.cu:1285-1301
            self.du = [self.du[a] * self.dudn[a] for a in range(3)]
        # A9
        if self.nonlinear:
            s = (2.0 * (self.rho[0] + self.rho[1] + self.rho[2]) + self.rho0) * self.dt
        else:
            s = self.dt * self.rho0
        self.rho = [self.pml[a] * (self.pml[a] * self.rho[a] - s * self.du[a]) for a in range(3)]
        # A10
        if int(_sc(pr.get("p_source_flag", 0))) > self.t:
            idx = np.asarray(pr["p_source_index"], dtype=np.int64).reshape(-1) - 1
            mode = int(_sc(pr["p_source_mode"]))
            many = int(_sc(pr["p_source_many"]))
            ndim = 2 if self.nz == 1 else 3  # SD::k2D: rho_x, rho_y only
            if mode == 2:
                sc = self._scaled(pr["p_source_input"], idx, many)
                self.rho = [r + sc if a < ndim else r for a, r in enumerate(self.rho)]
            else:
                v = self._src_values(pr["p_source_input"], idx.size, many)
                for a in range(ndim):
                    flat = self.rho[a].reshape(-1)
                    if mode == 0:
                        flat[idx] = v
                    else:
                        np.add.at(flat, idx, v)
        # A11
        S = self.rho[0] + self.rho[1] + self.rho[2]
        if not self.absorbing:
            if self.nonlinear:
                self.p = self.c2 * (S + self.bona * S * S / (2.0 * self.rho0))
            else:
                self.p = self.c2 * S
        else:
            D = self.du[0] + self.du[1] + self.du[2]
            tau_term = self.Fi(self.F(self.rho0 * D) * self.nabla1)
            eta_term = self.Fi(self.F(S) * self.nabla2)
            base = (self.bona * S * S / (2.0 * self.rho0) + S) if self.nonlinear else S
            self.p = self.c2 * (base + d * (tau_term * self.tau - eta_term * self.eta))
        # A12
        if self.t == 0 and int(_sc(pr.get("p0_source_flag", 0))) == 1:
            p0 = np.asarray(pr["p0_source_input"], dtype=np.float64)
            self.p = p0.copy()
            ndim = 2 if self.nz == 1 else 3  # SolverCudaKernels.cu:873-876 dimScalingFactor
            self.rho = [p0 / (ndim * self.c2) if a < ndim else np.zeros_like(p0) for a in range(3)]
            e = self.F(self.p) * self.kappa
            g = [self.Fi(e * self.ddx_pos), self.Fi(e * self.ddy_pos), self.Fi(e * self.ddz_pos)]
            self.u = [g[a] * self.dtrho[a] * d * 0.5 for a in range(3)]
        self.t += 1


def closed_form_pressure(pr: Dict[str, np.ndarray], n_steps: int) -> np.ndarray:
    """K1: homogeneous lossless linear medium, c0 == c_ref, PML == 1, p0 source only.

    p(n dt) = Fi{ cos(c |k| n dt) F{p0} } / N exactly for the k-space corrected scheme, where |k| is
    the wavenumber magnitude used by kappa (KSpaceFirstOrderSolver.cpp:2404-2452).
    Sample index n corresponds to the field after step n (storeSensorData follows the update, :930).
    """
    nx, ny, nz = (int(_sc(pr[k])) for k in ("Nx", "Ny", "Nz"))
    dx, dy, dz = (_sc(pr[k]) for k in ("dx", "dy", "dz"))
    c, dt = _sc(pr["c0"]), _sc(pr["dt"])
    fx = 0.5 - np.abs(0.5 - np.arange(nx // 2 + 1) / nx)
    fy = 0.5 - np.abs(0.5 - np.arange(ny) / ny)
    fz = 0.5 - np.abs(0.5 - np.arange(nz) / nz)
    k = 2.0 * math.pi * np.sqrt((fz ** 2 / dz ** 2).reshape(-1, 1, 1) + (fy ** 2 / dy ** 2).reshape(1, -1, 1)
                                + (fx ** 2 / dx ** 2).reshape(1, 1, -1))
    p0 = np.asarray(pr["p0_source_input"], dtype=np.float64)
    spec = np.fft.rfftn(p0, axes=(0, 1, 2)) * np.cos(c * k * n_steps * dt)
    return np.fft.irfftn(spec, s=(nz, ny, nx), axes=(0, 1, 2))


def absorbing_mode_recurrence(pr: Dict[str, np.ndarray], n_steps: int):
    """K8: homogeneous *absorbing* linear medium (scalar c0, rho0, alpha_coeff), PML == 1 (periodic box), p0 source only.

    Every operator of the step is then diagonal in k-space, so the whole scheme closes per Fourier mode on three
    scalars — p^, R^ = sum_i rho_i^ and S^ = sum_i ddk_i^- kappa u_i^ — independently of any FFT code, field kernel or
    generator loop of the restatements:
        S <- S + (dt/rho0) kappa^2 |k|^2 p          (A2-A4 then A6-A7: sum_i ddk_i^+ ddk_i^- = -|k|^2 by construction)
        R <- R - dt rho0 S                           (A9, linear)
        p <- c^2 ( R (1 - eta nabla2) + tau nabla1 rho0 S )          (A11 absorbing linear, SolverCudaKernels.cu:1978)
    starting after step 0 from  p = p0^, R = p0^/c^2, S = -(dt / 2 rho0) kappa^2 |k|^2 p0^   (A12: u = +dt/(2 rho0) grad p0).
    kappa = sinc(c_ref dt |k| / 2), nabla1 = |k|^(y-2), nabla2 = |k|^(y-1) (0 at k = 0), tau = -2 a c0^(y-1),
    eta = 2 a c0^y tan(pi y / 2), a = alpha_coeff * 100 (1e-6 / 2 pi)^y / (20 log10 e)  (KSpaceFirstOrderSolver.cpp:2514-2643).
    Returns (p, ux) in fp64 after step n_steps; ux from u_x^ <- u_x^ - (dt/rho0) ddx^+ kappa p^ alongside.
    """
    nx, ny, nz = (int(_sc(pr[k])) for k in ("Nx", "Ny", "Nz"))
    dx, dy, dz = (_sc(pr[k]) for k in ("dx", "dy", "dz"))
    c0, rho0, dt, c_ref = _sc(pr["c0"]), _sc(pr["rho0"]), _sc(pr["dt"]), _sc(pr["c_ref"])
    y, alpha = _sc(pr["alpha_power"]), _sc(pr["alpha_coeff"])
    kx = 2.0 * math.pi * np.fft.rfftfreq(nx, dx)
    ky = 2.0 * math.pi * np.fft.fftfreq(ny, dy)
    kz = 2.0 * math.pi * np.fft.fftfreq(nz, dz)
    k2 = (kz ** 2).reshape(-1, 1, 1) + (ky ** 2).reshape(1, -1, 1) + (kx ** 2).reshape(1, 1, -1)
    k = np.sqrt(k2)
    kappa = np.sinc(c_ref * dt * k / 2.0 / math.pi)  # np.sinc(x) = sin(pi x) / (pi x)
    with np.errstate(divide="ignore"):
        nabla1 = np.where(k > 0, k ** (y - 2.0), 0.0)
    nabla2 = np.where(k > 0, k ** (y - 1.0), 0.0)
    a_np = alpha * 100.0 * (1.0e-6 / (2.0 * math.pi)) ** y / (20.0 * math.log10(math.e))
    tau = -2.0 * a_np * c0 ** (y - 1.0)
    eta = 2.0 * a_np * c0 ** y * math.tan(math.pi * y / 2.0)
    ddx_pos = (1j * kx * np.exp(1j * kx * dx / 2.0)).reshape(1, 1, -1)
    p0 = np.fft.rfftn(np.asarray(pr["p0_source_input"], dtype=np.float64), axes=(0, 1, 2))
    g = (dt / rho0) * kappa ** 2 * k2
    p, R, S = p0.copy(), p0 / c0 ** 2, -0.5 * g * p0
    ux = 0.5 * (dt / rho0) * ddx_pos * kappa * p0
    for _ in range(n_steps):
        ux = ux - (dt / rho0) * ddx_pos * kappa * p
        S = S + g * p
        R = R - dt * rho0 * S
        p = c0 ** 2 * (R * (1.0 - eta * nabla2) + tau * nabla1 * rho0 * S)
    back = lambda a: np.fft.irfftn(a, s=(nz, ny, nx), axes=(0, 1, 2))
    return back(p), back(ux)
