"""Multi-GPU driver: Z-slab decomposition of the k-space step, one process per GPU.

New with this build (the reference is single-GPU, /root/reference/Readme.md:12-13).  Real-space arrays are split into
Z-slabs; every 3-D FFT of the fused pipeline (csrc/kw_fused.hip) does x- and y-passes locally, one all-to-all transpose,
the fused z-pass on `Ny/P` rows with all `Nz` planes, and the mirror image on the way back.  Four transports:
  "native"  the device library's own RCCL path (csrc/kw_comm.hip: ncclSend/ncclRecv groups on a communication stream,
            events against the compute stream).  Python only hands over the communicator id — nothing of it runs in
            the step loop.  The default on a GPU node.
  "p2p"     the device library's device-initiated transport (kw_comm_init_p2p): the ranks map each other's exchange
            buffers and one store kernel per exchange writes the peers' receive buffers directly.  Python's part is
            the all-gather of the buffer handles at set-up (any process-group backend; ranks may share a GPU).
  "torch"   `torch.distributed.all_to_all_single` (backend nccl = RCCL) as a callback (`kw_exchange_fn`).
  "host"    device -> pinned host -> gloo all-to-all -> device, for ranks that share one GPU (tests).

partition_problem() cuts a global problem dict (HDF5 dataset names) into the slab of one rank:
  * 3-D arrays, pml_z, pml_z_sgz, dzudzn* -> planes z0 <= z < z1
  * source / sensor index masks (1-based) -> entries inside the slab, re-based to the slab, original order kept
  * per-point source series, delay mask    -> the matching columns
  * everything else (scalars, x/y PML, ddx/ddy/ddz operators) unchanged; "Nz" becomes the local plane count.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, Optional, Tuple

import numpy as np

from . import capi

U64 = np.uint64


def slab_range(nz: int, rank: int, nranks: int) -> Tuple[int, int]:
    if nz % nranks:
        raise ValueError(f"Nz={nz} is not divisible by {nranks} ranks")
    nzl = nz // nranks
    return rank * nzl, (rank + 1) * nzl


def _sc(a) -> int:
    return int(np.asarray(a).reshape(-1)[0])


def partition_problem(pr: Dict[str, np.ndarray], rank: int, nranks: int, arrays_are_local: bool = False):
    """Return (local problem dict, info).  info["sensor_positions"] = positions of this rank's sensor points in the
    global sensor mask (to reassemble sampled series in the original order)."""
    nx, ny, nz = _sc(pr["Nx"]), _sc(pr["Ny"]), _sc(pr["Nz"])
    z0, z1 = slab_range(nz, rank, nranks)
    if ny % nranks:
        raise ValueError(f"Ny={ny} is not divisible by {nranks} ranks")
    plane = nx * ny
    lo, hi = z0 * plane, z1 * plane
    out: Dict[str, np.ndarray] = {}
    info = {"z0": z0, "z1": z1, "nz_global": nz}

    def local_index(idx1: np.ndarray):
        idx0 = idx1.reshape(-1).astype(np.int64) - 1
        sel = np.nonzero((idx0 >= lo) & (idx0 < hi))[0]
        return sel, (idx0[sel] - lo + 1).astype(U64)

    p_sel = u_sel = None
    if "p_source_index" in pr:
        p_sel, p_loc = local_index(pr["p_source_index"])
    if "u_source_index" in pr:
        u_sel, u_loc = local_index(pr["u_source_index"])

    for name, a in pr.items():
        a = np.asarray(a)
        if name == "Nz":
            out[name] = np.array([[[z1 - z0]]], dtype=U64)
        elif a.ndim == 3 and a.shape == (nz, ny, nx):
            out[name] = np.ascontiguousarray(a[z0:z1])
        elif arrays_are_local and a.ndim == 3 and a.shape == (z1 - z0, ny, nx):
            out[name] = a
        elif name in ("pml_z", "pml_z_sgz", "dzudzn", "dzudzn_sgz"):
            out[name] = np.ascontiguousarray(a.reshape(-1)[z0:z1])
        elif name == "p_source_index":
            out[name] = p_loc.reshape(1, 1, -1)
        elif name == "u_source_index":
            out[name] = u_loc.reshape(1, 1, -1)
        elif name == "p_source_input" and _sc(pr.get("p_source_many", 0)):
            nt_src = _sc(pr["p_source_flag"])
            out[name] = np.ascontiguousarray(a.reshape(nt_src, -1)[:, p_sel]).reshape(1, nt_src, -1)
        elif name in ("ux_source_input", "uy_source_input", "uz_source_input") and _sc(pr.get("u_source_many", 0)):
            nt_src = _sc(pr[name[:2] + "_source_flag"])
            out[name] = np.ascontiguousarray(a.reshape(nt_src, -1)[:, u_sel]).reshape(1, nt_src, -1)
        elif name == "delay_mask":
            out[name] = np.ascontiguousarray(a.reshape(-1)[u_sel]).reshape(1, 1, -1)
        elif name == "sensor_mask_index":
            s_sel, s_loc = local_index(a)
            out[name] = s_loc.reshape(1, 1, -1)
            info["sensor_positions"] = s_sel
        elif name == "sensor_mask_corners":
            # cuboids [x0 y0 z0 x1 y1 z1] (1-based, inclusive): this rank keeps the part of every cuboid inside its slab,
            # z re-based; info["cuboids"] = (cuboid, first plane, end plane) of each kept part within its cuboid
            rows, kept = [], []
            for c, (x0, y0, cz0, x1, y1, cz1) in enumerate(a.reshape(-1, 6).astype(np.int64)):
                lo, hi = max(cz0, z0 + 1), min(cz1, z1)
                if lo <= hi:
                    rows.append([x0, y0, lo - z0, x1, y1, hi - z0])
                    kept.append((c, int(lo - cz0), int(hi - cz0 + 1)))
            info["cuboids"] = kept
            info["cuboid_shapes"] = [(int(r[5] - r[2] + 1), int(r[4] - r[1] + 1), int(r[3] - r[0] + 1)) for r in a.reshape(-1, 6).astype(np.int64)]
            if rows:
                out[name] = np.array(rows, dtype=U64).reshape(1, len(rows), 6)
            else:  # no part of any cuboid here: this rank samples nothing
                out["sensor_mask_index"] = np.zeros((1, 1, 0), dtype=U64)
                out["sensor_mask_type"] = np.array([[[0]]], dtype=U64)
        elif name == "sensor_mask_type" and "sensor_mask_type" in out:
            pass  # set above: a rank without any cuboid part runs with an empty index mask
        else:
            out[name] = a
    if "sensor_mask_corners" in pr and not info.get("cuboids"):
        out["sensor_mask_type"] = np.array([[[0]]], dtype=U64)
        out.pop("sensor_mask_corners", None)
    return out, info


class SlabExchange:
    """The all-to-all handed to the C++ solver.  nccl (= RCCL): device tensors allocated here double as the pipeline's
    scratch, so the collective runs in place on them.  gloo: device -> pinned host, all-to-all on CPU, host -> device."""

    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    CB_START = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
    CB_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
    CB_PIECE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int)

    def __init__(self, nranks: int, device_index: int = 0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.nranks = nranks
        self.backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
        self.device_index = device_index
        self.tensors: Dict[int, "torch.Tensor"] = {}
        self.ctx = None
        self.stream = None
        self._host: Dict[int, tuple] = {}
        self._view_cache: Dict[tuple, tuple] = {}
        self.error: Optional[BaseException] = None  # what a callback raised (ctypes cannot propagate it)
        self.callback = self.CB(self._guard(self._exchange))
        self.calls = 0
        self.host_seconds = 0.0  # time the launching thread spent inside the split-phase callbacks (diagnostic)
        self.wait_seconds = 0.0  # ... of which in work.wait()
        # split-phase form (RCCL only): the transpose of one array is in flight while other arrays compute
        self.works: Dict[int, object] = {}
        self.start_callback = self.CB_START(self._guard(self._start)) if self.backend == "nccl" else None
        self.wait_callback = self.CB_WAIT(self._guard(self._wait)) if self.backend == "nccl" else None
        # strided pieces (plane chunks of an array), host-staged: lets the pipeline run its pipelined slab schedule
        # with several ranks on one GPU (tests of the schedule the library's own RCCL path runs)
        self.piece_callback = self.CB_PIECE(self._guard(self._piece)) if self.backend != "nccl" else None

    def _guard(self, fn):
        """A callback returns 0 / non-zero to the C++ step loop (which then stops with KW_ERR_COMM instead of running
        on with stale scratch data); the exception itself is kept for DistSolver to re-raise."""
        def call(*args):
            try:
                fn(*args)
                return 0
            except BaseException as e:  # noqa: BLE001 - must not unwind through the C frames
                if self.error is None:
                    self.error = e
                return 1
        return call

    def alloc_scratch(self, nbytes: int):
        """Six device buffers (s[3], t[3]) as torch tensors; returns their addresses (for HostSolver(scratch=...))."""
        torch = self.torch
        ptrs = []
        for _ in range(6):
            t = torch.zeros(nbytes // 4, dtype=torch.float32, device=f"cuda:{self.device_index}")
            self.tensors[t.data_ptr()] = t
            ptrs.append(t.data_ptr())
        torch.cuda.synchronize()
        return ptrs

    def bind(self, ctx, stream=None):
        """ctx: kw_ctx* of the solver; stream: torch stream every launch of the solver is moved to (nccl ordering).
        The stream also becomes torch's current stream of this thread, so the collectives below are ordered after the
        kernels already enqueued without a per-call stream context."""
        self.ctx = ctx
        self.stream = stream
        if stream is not None:
            capi.check(capi.load().kw_set_stream(ctx, C.c_void_p(stream.cuda_stream)))
            self.torch.cuda.set_stream(stream)

    def _slice(self, ptr, n):
        """n bytes at device address ptr as a view of the scratch tensor that holds them (the spectral rows start at
        the tensor's base, the x-Nyquist side array lies further in)."""
        for base, t in self.tensors.items():
            if base <= ptr and ptr + n <= base + t.numel() * 4:
                first = (ptr - base) // 4
                return t[first: first + n // 4]
        raise KeyError(f"exchange buffer {ptr:#x} (+{n} bytes) is not inside the scratch arrays")

    def _views(self, send, recv, n):
        key = (send, recv, n)
        v = self._view_cache.get(key)
        if v is None:
            v = (self._slice(send, n), self._slice(recv, n))
            self._view_cache[key] = v
        return v

    def _start(self, user, send, recv, bytes_per_peer, slot):
        """all_to_all_single(async_op=True): the RCCL stream waits for the work enqueued so far on the solver's stream
        and the call returns; later launches on the solver's stream overlap with the collective."""
        t0 = time.perf_counter()
        self.calls += 1
        src, dst = self._views(send, recv, bytes_per_peer * self.nranks)
        self.works[slot] = self.dist.all_to_all_single(dst, src, async_op=True)
        self.host_seconds += time.perf_counter() - t0

    def _wait(self, user, slot):
        """work.wait(): the solver's stream waits for the collective (the host does not block)."""
        t0 = time.perf_counter()
        self.works.pop(slot).wait()
        dt = time.perf_counter() - t0
        self.host_seconds += dt
        self.wait_seconds += dt

    def _piece(self, user, send, recv, stride, offset, nbytes, slot):
        """kw_exchange_piece_fn, blocking: per peer q the bytes at send + q * stride + offset go to rank q and arrive
        at recv + sender * stride + offset."""
        self.calls += 1
        torch, dist, hip = self.torch, self.dist, capi.load()
        n = nbytes * self.nranks
        key = ("piece", n)
        if key not in self._host:
            self._host[key] = (torch.empty(n // 4, dtype=torch.float32).pin_memory() if torch.cuda.is_available()
                               else torch.empty(n // 4, dtype=torch.float32),
                               torch.empty(n // 4, dtype=torch.float32))
        hin, hout = self._host[key]
        for q in range(self.nranks):
            capi.check(hip.kw_memcpy_d2h(self.ctx, C.c_void_p(hin.data_ptr() + q * nbytes),
                                         C.c_void_p(send + q * stride + offset), nbytes))
        dist.all_to_all_single(hout, hin)
        for q in range(self.nranks):
            capi.check(hip.kw_memcpy_h2d(self.ctx, C.c_void_p(recv + q * stride + offset),
                                         C.c_void_p(hout.data_ptr() + q * nbytes), nbytes))

    def _exchange(self, user, send, recv, bytes_per_peer):
        self.calls += 1
        torch, dist = self.torch, self.dist
        n = bytes_per_peer * self.nranks
        if self.backend == "nccl":
            src, dst = self._views(send, recv, n)
            dist.all_to_all_single(dst, src)
            return 0
        # gloo (ranks sharing one GPU, tests): stage through pinned host memory
        hip = capi.load()
        key = n
        if key not in self._host:
            self._host[key] = (torch.empty(n // 4, dtype=torch.float32).pin_memory() if torch.cuda.is_available()
                               else torch.empty(n // 4, dtype=torch.float32),
                               torch.empty(n // 4, dtype=torch.float32))
        hin, hout = self._host[key]
        capi.check(hip.kw_memcpy_d2h(self.ctx, C.c_void_p(hin.data_ptr()), C.c_void_p(send), n))
        dist.all_to_all_single(hout, hin)
        capi.check(hip.kw_memcpy_h2d(self.ctx, C.c_void_p(recv), C.c_void_p(hout.data_ptr()), n))


class DistSolver:
    """One rank of a slab-decomposed simulation (wraps solver.HostSolver).

    exchange: "native" | "torch" | "host" (module docstring); None picks "torch" for an nccl process group (kept for
    the callback path's tests) and "host" otherwise.  "native" needs one GPU per rank (RCCL refuses two ranks on one
    device) — or a single rank, which then exchanges with itself; "p2p" has no such limit."""

    def __init__(self, pr_local: Dict[str, np.ndarray], rank: int, nranks: int, nz_global: int, device_index: int = 0,
                 exchange: Optional[str] = None, **opts):
        from .solver import HostSolver
        import torch
        import torch.distributed as dist
        self.rank, self.nranks = rank, nranks
        backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
        if exchange is None:
            exchange = "torch" if backend == "nccl" else "host"
        if exchange not in ("native", "p2p", "torch", "host"):
            raise ValueError(f"unknown exchange {exchange!r}")
        if exchange == "torch" and backend != "nccl":
            raise ValueError("exchange='torch' needs an nccl (= RCCL) process group")
        self.transport = exchange
        nx, ny, nzl = (_sc(pr_local[k]) for k in ("Nx", "Ny", "Nz"))
        if exchange == "p2p":
            def allgather(mine: bytes) -> bytes:
                box = [None] * nranks
                dist.all_gather_object(box, mine)
                return b"".join(box)
            self.exchange = None
            self.sim = HostSolver(pr_local, slab_ranks=nranks, slab_rank=rank, nz_global=nz_global, comm_p2p=True,
                                  comm_allgather=allgather if nranks > 1 else None, device_idx=device_index, **opts)
            return
        if exchange == "native":
            # rank 0 draws the communicator id; it travels through the process group (any backend) as plain bytes
            box = [capi.comm_unique_id(opts.get("rccl_library")) if rank == 0 else None]
            if nranks > 1:
                dist.broadcast_object_list(box, src=0)
            self.exchange = None
            self.sim = HostSolver(pr_local, slab_ranks=nranks, slab_rank=rank, nz_global=nz_global, comm_unique_id=box[0],
                                  device_idx=device_index, **opts)
            return
        self.exchange = SlabExchange(nranks, device_index)
        scratch = None
        if exchange == "torch":
            torch.cuda.set_device(device_index)
            pitch = (nx // 2 + 1 + 15) // 16 * 16
            scratch = self.exchange.alloc_scratch(pitch * ny * nzl * 8)
        self.sim = HostSolver(pr_local, slab_ranks=nranks, slab_rank=rank, nz_global=nz_global,
                              exchange_fn=self.exchange.callback,
                              exchange_piece_fn=self.exchange.piece_callback if exchange == "host" else None,
                              exchange_start_fn=self.exchange.start_callback if exchange == "torch" else None,
                              exchange_wait_fn=self.exchange.wait_callback if exchange == "torch" else None,
                              scratch=scratch, device_idx=device_index, **opts)
        stream = torch.cuda.Stream(device=device_index) if exchange == "torch" else None
        self.exchange.bind(self.sim.ctx, stream)

    @property
    def exchanges(self) -> int:
        """all-to-all exchanges started so far"""
        return capi.comm_exchanges(self.sim.ctx) if self.exchange is None else self.exchange.calls

    def run(self, n_steps: int):
        try:
            self.sim.run(n_steps)
        except capi.KWaveError as e:
            if self.exchange is not None and self.exchange.error is not None:
                raise RuntimeError(f"slab exchange failed on rank {self.rank}: {self.exchange.error!r}") from e
            raise

    def step(self, n: int = 1):
        self.run(n)

    def __getattr__(self, name):
        return getattr(self.sim, name)
