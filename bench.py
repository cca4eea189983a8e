#!/usr/bin/env python3
"""bench.py — time-steps/s of the k-space first-order loop on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size n] [--no-cpu] [--no-512] [--weak | --strong]

N=1 workload: BASELINE config 3 — 256^3 heterogeneous (c0, rho0, BonA, alpha_coeff as arrays), power-law absorption +
nonlinear term, p0 source, p_raw + p_max sampled on one xy plane (the manual's benchmark setup).
A "step" is one pass of the per-step loop (KSpaceFirstOrderSolver.cpp:885-935) over the whole grid, with all inputs
resident in HBM before the timed region.  The C++ host loop (libkwave_host) drives libkwave_hip; there is no CPU
fallback.  One JSON line is printed by rank 0.

N>1 (python -m torch.distributed.run ... bench.py --gpus N): the same 256^3 workload as Z-slabs over N GPUs (strong
scaling: one curve with the N=1 line), with BASELINE config 4 (512^3: N GPUs vs one) in config.c4_512; see
run_distributed().

Extra objects: "roofline" (dominant kernel: algorithmic bytes per launch / HIP-event duration, vs the 8 TB/s HBM spec
peak, with this box's measured copy bandwidth beside it; "step" carries the whole-step figure with B_alg of SURVEY.md
§8d) and "cpu_baseline" (the CPU oracle on the host cores for a bounded number of steps of the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BASELINE_STEPS_PER_S_256 = 1000.0 / 49.72  # kspaceFirstOrder3D-CUDA, TITAN X, 256^3, manual Table C.4


def alg_bytes(n: int, *, het=True, nonlinear=True, absorbing=True):
    """Algorithmic bytes per step and per device entry point (SURVEY.md §8d convention: every element-wise stage is
    charged its array reads+writes once, every 3-D FFT one read of its input + one write of its output)."""
    N = n ** 3
    Nc = (n // 2 + 1) * n * n
    R, Cx, K = 4 * N, 8 * Nc, 4 * Nc
    h = int(het)
    per = {
        "fft_r2c_3d": R + Cx,
        "fft_c2r_3d": R + Cx,
        "compute_pressure_gradient": 4 * Cx + K,
        "compute_velocity": (9 + 3 * h) * R,
        "compute_velocity_gradient": 6 * Cx + K,
        "compute_density_nonlinear": (9 + h) * R,
        "compute_density_linear": (9 + h) * R,
        "compute_pressure_terms_nonlinear": (9 + 2 * h) * R,
        "compute_pressure_terms_linear": (8 + h) * R,
        "compute_absorbtion_term": 4 * Cx + 2 * K,
        "sum_pressure_terms": (4 + 3 * h) * R,
        "sum_pressure_nonlinear_lossless": (4 + 3 * h) * R,
        "sum_pressure_linear_lossless": (4 + h) * R,
    }
    # fused entry points = the reference stages they replace (same convention, so the sum over a step is unchanged)
    terms = per["compute_pressure_terms_nonlinear"] if nonlinear else per["compute_pressure_terms_linear"]
    per["fused_velocity"] = 4 * (R + Cx) + per["compute_pressure_gradient"] + per["compute_velocity"]
    per["fused_density"] = (6 * (R + Cx) + per["compute_velocity_gradient"] + per["compute_density_nonlinear"]
                            + (terms if absorbing else 0))
    per["fused_absorption_pressure"] = 4 * (R + Cx) + per["compute_absorbtion_term"] + per["sum_pressure_terms"]
    # kernels of the fused pipeline (kw_fused.hip): every array a kernel has to read or write, once
    nl = int(nonlinear)
    for na in (1, 2, 3):
        per[f"k_xfwd[{na}]"] = na * (R + Cx)
        per[f"k_ypass_fwd[{na}]"] = per[f"k_ypass_inv[{na}]"] = 2 * na * Cx
        per[f"k_zfused_vgrad[{na}]"] = 2 * na * Cx + K
        per[f"k_zfused_absorb[{na}]"] = na * (2 * Cx + K)
    per["k_zfused_pgrad"] = 3 * Cx + K            # F{p} in; Q and G_z out (d/dx, d/dy share Q)
    per["k_ypass_inv_pgrad[3]"] = 5 * Cx          # Q read once, three arrays written
    per["k_ypass_inv_pgrad[2]"] = 3 * Cx
    per["k_ypass_inv_pgrad[1]"] = 2 * Cx
    per["k_zfused_source"] = 2 * Cx + K
    per["k_xinv_velocity"] = 3 * Cx + (6 + 3 * h) * R
    per["k_xinv_velocity_chain"] = per["k_xinv_velocity"] + 3 * Cx
    per["k_xinv_density"] = 3 * Cx + (6 + h) * R + ((h * nl + 1 + 2) * R if absorbing else 0)
    per["k_xinv_density_chain"] = 3 * Cx + (6 + h) * R + (h * nl + 1) * R + 2 * Cx
    per["k_xinv_psum"] = 2 * Cx + (2 + 3 * h) * R
    per["k_xinv_psum_chain"] = per["k_xinv_psum"] + Cx
    n_fft = 10 + 4 * int(absorbing)
    b = n_fft * (R + Cx) + per["compute_pressure_gradient"] + per["compute_velocity"] + per["compute_velocity_gradient"]
    b += per["compute_density_nonlinear"]
    if absorbing:
        b += (per["compute_pressure_terms_nonlinear"] if nonlinear else per["compute_pressure_terms_linear"])
        b += per["compute_absorbtion_term"] + per["sum_pressure_terms"]
    else:
        b += per["sum_pressure_nonlinear_lossless"] if nonlinear else per["sum_pressure_linear_lossless"]
    return b, per


def cpu_baseline(pr, n, budget_s=20.0):
    """CPU oracle ("port" of the reference algorithm, own FFT — no FFTW/MKL in the image) on the host cores."""
    from oracle import oracle as orc
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or orc.host_threads()
    sim = orc.OracleSim(pr)
    sim.step(1)  # step 0 (p0 initialisation) is not part of the steady loop
    t0 = time.time()
    sim.step(1)
    one = time.time() - t0
    steps = max(1, min(100, int(budget_s / max(one, 1e-3)) - 1))
    t0 = time.time()
    sim.step(steps)
    dt = time.time() - t0
    sim.close()
    model = ""
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": steps / dt, "unit": "time-steps/s", "cores": cores, "kind": "port", "cpu_model": model,
            "host_logical_cpus": os.cpu_count(),
            "sample": f"{steps} steps of the same {n}^3 workload after 2 untimed steps, OpenMP on {cores} threads, "
                      f"in-repo FFT (no FFTW/MKL); manual Table C.3: 224.5 ms/step on 2x12-core Haswell + MKL"}


def _code_only(text):
    """C++ source without comments and blank lines (string and character literals kept as they are)"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
        elif text.startswith("/*", i):
            i = text.find("*/", i + 2)
            i = n if i < 0 else i + 2
        else:
            out.append(c)
            i += 1
    return "\n".join(line.rstrip() for line in "".join(out).splitlines() if line.strip())


def kernel_source_hash():
    """sha256 over the device sources' code (comments and blank lines do not count): PMC counters committed under
    profiles/ are only quoted for the kernels they were collected with"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "k-wave-fluid-cuda_amd", "csrc", "*"))):
        if os.path.isdir(f):
            continue
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def pmc_traffic_for_this_build():
    """newest profiles/r*_pmc_traffic.json whose "kernel_source_hash" is this build's; (None, None) when none matches"""
    import glob
    cur = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("kernel_source_hash") == cur:
            return f, d
    return None, None


WEAK_DIMS = {1: (256, 256, 256), 2: (256, 256, 512), 4: (256, 512, 512), 8: (512, 512, 512)}  # 256^3 voxels per GPU


def single_gpu_rate(n, K, W):
    """time-steps/s of the n^3 config-3 workload on this process's GPU alone (non-slab fused path)"""
    import kwave_amd  # noqa: F401
    from kwave_amd import synthetic
    from kwave_amd.solver import HostSolver
    pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=W + K + 8)
    sim = HostSolver(pr, p_raw=1, p_max=1)
    del pr
    sim.run(W)
    sim.sync()
    ms = sim.time_steps(K)
    sim.sync()
    sim.close()
    return K / (ms * 1e-3)


def run_distributed(args):
    """N>1: one process per GPU (torch.distributed.run), Z-slab decomposition, one all-to-all transpose per 3-D FFT.

    Default = strong scaling on the metric's own grid: `value` = time-steps/s of the 256^3 config-3 workload on N GPUs
    (same grid, unit and workload as the N=1 line, so the per-N values form one curve).  BASELINE config 4 rides along
    in config.c4_512: time-steps/s of the 512^3 grid on the same N GPUs, on one GPU (rank 0 alone, same run) and their
    ratio (north_star: >= 3.5x at 8 GPUs).  --weak: every GPU owns 256^3 voxels instead (N=8 is the 512^3 grid),
    value = N x global time-steps/s.  --strong --size n: one fixed n^3 grid only.

    The data path is the device library's own RCCL exchange (--exchange native: kw_comm_init, ncclSend/ncclRecv groups
    on a communication stream); the process group (gloo) only carries the communicator id, the barriers and the
    max-over-ranks of the timings.  The same grids are then timed over the library's device-initiated transport
    (kw_comm_init_p2p: mapped peer buffers, one store kernel per exchange) and reported beside it in config.p2p — a
    failure there is reported, not fatal.  --exchange p2p: that transport only (ranks may share a GPU: rehearsal on a
    one-GPU box).  --exchange torch: torch.distributed.all_to_all_single as a callback (nccl group)."""
    # the process-group and RCCL libraries print banners on stdout: everything but the result line goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import kwave_amd  # noqa: F401
    from kwave_amd import synthetic
    from kwave_amd.dist import DistSolver, partition_problem, slab_range

    state = {"exchange": args.exchange, "backend": "nccl" if args.exchange == "torch" else "gloo", "fallback": None}
    dist.init_process_group(state["backend"])
    rank, world = dist.get_rank(), dist.get_world_size()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= torch.cuda.device_count():
        local_rank = 0  # the launcher narrowed the visible devices to this rank's GPU
    torch.cuda.set_device(local_rank)
    K, W = args.steps, args.warmup

    def reduce(x, op):
        t = torch.tensor([x], dtype=torch.float64)
        if state["backend"] == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=op)
        return float(t.item())

    def open_solver(loc, nz, exchange=None):
        """DistSolver on every rank with the same transport.  If the device library's communicator cannot be created
        on some rank (say, an RCCL build the library cannot bind), ALL ranks move to torch.distributed's RCCL group as
        the callback transport — still RCCL over xGMI, still the HIP pipeline; the line reports which one ran."""
        sim, err = None, None
        extra = {"rccl_library": args.rccl_library} if (args.rccl_library and (exchange or state["exchange"]) == "native") else {}
        try:
            sim = DistSolver(loc, rank, world, nz, device_index=local_rank, exchange=exchange or state["exchange"], p_raw=1, p_max=1,
                             **extra)
        except Exception as e:  # noqa: BLE001 - decided collectively below
            err = e
        if reduce(0.0 if err is not None else 1.0, dist.ReduceOp.MIN) > 0.5:
            return sim
        if sim is not None:
            sim.close()
        if exchange is not None or state["exchange"] != "native":
            raise err if err is not None else RuntimeError("slab solver could not be created on another rank")
        print(f"[bench rank {rank}] native RCCL exchange unavailable ({err!r}); using the torch.distributed transport",
              file=sys.stderr, flush=True)
        state.update(exchange="torch", backend="nccl", fallback=repr(err) if err is not None else "failed on another rank")
        dist.barrier()
        dist.destroy_process_group()
        dist.init_process_group("nccl")
        return DistSolver(loc, rank, world, nz, device_index=local_rank, exchange="torch", p_raw=1, p_max=1)

    def slab_run(grid, k, w, exchange=None):
        """k timed steps of the config-3 workload on `grid`, Z-slabs over all ranks; (seconds, exchanges per step)"""
        nx, ny, nz = grid
        z0, z1 = slab_range(nz, rank, world)
        pr = synthetic.make_problem(nx, ny, nz, heterogeneous=True, nonlinear=True, absorbing=True, source="p0",
                                    nt=w + k + 8, zslab=(z0, z1))
        loc, _ = partition_problem(pr, rank, world, arrays_are_local=True)
        del pr
        sim = open_solver(loc, nz, exchange)

        def together(fn):
            """fn() on every rank, then agreement: a rank whose exchange failed (a transport time-out, say) must not leave
            the others alone in the next collective of the process group"""
            err = None
            try:
                fn()
            except Exception as e:  # noqa: BLE001
                err = e
            if reduce(0.0 if err is not None else 1.0, dist.ReduceOp.MIN) < 0.5:
                sim.close()
                raise err if err is not None else RuntimeError("the slab run failed on another rank")

        together(lambda: (sim.run(w), sim.sync(), torch.cuda.synchronize()))
        dist.barrier()
        took = {}

        def timed():
            t0 = time.perf_counter()
            sim.run(k)
            sim.sync()
            torch.cuda.synchronize()
            took["sec"] = time.perf_counter() - t0

        together(timed)
        sec = reduce(took["sec"], dist.ReduceOp.MAX)
        dist.barrier()
        per_step = sim.exchanges // max(k + w, 1)
        sim.close()
        return sec, per_step

    def schedule_of(grid):
        """which slab schedule the device library picks for this grid (the rules of kw_fused.hip's create_impl)"""
        from kwave_amd import capi
        nx, ny, nz = grid
        tn = capi.make_tuning()  # the defaults (+ KW_TUNING, which HostSolver applies the same way)
        if state["exchange"] not in ("native", "p2p") or not tn.slab_pipeline:
            return "whole-array exchanges, per-array pipelining (callback transport or slab_pipeline=0)"
        per_peer = (nz // world) * (ny // world) * (nx // 2 + 1) * 8
        if (tn.slab_batch == 1) if tn.slab_batch >= 0 else (per_peer < (4 << 20)):
            return f"batched: one exchange per stage and direction ({per_peer / 2 ** 20:.1f} MiB per peer and array)"
        chunks = max(1, int(tn.slab_chunks))
        return (f"per-array pipelining, forward transposes started by the producing stage, "
                f"{'whole-array exchanges' if chunks == 1 else str(chunks) + ' plane chunks'} "
                f"({per_peer / 2 ** 20:.1f} MiB per peer and array)")

    c4 = None
    if args.weak:
        if world not in WEAK_DIMS:
            raise SystemExit(f"weak-scaling grids are defined for 1/2/4/8 GPUs, not {world}")
        grid = WEAK_DIMS[world]
    elif args.strong:
        grid = (args.size or 512,) * 3
    else:
        grid = (args.size or 256,) * 3
        if not args.no_512 and world > 1:
            k5, w5 = max(5, K // 5), max(2, W // 5)
            one = single_gpu_rate(512, k5, w5) if rank == 0 else 0.0
            dist.barrier()
            sec5, ex5 = slab_run((512, 512, 512), k5, w5)
            c4 = {"grid": [512, 512, 512], "steps": k5, "warmup": w5, "n_gpus": world,
                  "steps_per_s": round(k5 / sec5, 3), "steps_per_s_1gpu": round(one, 3),
                  "speedup_vs_1gpu": round((k5 / sec5) / one, 3) if one > 0 else None, "exchanges_per_step": ex5,
                  "slab_schedule": schedule_of((512, 512, 512)),
                  "note": "BASELINE config 4 (north_star: >= 3.5x at 8 GPUs vs 1 on 512^3); 1-GPU figure measured by "
                          "rank 0 alone in this run (non-slab fused path)"}
    sec, ex = slab_run(grid, K, W)
    nx, ny, nz = grid
    global_rate = K / sec
    # the same grids over the device-initiated transport, beside the default's numbers (never instead of them)
    p2p = None
    if state["exchange"] == "native" and state["fallback"] is None and not args.no_p2p:
        p2p = {"transport": "kw_comm_init_p2p: mapped peer buffers (hipIpc), one store kernel per exchange on the communication stream"}
        try:
            sec_p, ex_p = slab_run(grid, K, W, exchange="p2p")
            p2p.update(steps_per_s=round(K / sec_p, 2), ms_per_step=round(1e3 * sec_p / K, 4), exchanges_per_step=ex_p)
            if c4 is not None:
                sec5p, _ = slab_run((512, 512, 512), c4["steps"], c4["warmup"], exchange="p2p")
                p2p["c4_512_steps_per_s"] = round(c4["steps"] / sec5p, 3)
                if c4["steps_per_s_1gpu"]:
                    p2p["c4_512_speedup_vs_1gpu"] = round(c4["steps"] / sec5p / c4["steps_per_s_1gpu"], 3)
        except Exception as e:  # noqa: BLE001 - reported in the line
            p2p["error"] = repr(e)[:400]
    if rank == 0:
        b_step, _ = alg_bytes(256)
        voxels = nx * ny * nz
        b_global = b_step * voxels / 256 ** 3
        value = global_rate * world if args.weak else global_rate
        out = {"metric": "time-steps/sec on 256^3 heterogeneous grid; achieved HBM GB/s vs roofline",
               "value": round(value, 2), "unit": "time-steps/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": round(1e3 * sec / K, 4), "higher_is_better": True,
               "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{nx}x{ny}x{nz} heterogeneous (c0,rho0,BonA,alpha_coeff arrays), power-law absorption "
                                      f"+ nonlinear, p0 source, p_raw+p_max on one xy plane; Z-slabs over {world} GPUs "
                                      f"({voxels // world} voxels per GPU), one all-to-all transpose per 3-D FFT",
                          "grid": [nx, ny, nz], "parallelism": f"zslab{world}",
                          "exchange": "RCCL inside libkwave_hip.so (ncclSend/ncclRecv groups on a communication stream)"
                          if state["exchange"] == "native" else
                          "P2P inside libkwave_hip.so (mapped peer buffers, one store kernel per exchange on a communication stream)"
                          if state["exchange"] == "p2p" else "torch.distributed.all_to_all_single (RCCL) callback",
                          "exchange_fallback": state["fallback"],
                          "global_steps_per_s": round(global_rate, 2),
                          "value_definition": "N x global time-steps/s (each GPU owns one 256^3-voxel block)" if args.weak
                          else "global time-steps/s of this grid on N GPUs",
                          "exchanges_per_step": ex, "slab_schedule": schedule_of(grid)},
               "roofline": {"bound": "hbm", "kernel": "step (all ranks)", "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                            "achieved": round(b_global / (sec / K) / 1e9, 1),
                            "frac": round(b_global / (sec / K) / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None}}
        if c4 is not None:
            out["config"]["c4_512"] = c4
        if p2p is not None:
            out["config"]["p2p"] = p2p
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=0, help="grid size n (n^3); default 256 at N=1, 512 at N>1")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--granular", action="store_true", help="one launch per reference kernel instead of fused kernels")
    ap.add_argument("--strong", action="store_true", help="N>1: one fixed --size^3 grid (default 512) only")
    ap.add_argument("--weak", action="store_true", help="N>1: 256^3 voxels per GPU (256x256x512 / 256x512x512 / 512^3)")
    ap.add_argument("--no-512", action="store_true", help="N>1: skip the config-4 (512^3) block of the line")
    ap.add_argument("--no-p2p", action="store_true", help="N>1: skip the P2P-transport leg of the line")
    ap.add_argument("--rccl-library", default=None, help="N>1: library the device layer's RCCL binding loads (default search otherwise)")
    ap.add_argument("--exchange", default=os.environ.get("KW_EXCHANGE", "native"), choices=("native", "p2p", "torch"),
                    help="N>1 data path: the device library's own RCCL exchange, or torch.distributed as a callback")
    ap.add_argument("--slab-selftest", action="store_true",
                    help="one rank through the N>1 code path (slab kernels + RCCL all-to-all with itself): rehearsal on a 1-GPU box")
    args = ap.parse_args()

    if args.slab_selftest:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
        return run_distributed(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without the launcher: become the parent of `torch.distributed.run` (nothing has touched the GPU yet)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29541"),
               os.path.abspath(__file__)] + sys.argv[1:]
        return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if args.gpus > 1 or world > 1:
        return run_distributed(args)

    import kwave_amd  # noqa: F401
    from kwave_amd import capi, synthetic
    from kwave_amd.solver import HostSolver

    n = args.size or 256
    K, W, P = args.steps, args.warmup, args.profile_steps
    t_gen = time.time()
    pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=W + K + P + 8)
    t_gen = time.time() - t_gen
    sim = HostSolver(pr, p_raw=1, p_max=1, fused_kernels=not args.granular)
    sim.run(W)
    sim.sync()
    ms = sim.time_steps(K)  # HIP events on the solver's stream around exactly K steps; synchronises on the stop event
    sim.sync()
    steps_per_s = K / (ms * 1e-3)

    # --- per entry point timing (live, same process, same workload) ---
    hip = capi.load()
    capi.check(hip.kw_profile_enable(sim.ctx, 1))
    sim.run(P)
    prof = capi.profile_collect(sim.ctx)
    capi.check(hip.kw_profile_enable(sim.ctx, 0))
    b_step, per = alg_bytes(n)
    table = {}
    for name, (calls, total_ms) in prof.items():
        avg = total_ms / max(calls, 1)
        ab = per.get(name)
        table[name] = {"calls_per_step": calls / P, "avg_ms": round(avg, 4),
                       "ms_per_step": round(total_ms / P, 4),
                       "alg_gbs": round(ab / (avg * 1e-3) / 1e9, 1) if ab else None}
        if name.startswith("fused_"):  # stage boundaries of the pipeline differ from the reference's (chained passes)
            table[name]["alg_gbs"] = None
    # dominant GPU kernel (fused pipeline: the "k_*" records; granular path: one kernel per entry point)
    kernels = [k for k in table if per.get(k) and (k.startswith("k_") or args.granular)]
    if not kernels:  # a grid the fused pipeline does not take (kw_fused_supported): the run used the rocFFT path
        kernels = [k for k in table if per.get(k)]
    dom = max(kernels, key=lambda k: table[k]["ms_per_step"])
    achieved = per[dom] / (table[dom]["avg_ms"] * 1e-3) / 1e9
    traffic, traffic_src, step_traffic = None, None, None
    pmc_file, pmc = pmc_traffic_for_this_build()
    if n == 256 and not args.granular and pmc is not None:
        traffic = pmc.get("traffic_bytes_per_bench_kernel", {}).get(dom)
        traffic_src = (f"profiles/{os.path.basename(pmc_file)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of "
                       "this workload with this build of the kernels (source hash checked), 128-B read requests counted x2 "
                       "(gfx950 correction, checked on probe kernels of known size)")
        step_traffic = pmc.get("traffic_bytes_per_step")
    # measured copy bandwidth of this box beside the spec peak (1 GiB buffers: four times the Infinity Cache)
    copy_gbs = C.c_double()
    capi.check(hip.kw_measure_copy_bandwidth(sim.ctx, 1 << 30, 10, C.byref(copy_gbs)))
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "measured_copy_gbs": round(copy_gbs.value, 1),
                "frac_of_measured_copy": round(achieved / copy_gbs.value, 4),
                "traffic_source": traffic_src,
                "alg_bytes_per_launch": per[dom], "avg_ms": table[dom]["avg_ms"],
                "step": {"alg_bytes": b_step, "achieved": round(b_step / (ms * 1e-3 / K) / 1e9, 1),
                         "frac": round(b_step / (ms * 1e-3 / K) / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_of_measured_copy": round(b_step / (ms * 1e-3 / K) / 1e9 / copy_gbs.value, 4),
                         "traffic": step_traffic},
                "entry_points": {k: v for k, v in table.items() if not k.startswith("k_")},
                "kernels": {k: v for k, v in table.items() if k.startswith("k_")}}
    info = capi.DeviceInfo()
    capi.check(hip.kw_device_info_get(sim.ctx, info))
    sim.close()
    del sim
    c4_one = None
    if n == 256 and not args.granular and not args.no_512:
        # BASELINE config 4's single-GPU point (the denominator of the >= 3.5x at 8 GPUs target), same process
        k5 = max(5, K // 5)
        c4_one = {"grid": [512, 512, 512], "steps": k5, "n_gpus": 1, "steps_per_s": round(single_gpu_rate(512, k5, 2), 3)}
        c4_one["frac"] = round(alg_bytes(512)[0] * c4_one["steps_per_s"] / 1e9 / HBM_PEAK_GBS, 4)

    out = {"metric": "time-steps/sec on 256^3 heterogeneous grid; achieved HBM GB/s vs roofline",
           "value": round(steps_per_s, 2), "unit": "time-steps/s", "n_gpus": 1, "steps": K, "warmup": W,
           "ms_per_step": round(ms / K, 4), "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
           "vs_baseline": round(steps_per_s / BASELINE_STEPS_PER_S_256, 2) if n == 256 else None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{n}^3 heterogeneous (c0,rho0,BonA,alpha_coeff arrays), power-law absorption + "
                                  f"nonlinear, p0 source, p_raw+p_max on one xy plane ({n * n} points)",
                      "grid": [n, n, n], "fused_kernels": not args.granular,
                      "fft": "rocFFT 3-D R2C/C2R" if args.granular else "hand-written fused FFT passes (kw_fused.hip)",
                      "device": (info.name.decode() or info.arch.decode()), "baseline_ref": "BASELINE.md: 49.72 ms/step, TITAN X, "
                      "kspaceFirstOrder3D-CUDA v1.1 (manual Table C.4)", "input_generation_s": round(t_gen, 1)},
           "roofline": roofline}
    if c4_one is not None:
        out["config"]["c4_512"] = c4_one
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(pr, n)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
