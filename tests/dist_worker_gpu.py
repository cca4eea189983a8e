"""Worker for tests/test_gpu_dist.py: P ranks share one MI355X, all-to-all through gloo (host staging); with
backend nccl (one GPU per rank) the same code runs the RCCL path.  Rank 0 writes the gathered fields to --out."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.dist import DistSolver, partition_problem  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs=3, default=[32, 32, 32])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--source", default="p0")
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--medium", default="111", help="heterogeneous, nonlinear, absorbing as three 0/1 digits")
    ap.add_argument("--exchange", default=None, help="native | torch | host (DistSolver's default when omitted)")
    ap.add_argument("--config5", action="store_true",
                    help="also store the non-staggered velocity, compression and intensity streams (BASELINE config 5's set)")
    ap.add_argument("--pml", type=int, default=4)
    ap.add_argument("--sensor", default="random")
    ap.add_argument("--per-rank", action="store_true",
                    help="full-size runs: every rank builds only its own slab of the problem and writes <out>.rank<r>.npz")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    dist.init_process_group(a.backend)
    rank, P = dist.get_rank(), dist.get_world_size()
    dev = int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0
    nx, ny, nz = a.dims
    het, nonlin, absorb = (c == "1" for c in a.medium)
    zslab = (rank * nz // P, (rank + 1) * nz // P) if a.per_rank else None
    pr = synthetic.make_problem(nx, ny, nz, heterogeneous=het, nonlinear=nonlin, absorbing=absorb, source=a.source,
                                source_mode=a.mode, source_many=1, nt=a.steps, pml_size=a.pml, sensor=a.sensor, zslab=zslab)
    loc, info = partition_problem(pr, rank, P, arrays_are_local=a.per_rank)
    del pr
    extra = {}
    if a.config5:
        dt = float(np.asarray(loc["dt"]).ravel()[0])
        extra = dict(u_non_staggered_raw=1, p_c=1, u_non_staggered_c=1, i_avg_c=1, i_avg=1, q_term=1, q_term_c=1,
                     period=1.0 / (1.0e6 * dt) / 2.0, mos=1, harmonics=2)
    sim = DistSolver(loc, rank, P, nz, device_index=dev, exchange=a.exchange, p_raw=1, p_max=1, **extra)
    sim.run(a.steps)
    sim.finish()
    fields = {k: sim.field(k) for k in ("p", "ux", "uz", "rhoy")}
    series = sim.stream("p") if info["sensor_positions"].size else np.zeros((a.steps, 0), dtype=np.float32)
    if a.per_rank:
        np.savez(f"{a.out}.rank{rank}.npz", series=series, pos=info["sensor_positions"],
                 exchanges=np.array([sim.exchanges]), **fields)
        sim.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    streams = {}
    if a.config5 and info["sensor_positions"].size:
        for name in sim.stream_names():
            if name not in ("p", "p_max"):
                streams[name] = sim.stream(name)
        fields["uz_shifted"] = sim.field("uz_shifted")
    gathered = [None] * P if rank == 0 else None
    dist.gather_object({"fields": fields, "series": series, "pos": info["sensor_positions"], "streams": streams}, gathered, dst=0)
    if rank == 0:
        out = {k: np.concatenate([g["fields"][k] for g in gathered], axis=0) for k in fields}
        n_sens = sum(g["pos"].size for g in gathered)
        full = np.zeros((a.steps, n_sens), dtype=np.float32)
        for g in gathered:
            if g["pos"].size:
                full[:, g["pos"]] = g["series"]
        out["series"] = full
        if a.config5:
            out["uz_shifted"] = np.concatenate([g["fields"]["uz_shifted"] for g in gathered], axis=0)
            names = sorted({k for g in gathered for k in g["streams"]})
            for name in names:  # per sensor point: w floats per step (1, or 2 * harmonics for compression frames)
                some = next(g for g in gathered if name in g["streams"])
                a0 = np.asarray(some["streams"][name])
                steps = a0.shape[0] if a0.ndim == 2 else 1
                w = a0.size // (steps * some["pos"].size)
                full_s = np.zeros((steps, n_sens, w), dtype=np.float32)
                for g in gathered:
                    if name in g["streams"]:
                        full_s[:, g["pos"], :] = np.asarray(g["streams"][name]).reshape(steps, g["pos"].size, w)
                out["stream_" + name] = full_s
        out["exchanges"] = np.array([sim.exchanges])
        np.savez(a.out, **out)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
