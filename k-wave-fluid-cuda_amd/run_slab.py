"""File-driven multi-GPU run: one process per GPU, Z-slab decomposition (dist.py), k-Wave HDF5 input and output files.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      -m kwave_amd.run_slab -i in.h5 -o out.h5 --p_raw --p_max --p_final

Every rank reads its own planes of the grid-sized input arrays (and the small datasets whole), runs its slab, and
rank 0 writes the output file from the gathered pieces: sampled series / aggregates in the order of the global sensor
mask, whole-domain streams and final fields with the slabs stacked along z, and the scalars of the reference's output
file (Parameters.cpp:559-647).  The reference is single-GPU; the flags are the subset of its command line
(CommandLineParameters.cpp:264-292) that the slab path carries: index and corner (cuboid) sensor masks, p / u raw and aggregated streams,
compression and intensity streams, checkpoint / restart (one checkpoint file per rank).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

STREAM_FLAGS = ("p_raw", "p_rms", "p_max", "p_min", "p_max_all", "p_min_all", "p_final",
                "u_raw", "u_rms", "u_max", "u_min", "u_max_all", "u_min_all", "u_final",
                "u_non_staggered_raw", "p_c", "u_c", "u_non_staggered_c", "I_avg_c", "I_avg", "Q_term", "Q_term_c",
                "no_overlap")
OUTPUT_SCALARS = ("Nx", "Ny", "Nz", "Nt", "dt", "dx", "dy", "dz", "c_ref", "pml_x_size", "pml_y_size", "pml_z_size",
                  "pml_x_alpha", "pml_y_alpha", "pml_z_alpha", "p_source_flag", "p0_source_flag", "transducer_source_flag",
                  "ux_source_flag", "uy_source_flag", "uz_source_flag", "nonuniform_grid_flag", "absorbing_flag",
                  "nonlinear_flag", "u_source_many", "u_source_mode", "p_source_many", "p_source_mode", "alpha_power",
                  "sensor_mask_type")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="kwave_amd.run_slab", description=__doc__.split("\n")[0])
    ap.add_argument("-i", dest="input", required=True)
    ap.add_argument("-o", dest="output", required=True)
    ap.add_argument("-s", dest="start", type=int, default=1, help="first sampled time step (1-based, as on the reference's command line)")
    ap.add_argument("--benchmark", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL) or gloo (ranks sharing a GPU)")
    ap.add_argument("-p", action="store_true", help="same as --p_raw")
    ap.add_argument("-u", action="store_true", help="same as --u_raw")
    for f in STREAM_FLAGS:
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--period", type=float, default=0.0)
    ap.add_argument("--frequency", type=float, default=0.0)
    ap.add_argument("--mos", type=int, default=1)
    ap.add_argument("--harmonics", type=int, default=1)
    ap.add_argument("--checkpoint_file", default=None,
                    help="restart point of the run (CommandLineParameters.cpp:304-312): every rank keeps its slab in <file>.rank<r>of<P>")
    ap.add_argument("--checkpoint_timesteps", type=int, default=0, help="stop and checkpoint after this many steps of this launch")
    ap.add_argument("--checkpoint_interval", type=float, default=0.0, help="stop and checkpoint after this many seconds of this launch")
    a = ap.parse_args(argv)
    if (a.checkpoint_timesteps or a.checkpoint_interval) and not a.checkpoint_file:
        raise SystemExit("Error: --checkpoint_timesteps / --checkpoint_interval need --checkpoint_file.")
    a.p_raw |= a.p
    a.u_raw |= a.u

    import torch
    import torch.distributed as dist
    import kwave_amd  # noqa: F401
    from kwave_amd import h5io
    from kwave_amd.dist import DistSolver, partition_problem, slab_range

    dist.init_process_group(a.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0
    if dev >= torch.cuda.device_count():
        dev = 0
    torch.cuda.set_device(dev)

    nz = int(h5io.read_dataset(a.input, "Nz").ravel()[0])
    z0, z1 = slab_range(nz, rank, world)
    pr = h5io.read_problem(a.input, zslab=(z0, z1))
    loc, info = partition_problem(pr, rank, world, arrays_are_local=True)
    opts = {f.lower(): 1 for f in STREAM_FLAGS if getattr(a, f)}
    opts.update(period=a.period, frequency=a.frequency, mos=a.mos, harmonics=a.harmonics)
    if a.start < 1:
        raise SystemExit("Error: The beginning of data sampling is out of the simulation time span <1, Nt>.")
    sim = DistSolver(loc, rank, world, nz, device_index=dev, sampling_start=a.start - 1, benchmark_steps=a.benchmark, **opts)
    nt = a.benchmark or int(np.asarray(pr["Nt"]).ravel()[0])

    # ---- checkpoint / restart (KSpaceFirstOrderSolver.cpp:186-228, 1176-1224 per slab): the seven state arrays, the
    # time index and the stream accumulators of this rank's slab, one file per rank, written beside its target and
    # renamed; a launch that finds the files of ALL ranks continues from them, and removes them once the run is complete
    import pickle
    import time
    ckpt = f"{a.checkpoint_file}.rank{rank}of{world}" if a.checkpoint_file else None

    def all_ranks(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        if a.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    if ckpt and all_ranks(os.path.exists(ckpt)):
        with open(ckpt, "rb") as fh:
            st = pickle.load(fh)
        if st.get("grid") != (int(np.asarray(pr["Nx"]).ravel()[0]), int(np.asarray(pr["Ny"]).ravel()[0]), nz, world, nt):
            raise SystemExit(f"Error: {ckpt} belongs to another run (grid, rank count or Nt differ).")
        sim.sim.restore_state(st["state"])
    t_launch = time.time()
    done_here = 0
    while sim.t < nt:
        leg = min(nt - sim.t, 20)
        if a.checkpoint_timesteps:
            leg = min(leg, a.checkpoint_timesteps - done_here)
        sim.run(leg)
        done_here += leg
        stop = bool(a.checkpoint_timesteps and done_here >= a.checkpoint_timesteps)
        if a.checkpoint_interval:
            sim.sync()
            stop = stop or (time.time() - t_launch >= a.checkpoint_interval)
        if not all_ranks(not stop) and sim.t < nt:   # any rank out of time: every rank stops here
            sim.sync()
            state = {"grid": (int(np.asarray(pr["Nx"]).ravel()[0]), int(np.asarray(pr["Ny"]).ravel()[0]), nz, world, nt),
                     "state": sim.sim.checkpoint_state()}
            with open(ckpt + ".partial", "wb") as fh:
                pickle.dump(state, fh, protocol=4)
            os.replace(ckpt + ".partial", ckpt)
            dist.barrier()
            if rank == 0:
                print(f"time steps: {sim.t} of {nt}, checkpoint: {a.checkpoint_file}.rank*of{world}")
            sim.close()
            dist.destroy_process_group()
            return 0
    sim.finish()

    corners = "sensor_mask_corners" in pr
    piece = {"pos": info.get("sensor_positions", np.zeros(0, dtype=np.int64)), "streams": {}, "fields": {},
             "cuboids": info.get("cuboids", [])}
    for name in sim.stream_names():
        piece["streams"][name] = sim.stream(name)
    if a.p_final:
        piece["fields"]["p_final"] = sim.field("p")
    if a.u_final:
        for c in "xyz":
            piece["fields"][f"u{c}_final"] = sim.field("u" + c)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(piece, gathered, dst=0)
    if rank == 0:
        nx, ny = (int(np.asarray(pr[k]).ravel()[0]) for k in ("Nx", "Ny"))
        plane = nx * ny
        out = {}
        for k in OUTPUT_SCALARS:
            if k in pr:
                out[k] = np.asarray(pr[k]).reshape(1, 1, 1)
        out["Nz"] = np.array([[[nz]]], dtype=np.uint64)
        out["Nt"] = np.array([[[nt]]], dtype=np.uint64)
        out["t_index"] = np.array([[[sim.t]]], dtype=np.uint64)
        nsens = int(np.asarray(pr["sensor_mask_index"]).size) if "sensor_mask_index" in pr else 0
        names = sorted({n for g in gathered for n in g["streams"]})
        cuboid_sets = {}   # corner masks: stream -> [(cuboid, array (steps, nz, ny, nx) or (nz, ny, nx)), ...]
        for name in names:
            parts = [g["streams"].get(name) for g in gathered]
            if name.endswith("_all"):      # whole-domain aggregate: the slabs stacked along z
                out[name] = np.concatenate([p.reshape(-1, ny, nx) for p in parts], axis=0)
                continue
            if corners:
                # every rank holds, per stored step, its parts of the cuboids back to back (x fastest, then y, then z):
                # a cuboid is its parts stacked along z
                shapes = info["cuboid_shapes"]
                steps = max((np.asarray(p).shape[0] if np.asarray(p).ndim == 2 else 1) for p, g in zip(parts, gathered)
                            if p is not None and g["cuboids"])
                series = any(np.asarray(p).ndim == 2 for p in parts if p is not None)
                full = [np.zeros((steps,) + shp, dtype=np.float32) for shp in shapes]
                for p, g in zip(parts, gathered):
                    if p is None or not g["cuboids"]:
                        continue
                    rows, off = np.asarray(p).reshape(steps, -1), 0
                    for c, zlo, zhi in g["cuboids"]:
                        n = (zhi - zlo) * shapes[c][1] * shapes[c][2]
                        full[c][:, zlo:zhi] = rows[:, off:off + n].reshape(steps, zhi - zlo, shapes[c][1], shapes[c][2])
                        off += n
                cuboid_sets[name] = [(c, a if series else a[0], series) for c, a in enumerate(full)]
                continue
            # per sensor point and stored step: one value, or 2 * harmonics coefficients of a compression frame
            shapes = [(np.asarray(p).shape[0] if np.asarray(p).ndim == 2 else 1) for p, g in zip(parts, gathered) if g["pos"].size]
            steps = max(shapes, default=0)
            w = max((np.asarray(p).size // max(steps * g["pos"].size, 1) for p, g in zip(parts, gathered) if g["pos"].size), default=1)
            full = np.zeros((steps, nsens, w), dtype=np.float32)
            for p, g in zip(parts, gathered):
                if g["pos"].size:
                    full[:, g["pos"], :] = np.asarray(p).reshape(steps, g["pos"].size, w)
            out[name] = full.reshape(1, steps, nsens * w)     # dataset dims (Nsens [* 2 * harmonics], steps, 1)
        for name in piece["fields"]:
            out[name] = np.concatenate([g["fields"][name].reshape(-1, ny, nx) for g in gathered], axis=0)
        assert all(v.size == plane * nz for k, v in out.items() if k.endswith(("_final", "_all")))
        h5io.write_file(out, a.output, "output", f"k-Wave output written by kwave_amd.run_slab ({world} slab ranks)")
        for name, sets in cuboid_sets.items():
            for c, arr, series in sets:
                h5io.append_cuboid(a.output, name, c + 1, arr, series)
        print(f"time steps: {sim.t}, ranks: {world}, output: {a.output}")
    if ckpt and os.path.exists(ckpt):
        os.remove(ckpt)  # the run is complete (main.cpp:239-241 removes the checkpoint file likewise)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
