"""Worker for tests/test_gpu_dist.py::test_native_exchange_with_many_ranks_on_one_gpu: P ranks as P THREADS of this process,
each with its own solver (own device context, own communication stream), all on the one GPU, exchanging through the
device library's own paths (csrc/kw_comm.hip):
  --rccl-library <tests/native/libmock_rccl.so>   the RCCL path bound to the test double (the real RCCL refuses two ranks
                                                  on one device);
  --transport p2p                                 the device-initiated transport: the ranks trade their export blobs
                                                  through a barrier of this process and map each other's buffers.
Writes the gathered fields and sensor series to --out."""
import argparse
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kwave_amd  # noqa: E402,F401
from kwave_amd import capi, synthetic  # noqa: E402
from kwave_amd.dist import partition_problem  # noqa: E402
from kwave_amd.solver import HostSolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--dims", type=int, nargs=3, default=[32, 64, 32])
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--source", default="p0")
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--pml", type=int, default=4)
    ap.add_argument("--per-rank", action="store_true",
                    help="full-size runs: every rank's slab is generated on its own and written to <out>.rank<r>.npz")
    ap.add_argument("--out", required=True)
    ap.add_argument("--transport", default="rccl", choices=("rccl", "p2p"))
    ap.add_argument("--rccl-library", default=None, help="library the RCCL binding loads (the test double)")
    ap.add_argument("--absent", type=int, default=-1, help="p2p: this rank never starts its run (time-out test)")
    ap.add_argument("--tuning", default=None)
    a = ap.parse_args()
    assert a.transport == "p2p" or "mock" in (a.rccl_library or ""), "the RCCL transport of this worker is for the test double only"
    P = a.ranks
    nx, ny, nz = a.dims
    kw = dict(heterogeneous=True, nonlinear=True, absorbing=True, source=a.source, source_mode=a.mode, source_many=1,
              nt=a.steps, pml_size=a.pml, sensor="random")
    if a.per_rank:
        parts = []
        for r in range(P):
            pr = synthetic.make_problem(nx, ny, nz, zslab=(r * nz // P, (r + 1) * nz // P), **kw)
            parts.append(partition_problem(pr, r, P, arrays_are_local=True))
            del pr
    else:
        pr = synthetic.make_problem(nx, ny, nz, **kw)
        parts = [partition_problem(pr, r, P) for r in range(P)]
    comm_id = capi.comm_unique_id(a.rccl_library) if a.transport == "rccl" else None
    results, errors = [None] * P, []
    blobs, gate = [None] * P, threading.Barrier(P)

    def allgather_of(r):
        def gather(mine: bytes) -> bytes:
            blobs[r] = mine
            gate.wait(timeout=120)
            return b"".join(blobs)
        return gather

    def rank_main(r):
        try:
            loc, info = parts[r]
            if a.transport == "p2p":
                sim = HostSolver(loc, slab_ranks=P, slab_rank=r, nz_global=nz, comm_p2p=True, comm_allgather=allgather_of(r),
                                 p_raw=1, p_max=1, tuning=a.tuning)
            else:
                sim = HostSolver(loc, slab_ranks=P, slab_rank=r, nz_global=nz, comm_unique_id=comm_id, rccl_library=a.rccl_library,
                                 p_raw=1, p_max=1, tuning=a.tuning)
            sim.run(0)  # set-up only (communicator, buffers, the ranks' hand-over of handles): collective
            if a.transport == "p2p":
                assert capi.comm_transport(sim.ctx) == "p2p", capi.comm_transport(sim.ctx)
            if r == a.absent:
                gate.wait(timeout=120)  # stays away until the others have given up
                sim.close()
                results[r] = "absent"
                return
            sim.run(a.steps)
            sim.finish()
            fields = {k: sim.field(k) for k in ("p", "ux", "uz", "rhoy")}
            series = sim.stream("p") if info["sensor_positions"].size else np.zeros((a.steps, 0), dtype=np.float32)
            results[r] = (fields, series, info["sensor_positions"], capi.comm_exchanges(sim.ctx))
            sim.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            if a.absent >= 0:
                try:
                    gate.wait(timeout=120)
                except threading.BrokenBarrierError:
                    pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    if a.absent >= 0:  # every present rank must have failed with the transport's time-out message
        ok = len(errors) == P - 1 and all("gave up waiting" in e for _, e in errors) and not any(t.is_alive() for t in threads)
        print("TIMEOUT-OK" if ok else "TIMEOUT-BAD", errors, flush=True)
        os._exit(0 if ok else 1)
    if errors or any(t.is_alive() for t in threads) or any(x is None for x in results):
        print("FAILED", errors, [t.is_alive() for t in threads], flush=True)
        os._exit(1)  # a stuck rank thread must not keep the process alive
    if a.per_rank:
        for r, res in enumerate(results):
            np.savez(f"{a.out}.rank{r}.npz", series=res[1], pos=res[2], exchanges=np.array([res[3]]), **res[0])
        return
    out = {k: np.concatenate([res[0][k] for res in results], axis=0) for k in ("p", "ux", "uz", "rhoy")}
    n_sens = sum(res[2].size for res in results)
    full = np.zeros((a.steps, n_sens), dtype=np.float32)
    for res in results:
        if res[2].size:
            full[:, res[2]] = res[1]
    out["series"] = full
    out["exchanges"] = np.array([results[0][3]])
    np.savez(a.out, **out)


if __name__ == "__main__":
    main()
