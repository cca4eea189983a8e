"""Worker for tests/test_gpu_dist.py::test_native_exchange_with_many_ranks_on_one_gpu: P ranks as P THREADS of this process,
each with its own solver (own device context, own communication stream), all on the one GPU, exchanging through the
device library's own path (csrc/kw_comm.hip) bound to tests/native/mock_rccl.cpp instead of RCCL (KW_RCCL_LIB): the
real RCCL refuses two ranks on one device.  Writes the gathered fields and sensor series to --out."""
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
    a = ap.parse_args()
    assert "mock" in os.environ.get("KW_RCCL_LIB", ""), "this worker is for the mock exchange library only"
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
    comm_id = capi.comm_unique_id()
    results, errors = [None] * P, []

    def rank_main(r):
        try:
            loc, info = parts[r]
            sim = HostSolver(loc, slab_ranks=P, slab_rank=r, nz_global=nz, comm_unique_id=comm_id, p_raw=1, p_max=1)
            sim.run(a.steps)
            sim.finish()
            fields = {k: sim.field(k) for k in ("p", "ux", "uz", "rhoy")}
            series = sim.stream("p") if info["sensor_positions"].size else np.zeros((a.steps, 0), dtype=np.float32)
            results[r] = (fields, series, info["sensor_positions"], capi.comm_exchanges(sim.ctx))
            sim.close()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
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
