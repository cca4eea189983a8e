#!/usr/bin/env python3
"""Timeline of ONE rank of an N-GPU slab run on a one-GPU box: the rank's real kernels at their real sizes, the wire
replaced by a link model (values are meaningless).  A model to compare the slab SCHEDULES and the two transports with
(KW_TUNING="slab_pipeline=0" / "slab_batch=0,slab_chunks=2" ...), not a measurement of a node.

  --transport p2p   the device library's own model (kw_comm_p2p_emulate): the P2P exchange kernel copies every absent
                    peer's chunk locally and holds that peer's blocks for latency + bytes / link rate; host cost per
                    exchange = what the real launch costs (nothing is modelled on the host)
  --transport rccl  tests/native/mock_rccl.cpp in emulation mode behind the RCCL path: modelled link time on the
                    communication stream, --host-us of the launching thread per group (an RCCL group was measured at
                    44 us), the transfer's local HBM traffic as a device copy

  python tools/emulate_rank.py --grid 512 --ranks 8 [--rank 3] [--steps 20] [--link-gbs 60] [--latency-us 10] [--transport p2p]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--link-gbs", type=float, default=60.0)
    ap.add_argument("--host-us", type=float, default=44.0)
    ap.add_argument("--latency-us", type=float, default=10.0)
    ap.add_argument("--transport", default="p2p", choices=("p2p", "rccl"))
    a = ap.parse_args()
    mock = os.path.join(ROOT, "tests", "native", "libmock_rccl.so")
    if a.transport == "rccl":
        os.environ.update(MOCK_RCCL_EMULATE="1", MOCK_LINK_GBS=str(a.link_gbs), MOCK_GROUP_HOST_US=str(a.host_us),
                          MOCK_LINK_LATENCY_US=str(a.latency_us))
    import kwave_amd  # noqa: F401
    from kwave_amd import capi, synthetic
    from kwave_amd.dist import partition_problem, slab_range
    from kwave_amd.solver import HostSolver
    n, P, r = a.grid, a.ranks, min(a.rank, a.ranks - 1)
    z0, z1 = slab_range(n, r, P)
    pr = synthetic.make_problem(n, n, n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=a.steps + 12,
                                zslab=(z0, z1))
    loc, _ = partition_problem(pr, r, P, arrays_are_local=True)
    del pr
    if a.transport == "rccl":
        sim = HostSolver(loc, slab_ranks=P, slab_rank=r, nz_global=n, comm_unique_id=capi.comm_unique_id(mock), rccl_library=mock,
                         p_max=1)
    else:
        sim = HostSolver(loc, slab_ranks=P, slab_rank=r, nz_global=n, comm_p2p=True, p2p_emulate=(a.link_gbs, a.latency_us), p_max=1)
    sim.run(4)
    sim.sync()
    t0 = time.perf_counter()
    sim.run(a.steps)
    t1 = time.perf_counter()
    sim.sync()
    t2 = time.perf_counter()
    groups = capi.comm_exchanges(sim.ctx) // (a.steps + 4)
    tuning = os.environ.get("KW_TUNING") or "default schedule"
    host = f", {a.host_us:g} us per group on the host" if a.transport == "rccl" else ""
    print(f"{n}^3 on {P} ranks (rank {r}), {a.transport}, {tuning}: {1e3 * (t2 - t0) / a.steps:.3f} ms/step "
          f"({a.steps / (t2 - t0):.1f} steps/s), enqueue {1e3 * (t1 - t0) / a.steps:.3f} ms/step, {groups} exchange groups/step; "
          f"model: {a.link_gbs:g} GB/s per link, {a.latency_us:g} us latency{host}")
    sim.close()


if __name__ == "__main__":
    main()
