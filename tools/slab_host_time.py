#!/usr/bin/env python3
"""How much of a slab-mode step is spent on the launching thread: one rank through the N > 1 code path (RCCL all-to-all
with itself), wall time until run() returns (everything enqueued) vs until the device is idle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.dist import DistSolver, partition_problem  # noqa: E402

dist.init_process_group("nccl")
torch.cuda.set_device(0)
n, K = 256, int(os.environ.get("K", "100"))
pr = synthetic.make_problem(n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=2 * K + 20)
loc, _ = partition_problem(pr, 0, 1)
streams = dict(p_max=1) if os.environ.get("NO_RAW") else dict(p_raw=1, p_max=1)
sim = DistSolver(loc, 0, 1, n, device_index=0, **streams)
sim.run(10)
sim.sync()
torch.cuda.synchronize()
for _ in range(2):
    sim.exchange.host_seconds = sim.exchange.wait_seconds = 0.0
    t0 = time.perf_counter()
    sim.run(K)
    t1 = time.perf_counter()
    sim.sync()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, until idle {1e3 * (t2 - t0) / K:.3f} ms/step, of which inside the exchange callbacks {1e3 * sim.exchange.host_seconds / K:.3f} ms/step (wait(): {1e3 * sim.exchange.wait_seconds / K:.3f})", flush=True)
sim.close()
dist.destroy_process_group()
