#!/usr/bin/env python3
"""One rank through the slab path (exchanging with itself over RCCL or over the P2P transport): host enqueue time vs total
time per step — is the launching thread or the GPU the limit?
  python tools/slab_host_time.py [n] [steps] [native|p2p]      (KW_TUNING="slab_chunks=2,slab_batch=0" ... applies)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29571"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
    os.environ.setdefault(k, v)
import torch.distributed as dist  # noqa: E402
import kwave_amd  # noqa: E402,F401
from kwave_amd import synthetic  # noqa: E402
from kwave_amd.dist import DistSolver, partition_problem  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dist.init_process_group("gloo")
    pr = synthetic.make_problem(n, n, n, heterogeneous=True, nonlinear=True, absorbing=True, source="p0", nt=k + 20)
    loc, _ = partition_problem(pr, 0, 1)
    exchange = sys.argv[3] if len(sys.argv) > 3 else "native"
    sim = DistSolver(loc, 0, 1, n, exchange=exchange, p_max=1)
    sim.run(5)
    sim.sync()
    t0 = time.perf_counter()
    sim.run(k)
    t1 = time.perf_counter()
    sim.sync()
    t2 = time.perf_counter()
    print(f"{exchange}, {os.environ.get('KW_TUNING') or 'default schedule'}: "
          f"enqueue {1e3 * (t1 - t0) / k:.3f} ms/step, total {1e3 * (t2 - t0) / k:.3f} ms/step, "
          f"exchange groups per step {sim.exchanges // (k + 5)}")
    sim.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
