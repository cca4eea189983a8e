"""N>1 path on CPU: world_size-2 and -4 gloo runs of the slab partition + all-to-all transpose model."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [2, 4])
def test_slab_decomposition_gloo(world):
    port = 29600 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_worker_cpu.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "DIST_CPU_OK=1" in r.stdout, r.stdout[-3000:]


def test_partition_rejects_indivisible(syn):
    import kwave_amd  # noqa: F401
    from kwave_amd.dist import partition_problem
    pr = syn.make_problem(16, 16, 12, nt=4, pml_size=2)
    with pytest.raises(ValueError):
        partition_problem(pr, 0, 8)
